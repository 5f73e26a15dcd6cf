/*
 * oracle/ipm_dense_ref.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * Plain-C, single-LP-at-a-time CPU restatement of the reference's batched dense
 * primal-normal-equations interior-point path (pycllp/cl/primal_normal.cl +
 * pycllp/cl/ldl.cl, hosted by pycllp/solvers/cl.py).  It exists so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the
 * HIP kernels against an independent implementation of the same algorithm.
 * Nothing under pycllp_amd/ may import, link or call this file.
 *
 * Parity pinning: the objectives this restatement produces are pinned against
 * the reference's own CPU solver (pycllp/ipo.py -> ipo/hsd.c, compiled from
 * the reference sources into oracle/_ref/ by oracle/Makefile) through
 * tests/golden/*.npz (tools/gen_golden.py) and tests/test_oracle.py, and the
 * Newton-step solve is pinned by the known-answer formula of the reference's
 * tests/test_ldl.py:196-216.
 *
 * Semantic picks where the OpenCL kernel and its CPU twin (solvers/normal_eqns.py,
 * _ldl.pyx) disagree follow SURVEY.md section 8(a) "divergence" table:
 *   - stopping rule: RELATIVE eps on |rho|, |sigma|, gamma (reference: absolute
 *     EPS 1e-7f, primal_normal.cl:8,256) -- required to reach 1e-8 objective parity
 *     with ipo.py; the 10x growth exits (primal_normal.cl:261-269) are kept, with the
 *     reference's floor (1e-7 absolute = 1e3 x the relative tolerance for |b| ~ 1);
 *   - centering DELTA=0.02, mu = delta*gamma/(n+m) (primal_normal.cl:10,272);
 *   - step: theta starts at 0 so the step is clamped to <=1 (primal_normal.cl:134,143);
 *   - beta = sqrt(max |diag M|) (ldl.cl:280-294);
 *   - refinement: <=5 passes (ldl.cl:642-652) driven by max|residual|, but on the x-space form of the
 *     residual and with a relative tolerance -- see newton_dy();
 *   - NaN guard -> status 3 (normal_eqns.py:85-87).
 *
 * Unlike the reference, M = A diag(x/z) A' is formed once per iteration instead of
 * being recomputed entry by entry inside factor/rhs/residual (ldl.cl:110-138,198-219,
 * 577-599); the arithmetic per entry is the same sum over columns.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct oracle_opts {
    double eps;         /* relative stopping tolerance (default 1e-10)            */
    double delta;       /* centering parameter DELTA   (primal_normal.cl:10)       */
    double r;           /* step fraction R             (primal_normal.cl:11)       */
    double pivot_floor; /* LDL' diagonal floor `delta` (primal_normal.cl:275)      */
    double refine_tol;  /* refinement tolerance, relative to 1+|b| (ldl.cl:645 is 1e-8 absolute) */
    int max_iter;       /* MAX_ITER                    (primal_normal.cl:9)        */
    int max_refine;     /* refinement passes           (ldl.cl:645)                */
    int flags;          /* bit0: warm start (x,z,y are in/out, primal_normal.cl:213-219);
                           bit3 (8): autoscale -- solve with b/max|b|, c/max|c| and scale the results back;
                           bit5 (32): homogeneous self-dual embedding, see hsd_one_raw();
                           bit7 (128): Mehrotra predictor-corrector on the plain path, see ipm_one_pc() */
} oracle_opts;

void oracle_default_opts(oracle_opts *o) {
    o->eps = 1e-10;
    o->delta = 0.02;
    o->r = 0.9;
    o->pivot_floor = 1e-6;
    o->refine_tol = 1e-11;
    o->max_iter = 200;
    o->max_refine = 5;
    o->flags = 0;
}

/* packed lower-triangular index, ldl.cl:12-18 (per-LP, no batch interleave) */
static inline int tri(int i, int j) { return i * (i + 1) / 2 + j; }

/* ---- stand-alone LDL' kernels on explicit matrices (ldl.cl:28-55, 57-107; ldl.py:58-112) ---- */

/* plain LDL' of a dense symmetric n x n matrix, row-major; L packed, unit diagonal */
void oracle_ldl(int n, const double *A, double *L, double *D) {
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < i; j++) {
            double l = A[i * n + j];
            for (int k = 0; k < j; k++) l -= L[tri(i, k)] * L[tri(j, k)] * D[k];
            L[tri(i, j)] = l / D[j];
        }
        double d = A[i * n + i];
        for (int k = 0; k < i; k++) d -= D[k] * L[tri(i, k)] * L[tri(i, k)];
        D[i] = d;
        L[tri(i, i)] = 1.0;
    }
}

/* modified LDL' (Nocedal & Wright alg. 3.4 diagonal guard), ldl.cl:57-107.  The pivot floor of column j is
 * max(delta, delta_rel |A_jj|): the reference has the absolute delta only (delta_rel = 0); the homogeneous self-dual
 * path uses a floor relative to each pivot's own original diagonal entry instead (see hsd_one_raw). */
static void modified_ldl_core(int n, const double *A, double *L, double *D, double beta, double delta, double delta_rel) {
    for (int j = 0; j < n; j++) {
        double Dj = A[j * n + j];
        for (int k = 0; k < j; k++) Dj -= D[k] * L[tri(j, k)] * L[tri(j, k)];
        double theta = 0.0;
        for (int i = j + 1; i < n; i++) {
            double l = A[i * n + j];
            for (int k = 0; k < j; k++) l -= L[tri(i, k)] * L[tri(j, k)] * D[k];
            theta = fmax(theta, fabs(l));
            L[tri(i, j)] = l;
        }
        double tb = theta / beta;
        Dj = fmax(fabs(Dj), fmax(tb * tb, fmax(delta, delta_rel * fabs(A[j * n + j]))));
        for (int i = j + 1; i < n; i++) L[tri(i, j)] /= Dj;
        D[j] = Dj;
        L[tri(j, j)] = 1.0;
    }
}

void oracle_modified_ldl(int n, const double *A, double *L, double *D, double beta, double delta) {
    modified_ldl_core(n, A, L, D, beta, delta, 0.0);
}

/* S <- (L D L')^-1 S, forward then backward substitution, ldl.cl:505-537 */
void oracle_forward_backward(int m, const double *L, const double *D, double *S) {
    for (int i = 0; i < m; i++) {
        double s = S[i];
        for (int j = 0; j < i; j++) s -= S[j] * L[tri(i, j)] * D[j];
        S[i] = s / D[i];
    }
    for (int j = m - 1; j >= 0; j--) {
        double s = S[j];
        for (int i = j + 1; i < m; i++) s -= S[i] * L[tri(i, j)];
        S[j] = s;
    }
}

/* ---- the Newton step of the primal normal equations (ldl.cl:602-653) ---- */

typedef struct work {
    double *M, *L, *D, *S, *rhs, *dy, *d, *t, *w, *rho, *sigma;
} work;

static void work_alloc(work *wk, int m, int N) {
    wk->M = (double *)malloc(sizeof(double) * m * m);
    wk->L = (double *)malloc(sizeof(double) * (m * (m + 1) / 2));
    wk->D = (double *)malloc(sizeof(double) * m);
    wk->S = (double *)malloc(sizeof(double) * m);
    wk->rhs = (double *)malloc(sizeof(double) * m);
    wk->dy = (double *)malloc(sizeof(double) * m);
    wk->rho = (double *)malloc(sizeof(double) * m);
    wk->d = (double *)malloc(sizeof(double) * N);
    wk->t = (double *)malloc(sizeof(double) * N);
    wk->w = (double *)malloc(sizeof(double) * N);
    wk->sigma = (double *)malloc(sizeof(double) * N);
}

static void work_free(work *wk) {
    free(wk->M); free(wk->L); free(wk->D); free(wk->S); free(wk->rhs); free(wk->dy);
    free(wk->rho); free(wk->d); free(wk->t); free(wk->w); free(wk->sigma);
}

/* M = A diag(x/z) A' (ldl.cl:110-138), lower triangle mirrored */
static void gram(int m, int N, const double *A, const double *d, double *M) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j <= i; j++) {
            double a = 0.0;
            for (int k = 0; k < N; k++) a += A[i * N + k] * d[k] * A[j * N + k];
            M[i * m + j] = a;
            M[j * m + i] = a;
        }
}

/* factor (ldl.cl:314-378): beta from the diagonal (ldl.cl:280-294), then modified LDL' */
static void factor(int m, const double *M, double *L, double *D, double floor_, double floor_rel) {
    double beta = 0.0;
    for (int j = 0; j < m; j++) beta = fmax(beta, fabs(M[j * m + j]));
    beta = sqrt(beta);
    modified_ldl_core(m, M, L, D, beta, floor_, floor_rel);
}

/*
 * Newton step of the primal normal equations (ldl.cl:602-653):
 *   M dy = -(b - A x - A (x/z)(c - A'y + mu/x)),   dx = (c - A'y + mu/x - A'dy) x/z   (primal_normal.cl:142)
 * followed by <= max_refine passes of iterative refinement.  The reference refines on r = rhs - M dy with
 * M re-formed from x/z (ldl.cl:577-599, 642-652); in floating point that residual cannot fall below
 * eps*|M|*|dy| (|M| ~ 1/mu), which is what stalls the reference's primal feasibility near convergence
 * (SURVEY section 7 hard part 1).  The restatement refines the SAME equation in x-space instead:
 *   e = (b - A x) - A dx ;  M eta = e ;  dx += (x/z) A'eta ;  dy -= eta
 * (mathematically r == e), stopping when max|e| <= refine_tol * (1 + |b|).  dx is accumulated, so the
 * rounding error of every pass scales with the size of that pass's correction.
 * Outputs: wk->dy, wk->w (= dx).  Returns the number of refinement passes used.
 */
static int newton_dy(int m, int N, const double *A, const double *x, const double *z, const double *y,
                     const double *b, const double *c, double mu, const oracle_opts *o, work *wk) {
    double *d = wk->d, *t = wk->t, *dx = wk->w;
    double nb = 0.0;
    for (int i = 0; i < m; i++) nb += b[i] * b[i];
    const double etol = o->refine_tol * (1.0 + sqrt(nb));
    for (int k = 0; k < N; k++) {
        double aty = 0.0;
        for (int i = 0; i < m; i++) aty += A[i * N + k] * y[i];
        d[k] = x[k] / z[k];
        t[k] = c[k] - aty + mu / x[k];
    }
    gram(m, N, A, d, wk->M);
    factor(m, wk->M, wk->L, wk->D, o->pivot_floor, 0.0);
    for (int i = 0; i < m; i++) {
        double rho = b[i], adt = 0.0;
        for (int k = 0; k < N; k++) {
            rho -= A[i * N + k] * x[k];
            adt += A[i * N + k] * d[k] * t[k];
        }
        wk->rho[i] = rho;
        wk->S[i] = adt - rho; /* = -(b - Ax - A d t), ldl.cl:198-219 */
    }
    oracle_forward_backward(m, wk->L, wk->D, wk->S);
    for (int i = 0; i < m; i++) wk->dy[i] = wk->S[i];
    for (int k = 0; k < N; k++) {
        double atdy = 0.0;
        for (int i = 0; i < m; i++) atdy += A[i * N + k] * wk->dy[i];
        dx[k] = (t[k] - atdy) * d[k];
    }
    int nref = 0;
    for (;;) {
        double maxe = 0.0;
        for (int i = 0; i < m; i++) {
            double adx = 0.0;
            for (int k = 0; k < N; k++) adx += A[i * N + k] * dx[k];
            wk->S[i] = wk->rho[i] - adx;
            maxe = fmax(maxe, fabs(wk->S[i]));
        }
        if (!(maxe > etol) || nref >= o->max_refine) break;
        oracle_forward_backward(m, wk->L, wk->D, wk->S);
        for (int i = 0; i < m; i++) wk->dy[i] -= wk->S[i];
        for (int k = 0; k < N; k++) {
            double ate = 0.0;
            for (int i = 0; i < m; i++) ate += A[i * N + k] * wk->S[i];
            dx[k] += d[k] * ate;
        }
        nref++;
    }
    return nref;
}

/* exported single Newton step: mirrors the `solve_primal_normal` kernel as launched by
 * the reference tests (tests/test_ldl.py:219-273) */
int oracle_solve_primal_normal(int m, int N, const double *A, const double *x, const double *z,
                               const double *y, const double *b, const double *c, double mu,
                               double pivot_floor, double *dy) {
    oracle_opts o;
    oracle_default_opts(&o);
    o.pivot_floor = pivot_floor;
    work wk;
    work_alloc(&wk, m, N);
    int nref = newton_dy(m, N, A, x, z, y, b, c, mu, &o, &wk);
    memcpy(dy, wk.dy, sizeof(double) * m);
    work_free(&wk);
    return nref;
}

/*
 * One LP: max c'x s.t. Ax = b, x >= 0 (equality form, lp.py:306-330), the loop of
 * primal_normal.cl:201-284 with the step of primal_normal.cl:122-156.
 */
static int ipm_one_path(int m, int N, const double *A, const double *b, const double *c, double *x, double *y,
                        double *z, double *pobj, double *dobj, int *iters, int *nrefs, const oracle_opts *o,
                        work *wk);
static int hsd_one_raw(int m, int N, const double *A, const double *b, const double *c, double *x, double *y,
                       double *z, double *pobj, double *dobj, int *iters, int *nrefs, const oracle_opts *o,
                       work *wk);

static int ipm_one_pc(int m, int N, const double *A, const double *b, const double *c, double *x, double *y,
                      double *z, double *pobj, double *dobj, int *iters, int *nrefs, const oracle_opts *o,
                      work *wk);

static int ipm_one_raw(int m, int N, const double *A, const double *b, const double *c, double *x, double *y,
                       double *z, double *pobj, double *dobj, int *iters, int *nrefs, const oracle_opts *o,
                       work *wk) {
    if (o->flags & 32) return hsd_one_raw(m, N, A, b, c, x, y, z, pobj, dobj, iters, nrefs, o, wk);
    if (o->flags & 128) return ipm_one_pc(m, N, A, b, c, x, y, z, pobj, dobj, iters, nrefs, o, wk);
    return ipm_one_path(m, N, A, b, c, x, y, z, pobj, dobj, iters, nrefs, o, wk);
}

/* wrapper implementing the optional scaling (not in the reference): b/max|b|, c/max|c| */
static int ipm_one(int m, int N, const double *A, const double *b, const double *c, double *x, double *y,
                   double *z, double *pobj, double *dobj, int *iters, int *nrefs, const oracle_opts *o,
                   work *wk) {
    if (!(o->flags & 8)) return ipm_one_raw(m, N, A, b, c, x, y, z, pobj, dobj, iters, nrefs, o, wk);
    double sb = 0.0, sc = 0.0;
    for (int i = 0; i < m; i++) sb = fmax(sb, fabs(b[i]));
    for (int j = 0; j < N; j++) sc = fmax(sc, fabs(c[j]));
    if (!(sb > 0.0)) sb = 1.0;
    if (!(sc > 0.0)) sc = 1.0;
    double *bs = (double *)malloc(sizeof(double) * m), *cs = (double *)malloc(sizeof(double) * N);
    for (int i = 0; i < m; i++) bs[i] = b[i] / sb;
    for (int j = 0; j < N; j++) cs[j] = c[j] / sc;
    if (o->flags & 1) {
        for (int i = 0; i < m; i++) y[i] = y[i] / sc;
        for (int j = 0; j < N; j++) { x[j] = x[j] / sb; z[j] = z[j] / sc; }
    }
    int st = ipm_one_raw(m, N, A, bs, cs, x, y, z, pobj, dobj, iters, nrefs, o, wk);
    for (int i = 0; i < m; i++) y[i] = y[i] * sc;
    for (int j = 0; j < N; j++) { x[j] = x[j] * sb; z[j] = z[j] * sc; }
    *pobj = *pobj * (sb * sc);
    *dobj = *dobj * (sb * sc);
    free(bs); free(cs);
    return st;
}

static int ipm_one_path(int m, int N, const double *A, const double *b, const double *c, double *x, double *y,
                        double *z, double *pobj, double *dobj, int *iters, int *nrefs, const oracle_opts *o,
                        work *wk) {
    int stat = 5;
    if (!(o->flags & 1)) { /* initialize_xzyw, primal_normal.cl:14-28 */
        for (int j = 0; j < N; j++) { x[j] = 1.0; z[j] = 1.0; }
        for (int i = 0; i < m; i++) y[i] = 1.0;
    }
    double nb = 0.0, nc = 0.0;
    for (int i = 0; i < m; i++) nb += b[i] * b[i];
    for (int j = 0; j < N; j++) nc += c[j] * c[j];
    const double tol_r = o->eps * (1.0 + sqrt(nb));
    const double tol_s = o->eps * (1.0 + sqrt(nc));
    double normr0 = 1e300, norms0 = 1e300;
    int it, totref = 0;
    double po = 0.0, du = 0.0;
    for (it = 0; it < o->max_iter; it++) {
        /* primal_infeasibility, primal_normal.cl:30-48 */
        double normr = 0.0;
        for (int i = 0; i < m; i++) {
            double rho = b[i];
            for (int j = 0; j < N; j++) rho -= A[i * N + j] * x[j];
            wk->rho[i] = rho;
            normr += rho * rho;
        }
        normr = sqrt(normr);
        /* dual_infeasibility, primal_normal.cl:76-94 */
        double norms = 0.0;
        for (int j = 0; j < N; j++) {
            double sigma = c[j] + z[j];
            for (int i = 0; i < m; i++) sigma += -A[i * N + j] * y[i];
            norms += sigma * sigma;
        }
        norms = sqrt(norms);
        /* complementarity, primal_normal.cl:245-248 */
        double gamma = 0.0;
        po = 0.0; du = 0.0;
        for (int j = 0; j < N; j++) { gamma += z[j] * x[j]; po += c[j] * x[j]; }
        for (int i = 0; i < m; i++) du += b[i] * y[i];

        if (!(isfinite(normr) && isfinite(norms) && isfinite(gamma))) { stat = 3; break; }
        if (normr <= tol_r && norms <= tol_s && gamma <= o->eps * (1.0 + fabs(po))) { stat = 0; break; }
        /* growth exits with the reference's own floor: EPS = 1e-7f absolute (primal_normal.cl:8,261-269) = 1e3 x the
         * relative stopping tolerance used here, for |b|, |c| ~ 1 */
        if (normr > 10 * normr0 && normr > 1e3 * tol_r) { stat = 2; break; }
        if (norms > 10 * norms0 && norms > 1e3 * tol_s) { stat = 4; break; }

        double mu = o->delta * gamma / (N + m); /* primal_normal.cl:272 */
        totref += newton_dy(m, N, A, x, z, y, b, c, mu, o, wk);
        const double *dy = wk->dy;
        int bad = 0;
        for (int i = 0; i < m; i++) if (!isfinite(dy[i])) bad = 1;
        if (bad) { stat = 3; break; }

        /* primal_normal_step, primal_normal.cl:122-156 (dx comes refined from newton_dy) */
        double theta = 0.0;
        double *dx = wk->w, *dz = wk->sigma;
        for (int j = 0; j < N; j++) {
            dz[j] = (mu - z[j] * dx[j]) / x[j] - z[j];
            theta = fmax(theta, fmax(-dz[j] / z[j], -dx[j] / x[j]));
        }
        theta = fmin(o->r / theta, 1.0);
        for (int i = 0; i < m; i++) y[i] += theta * dy[i];
        for (int j = 0; j < N; j++) { z[j] += theta * dz[j]; x[j] += theta * dx[j]; }
        normr0 = normr;
        norms0 = norms;
    }
    *pobj = po;
    *dobj = du;
    *iters = it;
    if (nrefs) *nrefs = totref;
    return stat;
}

/*
 * Predictor-corrector variant of the same path (flag 128, PYCLLP_FLAG_PREDCORR; not in the reference's OpenCL kernel).
 * The reference's CPU solver alternates a pure predictor iteration (delta = 0) with a pure centering one (delta = 1),
 * each with its own factorisation (ipo/hsd.c:133-143, 222-260); Mehrotra's rule does both with ONE factorisation per
 * iteration -- the second solve costs a forward/back substitution, not a factorisation:
 *   predictor (mu = 0):  M dy_a = A(d t_a) - rho, t_a = c - A'y;  dx_a = d (t_a - A'dy_a);  dz_a = -z - z dx_a / x
 *   theta_a = min(1, 1 / max(-dx_a/x, -dz_a/z));  gamma_a = (x + theta_a dx_a)'(z + theta_a dz_a)
 *   centering from the predictor's success:  sigma = (gamma_a / gamma)^3,  mu = sigma gamma / N
 *   corrector:  t = t_a + (mu - dx_a dz_a) / x;  M dy = A(d t) - rho;  dx = d (t - A'dy)  (+ the x-space refinement of
 *   newton_dy);  dz = (mu - dx_a dz_a - z dx) / x - z;  step theta = min(r / max(-dx/x, -dz/z), 1) as on the plain path.
 * Everything else (start, stop tests, statuses, growth exits, objectives) is ipm_one_path's.
 */
static int ipm_one_pc(int m, int N, const double *A, const double *b, const double *c, double *x, double *y,
                      double *z, double *pobj, double *dobj, int *iters, int *nrefs, const oracle_opts *o,
                      work *wk) {
    int stat = 5;
    if (!(o->flags & 1)) {
        for (int j = 0; j < N; j++) { x[j] = 1.0; z[j] = 1.0; }
        for (int i = 0; i < m; i++) y[i] = 1.0;
    }
    double nb = 0.0, nc = 0.0;
    for (int i = 0; i < m; i++) nb += b[i] * b[i];
    for (int j = 0; j < N; j++) nc += c[j] * c[j];
    const double tol_r = o->eps * (1.0 + sqrt(nb));
    const double tol_s = o->eps * (1.0 + sqrt(nc));
    const double etol = o->refine_tol * (1.0 + sqrt(nb));
    double normr0 = 1e300, norms0 = 1e300;
    int it, totref = 0;
    double po = 0.0, du = 0.0;
    double *d = wk->d, *t = wk->t, *dx = wk->w, *dz = wk->sigma;
    double *cor = (double *)malloc(sizeof(double) * N), *aty = (double *)malloc(sizeof(double) * N);
    for (it = 0; it < o->max_iter; it++) {
        double normr = 0.0, norms = 0.0, gamma = 0.0;
        for (int i = 0; i < m; i++) {
            double rho = b[i];
            for (int j = 0; j < N; j++) rho -= A[i * N + j] * x[j];
            wk->rho[i] = rho;
            normr += rho * rho;
        }
        normr = sqrt(normr);
        po = 0.0; du = 0.0;
        for (int j = 0; j < N; j++) {
            double a = 0.0;
            for (int i = 0; i < m; i++) a += A[i * N + j] * y[i];
            aty[j] = a;
            const double sigma = c[j] - a + z[j];
            norms += sigma * sigma;
            gamma += z[j] * x[j];
            po += c[j] * x[j];
        }
        norms = sqrt(norms);
        for (int i = 0; i < m; i++) du += b[i] * y[i];
        if (!(isfinite(normr) && isfinite(norms) && isfinite(gamma))) { stat = 3; break; }
        if (normr <= tol_r && norms <= tol_s && gamma <= o->eps * (1.0 + fabs(po))) { stat = 0; break; }
        if (normr > 10 * normr0 && normr > 1e3 * tol_r) { stat = 2; break; }
        if (norms > 10 * norms0 && norms > 1e3 * tol_s) { stat = 4; break; }

        for (int k = 0; k < N; k++) { d[k] = x[k] / z[k]; t[k] = c[k] - aty[k]; }
        gram(m, N, A, d, wk->M);
        factor(m, wk->M, wk->L, wk->D, o->pivot_floor, 0.0);
        /* predictor */
        for (int i = 0; i < m; i++) {
            double adt = 0.0;
            for (int k = 0; k < N; k++) adt += A[i * N + k] * d[k] * t[k];
            wk->S[i] = adt - wk->rho[i];
        }
        oracle_forward_backward(m, wk->L, wk->D, wk->S);
        double tha = 0.0;
        for (int k = 0; k < N; k++) {
            double atdy = 0.0;
            for (int i = 0; i < m; i++) atdy += A[i * N + k] * wk->S[i];
            const double dxa = (t[k] - atdy) * d[k];
            const double dza = -z[k] - z[k] * dxa / x[k];
            dx[k] = dxa; dz[k] = dza;
            tha = fmax(tha, fmax(-dza / z[k], -dxa / x[k]));
        }
        tha = fmin(1.0 / tha, 1.0);      /* (tha = 0: 1/0 = inf -> 1) */
        double ga = 0.0;
        for (int k = 0; k < N; k++) ga += (x[k] + tha * dx[k]) * (z[k] + tha * dz[k]);
        const double sg = ga / gamma;
        const double mu = sg * sg * sg * gamma / N;
        /* corrector */
        for (int k = 0; k < N; k++) { cor[k] = mu - dx[k] * dz[k]; t[k] = t[k] + cor[k] / x[k]; }
        for (int i = 0; i < m; i++) {
            double adt = 0.0;
            for (int k = 0; k < N; k++) adt += A[i * N + k] * d[k] * t[k];
            wk->S[i] = adt - wk->rho[i];
        }
        oracle_forward_backward(m, wk->L, wk->D, wk->S);
        for (int i = 0; i < m; i++) wk->dy[i] = wk->S[i];
        for (int k = 0; k < N; k++) {
            double atdy = 0.0;
            for (int i = 0; i < m; i++) atdy += A[i * N + k] * wk->dy[i];
            dx[k] = (t[k] - atdy) * d[k];
        }
        int nref = 0;
        for (;;) {
            double maxe = 0.0;
            for (int i = 0; i < m; i++) {
                double adx = 0.0;
                for (int k = 0; k < N; k++) adx += A[i * N + k] * dx[k];
                wk->S[i] = wk->rho[i] - adx;
                maxe = fmax(maxe, fabs(wk->S[i]));
            }
            if (!(maxe > etol) || nref >= o->max_refine) break;
            oracle_forward_backward(m, wk->L, wk->D, wk->S);
            for (int i = 0; i < m; i++) wk->dy[i] -= wk->S[i];
            for (int k = 0; k < N; k++) {
                double ate = 0.0;
                for (int i = 0; i < m; i++) ate += A[i * N + k] * wk->S[i];
                dx[k] += d[k] * ate;
            }
            nref++;
        }
        totref += nref;
        int bad = 0;
        for (int i = 0; i < m; i++) if (!isfinite(wk->dy[i])) bad = 1;
        if (bad) { stat = 3; break; }
        double theta = 0.0;
        for (int j = 0; j < N; j++) {
            dz[j] = (cor[j] - z[j] * dx[j]) / x[j] - z[j];
            theta = fmax(theta, fmax(-dz[j] / z[j], -dx[j] / x[j]));
        }
        theta = fmin(o->r / theta, 1.0);
        for (int i = 0; i < m; i++) y[i] += theta * wk->dy[i];
        for (int j = 0; j < N; j++) { z[j] += theta * dz[j]; x[j] += theta * dx[j]; }
        normr0 = normr;
        norms0 = norms;
    }
    free(cor); free(aty);
    *pobj = po;
    *dobj = du;
    *iters = it;
    if (nrefs) *nrefs = totref;
    return stat;
}

/*
 * The same path on the homogeneous self-dual embedding: the model of the reference's CPU solver
 * ipo/hsd.c:27-312 (variables x, z, y plus the homogenising pair tau = `phi`, kappa = `psi`) re-derived on the
 * normal equations so that it reuses gram/factor/forward-backward unchanged:
 *     A x - b tau = 0,   c tau - A'y + z = 0,   c'x - b'y - kappa = 0,   x, z, tau, kappa >= 0.
 * With rho = b tau - A x, sigma = c tau - A'y + z, phi = b'y - c'x + kappa, eta = 1 - delta, d = x/z and
 * r1 = delta mu/x - z + eta sigma, the Newton system reduces to two solves with the same factor,
 *     M p = A(d c) - b,      M q = A(d r1) - eta rho,       dy = p dtau + q,     dx = u dtau + v,
 *     u = d (c - A'p),  v = d (r1 - A'q),
 *     dtau = (eta phi - c'v + b'q + delta mu/tau - kappa) / (|sqrt(d)(c - A'p)|^2 + kappa/tau)
 * -- the counterpart of hsd.c:222-240, which solves the reduced KKT system twice (fx,fy and gx,gy) and combines
 * them through dphi.  Differences from hsd.c, all deliberate: a fixed centering delta (hsd.c:133-137 alternates 0
 * and 1) and step fraction r as on the non-homogeneous path; the stopping rule is the same relative eps on the
 * tau-scaled residuals as ipm_one_path (hsd.c:156 stops on mu < 1e-12); infeasibility is declared from the
 * certificate itself -- status 4 when c'x > 0 and |b| tau + |rho| <= 100 eps c'x (x is then a primal ray),
 * status 2 when b'y < 0 and |c| tau + |sigma| <= 100 eps (-b'y) -- instead of from the signs of the objectives
 * once mu < 1e-12 (hsd.c:156-177); when both hold the larger certificate wins; the LDL' pivot floor is relative to the
 * pivot's own original diagonal entry (see the factor() call below).  On exit with status 0 (and 5) x, y, z
 * are divided by tau (hsd.c:266-273); with status 2/4 they are the certificate as it stands.
 */
static int hsd_one_raw(int m, int N, const double *A, const double *b, const double *c, double *x, double *y,
                       double *z, double *pobj, double *dobj, int *iters, int *nrefs, const oracle_opts *o,
                       work *wk) {
    int stat = 5;
    double tau = 1.0, kap = 1.0;
    if (!(o->flags & 1)) {
        for (int j = 0; j < N; j++) { x[j] = 1.0; z[j] = 1.0; }
        for (int i = 0; i < m; i++) y[i] = 0.0;
    } else {
        double g = 0.0;
        for (int j = 0; j < N; j++) g += x[j] * z[j];
        kap = g / N;
    }
    double nb = 0.0, nc = 0.0;
    for (int i = 0; i < m; i++) nb += b[i] * b[i];
    for (int j = 0; j < N; j++) nc += c[j] * c[j];
    nb = sqrt(nb); nc = sqrt(nc);
    const double tol_r = o->eps * (1.0 + nb), tol_s = o->eps * (1.0 + nc), einf = 100.0 * o->eps;
    const double eta = 1.0 - o->delta;
    double *p = (double *)malloc(sizeof(double) * (3 * m + 5 * N));
    double *q = p + m, *e = q + m, *d = e + m, *r1 = d + N, *atp = r1 + N, *dx = atp + N, *dz = dx + N;
    int it, totref = 0;
    double po = 0.0, du = 0.0;
    for (it = 0; it < o->max_iter; it++) {
        double normr = 0.0, norms = 0.0, gamma = 0.0;
        for (int i = 0; i < m; i++) {
            double rho = b[i] * tau;
            for (int j = 0; j < N; j++) rho -= A[i * N + j] * x[j];
            wk->rho[i] = rho;
            normr += rho * rho;
        }
        for (int j = 0; j < N; j++) {
            double sigma = c[j] * tau + z[j];
            for (int i = 0; i < m; i++) sigma -= A[i * N + j] * y[i];
            wk->sigma[j] = sigma;
            norms += sigma * sigma;
        }
        normr = sqrt(normr); norms = sqrt(norms);
        po = 0.0; du = 0.0;
        for (int j = 0; j < N; j++) { gamma += z[j] * x[j]; po += c[j] * x[j]; }
        for (int i = 0; i < m; i++) du += b[i] * y[i];
        if (!(isfinite(normr) && isfinite(norms) && isfinite(gamma) && isfinite(tau) && isfinite(kap))) { stat = 3; break; }
        if (normr <= tol_r * tau && norms <= tol_s * tau && gamma <= o->eps * tau * (tau + fabs(po))) { stat = 0; break; }
        const int p_ray = po > 0.0 && nb * tau + normr <= einf * po;
        const int d_ray = du < 0.0 && nc * tau + norms <= einf * -du;
        if (p_ray || d_ray) { stat = (p_ray && d_ray) ? (-du > po ? 2 : 4) : (p_ray ? 4 : 2); break; }

        const double mu = (gamma + tau * kap) / (N + 1), dmu = o->delta * mu;
        const double phi = du - po + kap;
        for (int k = 0; k < N; k++) {
            d[k] = x[k] / z[k];
            r1[k] = dmu / x[k] - z[k] + eta * wk->sigma[k];
        }
        gram(m, N, A, d, wk->M);
        /* The pivot floor (primal_normal.cl:275, 1e-6 absolute) does not fit this path: a dual ray drives z up and the
         * whole of M = A (x/z) A' down, and nearly-feasible infeasible LPs need pivots of 1e-7 max|M| resolved -- while a
         * rank-deficient A (duplicated rows) needs its numerically-zero pivots caught.  Both are served by a floor
         * RELATIVE TO EACH PIVOT'S OWN ORIGINAL DIAGONAL ENTRY: D_j >= pivot_floor^2 |M_jj|  (1e-12 |M_jj|). */
        factor(m, wk->M, wk->L, wk->D, 0.0, o->pivot_floor * o->pivot_floor);
        for (int i = 0; i < m; i++) {
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < N; k++) {
                s1 += A[i * N + k] * d[k] * c[k];
                s2 += A[i * N + k] * d[k] * r1[k];
            }
            p[i] = s1 - b[i];
            q[i] = s2 - eta * wk->rho[i];
        }
        oracle_forward_backward(m, wk->L, wk->D, p);
        oracle_forward_backward(m, wk->L, wk->D, q);
        double den = kap / tau, num = eta * phi + dmu / tau - kap;
        for (int i = 0; i < m; i++) num += b[i] * q[i];
        for (int k = 0; k < N; k++) {
            double ap = 0.0, aq = 0.0;
            for (int i = 0; i < m; i++) { ap += A[i * N + k] * p[i]; aq += A[i * N + k] * q[i]; }
            atp[k] = d[k] * (c[k] - ap);           /* u */
            dx[k] = d[k] * (r1[k] - aq);           /* v */
            den += atp[k] * (c[k] - ap);
            num -= c[k] * dx[k];
        }
        const double dtau = num / den;
        for (int i = 0; i < m; i++) wk->dy[i] = p[i] * dtau + q[i];
        for (int k = 0; k < N; k++) dx[k] += atp[k] * dtau;
        /* refinement on the x-space residual of  A dx - b dtau = eta rho  (as newton_dy) */
        const double etol = o->refine_tol * (1.0 + nb) * fmax(tau, kap);
        int nref = 0;
        for (;;) {
            double maxe = 0.0;
            for (int i = 0; i < m; i++) {
                double adx = 0.0;
                for (int k = 0; k < N; k++) adx += A[i * N + k] * dx[k];
                e[i] = eta * wk->rho[i] + b[i] * dtau - adx;
                maxe = fmax(maxe, fabs(e[i]));
            }
            if (!(maxe > etol) || nref >= o->max_refine) break;
            oracle_forward_backward(m, wk->L, wk->D, e);
            for (int i = 0; i < m; i++) wk->dy[i] -= e[i];
            for (int k = 0; k < N; k++) {
                double ate = 0.0;
                for (int i = 0; i < m; i++) ate += A[i * N + k] * e[i];
                dx[k] += d[k] * ate;
            }
            nref++;
        }
        totref += nref;
        int bad = !isfinite(dtau);
        for (int i = 0; i < m; i++) if (!isfinite(wk->dy[i])) bad = 1;
        if (bad) { stat = 3; break; }
        const double dkap = dmu / tau - kap - kap / tau * dtau;
        double theta = fmax(-dtau / tau, -dkap / kap);
        theta = fmax(theta, 0.0);
        for (int j = 0; j < N; j++) {
            dz[j] = (dmu - z[j] * dx[j]) / x[j] - z[j];
            theta = fmax(theta, fmax(-dz[j] / z[j], -dx[j] / x[j]));
        }
        theta = fmin(o->r / theta, 1.0);
        for (int i = 0; i < m; i++) y[i] += theta * wk->dy[i];
        for (int j = 0; j < N; j++) { z[j] += theta * dz[j]; x[j] += theta * dx[j]; }
        tau += theta * dtau;
        kap += theta * dkap;
    }
    if (stat == 0 || stat == 5) {
        for (int i = 0; i < m; i++) y[i] /= tau;
        for (int j = 0; j < N; j++) { x[j] /= tau; z[j] /= tau; }
        po /= tau; du /= tau;
    }
    free(p);
    *pobj = po;
    *dobj = du;
    *iters = it;
    if (nrefs) *nrefs = totref;
    return stat;
}

/*
 * Batched driver with the reference's problem-major layout (lp.py:338-347):
 * A [m,N] row-major shared; b [B,m]; c [B,N]; outputs x [B,N], y [B,m], z [B,N],
 * pobj/dobj [B], status/iters/nrefs [B].  nthreads<=1 -> serial.
 */
int oracle_dense_solve(int m, int N, const double *A, long B, const double *b, const double *c, double *x,
                       double *y, double *z, double *pobj, double *dobj, int *status, int *iters,
                       int *nrefs, const oracle_opts *opts, int nthreads) {
    oracle_opts o;
    if (opts) o = *opts; else oracle_default_opts(&o);
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        work wk;
        work_alloc(&wk, m, N);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
        for (long p = 0; p < B; p++) {
            int it = 0, nr = 0;
            status[p] = ipm_one(m, N, A, b + p * m, c + p * N, x + p * N, y + p * m, z + p * N, pobj + p,
                                dobj + p, &it, &nr, &o, &wk);
            iters[p] = it;
            if (nrefs) nrefs[p] = nr;
        }
        work_free(&wk);
    }
    return 0;
}
