// ipm_dense.hip -- batched dense primal-normal-equations interior-point LP solver for MI355X (gfx950).
//
// What it replaces in the reference (jetuk/pycllp): the OpenCL kernels pycllp/cl/primal_normal.cl
// (standard_primal_normal :201-284, primal_normal_step :122-156, initialize_xzyw :14-28) and
// pycllp/cl/ldl.cl (factor_primal_normal :314-378, forward_backward_primal_normal :505-537,
// residual_primal_normal :577-599, solve_primal_normal :602-653), hosted by pycllp/solvers/cl.py.
//
// Design (CDNA4-first, not a translation): the reference maps one LP to one work-item and re-reads
// x, z from global memory for every matrix entry.  Here ONE LP IS OWNED BY ONE 64-LANE WAVEFRONT for
// its whole solve; its state never leaves the CU:
//   * N-vectors (x, z, c, A'y, ...) live in registers, lane = column (columns lane and lane+64);
//   * m-vectors (y, b, rho, dy, ...) live in registers, lane = row (lane mod MP);
//   * the shared constraint matrix A sits in LDS once per workgroup, in two images: the operand order
//     of v_mfma_f64_16x16x4_f64 (for the Gram product and the A*v products) and row-major (for A'u);
//   * the Gram matrix M = A diag(x/z) A' is a true dense contraction and runs on the matrix cores
//     (MFMA f64 16x16x4, symmetric: only the lower 16x16 blocks), fused with A*x and A*(d.t);
//   * M goes through LDS once to turn the MFMA accumulator layout into "lane = row"; the modified
//     LDL' factorisation then runs out of registers (row i of the trailing matrix in lane i), the
//     transposed factor is staged in the wave's private LDS slab for the triangular solves;
//   * HBM traffic is the compulsory b, c in and x, y, z, objectives, status out: 16(m+N)+8N+24 B/LP.
// Workgroups are persistent: each wave strides over the batch.
//
// Numerical semantics follow oracle/ipm_dense_ref.c (the CPU restatement used by the tests), i.e. the
// reference algorithm with the SURVEY section 8(a) picks: relative stopping tolerance, DELTA/R of the
// OpenCL kernel, Nocedal-Wright diagonal guard, |r|-driven iterative refinement, NaN guard.
#include "wreg.h"
#include "big.h"
#include <mutex>

// ------------------------------------------------------------------------------------------------
// compile-time geometry
// ------------------------------------------------------------------------------------------------
template <int MP, int NP>
struct Geo {
    static_assert(MP == 16 || MP == 32, "row padding must be 16 or 32");
    static_assert(NP % 8 == 0 && NP >= 8 && NP <= 128, "column padding must be a multiple of 8, <= 128");
    static constexpr int JB = MP / 16;           // 16-row blocks of A
    static constexpr int KS = NP / 4;            // k-steps of the 16x16x4 MFMA
    static constexpr int NC = (NP > 64) ? 2 : 1; // columns per lane
    static constexpr int NL = 64 * NC;           // row-major image row length
    static constexpr int MS = MP + 2;            // row stride of the wave's matrix slab (16-B aligned rows,
                                                 // conflict-free b128 row reads)
    static constexpr int AMF = JB * KS * 64;     // doubles in the MFMA-order image
    static constexpr int ARM = MP * NL;          // doubles in the row-major image
    static constexpr int APACK = AMF + ARM;
    static constexpr int WSLAB = MP * MS + 3 * NP; // per-wave doubles: matrix slab + 3 k-layout vectors
    static constexpr size_t lds_bytes(int wpb) { return sizeof(double) * (size_t)(APACK + wpb * WSLAB); }
    // position of column j in a k-layout vector: lane group g = j&3 reads KS consecutive values
    __host__ __device__ static constexpr int kpos(int j) { return (j & 3) * KS + (j >> 2); }
};

// ------------------------------------------------------------------------------------------------
// pack kernel: A [m,n] row-major  ->  [ MFMA-operand image | row-major padded image ]
//   Amf[J][s][l] = A[16J + (l&15)][4s + (l>>4)]   (A/B operand lane map of v_mfma_f64_16x16x4_f64)
//   Arm[i][j]    = A[i][j]   (zero padded to MP x NL)
// ------------------------------------------------------------------------------------------------
template <int MP, int NP>
__global__ void pack_A_kernel(int m, int n, const double* __restrict__ A, double* __restrict__ pack) {
    using G = Geo<MP, NP>;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < G::AMF) {
        int l = idx & 63, s = (idx >> 6) % G::KS, J = (idx >> 6) / G::KS;
        int row = 16 * J + (l & 15), col = 4 * s + (l >> 4);
        pack[idx] = (row < m && col < n) ? A[(size_t)row * n + col] : 0.0;
    } else if (idx < G::APACK) {
        int k = idx - G::AMF;
        int row = k / G::NL, col = k % G::NL;
        pack[idx] = (row < m && col < n) ? A[(size_t)row * n + col] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
// Per-wave Newton machinery
// ------------------------------------------------------------------------------------------------
template <int MP, int NP>
struct Wave {
    using G = Geo<MP, NP>;
    static constexpr int JB = G::JB, KS = G::KS, NC = G::NC, MS = G::MS;

    const double* Amf;  // LDS, MFMA-order image
    const double* Arm;  // LDS, row-major image
    double* slab;       // LDS, wave-private MP x MS matrix slab
    double* kx;         // LDS, wave-private k-layout vectors
    double* kd;
    double* kdt;
    int lane, ri, m, n;

    // v_q = sum_i A[i][col_q] * u[i]   (u in lane=row layout)  -- A'u, reference primal_normal.cl:138-141
    __device__ __forceinline__ void At_times(double u, double out[NC]) const {
#pragma unroll
        for (int q = 0; q < NC; q++) out[q] = 0.0;
        // chunks of 8 rows: bounded unrolling keeps the LDS loads the scheduler hoists (and the broadcast SGPRs) few
#pragma unroll 1
        for (int i0 = 0; i0 < MP; i0 += 8) {
#pragma unroll
            for (int ii = 0; ii < 8; ii++) {
                const int i = i0 + ii;
                const double ui = readlane_d(u, i);
#pragma unroll
                for (int q = 0; q < NC; q++) out[q] = fma(Arm[i * G::NL + lane + 64 * q], ui, out[q]);
            }
        }
    }

    // Gram product on the matrix cores, fused with A*x and A*(d.t).
    // In: kx, kd, kdt hold x, d = x/z and d*t in k-layout.  Out: M written to the slab (full symmetric),
    // Ax / Adt in lane=row layout.
    __device__ __forceinline__ void gram_fused(double& Ax, double& Adt) const {
        const int g = lane >> 4;
        double4_t acc[JB][JB];
        double axp[JB], adp[JB];
#pragma unroll
        for (int I = 0; I < JB; I++) {
            axp[I] = 0.0; adp[I] = 0.0;
#pragma unroll
            for (int J = 0; J < JB; J++) acc[I][J] = (double4_t){0.0, 0.0, 0.0, 0.0};
        }
        const double* px = kx + g * KS;
        const double* pd = kd + g * KS;
        const double* pt = kdt + g * KS;
        static_assert(KS % 2 == 0, "k-steps are processed in pairs");
        constexpr int SC = (KS % 4 == 0) ? 4 : 2;   // k-steps per chunk (bounded unrolling: see At_times)
#pragma unroll 1
        for (int s0 = 0; s0 < KS; s0 += SC) {
#pragma unroll
            for (int ss = 0; ss < SC; ss++) {
                const int s = s0 + ss;
                const double xk = px[s], dk = pd[s], tk = pt[s];
                double a[JB], ad[JB];
#pragma unroll
                for (int J = 0; J < JB; J++) {
                    a[J] = Amf[(J * KS + s) * 64 + lane];
                    axp[J] = fma(a[J], xk, axp[J]);
                    adp[J] = fma(a[J], tk, adp[J]);
                    ad[J] = a[J] * dk;
                }
#pragma unroll
                for (int I = 0; I < JB; I++)
#pragma unroll
                    for (int J = 0; J <= I; J++)
                        acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[I], a[J], acc[I][J], 0, 0, 0);
            }
        }
        // accumulator (I,J), register r4: row 16I + (lane>>4) + 4*r4, col 16J + (lane&15)
        const int cc = lane & 15;
#pragma unroll
        for (int I = 0; I < JB; I++)
#pragma unroll
            for (int J = 0; J <= I; J++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    int row = 16 * I + g + 4 * r4, col = 16 * J + cc;
                    double v = acc[I][J][r4];
                    slab[row * MS + col] = v;
                    if (I != J) slab[col * MS + row] = v;
                }
        // reduce the A*v partials over the 4 lane groups; pick this lane's row block
#pragma unroll
        for (int J = 0; J < JB; J++) {
            axp[J] += __shfl_xor(axp[J], 16, WAVE);
            axp[J] += __shfl_xor(axp[J], 32, WAVE);
            adp[J] += __shfl_xor(adp[J], 16, WAVE);
            adp[J] += __shfl_xor(adp[J], 32, WAVE);
        }
        if (JB == 1) { Ax = axp[0]; Adt = adp[0]; }
        else {
            const bool hiblk = (ri >> 4) & 1;
            Ax = hiblk ? axp[JB - 1] : axp[0];
            Adt = hiblk ? adp[JB - 1] : adp[0];
        }
    }

    // Modified LDL' (Nocedal-Wright 3.4 guard, ldl.cl:314-378) of the matrix whose row `ri` is in W.
    // On exit: slab[j*MS + i] = L[i][j] for i>j (0 elsewhere), rdiag = 1/D[ri].
    __device__ __forceinline__ void factor(double (&W)[MP], double beta2, double floor_, double& rdiag) const {
        rdiag = 1.0;
#pragma unroll
        for (int j = 0; j < MP; j++) {
            const double u = W[j];
            const double piv = readlane_d(u, j);
            double aD = fmax(fabs(piv), floor_);
            const bool below = ri > j;
            // guard: D_j = max(|D_j|, (theta/beta)^2, floor) with theta = max_{i>j} |u_i|; it only bites when
            // some u_i^2 > beta^2 * max(|D_j|, floor), which one ballot detects
            if (__any(below && (u * u > beta2 * aD))) {
                double theta = wave_max(below ? fabs(u) : 0.0);
                aD = fmax(aD, theta * theta / beta2);
            }
            const double rD = fast_rcp(aD);
            const double l = below ? u * rD : 0.0;
            if (ri == j) rdiag = rD;
            slab[j * MS + ri] = l;  // lanes >= MP write the same value as their twin: benign
            if (j + 1 < MP) {
                wave_lds_sync();
                const double* Lj = slab + j * MS;
                double lk[MP];
#pragma unroll
                for (int k = j + 1; k < MP; k++) lk[k] = Lj[k];          // broadcast reads, batched
#pragma unroll
                for (int k = j + 1; k < MP; k++) W[k] = fma(-u, lk[k], W[k]);
                // pin the update to column j: otherwise LLVM sinks the FMA chains to the column that consumes
                // them (a left-looking sweep), keeps every broadcast L value alive and spills ~2 KB per lane
#pragma unroll
                for (int k = j + 1; k < MP; k++) asm volatile("" : "+v"(W[k]));
            }
        }
        wave_lds_sync();
    }

    // s <- (L D L')^-1 s  (ldl.cl:505-537), s in lane=row layout
    __device__ __forceinline__ double fwd_back(double s, double rdiag) const {
        // forward, column oriented: t_i -= L[i][k] t_k
#pragma unroll 1
        for (int k0 = 0; k0 < MP; k0 += 8) {
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const int k = k0 + kk;
                const double tk = readlane_d(s, k);
                s = fma(-slab[k * MS + ri], tk, s);   // row MP-1 of the slab is all zero: harmless last step
            }
        }
        s *= rdiag;
        // backward, column oriented: s_j -= L[i][j] s_i for j < i; lane j owns row j of the slab
        const double* row = slab + ri * MS;
#pragma unroll 1
        for (int i0 = MP - 8; i0 >= 0; i0 -= 8) {
#pragma unroll
            for (int ii = 7; ii >= 0; ii--) {
                const int i = i0 + ii;
                const double si = readlane_d(s, i);
                s = fma(-row[i], si, s);              // row[i] = L[i][ri] for i > ri, 0 otherwise
            }
        }
        return s;
    }

    // u = A * vq  (vq per column -> lane=row layout) through the MFMA-order image: the k-layout copy of vq is
    // staged in the wave's kx slot, every lane accumulates its 4-column slice, two shuffles fold the 4 lane groups
    __device__ __forceinline__ double A_times(const double (&vq)[NC]) const {
#pragma unroll
        for (int q = 0; q < NC; q++) {
            const int j = lane + 64 * q;
            if (j < NP) kx[G::kpos(j)] = (j < n) ? vq[q] : 0.0;
        }
        wave_lds_sync();
        const double* px = kx + (lane >> 4) * KS;
        double acc[JB];
#pragma unroll
        for (int J = 0; J < JB; J++) acc[J] = 0.0;
        constexpr int SC = (KS % 4 == 0) ? 4 : 2;
#pragma unroll 1
        for (int s0 = 0; s0 < KS; s0 += SC) {
#pragma unroll
            for (int ss = 0; ss < SC; ss++) {
                const double xk = px[s0 + ss];
#pragma unroll
                for (int J = 0; J < JB; J++) acc[J] = fma(Amf[(J * KS + s0 + ss) * 64 + lane], xk, acc[J]);
            }
        }
#pragma unroll
        for (int J = 0; J < JB; J++) {
            acc[J] += __shfl_xor(acc[J], 16, WAVE);
            acc[J] += __shfl_xor(acc[J], 32, WAVE);
        }
        wave_lds_sync();
        if (JB == 1) return acc[0];
        return ((ri >> 4) & 1) ? acc[JB - 1] : acc[0];
    }

    // Newton step of the primal normal equations (ldl.cl:602-653) given x, z, c, v=A'y, b, mu:
    //   M dy = A d t - rho,  dx = d (t - A'dy),  then <= max_refine passes of x-space refinement
    //   e = rho - A dx ; M eta = e ; dx += d A'eta ; dy -= eta      (see oracle/ipm_dense_ref.c newton_dy)
    // Returns dy (lane=row); outputs dx, wv = A'dy (per column), rho and the refinement count.
    __device__ __forceinline__ double newton(const double (&x)[NC], const double (&z)[NC], const double (&c)[NC],
                                             const double (&v)[NC], double b, double mu, double etol, const DevOpts& o,
                                             double (&dx)[NC], double (&wv)[NC], double& rho, int& nref STAMP_ARGS) const {
        double d[NC], t[NC];
#pragma unroll
        for (int q = 0; q < NC; q++) {
            const int j = lane + 64 * q;
            const bool ok = j < n;
            d[q] = ok ? x[q] / z[q] : 0.0;
            t[q] = ok ? c[q] - v[q] + mu / x[q] : 0.0;
            if (j < NP) {
                const int p = G::kpos(j);
                kx[p] = ok ? x[q] : 0.0;
                kd[p] = d[q];
                kdt[p] = d[q] * t[q];
            }
        }
        wave_lds_sync();
        STAMP(1)
        double Ax, Adt;
        gram_fused(Ax, Adt);
        STAMP(2)
        rho = b - Ax;                       // primal_normal.cl:30-48
        const double rhs = Adt - rho;       // -(b - Ax - A d t), ldl.cl:198-219
        wave_lds_sync();
        double rdiag;
        {
            // pull row ri of M out of the slab; padded rows become identity rows
            double W[MP];
            const double* row = slab + ri * MS;
#pragma unroll
            for (int k = 0; k < MP; k++) W[k] = row[k];
            double diag = slab[ri * MS + ri];
            if (ri >= m) {
#pragma unroll
                for (int k = 0; k < MP; k++) W[k] = (k == ri) ? 1.0 : W[k];
                diag = 0.0;
            }
            const double beta2 = wave_max(fabs(diag));  // beta^2 = max |M_ii|, ldl.cl:280-294
            wave_lds_sync();
            STAMP(3)
            factor(W, beta2, o.pivot_floor, rdiag);
            STAMP(4)
        }
        double dy = fwd_back(rhs, rdiag);
        STAMP(5)
        At_times(dy, wv);
        STAMP(6)
#pragma unroll
        for (int q = 0; q < NC; q++) dx[q] = (t[q] - wv[q]) * d[q];   // primal_normal.cl:142
        nref = 0;
        for (;;) {
            const double e = rho - A_times(dx);
            const double maxe = wave_max(fabs(e));
            STAMP(7)
            if (!(maxe > etol) || nref >= o.max_refine) break;
            const double eta = fwd_back(e, rdiag);
            double w2[NC];
            At_times(eta, w2);
#pragma unroll
            for (int q = 0; q < NC; q++) {
                dx[q] = fma(d[q], w2[q], dx[q]);
                wv[q] -= w2[q];
            }
            dy -= eta;
            nref++;
        }
        return dy;
    }
};

// ------------------------------------------------------------------------------------------------
// solve kernel: standard_primal_normal (primal_normal.cl:201-284), one LP per wavefront, persistent.
// FIRST GENERATION (round 1), superseded by ipm_group_kernel: compiled only with -DPYCLLP_FIRST_GEN (diagnostic A/B
// builds); the default library answers PYCLLP_FLAG_WAVE_KERNEL with PYCLLP_E_UNSUPPORTED.
// ------------------------------------------------------------------------------------------------
#ifdef PYCLLP_FIRST_GEN
template <int MP, int NP>
__global__ void __launch_bounds__(512)
ipm_solve_kernel(int m, int n, long B, const double* __restrict__ pack, const double* __restrict__ bg,
                 const double* __restrict__ cg, double* __restrict__ xg, double* __restrict__ yg,
                 double* __restrict__ zg, double* __restrict__ pobj, double* __restrict__ dobj,
                 int* __restrict__ status, int* __restrict__ iters, DevOpts o) {
    using G = Geo<MP, NP>;
    constexpr int NC = G::NC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int wpb = blockDim.x / WAVE;
    // stage both images of A (shared by the workgroup's waves)
    for (int i = tid; i < G::APACK; i += blockDim.x) lds[i] = pack[i];
    __syncthreads();

    Wave<MP, NP> w;
    const int wave = tid / WAVE;
    w.lane = tid & 63;
    w.ri = w.lane & (MP - 1);
    w.m = m; w.n = n;
    w.Amf = lds;
    w.Arm = lds + G::AMF;
    w.slab = lds + G::APACK + wave * G::WSLAB;
    w.kx = w.slab + MP * G::MS;
    w.kd = w.kx + NP;
    w.kdt = w.kd + NP;
    const int lane = w.lane, ri = w.ri;
    const bool rowok = ri < m;
    const bool warm = (o.flags & PYCLLP_FLAG_WARM_START) != 0;
    STAMP_DECL

    for (long lp = (long)blockIdx.x * wpb + wave; lp < B; lp += (long)gridDim.x * wpb) {
        double x[NC], z[NC], c[NC], v[NC];
        bool ok[NC];
#pragma unroll
        for (int q = 0; q < NC; q++) {
            const int j = lane + 64 * q;
            ok[q] = j < n;
            c[q] = ok[q] ? cg[lp * n + j] : 0.0;
            x[q] = (warm && ok[q]) ? xg[lp * n + j] : 1.0;
            z[q] = (warm && ok[q]) ? zg[lp * n + j] : 1.0;
        }
        const double b = rowok ? bg[lp * m + ri] : 0.0;
        double y = rowok ? ((warm && yg) ? yg[lp * m + ri] : 1.0) : 0.0;
        w.At_times(y, v);

        double nb2 = wave_sum((lane < MP) ? b * b : 0.0);
        double nc2 = 0.0;
#pragma unroll
        for (int q = 0; q < NC; q++) nc2 += c[q] * c[q];
        nc2 = wave_sum(nc2);
        const double tol_r = o.eps * (1.0 + sqrt(nb2));
        const double tol_s = o.eps * (1.0 + sqrt(nc2));
        const double etol = o.refine_tol * (1.0 + sqrt(nb2));
        double normr0 = 1e300, norms0 = 1e300;
        int stat = PYCLLP_STATUS_ITERATION_LIMIT, it = 0;
        double po = 0.0, du = 0.0;
        STAMP(9)

        for (it = 0; it < o.max_iter; it++) {
            // dual infeasibility, complementarity, objectives (primal_normal.cl:76-94, 245-248)
            double s2 = 0.0, gam = 0.0, pp = 0.0;
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const double sg = ok[q] ? c[q] - v[q] + z[q] : 0.0;
                s2 = fma(sg, sg, s2);
                gam += ok[q] ? x[q] * z[q] : 0.0;
                pp += c[q] * (ok[q] ? x[q] : 0.0);
            }
            s2 = wave_sum(s2); gam = wave_sum(gam); po = wave_sum(pp);
            du = wave_sum((lane < MP) ? b * y : 0.0);
            const double norms = sqrt(s2);
            const double mu = o.delta * gam / (double)(n + m);  // primal_normal.cl:272

            // Newton step (also yields rho = b - Ax for the stop test of THIS point)
            double dx[NC], wv[NC], rho;
            int nref;
            STAMP(0)
            const double dy = w.newton(x, z, c, v, b, mu, etol, o, dx, wv, rho, nref STAMP_PASS);
            const double normr = sqrt(wave_sum((lane < MP) ? rho * rho : 0.0));

            if (!(isfinite(normr) && isfinite(norms) && isfinite(gam))) { stat = PYCLLP_STATUS_NUMERICAL; break; }
            if (normr <= tol_r && norms <= tol_s && gam <= o.eps * (1.0 + fabs(po))) { stat = PYCLLP_STATUS_OPTIMAL; break; }
            if (normr > 10.0 * normr0 && normr > PYCLLP_GROWTH_FLOOR * tol_r) { stat = PYCLLP_STATUS_PRIMAL_INFEASIBLE; break; }
            if (norms > 10.0 * norms0 && norms > PYCLLP_GROWTH_FLOOR * tol_s) { stat = PYCLLP_STATUS_DUAL_INFEASIBLE; break; }
            if (__any(!isfinite(dy))) { stat = PYCLLP_STATUS_NUMERICAL; break; }

            // step (primal_normal.cl:122-156)
            double dz[NC];
            double th = 0.0;
#pragma unroll
            for (int q = 0; q < NC; q++) {
                dz[q] = ok[q] ? (mu - z[q] * dx[q]) / x[q] - z[q] : 0.0;
                if (ok[q]) th = fmax(th, fmax(-dz[q] / z[q], -dx[q] / x[q]));
            }
            th = wave_max(th);
            const double theta = fmin(o.r / th, 1.0);
            y = fma(theta, dy, y);
#pragma unroll
            for (int q = 0; q < NC; q++) {
                x[q] = fma(theta, dx[q], x[q]);
                z[q] = fma(theta, dz[q], z[q]);
                v[q] = fma(theta, wv[q], v[q]);  // A'y carried along: A'(y + theta dy)
            }
            normr0 = normr;
            norms0 = norms;
            STAMP(8)
        }

#pragma unroll
        for (int q = 0; q < NC; q++) {
            const int j = lane + 64 * q;
            if (ok[q]) {
                xg[lp * n + j] = x[q];
                if (zg) zg[lp * n + j] = z[q];
            }
        }
        if (yg && lane < MP && rowok) yg[lp * m + ri] = y;
        if (lane == 0) {
            if (pobj) pobj[lp] = po;
            if (dobj) dobj[lp] = du;
            status[lp] = stat;
            if (iters) iters[lp] = it;
        }
        STAMP(9)
    }
    STAMP_FLUSH(o, blockIdx.x * wpb + wave)
}
#endif  // PYCLLP_FIRST_GEN

// ------------------------------------------------------------------------------------------------
// stand-alone Newton step kernel: solve_primal_normal (ldl.cl:602-653) as launched by the reference's
// tests/test_ldl.py:219-273
// ------------------------------------------------------------------------------------------------
template <int MP, int NP>
__global__ void __launch_bounds__(512)
newton_kernel(int m, int n, long B, const double* __restrict__ pack, const double* __restrict__ xg,
              const double* __restrict__ zg, const double* __restrict__ yg, const double* __restrict__ bg,
              const double* __restrict__ cg, double mu, double* __restrict__ dyg, int* __restrict__ nrefg,
              DevOpts o) {
    using G = Geo<MP, NP>;
    constexpr int NC = G::NC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int wpb = blockDim.x / WAVE;
    for (int i = tid; i < G::APACK; i += blockDim.x) lds[i] = pack[i];
    __syncthreads();
    Wave<MP, NP> w;
    const int wave = tid / WAVE;
    w.lane = tid & 63;
    w.ri = w.lane & (MP - 1);
    w.m = m; w.n = n;
    w.Amf = lds;
    w.Arm = lds + G::AMF;
    w.slab = lds + G::APACK + wave * G::WSLAB;
    w.kx = w.slab + MP * G::MS;
    w.kd = w.kx + NP;
    w.kdt = w.kd + NP;
    const int lane = w.lane, ri = w.ri;
    for (long lp = (long)blockIdx.x * wpb + wave; lp < B; lp += (long)gridDim.x * wpb) {
        double x[NC], z[NC], c[NC], v[NC], dx[NC], wv[NC];
#pragma unroll
        for (int q = 0; q < NC; q++) {
            const int j = lane + 64 * q;
            const bool ok = j < n;
            x[q] = ok ? xg[lp * n + j] : 1.0;
            z[q] = ok ? zg[lp * n + j] : 1.0;
            c[q] = ok ? cg[lp * n + j] : 0.0;
        }
        const double b = (ri < m) ? bg[lp * m + ri] : 0.0;
        const double y = (ri < m) ? yg[lp * m + ri] : 0.0;
        w.At_times(y, v);
        const double etol = o.refine_tol * (1.0 + sqrt(wave_sum((lane < MP) ? b * b : 0.0)));
        double rho;
        int nref;
#ifdef PYCLLP_PROFILE
        unsigned long long t_prev_ = 0, t_acc_[NPHASE] = {0};
#endif
        const double dy = w.newton(x, z, c, v, b, mu, etol, o, dx, wv, rho, nref STAMP_PASS);
        if (lane < MP && ri < m) dyg[lp * m + ri] = dy;
        if (nrefg && lane == 0) nrefg[lp] = nref;
    }
}

#include "ipm_group.inc"
#include "ipm_group_hsd.inc"
#include "ldl_batched.inc"
#include "ipm_block.inc"

// ------------------------------------------------------------------------------------------------
// host side: C ABI
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int set_err(int code, const char* what) {
    if (code > 0) snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString((hipError_t)code));
    else snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

#define HIP_TRY(expr)                                             \
    do {                                                          \
        hipError_t e_ = (expr);                                   \
        if (e_ != hipSuccess) return set_err((int)e_, #expr);     \
    } while (0)

static constexpr unsigned kQueueRing = 64;   // work-queue counters per handle = launches that may be in flight at once

// Device work-queue heads of the persistent kernels: a ring of kQueueRing counters, one per launch in flight, so that solves
// issued on different streams / from different host threads with one handle never share a counter.  Every slot carries an
// event recorded behind the launch that used it; a slot is handed out again only once that launch has finished (the
// 65th launch in flight makes its caller wait for the oldest one instead of silently sharing its counter).
struct QueueRing {
    int* dev = nullptr;
    hipEvent_t ev[kQueueRing] = {};
    bool armed[kQueueRing] = {};
    std::atomic<unsigned> next{0};
    std::mutex mu[kQueueRing];
    hipError_t create() {
        hipError_t e = hipMalloc((void**)&dev, sizeof(int) * kQueueRing);
        for (unsigned i = 0; e == hipSuccess && i < kQueueRing; i++) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
        return e;
    }
    void destroy() {
        for (unsigned i = 0; i < kQueueRing; i++) if (ev[i]) (void)hipEventDestroy(ev[i]);
        if (dev) (void)hipFree(dev);
        dev = nullptr;
    }
    // zeroed counter for a launch on `st`; *slot identifies it for release()
    hipError_t acquire(hipStream_t st, int** head, unsigned* slot) {
        const unsigned s = next.fetch_add(1u) % kQueueRing;
        mu[s].lock();                                   // held until release(): two threads 64 launches apart
        if (armed[s]) {
            hipError_t q = hipEventSynchronize(ev[s]);   // returns at once unless 64 later launches overtook this one
            if (q != hipSuccess) { mu[s].unlock(); return q; }
        }
        *head = dev + s; *slot = s;
        hipError_t e = hipMemsetAsync(dev + s, 0, sizeof(int), st);
        if (e != hipSuccess) mu[s].unlock();
        return e;
    }
    hipError_t release(unsigned s, hipStream_t st) {
        hipError_t e = hipEventRecord(ev[s], st);
        armed[s] = (e == hipSuccess);
        mu[s].unlock();
        return e;
    }
};

struct pycllp_hip_dense {
    int m = 0, n = 0, mp = 0, np = 0, variant = -1;
    double* pack = nullptr;
    double* a_rm = nullptr;   // row-major copy of A [m,n] for the group kernel
    QueueRing ring; // device work-queue heads of the group kernel
    mutable std::mutex info_mu;   // guards grid/block/lds/mp/np (what launch_info reports: the LAST launch)
    int variant_sl = -1; // index into kSlackVariants when the last m columns of A are the identity, else -1
    int grid = 0, block = 0, lds = 0;
    int num_cu = 0;
    int max_lds = 0;
    struct pycllp_hip_sparse* sp = nullptr;   // LPs beyond the lane-group kernels (m <= 128, n <= 512): served by the sparse path's kernels
};

typedef hipError_t (*solve_launch_fn)(pycllp_hip_dense*, long, const double*, const double*, double*, double*,
                                      double*, double*, double*, int*, int*, DevOpts, hipStream_t);
typedef hipError_t (*newton_launch_fn)(pycllp_hip_dense*, long, const double*, const double*, const double*,
                                       const double*, const double*, double, double*, int*, DevOpts, hipStream_t);
typedef hipError_t (*pack_launch_fn)(pycllp_hip_dense*, const double*, hipStream_t);

template <int MP, int NP>
static int pick_wpb(const pycllp_hip_dense* h) {
    using G = Geo<MP, NP>;
    int wpb = 8;
    while (wpb > 1 && G::lds_bytes(wpb) > (size_t)h->max_lds) wpb >>= 1;
    return wpb;
}

struct LaunchPlan { int grid, block, lds, mp, np; };

static void publish(pycllp_hip_dense* h, const LaunchPlan& p) {
    std::lock_guard<std::mutex> g(h->info_mu);
    h->grid = p.grid; h->block = p.block; h->lds = p.lds; h->mp = p.mp; h->np = p.np;
}

template <int MP, int NP>
static LaunchPlan plan(const pycllp_hip_dense* h, long B) {
    using G = Geo<MP, NP>;
    const int wpb = pick_wpb<MP, NP>(h);
    long blocks = (B + wpb - 1) / wpb;
    const long resident = (long)h->num_cu * ((size_t)h->max_lds / G::lds_bytes(wpb) >= 2 ? 2 : 1);
    if (blocks > resident) blocks = resident;
    if (blocks < 1) blocks = 1;
    return LaunchPlan{(int)blocks, wpb * WAVE, (int)G::lds_bytes(wpb), MP, NP};
}

template <int MP, int NP>
static hipError_t launch_pack(pycllp_hip_dense* h, const double* A, hipStream_t st) {
    using G = Geo<MP, NP>;
    const int threads = 256, blocks = (G::APACK + threads - 1) / threads;
    hipLaunchKernelGGL((pack_A_kernel<MP, NP>), dim3(blocks), dim3(threads), 0, st, h->m, h->n, A, h->pack);
    return hipGetLastError();
}

#ifdef PYCLLP_FIRST_GEN
template <int MP, int NP>
static hipError_t launch_solve(pycllp_hip_dense* h, long B, const double* b, const double* c, double* x, double* y,
                               double* z, double* pobj, double* dobj, int* status, int* iters, DevOpts o,
                               hipStream_t st) {
    const LaunchPlan p = plan<MP, NP>(h, B);
    hipError_t e = set_dyn_lds((const void*)ipm_solve_kernel<MP, NP>, p.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ipm_solve_kernel<MP, NP>), dim3(p.grid), dim3(p.block), p.lds, st, h->m, h->n, B,
                       h->pack, b, c, x, y, z, pobj, dobj, status, iters, o);
    publish(h, p);
    return hipGetLastError();
}
#define FIRST_GEN_SOLVE(MP, NP) launch_solve<MP, NP>
#else
#define FIRST_GEN_SOLVE(MP, NP) nullptr
#endif

template <int MP, int NP, bool SL, bool HSD = false, bool PC = false>
static hipError_t launch_solve_group(pycllp_hip_dense* h, long B, const double* b, const double* c, double* x, double* y,
                                     double* z, double* pobj, double* dobj, int* status, int* iters, DevOpts o,
                                     hipStream_t st) {
    using G = GeoG<MP, NP, SL>;
    int wpb = HSD ? hsd_wpb<MP>() : PYCLLP_WPB;
    while (wpb > 1 && G::lds_bytes(wpb) > (size_t)h->max_lds) wpb--;
    // one persistent workgroup per CU: its 8 waves already use the whole register file, so a second workgroup could not
    // become resident whatever the LDS says
    const long resident = (long)h->num_cu - o.reserve_cus > 0 ? (long)h->num_cu - o.reserve_cus : 1;
    // a batch too small to fill every CU's eight waves (round 3).  Two waves on one SIMD share its issue port and
    // the lane groups of a wave share its instruction stream, so: up to one wave per SIMD everywhere (4 per CU) every LP gets a
    // wavefront of its own -- the kernel's slots are group-major, idle lane groups skip their Gram product -- then the lane
    // groups fill, and only then does a SIMD get its second wave.
    {
        const long per_cu = (B + resident - 1) / resident;                               // LPs per CU
        long want = (per_cu <= 4 * (long)G::G) ? (per_cu < 4 ? per_cu : 4) : (per_cu + G::G - 1) / G::G;
        if (want < 1) want = 1;
        if (want < wpb) wpb = (int)want;
    }
    // (workgroups: enough for one LP per wave while the batch is that small, else enough for all lane groups)
    long blocks = (B + wpb - 1) / wpb;
    if (blocks > resident) blocks = resident;
    if (blocks < 1) blocks = 1;
    const LaunchPlan p{(int)blocks, wpb * WAVE, (int)G::lds_bytes(wpb), MP, NP};
    auto kernel = HSD ? hsd_group_kernel<MP, NP, SL> : ipm_group_kernel<MP, NP, SL, PC>;
    hipError_t e = set_dyn_lds((const void*)kernel, p.lds);
    if (e != hipSuccess) return e;
    int* qhead = nullptr; unsigned slot = 0;
    e = h->ring.acquire(st, &qhead, &slot);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(p.grid), dim3(p.block), p.lds, st, h->m, h->n, B,
                       h->a_rm, b, c, x, y, z, pobj, dobj, status, iters, qhead, o);
    e = hipGetLastError();
    hipError_t e2 = h->ring.release(slot, st);
    publish(h, p);
    return e != hipSuccess ? e : e2;
}

template <int MP, int NP>
static hipError_t launch_newton(pycllp_hip_dense* h, long B, const double* x, const double* z, const double* y,
                                const double* b, const double* c, double mu, double* dy, int* nref, DevOpts o,
                                hipStream_t st) {
    const LaunchPlan p = plan<MP, NP>(h, B);
    hipError_t e = set_dyn_lds((const void*)newton_kernel<MP, NP>, p.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((newton_kernel<MP, NP>), dim3(p.grid), dim3(p.block), p.lds, st, h->m, h->n, B,
                       h->pack, x, z, y, b, c, mu, dy, nref, o);
    return hipGetLastError();
}

struct Variant {
    int mp, np, apack;
    pack_launch_fn pack;
    solve_launch_fn solve;        // wave-per-LP kernel (first generation)
    solve_launch_fn solve_group;  // group-per-LP kernel (default)
    solve_launch_fn solve_hsd;    // group-per-LP kernel on the homogeneous self-dual embedding (PYCLLP_FLAG_HSD)
    solve_launch_fn solve_pc;     // group-per-LP kernel with Mehrotra's predictor-corrector (PYCLLP_FLAG_PREDCORR)
    newton_launch_fn newton;
};

#define VARIANT(MP, NP) \
    { MP, NP, Geo<MP, NP>::APACK, launch_pack<MP, NP>, FIRST_GEN_SOLVE(MP, NP), launch_solve_group<MP, NP, false>, \
      launch_solve_group<MP, NP, false, true>, launch_solve_group<MP, NP, false, false, true>, launch_newton<MP, NP> }

// ordered by cost: the first variant that covers (m, n) is used
static const Variant kVariants[] = {
#ifdef PYCLLP_DEV_ONLY_3296   // development builds (tools/ab_*.sh): only the headline shape, compiles in a fraction of the time
    VARIANT(32, 96),
#elif defined(PYCLLP_DEV_ONLY_1648)
    VARIANT(16, 48),
#else
    VARIANT(16, 32), VARIANT(16, 48), VARIANT(16, 64), VARIANT(32, 64),
    VARIANT(32, 96), VARIANT(32, 128),
#endif
};
static const int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

// slack-aware group kernels: (MP, NP) with NP - MP padded dense columns + the m identity columns
struct SlackVariant { int mp, np; solve_launch_fn solve_group, solve_hsd, solve_pc; };
#define SLACK_VARIANT(MP, NP) { MP, NP, launch_solve_group<MP, NP, true>, launch_solve_group<MP, NP, true, true>, \
                                launch_solve_group<MP, NP, true, false, true> }
static const SlackVariant kSlackVariants[] = {
#ifdef PYCLLP_DEV_ONLY_3296
    SLACK_VARIANT(32, 96),
#elif defined(PYCLLP_DEV_ONLY_1648)
    SLACK_VARIANT(16, 48),
#else
    SLACK_VARIANT(16, 32), SLACK_VARIANT(16, 48), SLACK_VARIANT(16, 64), SLACK_VARIANT(32, 64),
    SLACK_VARIANT(32, 96), SLACK_VARIANT(32, 128),
#endif
};
static const int kNumSlackVariants = sizeof(kSlackVariants) / sizeof(kSlackVariants[0]);

static unsigned long long* g_prof = nullptr;  // diagnostic build only
#if defined(PYCLLP_PROFILE) || defined(PYCLLP_WREG_DEBUG)
extern "C" void pycllp_hip_debug_set_prof(void* p) { g_prof = (unsigned long long*)p; }
#endif

static DevOpts to_dev(const pycllp_hip_opts* opts) {
    pycllp_hip_opts d;
    pycllp_hip_default_opts(&d);
    if (opts) d = *opts;
    DevOpts o;
    o.eps = d.eps; o.delta = d.delta; o.r = d.r; o.pivot_floor = d.pivot_floor; o.refine_tol = d.refine_tol;
    o.max_iter = d.max_iter; o.flags = d.flags;
    // PYCLLP_MAX_REFINE_AUTO: the reference's cap of 5 passes (ldl.cl:645) on its own path; 20 on the homogeneous self-dual
    // variant, whose last systems are harder (with 5, LP 7557 of config 5's share stalls for 70-150 iterations)
    o.max_refine = d.max_refine >= 0 ? d.max_refine : ((d.flags & PYCLLP_FLAG_HSD) ? PYCLLP_MAX_REFINE_HSD : PYCLLP_MAX_REFINE_PLAIN);
    o.reserve_cus = d.reserve_cus < 0 ? 0 : d.reserve_cus;
    o.prof = g_prof;
    return o;
}

// ---- sparse shared-A path (one LP per workgroup) ---------------------------------------------------------------
struct pycllp_hip_sparse {
    BlockA desc = {};
    void* dev_blob = nullptr;     // one allocation holding every device array of desc
    QueueRing ring;     // work-queue heads (see pycllp_hip_dense)
    mutable std::mutex info_mu;
    int lds = 0, num_cu = 0, grid = 0;
    int lds_with_a = 0;     // LDS bytes with A's arrays in LDS (what per-problem values need), 0 when that does not fit
    int last_wreg = 0;      // 1 when the last solve ran on the wave kernel
    WregPlan* wreg = nullptr;     // tables of the register-resident wave kernel (ipm_wreg.hip), or nullptr when it does not cover A
    BigPlan* big = nullptr;       // LPs beyond m = 128 / n = 512: the workgroup-per-LP kernel of ipm_big.hip serves the handle alone
    WregPlan* wreg_pa = nullptr;  // its per-problem-A plan (structure tables only), built by the first pycllp_hip_sparse_solve_batch
    WregPlan* last_plan = nullptr;
    bool wreg_pa_tried = false;
    int max_lds = 0;
    std::vector<double> host_val; std::vector<int> host_ptr, host_col;   // host CSR copy (what a PA plan is built from)
};

template <typename T>
static T* blob_put(char* host, size_t& off, const std::vector<T>& v, char* dev_base) {
    off = (off + 15) & ~(size_t)15;
    if (!v.empty()) memcpy(host + off, v.data(), v.size() * sizeof(T));
    T* p = (T*)(dev_base + off);
    off += v.size() * sizeof(T);
    return p;
}

extern "C" {

int pycllp_hip_abi_version(void) { return PYCLLP_HIP_ABI_VERSION; }

const char* pycllp_hip_last_error(void) { return g_err; }

void pycllp_hip_default_opts(pycllp_hip_opts* o) {
    if (!o) return;
    o->eps = 1e-10;
    o->delta = 0.02;
    o->r = 0.9;
    o->pivot_floor = 1e-6;
    o->refine_tol = 1e-11;
    o->max_iter = 200;
    o->max_refine = PYCLLP_MAX_REFINE_AUTO;
    o->flags = 0;
    o->reserve_cus = 0;
}

int pycllp_hip_dense_max_rows(void) { return BIG_MAX_M; }   // m <= 32, n <= 128: lane-group kernels; beyond: the sparse path's kernels
int pycllp_hip_dense_max_cols(void) { return BIG_MAX_N; }

int pycllp_hip_dense_init(int m, int n, const double* A_dev, void* stream, pycllp_hip_dense** handle) {
    if (!A_dev || !handle || m <= 0 || n <= 0) return set_err(PYCLLP_E_BADARG, "pycllp_hip_dense_init: bad argument");
    int vi = -1;
    for (int i = 0; i < kNumVariants; i++)
        if (m <= kVariants[i].mp && n <= kVariants[i].np) { vi = i; break; }
    if (vi < 0) {
        // Beyond the lane-group kernels (m <= 32, n <= 128).  The reference's dense host has no size limit
        // (pycllp/solvers/cl.py:28-83; its own kernel test runs m = 100, N = 180): up to m = 256, n = 1280 the LP is handed
        // to the kernels of the sparse path -- the structural non-zeros of the dense A become its CSR arrays (m <= 128,
        // n <= 512: the wavefront-per-LP kernel on a dense image; beyond: the workgroup-per-LP kernel of ipm_big.hip).
        if (m > BIG_MAX_M || n > BIG_MAX_N) {
            snprintf(g_err, sizeof(g_err), "pycllp_hip_dense_init: (m=%d, n=%d) exceeds the compiled kernels (m<=%d, n<=%d)", m, n,
                     BIG_MAX_M, BIG_MAX_N);
            return PYCLLP_E_UNSUPPORTED;
        }
        hipStream_t st = (hipStream_t)stream;
        std::vector<double> Ah((size_t)m * n);
        HIP_TRY(hipMemcpyAsync(Ah.data(), A_dev, sizeof(double) * (size_t)m * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<double> val; std::vector<int> ptr(m + 1, 0), col;
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < n; j++) if (Ah[(size_t)i * n + j] != 0.0) { val.push_back(Ah[(size_t)i * n + j]); col.push_back(j); }
            ptr[i + 1] = (int)val.size();
        }
        if (val.empty()) return set_err(PYCLLP_E_BADARG, "pycllp_hip_dense_init: A is all zero");
        double* dval = nullptr; int* dptr = nullptr; int* dcol = nullptr;
        hipError_t e = hipMalloc((void**)&dval, sizeof(double) * val.size());
        if (e == hipSuccess) e = hipMalloc((void**)&dptr, sizeof(int) * ptr.size());
        if (e == hipSuccess) e = hipMalloc((void**)&dcol, sizeof(int) * col.size());
        if (e == hipSuccess) e = hipMemcpyAsync(dval, val.data(), sizeof(double) * val.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(dptr, ptr.data(), sizeof(int) * ptr.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(dcol, col.data(), sizeof(int) * col.size(), hipMemcpyHostToDevice, st);
        pycllp_hip_sparse* sp = nullptr;
        int rc = (e == hipSuccess) ? pycllp_hip_sparse_init(m, n, (int)val.size(), dval, dptr, dcol, stream, &sp) : set_err((int)e, "upload CSR of the dense A");
        if (dval) (void)hipFree(dval);
        if (dptr) (void)hipFree(dptr);
        if (dcol) (void)hipFree(dcol);
        if (rc != 0) return rc;
        pycllp_hip_dense* hs = new (std::nothrow) pycllp_hip_dense();
        if (!hs) { pycllp_hip_sparse_free(sp); return set_err(PYCLLP_E_NOMEM, "pycllp_hip_dense_init: out of host memory"); }
        hs->m = m; hs->n = n; hs->variant = -1; hs->variant_sl = -1; hs->sp = sp;
        *handle = hs;
        return 0;
    }
    pycllp_hip_dense* h = new (std::nothrow) pycllp_hip_dense();
    if (!h) return set_err(PYCLLP_E_NOMEM, "pycllp_hip_dense_init: out of host memory");
    h->m = m; h->n = n; h->variant = vi; h->mp = kVariants[vi].mp; h->np = kVariants[vi].np;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) { delete h; return set_err((int)e, "hipGetDeviceProperties"); }
    h->num_cu = prop.multiProcessorCount;
    h->max_lds = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (h->max_lds > 160 * 1024) h->max_lds = 160 * 1024;
    if (h->max_lds <= 0) h->max_lds = 64 * 1024;
    e = hipMalloc((void**)&h->pack, sizeof(double) * kVariants[vi].apack);
    if (e != hipSuccess) { delete h; return set_err((int)e, "hipMalloc(pack)"); }
    e = hipMalloc((void**)&h->a_rm, sizeof(double) * (size_t)m * n);
    if (e != hipSuccess) { (void)hipFree(h->pack); delete h; return set_err((int)e, "hipMalloc(A)"); }
    e = h->ring.create();
    if (e != hipSuccess) { h->ring.destroy(); (void)hipFree(h->pack); (void)hipFree(h->a_rm); delete h; return set_err((int)e, "hipMalloc(queue)"); }
    hipStream_t st = (hipStream_t)stream;
    e = hipMemcpyAsync(h->a_rm, A_dev, sizeof(double) * (size_t)m * n, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = kVariants[vi].pack(h, A_dev, st);
    // is A = [A_dense | I_m]?  (equality form of a StandardLP, pycllp/lp.py:551-567)
    h->variant_sl = -1;
    if (e == hipSuccess && n > m) {
        std::vector<double> Ah((size_t)m * n);
        e = hipMemcpyAsync(Ah.data(), A_dev, sizeof(double) * (size_t)m * n, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) {
            bool ident = true;
            for (int i = 0; i < m && ident; i++)
                for (int k = 0; k < m; k++)
                    if (Ah[(size_t)i * n + (n - m) + k] != (i == k ? 1.0 : 0.0)) { ident = false; break; }
            if (ident)
                for (int i = 0; i < kNumSlackVariants; i++)
                    if (m <= kSlackVariants[i].mp && n - m <= kSlackVariants[i].np - kSlackVariants[i].mp) { h->variant_sl = i; break; }
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(h->pack); (void)hipFree(h->a_rm); h->ring.destroy(); delete h; return set_err((int)e, "pack_A_kernel"); }
    *handle = h;
    return 0;
}

int pycllp_hip_dense_solve(pycllp_hip_dense* h, long B, const double* b_dev, const double* c_dev, double* x_dev,
                           double* y_dev, double* z_dev, double* pobj_dev, double* dobj_dev, int* status_dev,
                           int* iters_dev, const pycllp_hip_opts* opts, void* stream) {
    if (!h || B < 0) return set_err(PYCLLP_E_BADARG, "pycllp_hip_dense_solve: bad argument");
    if (B == 0) return 0;  // empty batch: nothing to do (pointers may be NULL)
    if (!b_dev || !c_dev || !x_dev || !status_dev)
        return set_err(PYCLLP_E_BADARG, "pycllp_hip_dense_solve: bad argument");
    if (h->sp) {   // beyond the lane-group kernels: the sparse path's kernels (flags that select lane-group variants do not apply)
        pycllp_hip_opts so;
        pycllp_hip_default_opts(&so);
        if (opts) so = *opts;
        if (so.flags & PYCLLP_FLAG_WAVE_KERNEL)
            return set_err(PYCLLP_E_BADARG, "pycllp_hip_dense_solve: PYCLLP_FLAG_WAVE_KERNEL is not available beyond m = 32, n = 128");
        so.flags &= ~PYCLLP_FLAG_NO_SLACK_PATH;
        return pycllp_hip_sparse_solve(h->sp, B, b_dev, c_dev, x_dev, y_dev, z_dev, pobj_dev, dobj_dev, status_dev, iters_dev, &so, stream);
    }
    DevOpts o = to_dev(opts);
    if ((o.flags & PYCLLP_FLAG_WARM_START) && (!y_dev || !z_dev))
        return set_err(PYCLLP_E_BADARG, "pycllp_hip_dense_solve: warm start needs y_dev and z_dev");
    if ((o.flags & PYCLLP_FLAG_AUTOSCALE) && (o.flags & PYCLLP_FLAG_WAVE_KERNEL))
        return set_err(PYCLLP_E_BADARG, "pycllp_hip_dense_solve: PYCLLP_FLAG_AUTOSCALE is not available with PYCLLP_FLAG_WAVE_KERNEL");
    if ((o.flags & PYCLLP_FLAG_HSD) && (o.flags & PYCLLP_FLAG_WAVE_KERNEL))
        return set_err(PYCLLP_E_BADARG, "pycllp_hip_dense_solve: PYCLLP_FLAG_HSD is not available with PYCLLP_FLAG_WAVE_KERNEL");
    if ((o.flags & PYCLLP_FLAG_PREDCORR) && (o.flags & (PYCLLP_FLAG_HSD | PYCLLP_FLAG_WAVE_KERNEL)))
        return set_err(PYCLLP_E_BADARG, "pycllp_hip_dense_solve: PYCLLP_FLAG_PREDCORR is an option of the reference's path (not with PYCLLP_FLAG_HSD)");
    const Variant& v = kVariants[h->variant];
    const bool hsd = (o.flags & PYCLLP_FLAG_HSD) != 0, pc = (o.flags & PYCLLP_FLAG_PREDCORR) != 0;
    solve_launch_fn fn = hsd ? v.solve_hsd : (pc ? v.solve_pc : v.solve_group);
    if (o.flags & PYCLLP_FLAG_WAVE_KERNEL) {
        fn = v.solve;
        if (!fn) return set_err(PYCLLP_E_UNSUPPORTED, "pycllp_hip_dense_solve: the first-generation kernel (PYCLLP_FLAG_WAVE_KERNEL) is not "
                                                      "part of this build (diagnostic builds: make EXTRA=-DPYCLLP_FIRST_GEN)");
    } else if (h->variant_sl >= 0 && !(o.flags & PYCLLP_FLAG_NO_SLACK_PATH))
        fn = hsd ? kSlackVariants[h->variant_sl].solve_hsd : (pc ? kSlackVariants[h->variant_sl].solve_pc : kSlackVariants[h->variant_sl].solve_group);
    hipError_t e = fn(h, B, b_dev, c_dev, x_dev, y_dev, z_dev, pobj_dev, dobj_dev, status_dev, iters_dev, o,
                      (hipStream_t)stream);
    if (e != hipSuccess) return set_err((int)e, "solve kernel launch");
    return 0;
}

int pycllp_hip_dense_newton(pycllp_hip_dense* h, long B, const double* x_dev, const double* z_dev,
                            const double* y_dev, const double* b_dev, const double* c_dev, double mu, double* dy_dev,
                            int* nrefine_dev, const pycllp_hip_opts* opts, void* stream) {
    if (!h || B < 0) return set_err(PYCLLP_E_BADARG, "pycllp_hip_dense_newton: bad argument");
    if (B == 0) return 0;
    if (!x_dev || !z_dev || !y_dev || !b_dev || !c_dev || !dy_dev)
        return set_err(PYCLLP_E_BADARG, "pycllp_hip_dense_newton: bad argument");
    if (h->sp) return pycllp_hip_sparse_newton(h->sp, B, x_dev, z_dev, y_dev, b_dev, c_dev, mu, dy_dev, nrefine_dev, opts, stream);
    DevOpts o = to_dev(opts);
    hipError_t e = kVariants[h->variant].newton(h, B, x_dev, z_dev, y_dev, b_dev, c_dev, mu, dy_dev, nrefine_dev, o,
                                                (hipStream_t)stream);
    if (e != hipSuccess) return set_err((int)e, "newton_kernel launch");
    return 0;
}

int pycllp_hip_dense_launch_info(const pycllp_hip_dense* h, int* grid, int* block, int* lds_bytes, int* m_pad,
                                 int* n_pad) {
    if (!h) return set_err(PYCLLP_E_BADARG, "pycllp_hip_dense_launch_info: bad argument");
    if (h->sp) {
        int kern = 0;
        const int rc = pycllp_hip_sparse_launch_info(h->sp, grid, block, lds_bytes, &kern);
        if (m_pad) *m_pad = BLK_MAX_M;
        if (n_pad) *n_pad = BLK_MAX_N;
        return rc;
    }
    std::lock_guard<std::mutex> g(h->info_mu);
    if (grid) *grid = h->grid;
    if (block) *block = h->block;
    if (lds_bytes) *lds_bytes = h->lds;
    if (m_pad) *m_pad = h->mp;
    if (n_pad) *n_pad = h->np;
    return 0;
}

int pycllp_hip_dense_kernel_kind(const pycllp_hip_dense* h) {
    if (!h || !h->sp) return -1;
    int kern = 0;
    if (pycllp_hip_sparse_launch_info(h->sp, nullptr, nullptr, nullptr, &kern) != 0) return -1;
    return kern;
}

int pycllp_hip_ldl(int n, long B, const double* A_dev, double* L_dev, double* D_dev, int modified, double beta,
                   double delta, void* stream) {
    if (n <= 0 || B < 0) return set_err(PYCLLP_E_BADARG, "pycllp_hip_ldl: bad argument");
    if (B == 0) return 0;
    if (!A_dev || !L_dev || !D_dev) return set_err(PYCLLP_E_BADARG, "pycllp_hip_ldl: bad argument");
    if (n > LDL_MAX_N) {
        snprintf(g_err, sizeof(g_err), "pycllp_hip_ldl: n=%d exceeds the compiled kernel (n<=%d)", n, LDL_MAX_N);
        return PYCLLP_E_UNSUPPORTED;
    }
    if (modified && !(beta > 0.0)) return set_err(PYCLLP_E_BADARG, "pycllp_hip_ldl: beta must be positive");
    const int threads = (n <= 64) ? 64 : 128;
    const int lds = (int)(sizeof(double) * ((size_t)n * (n + 1) + 8));
    hipError_t e = set_dyn_lds((const void*)ldl_batched_kernel, lds);
    if (e != hipSuccess) return set_err((int)e, "hipFuncSetAttribute(ldl_batched_kernel)");
    long blocks = B < 4096 ? B : 4096;
    hipLaunchKernelGGL(ldl_batched_kernel, dim3((unsigned)blocks), dim3(threads), lds, (hipStream_t)stream, n, B, A_dev,
                       L_dev, D_dev, modified, beta, delta);
    e = hipGetLastError();
    if (e != hipSuccess) return set_err((int)e, "ldl_batched_kernel launch");
    return 0;
}


int pycllp_hip_sparse_init(int m, int n, int nnz, const double* Adata_dev, const int* Aindptr_dev,
                           const int* Aindices_dev, void* stream, pycllp_hip_sparse** handle) {
    if (!handle || !Adata_dev || !Aindptr_dev || !Aindices_dev || m <= 0 || n <= 0 || nnz <= 0)
        return set_err(PYCLLP_E_BADARG, "pycllp_hip_sparse_init: bad argument");
    if (m > BIG_MAX_M || n > BIG_MAX_N) {
        snprintf(g_err, sizeof(g_err), "pycllp_hip_sparse_init: (m=%d, n=%d) exceeds the compiled kernels (m<=%d, n<=%d)", m, n,
                 BIG_MAX_M, BIG_MAX_N);
        return PYCLLP_E_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    std::vector<double> val(nnz);
    std::vector<int> ptr(m + 1), col(nnz);
    HIP_TRY(hipMemcpyAsync(val.data(), Adata_dev, sizeof(double) * nnz, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(ptr.data(), Aindptr_dev, sizeof(int) * (m + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(col.data(), Aindices_dev, sizeof(int) * nnz, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (ptr[0] != 0 || ptr[m] != nnz) return set_err(PYCLLP_E_BADARG, "pycllp_hip_sparse_init: malformed CSR row pointer");
    for (int i = 0; i < m; i++) if (ptr[i + 1] < ptr[i]) return set_err(PYCLLP_E_BADARG, "pycllp_hip_sparse_init: malformed CSR row pointer");
    for (int e = 0; e < nnz; e++) if (col[e] < 0 || col[e] >= n) return set_err(PYCLLP_E_BADARG, "pycllp_hip_sparse_init: column index out of range");
    if (m > BLK_MAX_M || n > BLK_MAX_N) {
        // beyond the wavefront-per-LP and block kernels: the workgroup-per-LP kernel of ipm_big.hip (blocks of the factor in
        // LDS or an L2-resident workspace)
        pycllp_hip_sparse* hb = new (std::nothrow) pycllp_hip_sparse();
        if (!hb) return set_err(PYCLLP_E_NOMEM, "pycllp_hip_sparse_init: out of host memory");
        int devb = 0;
        hipDeviceProp_t propb;
        hipError_t eb = hipGetDevice(&devb);
        if (eb == hipSuccess) eb = hipGetDeviceProperties(&propb, devb);
        if (eb == hipSuccess) eb = hb->ring.create();
        if (eb != hipSuccess) { hb->ring.destroy(); delete hb; return set_err((int)eb, "pycllp_hip_sparse_init (large LP)"); }
        hb->num_cu = propb.multiProcessorCount;
        int max_lds_b = (int)propb.maxSharedMemoryPerMultiProcessor;
        if (max_lds_b > 160 * 1024 || max_lds_b <= 0) max_lds_b = 160 * 1024;
        hb->max_lds = max_lds_b;
        hb->desc.m = m; hb->desc.n = n; hb->desc.nnz = nnz;
        const int rc = big_plan_create(m, n, nnz, val.data(), ptr.data(), col.data(), max_lds_b, st, &hb->big);
        if (rc != 0) {
            hb->ring.destroy(); delete hb;
            if (rc >= 1000) return set_err(rc - 1000, "big_plan_create");
            return set_err(PYCLLP_E_UNSUPPORTED, "pycllp_hip_sparse_init: the LP does not fit the large-LP kernel");
        }
        hb->lds = big_lds_bytes(hb->big);
        *handle = hb;
        return 0;
    }
    // CSC by counting sort (rows ascending inside a column)
    std::vector<int> cptr(n + 1, 0), crow(nnz), csrc(nnz);
    std::vector<double> cval(nnz);
    for (int e = 0; e < nnz; e++) cptr[col[e] + 1]++;
    for (int j = 0; j < n; j++) cptr[j + 1] += cptr[j];
    {
        std::vector<int> fill(cptr.begin(), cptr.end() - 1);
        for (int i = 0; i < m; i++)
            for (int e = ptr[i]; e < ptr[i + 1]; e++) { const int q = fill[col[e]]++; crow[q] = i; cval[q] = val[e]; csrc[q] = e; }
    }
    // Gram term list: entry (i,k), i >= k, gets a term a_ij a_kj for every column j holding both rows
    struct Term { int key, colj; double w; int ia, ib; };
    std::vector<Term> terms;
    for (int j = 0; j < n; j++)
        for (int a = cptr[j]; a < cptr[j + 1]; a++)
            for (int b2 = cptr[j]; b2 <= a; b2++) {
                const int i = crow[a], k = crow[b2];
                const int hi = i > k ? i : k, lo = i > k ? k : i;
                terms.push_back({hi * (hi + 1) / 2 + lo, j, cval[a] * cval[b2], csrc[a], csrc[b2]});
                if (terms.size() > (size_t)8 << 20) {
                    snprintf(g_err, sizeof(g_err), "pycllp_hip_sparse_init: A is too dense for the term-list Gram assembly");
                    return PYCLLP_E_UNSUPPORTED;
                }
            }
    std::stable_sort(terms.begin(), terms.end(), [](const Term& x, const Term& y) { return x.key < y.key; });
    std::vector<int> ent_tri, ent_ptr, term_col(terms.size()), term_ia(terms.size()), term_ib(terms.size());
    std::vector<double> term_w(terms.size());
    for (size_t t = 0; t < terms.size(); t++) {
        if (t == 0 || terms[t].key != terms[t - 1].key) { ent_tri.push_back(terms[t].key); ent_ptr.push_back((int)t); }
        term_col[t] = terms[t].colj; term_w[t] = terms[t].w; term_ia[t] = terms[t].ia; term_ib[t] = terms[t].ib;
    }
    ent_ptr.push_back((int)terms.size());

    pycllp_hip_sparse* h = new (std::nothrow) pycllp_hip_sparse();
    if (!h) return set_err(PYCLLP_E_NOMEM, "pycllp_hip_sparse_init: out of host memory");
    int dev = 0;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) { delete h; return set_err((int)e, "hipGetDeviceProperties"); }
    h->num_cu = prop.multiProcessorCount;
    int max_lds = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (max_lds > 160 * 1024 || max_lds <= 0) max_lds = 160 * 1024;
    const size_t total = 64 + sizeof(double) * (val.size() + cval.size() + term_w.size()) +
                         sizeof(int) * (ptr.size() + col.size() + cptr.size() + crow.size() + ent_tri.size() + ent_ptr.size() +
                                        term_col.size() + csrc.size() + term_ia.size() + term_ib.size()) + 16 * 16;
    std::vector<char> host(total);
    e = hipMalloc(&h->dev_blob, total);
    if (e == hipSuccess) e = h->ring.create();
    if (e != hipSuccess) { h->ring.destroy(); if (h->dev_blob) (void)hipFree(h->dev_blob); delete h; return set_err((int)e, "hipMalloc(sparse A)"); }
    size_t off = 0;
    char* db = (char*)h->dev_blob;
    BlockA& d = h->desc;
    d.m = m; d.n = n; d.nnz = nnz;
    d.csr_val = blob_put(host.data(), off, val, db);   d.csr_ptr = blob_put(host.data(), off, ptr, db);
    d.csr_col = blob_put(host.data(), off, col, db);   d.csc_val = blob_put(host.data(), off, cval, db);
    d.csc_ptr = blob_put(host.data(), off, cptr, db);  d.csc_row = blob_put(host.data(), off, crow, db);
    d.ent_tri = blob_put(host.data(), off, ent_tri, db); d.ent_ptr = blob_put(host.data(), off, ent_ptr, db);
    d.term_w = blob_put(host.data(), off, term_w, db);   d.term_col = blob_put(host.data(), off, term_col, db);
    d.csc_src = blob_put(host.data(), off, csrc, db);
    d.term_ia = blob_put(host.data(), off, term_ia, db); d.term_ib = blob_put(host.data(), off, term_ib, db);
    d.n_entries = (int)ent_tri.size(); d.n_terms = (int)terms.size();
    e = hipMemcpyAsync(h->dev_blob, host.data(), off, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(h->dev_blob); h->ring.destroy(); delete h; return set_err((int)e, "upload sparse A"); }
    // LDS plan: two workgroups per CU hide each other's LDS latency, so the CSR/CSC copy goes into LDS only when the
    // workgroup still fits in half a CU (or when it cannot be paired anyway)
    const size_t mp8 = ((size_t)m + 7) & ~(size_t)7;
    const size_t base = sizeof(double) * ((size_t)m * (m + 1) / 2 + 1 + 2 * ((size_t)n + 1) + 7 * mp8 + 8);
    const size_t with_a = base + sizeof(double) * 2 * (size_t)nnz + sizeof(int) * (2 * (size_t)nnz + m + n + 2) + 16;
    const size_t half = (size_t)max_lds / 2;
    if (with_a <= half) d.a_in_lds = 1;
    else if (base <= half) d.a_in_lds = 0;
    else d.a_in_lds = with_a <= (size_t)max_lds ? 1 : 0;
    h->lds = (int)(d.a_in_lds ? with_a : base);
    h->lds_with_a = with_a <= (size_t)max_lds ? (int)with_a : 0;
    if ((size_t)h->lds > (size_t)max_lds) {
        (void)hipFree(h->dev_blob); h->ring.destroy(); delete h;
        return set_err(PYCLLP_E_UNSUPPORTED, "pycllp_hip_sparse_init: problem does not fit in LDS");
    }
    // the register-resident one-LP-per-wavefront kernel takes over whenever its tables fit (ipm_wreg.hip)
    {
        WregPlan* wp = nullptr;
        const int rc = wreg_plan_create(m, n, nnz, val.data(), ptr.data(), col.data(), max_lds, 0, st, &wp);
        if (rc >= 1000) {
            (void)hipFree(h->dev_blob); h->ring.destroy(); delete h;
            return set_err(rc - 1000, "wreg_plan_create");
        }
        h->wreg = (rc == 0) ? wp : nullptr;
    }
    h->max_lds = max_lds;
    h->host_val = val; h->host_ptr = ptr; h->host_col = col;
    *handle = h;
    return 0;
}

// LPs the wave kernel deferred (status -1) when no guarded kernel can take them: PYCLLP_STATUS_NUMERICAL
__global__ void deferred_to_numerical_kernel(const int* __restrict__ worklist, int* __restrict__ status) {
    const int n = worklist[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) status[worklist[1 + i]] = PYCLLP_STATUS_NUMERICAL;
}

// which kernels of the sparse path implement the predictor-corrector step
static bool sparse_predcorr_available(const pycllp_hip_sparse* h, bool per_problem_a, int flags) {
    if (h->big) return true;
    if (flags & PYCLLP_FLAG_BLOCK_KERNEL) return false;
    // (per-problem A: the PA plan is built lazily by the first solve_batch; its kernels all have the variant)
    return per_problem_a ? true : wreg_has_predcorr(h->wreg) != 0;
}

static int sparse_solve_impl(pycllp_hip_sparse* h, long B, const double* a_batch, const double* b_dev, const double* c_dev,
                             double* x_dev, double* y_dev, double* z_dev, double* pobj_dev, double* dobj_dev, int* status_dev,
                             int* iters_dev, const pycllp_hip_opts* opts, void* stream) {
    if (!h || B < 0) return set_err(PYCLLP_E_BADARG, "pycllp_hip_sparse_solve: bad argument");
    if (B == 0) return 0;
    if (!b_dev || !c_dev || !x_dev || !status_dev) return set_err(PYCLLP_E_BADARG, "pycllp_hip_sparse_solve: bad argument");
    DevOpts o = to_dev(opts);
    if ((o.flags & PYCLLP_FLAG_WARM_START) && (!y_dev || !z_dev))
        return set_err(PYCLLP_E_BADARG, "pycllp_hip_sparse_solve: warm start needs y_dev and z_dev");
    hipStream_t st = (hipStream_t)stream;
    if ((o.flags & PYCLLP_FLAG_PREDCORR) && (o.flags & PYCLLP_FLAG_HSD))
        return set_err(PYCLLP_E_BADARG, "pycllp_hip_sparse_solve: PYCLLP_FLAG_PREDCORR is an option of the reference's path (not with PYCLLP_FLAG_HSD)");
    if ((o.flags & PYCLLP_FLAG_PREDCORR) && !sparse_predcorr_available(h, a_batch != nullptr, o.flags))
        return set_err(PYCLLP_E_UNSUPPORTED, "pycllp_hip_sparse_solve: PYCLLP_FLAG_PREDCORR is not available on the kernel that serves this LP");
    if (h->big) {
        if (a_batch) return set_err(PYCLLP_E_UNSUPPORTED, "pycllp_hip_sparse_solve_batch: per-problem values of A stop at m = 128, n = 512");
        int* qb = nullptr; unsigned sb_ = 0; int grid_b = 0;
        hipError_t eb = h->ring.acquire(st, &qb, &sb_);
        if (eb == hipSuccess) {
            eb = big_launch_solve(h->big, B, b_dev, c_dev, x_dev, y_dev, z_dev, pobj_dev, dobj_dev, status_dev, iters_dev, qb, o,
                                  h->num_cu, st, &grid_b);
            hipError_t er = h->ring.release(sb_, st);
            if (eb == hipSuccess) eb = er;
        }
        { std::lock_guard<std::mutex> g(h->info_mu); h->last_wreg = 0; h->last_plan = nullptr; h->grid = grid_b; }
        if (eb != hipSuccess) return set_err((int)eb, "ipm_big_kernel launch");
        return 0;
    }
    int* worklist = nullptr;
    int grid_w = 0;
    BlockA desc = h->desc;
    int lds = h->lds;
    if (a_batch && h->lds_with_a) { desc.a_in_lds = 1; lds = h->lds_with_a; }
    // per-problem values: the PA plan of the wave kernel (structure tables; built on first use), else the shared-A plan
    if (a_batch && !(o.flags & PYCLLP_FLAG_BLOCK_KERNEL)) {
        std::lock_guard<std::mutex> g(h->info_mu);
        if (!h->wreg_pa_tried) {
            h->wreg_pa_tried = true;
            WregPlan* wp = nullptr;
            const int rc = wreg_plan_create(h->desc.m, h->desc.n, h->desc.nnz, h->host_val.data(), h->host_ptr.data(), h->host_col.data(),
                                            h->max_lds, 1, st, &wp);
            if (rc >= 1000) return set_err(rc - 1000, "wreg_plan_create (per-problem A)");
            h->wreg_pa = (rc == 0) ? wp : nullptr;
        }
    }
    WregPlan* wplan = a_batch ? h->wreg_pa : h->wreg;
    const bool use_wreg = wplan && !(o.flags & PYCLLP_FLAG_BLOCK_KERNEL);
    // per-problem values on the block kernel need A's arrays in LDS next to the packed factor; a matrix too large for that is
    // served by the wave kernel's PA plan alone, and an LP it defers (guard would have bitten: never observed) ends NUMERICAL
    if ((o.flags & PYCLLP_FLAG_PREDCORR) && !use_wreg)
        return set_err(PYCLLP_E_UNSUPPORTED, "pycllp_hip_sparse_solve: PYCLLP_FLAG_PREDCORR is not available on the kernel that serves this LP");
    const bool block_can = !a_batch || h->lds_with_a != 0;
    if (!block_can && !use_wreg)
        return set_err(PYCLLP_E_UNSUPPORTED, "pycllp_hip_sparse_solve_batch: per-problem values of this A fit neither the wavefront-per-LP kernel's tables nor the block kernel's LDS");
    if (use_wreg) {
        // wave kernel first; whatever it defers (guard would have bitten) goes through the block kernel's guarded path
        HIP_TRY(hipMallocAsync((void**)&worklist, sizeof(int) * (size_t)(B + 1), st));
        int* qw = nullptr; unsigned sw = 0;
        hipError_t ew = h->ring.acquire(st, &qw, &sw);
        if (ew == hipSuccess) {
            ew = wreg_launch_solve(wplan, B, a_batch, b_dev, c_dev, x_dev, y_dev, z_dev, pobj_dev, dobj_dev, status_dev, iters_dev, qw,
                                   worklist, o, h->num_cu, st, &grid_w);
            hipError_t er = h->ring.release(sw, st);
            if (ew == hipSuccess) ew = er;
        }
        if (ew != hipSuccess) { (void)hipFreeAsync(worklist, st); return set_err((int)ew, "ipm_wreg_kernel launch"); }
    }
    long blocks = 0;
    hipError_t e = hipSuccess;
    if (!block_can) {
        hipLaunchKernelGGL(deferred_to_numerical_kernel, dim3(64), dim3(256), 0, st, worklist, status_dev);
        e = hipGetLastError();
        hipError_t e2 = hipFreeAsync(worklist, st); if (e == hipSuccess) e = e2;
        { std::lock_guard<std::mutex> g(h->info_mu); h->last_wreg = 1; h->last_plan = wplan; h->grid = grid_w; }
        if (e != hipSuccess) return set_err((int)e, "deferred_to_numerical_kernel launch");
        return 0;
    }
    HIP_TRY(set_dyn_lds((const void*)ipm_block_kernel, lds));
    const long per_cu = (160 * 1024) / lds >= 4 ? 4 : ((160 * 1024) / lds >= 2 ? 2 : 1);
    const long free_cus = (long)h->num_cu - o.reserve_cus > 0 ? (long)h->num_cu - o.reserve_cus : 1;
    blocks = free_cus * per_cu;
    if (blocks > B) blocks = B;
    int* qhead = nullptr; unsigned slot = 0;
    e = h->ring.acquire(st, &qhead, &slot);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(ipm_block_kernel, dim3((unsigned)blocks), dim3(BLK_T), lds, st, desc, B, b_dev, c_dev, x_dev,
                           y_dev, z_dev, pobj_dev, dobj_dev, status_dev, iters_dev, qhead, worklist, 0.0, nullptr, nullptr, a_batch, o);
        e = hipGetLastError();
        hipError_t er = h->ring.release(slot, st);
        if (e == hipSuccess) e = er;
    }
    if (worklist) { hipError_t e2 = hipFreeAsync(worklist, st); if (e == hipSuccess) e = e2; }
    {
        std::lock_guard<std::mutex> g(h->info_mu);
        h->last_wreg = use_wreg ? 1 : 0;
        h->last_plan = use_wreg ? wplan : nullptr;
        h->grid = use_wreg ? grid_w : (int)blocks;
    }
    if (e != hipSuccess) return set_err((int)e, "ipm_block_kernel launch");
    return 0;
}

int pycllp_hip_sparse_solve(pycllp_hip_sparse* h, long B, const double* b_dev, const double* c_dev, double* x_dev,
                            double* y_dev, double* z_dev, double* pobj_dev, double* dobj_dev, int* status_dev,
                            int* iters_dev, const pycllp_hip_opts* opts, void* stream) {
    return sparse_solve_impl(h, B, nullptr, b_dev, c_dev, x_dev, y_dev, z_dev, pobj_dev, dobj_dev, status_dev, iters_dev, opts, stream);
}

int pycllp_hip_sparse_solve_batch(pycllp_hip_sparse* h, long B, const double* Adata_dev, const double* b_dev,
                                  const double* c_dev, double* x_dev, double* y_dev, double* z_dev, double* pobj_dev,
                                  double* dobj_dev, int* status_dev, int* iters_dev, const pycllp_hip_opts* opts, void* stream) {
    if (B > 0 && !Adata_dev) return set_err(PYCLLP_E_BADARG, "pycllp_hip_sparse_solve_batch: bad argument");
    return sparse_solve_impl(h, B, Adata_dev, b_dev, c_dev, x_dev, y_dev, z_dev, pobj_dev, dobj_dev, status_dev, iters_dev, opts, stream);
}

int pycllp_hip_sparse_newton(pycllp_hip_sparse* h, long B, const double* x_dev, const double* z_dev, const double* y_dev,
                             const double* b_dev, const double* c_dev, double mu, double* dy_dev, int* nrefine_dev,
                             const pycllp_hip_opts* opts, void* stream) {
    if (!h || B < 0) return set_err(PYCLLP_E_BADARG, "pycllp_hip_sparse_newton: bad argument");
    if (B == 0) return 0;
    if (!x_dev || !z_dev || !y_dev || !b_dev || !c_dev || !dy_dev)
        return set_err(PYCLLP_E_BADARG, "pycllp_hip_sparse_newton: bad argument");
    DevOpts o = to_dev(opts);
    hipStream_t st = (hipStream_t)stream;
    if (h->big) {
        int* qb = nullptr; unsigned sb_ = 0;
        hipError_t eb = h->ring.acquire(st, &qb, &sb_);
        if (eb == hipSuccess) {
            eb = big_launch_newton(h->big, B, x_dev, z_dev, y_dev, b_dev, c_dev, mu, dy_dev, nrefine_dev, qb, o, h->num_cu, st);
            hipError_t er = h->ring.release(sb_, st);
            if (eb == hipSuccess) eb = er;
        }
        if (eb != hipSuccess) return set_err((int)eb, "ipm_big_kernel (Newton mode) launch");
        return 0;
    }
    if (h->wreg && !(o.flags & PYCLLP_FLAG_BLOCK_KERNEL)) {
        hipError_t e = wreg_launch_newton(h->wreg, B, x_dev, z_dev, y_dev, b_dev, c_dev, mu, dy_dev, nrefine_dev, o, h->num_cu, st);
        if (e != hipSuccess) return set_err((int)e, "newton_wreg_kernel launch");
        return 0;
    }
    // matrices the wave kernel does not cover (dense, or tables larger than LDS): the block kernel in its Newton mode
    HIP_TRY(set_dyn_lds((const void*)ipm_block_kernel, h->lds));
    const long per_cu = (160 * 1024) / h->lds >= 4 ? 4 : ((160 * 1024) / h->lds >= 2 ? 2 : 1);
    long blocks = (long)h->num_cu * per_cu;
    if (blocks > B) blocks = B;
    int* qhead = nullptr; unsigned slot = 0;
    hipError_t e = h->ring.acquire(st, &qhead, &slot);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(ipm_block_kernel, dim3((unsigned)blocks), dim3(BLK_T), h->lds, st, h->desc, B, b_dev, c_dev,
                           (double*)x_dev, (double*)y_dev, (double*)z_dev, (double*)nullptr, (double*)nullptr, (int*)nullptr,
                           (int*)nullptr, qhead, (const int*)nullptr, mu, dy_dev, nrefine_dev, (const double*)nullptr, o);
        e = hipGetLastError();
        hipError_t er = h->ring.release(slot, st);
        if (e == hipSuccess) e = er;
    }
    if (e != hipSuccess) return set_err((int)e, "ipm_block_kernel (Newton mode) launch");
    return 0;
}

int pycllp_hip_sparse_launch_info(const pycllp_hip_sparse* h, int* grid, int* block, int* lds_bytes, int* kernel) {
    if (!h) return set_err(PYCLLP_E_BADARG, "pycllp_hip_sparse_launch_info: bad argument");
    std::lock_guard<std::mutex> g(h->info_mu);
    if (grid) *grid = h->grid;
    if (block) *block = h->last_wreg ? wreg_block_threads(h->last_plan) : BLK_T;
    if (lds_bytes) *lds_bytes = h->last_wreg ? wreg_lds_bytes(h->last_plan) : h->lds;
    if (kernel) *kernel = h->big ? (big_dense_mode(h->big) ? 4 : 3) : (h->last_wreg ? wreg_variant(h->last_plan) : 0);
    return 0;
}

int pycllp_hip_ldl_solve(int n, long B, const double* A_dev, const double* rhs_dev, double* x_dev, int modified, double beta,
                         double delta, void* stream) {
    if (n <= 0 || B < 0) return set_err(PYCLLP_E_BADARG, "pycllp_hip_ldl_solve: bad argument");
    if (B == 0) return 0;
    if (!A_dev || !rhs_dev || !x_dev) return set_err(PYCLLP_E_BADARG, "pycllp_hip_ldl_solve: bad argument");
    if (n > LDL_MAX_N) {
        snprintf(g_err, sizeof(g_err), "pycllp_hip_ldl_solve: n=%d exceeds the compiled kernel (n<=%d)", n, LDL_MAX_N);
        return PYCLLP_E_UNSUPPORTED;
    }
    if (modified && !(beta > 0.0)) return set_err(PYCLLP_E_BADARG, "pycllp_hip_ldl_solve: beta must be positive");
    hipStream_t st = (hipStream_t)stream;
    if (!modified) {
        int dev = 0, ncu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        hipError_t e = wreg_launch_ldl_solve(n, B, A_dev, rhs_dev, x_dev, 0.0, ncu, st);
        if (e != hipSuccess) return set_err((int)e, "ldl_solve_wreg_kernel launch");
        return 0;
    }
    const int lds = (int)(sizeof(double) * ((size_t)n * (n + 1) + n + 8));
    hipError_t e = set_dyn_lds((const void*)ldl_solve_batched_kernel, lds);
    if (e != hipSuccess) return set_err((int)e, "hipFuncSetAttribute(ldl_solve_batched_kernel)");
    const long blocks = B < 4096 ? B : 4096;
    hipLaunchKernelGGL(ldl_solve_batched_kernel, dim3((unsigned)blocks), dim3(n <= 64 ? 64 : 128), lds, st, n, B, A_dev, rhs_dev,
                       x_dev, modified, beta, delta);
    e = hipGetLastError();
    if (e != hipSuccess) return set_err((int)e, "ldl_solve_batched_kernel launch");
    return 0;
}

int pycllp_hip_forward_backward_ldl(int n, long B, const double* L_dev, const double* D_dev, const double* b_dev, double* x_dev,
                                    void* stream) {
    if (n <= 0 || B < 0) return set_err(PYCLLP_E_BADARG, "pycllp_hip_forward_backward_ldl: bad argument");
    if (B == 0) return 0;
    if (!L_dev || !D_dev || !b_dev || !x_dev) return set_err(PYCLLP_E_BADARG, "pycllp_hip_forward_backward_ldl: bad argument");
    if (n > LDL_MAX_N) {
        snprintf(g_err, sizeof(g_err), "pycllp_hip_forward_backward_ldl: n=%d exceeds the compiled kernel (n<=%d)", n, LDL_MAX_N);
        return PYCLLP_E_UNSUPPORTED;
    }
    const int lds = (int)(sizeof(double) * ((size_t)n * (n + 1) / 2 + n + 8));
    hipError_t e = set_dyn_lds((const void*)forward_backward_ldl_kernel, lds);
    if (e != hipSuccess) return set_err((int)e, "hipFuncSetAttribute(forward_backward_ldl_kernel)");
    const long blocks = B < 4096 ? B : 4096;
    hipLaunchKernelGGL(forward_backward_ldl_kernel, dim3((unsigned)blocks), dim3(n <= 64 ? 64 : 128), lds, (hipStream_t)stream, n, B,
                       L_dev, D_dev, b_dev, x_dev);
    e = hipGetLastError();
    if (e != hipSuccess) return set_err((int)e, "forward_backward_ldl_kernel launch");
    return 0;
}

void pycllp_hip_sparse_free(pycllp_hip_sparse* h) {
    if (!h) return;
    if (h->dev_blob) (void)hipFree(h->dev_blob);
    h->ring.destroy();
    wreg_plan_free(h->wreg);
    wreg_plan_free(h->wreg_pa);
    big_plan_free(h->big);
    delete h;
}

int pycllp_hip_sparse_max_rows(void) { return BIG_MAX_M; }
int pycllp_hip_sparse_max_cols(void) { return BIG_MAX_N; }

void pycllp_hip_dense_free(pycllp_hip_dense* h) {
    if (!h) return;
    if (h->sp) pycllp_hip_sparse_free(h->sp);
    if (h->pack) (void)hipFree(h->pack);
    if (h->a_rm) (void)hipFree(h->a_rm);
    h->ring.destroy();
    delete h;
}

}  // extern "C"
