/*
 * pycllp_hip.h -- C ABI of libpycllp_hip.so, the MI355X (gfx950) implementation of pycllp's batched
 * dense primal-normal-equations interior-point path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  Every pointer argument
 * named *_dev is a DEVICE pointer owned by the caller (the Python host passes torch tensors'
 * data_ptr()); `stream` is a hipStream_t passed as void* (NULL = default stream).  No entry point
 * synchronises the host with the device except where stated; all of them are re-entrant per handle: solves
 * may be issued from several host threads and on several streams with one handle (the caller keeps their output
 * buffers apart).  A handle keeps 64 device work-queue counters; when 64 launches of one handle are still in flight
 * the next call waits on the host for the oldest of them (the only place a solve entry may block).
 * *_launch_info report the plan of the LAST launch on the handle.
 * Return value: 0 on success, a negative PYCLLP_E_* code for argument errors, or a positive
 * hipError_t for runtime failures (pycllp_hip_last_error() gives the text).
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   pycllp_hip_dense_init   <- ClDensePrimalNormalSolver.init      pycllp/solvers/cl.py:28-83
 *                              (densify+upload A once, allocate per-solver device state)
 *   pycllp_hip_dense_solve  <- ClDensePrimalNormalSolver.solve     pycllp/solvers/cl.py:85-124
 *                              kernels initialize_xzyw              pycllp/cl/primal_normal.cl:14-28
 *                                      standard_primal_normal       pycllp/cl/primal_normal.cl:201-284
 *                                      (-> solve_primal_normal      pycllp/cl/ldl.cl:602-653,
 *                                          primal_normal_step       pycllp/cl/primal_normal.cl:122-156)
 *   pycllp_hip_dense_newton <- kernel solve_primal_normal launched stand-alone by the reference's
 *                              tests/test_ldl.py:219-273            pycllp/cl/ldl.cl:602-653
 *   pycllp_hip_dense_free   <- release of ClDensePrimalNormalSolver.buffers  pycllp/solvers/cl.py:26
 *   pycllp_hip_ldl          <- test kernels ldl / modified_ldl                pycllp/cl/ldl.cl:28-55, 57-107
 *                              (host: pycllp/ldl.py:58-128, launched by tests/test_ldl.py:139-193)
 *
 * Layouts are the problem-major ones of the LP container (pycllp/lp.py:338-347), NOT the
 * batch-interleaved transposes the OpenCL host builds (pycllp/solvers/cl.py:99,102):
 *   A [m, n] row-major (shared by the batch);  b [B, m];  c [B, n];  x, z [B, n];  y [B, m].
 * The LP is in equality form: maximise c'x subject to A x = b, x >= 0 (pycllp/lp.py:306-330).
 */
#ifndef PYCLLP_HIP_H
#define PYCLLP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PYCLLP_HIP_ABI_VERSION 1

/* per-LP status codes, identical to the reference (pycllp/cl/primal_normal.cl:225,257,262,267;
 * pycllp/solvers/normal_eqns.py:85-87; names in pycllp/common/main.c:21-30) */
#define PYCLLP_STATUS_OPTIMAL 0
#define PYCLLP_STATUS_PRIMAL_INFEASIBLE 2
#define PYCLLP_STATUS_NUMERICAL 3
#define PYCLLP_STATUS_DUAL_INFEASIBLE 4
#define PYCLLP_STATUS_ITERATION_LIMIT 5

#define PYCLLP_E_BADARG (-1)      /* NULL pointer / non-positive size                     */
#define PYCLLP_E_UNSUPPORTED (-2) /* (m, n) outside what the compiled kernels cover        */
#define PYCLLP_E_NOMEM (-3)

/* flags */
#define PYCLLP_FLAG_WARM_START 1 /* x, z, y are in/out: start from the caller's point instead of
                                    x=z=y=1 (intent of pycllp/cl/primal_normal.cl:213-219)  */
#define PYCLLP_FLAG_AUTOSCALE 8 /* solve every LP with b/max|b| and c/max|c| and scale the results back: makes the
                                   unit-floored tolerances and the x=z=y=1 start scale invariant (not in the reference;
                                   not available with PYCLLP_FLAG_WAVE_KERNEL)                                       */
#define PYCLLP_FLAG_HSD 32 /* solve on the homogeneous self-dual embedding (the model of the reference's CPU solver,
                              pycllp/ipo/hsd.c:27-312, re-derived on the normal equations): infeasible
                              and unbounded LPs end with a certificate after ~12-15 iterations -- status 2 with (y, z):
                              b'y < 0, A'y - z ~ 0; status 4 with x: c'x > 0, A x ~ 0 -- instead of through the
                              heuristic 10x-growth exits.  x, y, z of a status 2/4 LP are the certificate in homogeneous
                              scaling.  The LDL' pivot floor of column j on this path is pivot_floor^2 * |M_jj| (own original diagonal).
                              Dense and sparse solvers; not available with PYCLLP_FLAG_WAVE_KERNEL.                */
#define PYCLLP_FLAG_PREDCORR 128 /* Mehrotra's predictor-corrector on the reference's path (not in its OpenCL kernel; its CPU solver
                                    alternates predictor and centering iterations, pycllp/ipo/hsd.c:133-143, 222-260): per
                                    iteration ONE factorisation and two solves -- a predictor with mu = 0, the centering parameter
                                    (gamma_affine / gamma)^3 from how far it gets, the corrector with the second-order term.  Same
                                    optimum to the same tolerance in ~27 % fewer iterations at r = 0.9, ~45 % fewer at r = 0.99
                                    (45-50 % on config 5's structure).  An option: the default path stays the reference's rule.
                                    Not with PYCLLP_FLAG_HSD.  oracle/ipm_dense_ref.c ipm_one_pc is its restatement.          */
#define PYCLLP_FLAG_NO_SLACK_PATH 16 /* do not use the slack-aware kernel even when the last m columns of A are the
                                        identity (diagnostic: results must agree to rounding)                    */
#define PYCLLP_FLAG_FORCE_GUARD_PATH 4 /* diagnostic: always run the guarded (cold) LDL' path of the group
                                          kernel; results must not change when the guard is inactive */
#define PYCLLP_FLAG_BLOCK_KERNEL 64 /* sparse solver: use the workgroup-per-LP kernel (ipm_block_kernel) even where the
                                       register-resident wavefront-per-LP kernel covers the problem (diagnostic / A-B runs;
                                       results agree to rounding)                                                  */
#define PYCLLP_FLAG_WAVE_KERNEL 2 /* use the first-generation kernel (one LP per wavefront) instead
                                     of the default one (one LP per 16/32-lane group)          */

#define PYCLLP_MAX_REFINE_AUTO (-1)
#define PYCLLP_MAX_REFINE_PLAIN 5
#define PYCLLP_MAX_REFINE_HSD 20

typedef struct pycllp_hip_opts {
    double eps;         /* relative stopping tolerance on |rho|,|sigma|,gamma; default 1e-10.
                           (reference: absolute EPS 1e-7f, primal_normal.cl:8,256)           */
    double delta;       /* centering parameter DELTA, default 0.02 (primal_normal.cl:10)     */
    double r;           /* step fraction R, default 0.9 (primal_normal.cl:11)                */
    double pivot_floor; /* LDL' diagonal floor, default 1e-6 (primal_normal.cl:275)          */
    double refine_tol;  /* refinement tolerance on max|b-Ax-A dx|, relative to 1+|b|, default
                           1e-11 (reference: 1e-8 absolute on rhs-M dy, ldl.cl:645)          */
    int max_iter;       /* default 200 (primal_normal.cl:9)                                  */
    int max_refine;     /* refinement passes per Newton system.  Default PYCLLP_MAX_REFINE_AUTO (-1): resolved inside
                           every entry point to 5 (ldl.cl:645) on the reference's path and to 20 with PYCLLP_FLAG_HSD
                           (DESIGN.md section 9); any value >= 0 is taken as given               */
    int flags;          /* PYCLLP_FLAG_*                                                     */
    int reserve_cus;    /* compute units the solve leaves idle (default 0).  The solve kernels are persistent and fill every CU
                           completely (LDS and registers), so a kernel on another stream -- e.g. the RCCL copy kernels of the
                           result gather that overlaps the next solve -- finds no free CU until a solve ends; reserving a few
                           (8 = one per XCD) lets it run beside the solve                                    */
} pycllp_hip_opts;

typedef struct pycllp_hip_dense pycllp_hip_dense; /* opaque per-solver device state */

int pycllp_hip_abi_version(void);
const char *pycllp_hip_last_error(void);
void pycllp_hip_default_opts(pycllp_hip_opts *opts);

/* Largest (m, n) the compiled kernels accept (n counts ALL columns of the equality form). */
int pycllp_hip_dense_max_rows(void);
int pycllp_hip_dense_max_cols(void);

/* Upload/pack the shared constraint matrix.  A_dev: [m, n] row-major f64 on the device.
 * Synchronises `stream` before returning (A_dev may be freed by the caller afterwards). */
int pycllp_hip_dense_init(int m, int n, const double *A_dev, void *stream, pycllp_hip_dense **handle);

/* Solve B LPs.  Inputs b_dev [B,m], c_dev [B,n].  Outputs (any of y/z/pobj/dobj/iters may be NULL):
 *   x_dev [B,n], y_dev [B,m], z_dev [B,n]  primal, dual and dual-slack solutions
 *   pobj_dev, dobj_dev [B]                 c'x and b'y at exit (objective offset f NOT added)
 *   status_dev [B] i32, iters_dev [B] i32  status code and IPM iterations used
 * Asynchronous on `stream`. */
int pycllp_hip_dense_solve(pycllp_hip_dense *handle, long B, const double *b_dev, const double *c_dev,
                           double *x_dev, double *y_dev, double *z_dev, double *pobj_dev,
                           double *dobj_dev, int *status_dev, int *iters_dev,
                           const pycllp_hip_opts *opts, void *stream);

/* One Newton step of the primal normal equations for B independent states:
 *   dy <- solve( A diag(x/z) A' , -(b - A x - A diag(x/z) (c - A'y + mu/x)) )
 * x,z,c [B,n]; y,b,dy [B,m].  nrefine_dev [B] (optional) receives the refinement passes used. */
int pycllp_hip_dense_newton(pycllp_hip_dense *handle, long B, const double *x_dev, const double *z_dev,
                            const double *y_dev, const double *b_dev, const double *c_dev, double mu,
                            double *dy_dev, int *nrefine_dev, const pycllp_hip_opts *opts, void *stream);

/* Kernel-level statistics of the last solve launch on this handle (host values). */
int pycllp_hip_dense_launch_info(const pycllp_hip_dense *handle, int *grid, int *block, int *lds_bytes,
                                 int *m_pad, int *n_pad);

/* Which kernel family serves this handle: -1 = the lane-group kernels (m <= 32, n <= 128); otherwise the LP was handed to
 * the sparse path's kernels at init and the value is that of pycllp_hip_sparse_launch_info's `kernel` for the last launch
 * (0 = workgroup-per-LP block kernel, 1 = wavefront-per-LP kernel on term tables, 2 = the same on a dense image of A). */
int pycllp_hip_dense_kernel_kind(const pycllp_hip_dense *handle);

void pycllp_hip_dense_free(pycllp_hip_dense *handle);

/* Stand-alone batched LDL' (modified != 0: Nocedal-Wright modified LDL' with the given beta and delta) of B
 * explicit symmetric matrices A_dev [B, n, n] (row-major, only the lower triangle is read), n <= 128.
 * L_dev [B, n(n+1)/2]: packed lower triangle with unit diagonal, entry (i, j) at i(i+1)/2 + j; D_dev [B, n].
 * Replaces the reference's test kernels `ldl` / `modified_ldl` (pycllp/cl/ldl.cl:28-55, 57-107). */
int pycllp_hip_ldl(int n, long B, const double *A_dev, double *L_dev, double *D_dev, int modified, double beta,
                   double delta, void *stream);

/* Stand-alone LDL' solves of B explicit symmetric systems (numpy prototypes pycllp/ldl.py:147-281).
 * pycllp_hip_ldl_solve: x = A^-1 rhs through A = L D L' -- `solve_ldl` (ldl.py:202-239) when modified == 0 (one matrix
 *   per wavefront, factor held in registers), `forward_backward_modified_ldl` (ldl.py:242-281: Nocedal-Wright guard with
 *   the given beta and delta) when modified != 0.  A_dev [B, n, n] (lower triangle read), rhs_dev, x_dev [B, n]; n <= 128.
 * pycllp_hip_forward_backward_ldl: x = (L D L')^-1 b for given factors -- `forward_backward_ldl` (ldl.py:165-180).
 *   L_dev [B, n(n+1)/2] packed lower triangle as written by pycllp_hip_ldl (unit diagonal), D_dev, b_dev, x_dev [B, n]. */
int pycllp_hip_ldl_solve(int n, long B, const double *A_dev, const double *rhs_dev, double *x_dev, int modified,
                         double beta, double delta, void *stream);
int pycllp_hip_forward_backward_ldl(int n, long B, const double *L_dev, const double *D_dev, const double *b_dev,
                                    double *x_dev, void *stream);

/* ---- sparse shared-A path (BASELINE config 5): one LP per workgroup, A in CSR, dense packed factor in LDS ----
 * Replaces ClSparsePrimalNormalSolver (pycllp/solvers/cl.py:127-278) and the sparse_* kernels
 * (pycllp/cl/primal_normal.cl:287-375, pycllp/cl/ldl.cl:140-196,221-257,381-502,540-574,656-712).
 * init takes the CSR arrays the reference uploads (cl.py:175-178: Adata f64[nnz], Aindptr i32[m+1], Aindices i32[nnz],
 * device pointers); A' and the structure of A diag(x/z) A' are derived inside (cl.py:180-196 does this on the host).
 * Limits: m <= 128, n <= 512 (equality form).  solve has the semantics and layouts of pycllp_hip_dense_solve. */
typedef struct pycllp_hip_sparse pycllp_hip_sparse;
int pycllp_hip_sparse_max_rows(void);
int pycllp_hip_sparse_max_cols(void);
int pycllp_hip_sparse_init(int m, int n, int nnz, const double *Adata_dev, const int *Aindptr_dev,
                           const int *Aindices_dev, void *stream, pycllp_hip_sparse **handle);
int pycllp_hip_sparse_solve(pycllp_hip_sparse *handle, long B, const double *b_dev, const double *c_dev,
                            double *x_dev, double *y_dev, double *z_dev, double *pobj_dev, double *dobj_dev,
                            int *status_dev, int *iters_dev, const pycllp_hip_opts *opts, void *stream);
/* The same solve with PER-PROBLEM VALUES of A on the shared structure -- SparseMatrix.data[nproblems, nnz] of the
 * reference's container (pycllp/lp.py:16-54, 274-281), which its LP classes still refuse (lp.py:335-336).
 * Adata_dev [B, nnz]: the values of LP k in the CSR order of the arrays given to pycllp_hip_sparse_init (whose values
 * only fixed the structure).  One LP per workgroup; its values travel HBM -> LDS once per LP (8 nnz bytes, next to the
 * 16 (m + n) + 24 of b, c, x, y). */
int pycllp_hip_sparse_solve_batch(pycllp_hip_sparse *handle, long B, const double *Adata_dev, const double *b_dev,
                                  const double *c_dev, double *x_dev, double *y_dev, double *z_dev, double *pobj_dev,
                                  double *dobj_dev, int *status_dev, int *iters_dev, const pycllp_hip_opts *opts,
                                  void *stream);
/* One Newton step of the primal normal equations for B independent states with the sparse shared A: the reference's
 * stand-alone kernel sparse_solve_primal_normal (pycllp/cl/ldl.cl:656-712) as launched by its tests/test_ldl.py:276-361.
 * Arguments as pycllp_hip_dense_newton. */
int pycllp_hip_sparse_newton(pycllp_hip_sparse *handle, long B, const double *x_dev, const double *z_dev,
                             const double *y_dev, const double *b_dev, const double *c_dev, double mu, double *dy_dev,
                             int *nrefine_dev, const pycllp_hip_opts *opts, void *stream);
/* grid (workgroups), threads per workgroup, LDS bytes per workgroup and kernel (0 = workgroup-per-LP block kernel,
 * 1 = register-resident wavefront-per-LP kernel on term tables, 2 = the same on a dense image of A) of the last solve
 * launch on this handle (host values; not thread-safe). */
int pycllp_hip_sparse_launch_info(const pycllp_hip_sparse *handle, int *grid, int *block, int *lds_bytes, int *kernel);
void pycllp_hip_sparse_free(pycllp_hip_sparse *handle);

#ifdef __cplusplus
}
#endif
#endif /* PYCLLP_HIP_H */
