// wreg.h -- internal interface between ipm_dense.hip (C ABI, handles) and ipm_wreg.hip (the register-resident
// one-LP-per-wavefront kernel of the sparse shared-A path).  Not part of the public ABI.
#ifndef PYCLLP_WREG_H
#define PYCLLP_WREG_H
#include "wave_common.h"

struct WregPlan;   // host tables + device copies for one shared constraint matrix

// Builds the plan from a host CSR copy of A (m rows, n columns, equality form).  Returns 0 and *out on success,
// 1 when the register-resident kernel does not cover the problem (too many rows/columns, tables larger than LDS):
// the caller then stays on ipm_block_kernel.  A positive hipError_t is returned as (1000 + error).
int wreg_plan_create(int m, int n, int nnz, const double* val, const int* ptr, const int* col, int max_lds,
                     hipStream_t st, WregPlan** out);
void wreg_plan_free(WregPlan* p);

// Solve B LPs (same argument meaning as pycllp_hip_sparse_solve).  LPs whose factorisation would have needed the
// Nocedal-Wright guard are NOT solved: their indices are appended to defer[1..] (defer[0] = count, zeroed here) and
// their status is left at -1; the caller runs them through the guarded kernel afterwards.
hipError_t wreg_launch_solve(WregPlan* p, long B, const double* b, const double* c, double* x, double* y, double* z,
                             double* pobj, double* dobj, int* status, int* iters, int* qhead, int* defer, DevOpts o,
                             int num_cu, hipStream_t st, int* grid_out);

// One Newton step for B states (semantics of pycllp_hip_dense_newton).  guard_hit[0] is set to 1 if any state would
// have needed the guard (the stand-alone step then simply ran without it).
hipError_t wreg_launch_newton(WregPlan* p, long B, const double* x, const double* z, const double* y, const double* b,
                              const double* c, double mu, double* dy, int* nref, DevOpts o, int num_cu, hipStream_t st);

// out = (L D L')^-1 rhs with L D L' = A for B explicit dense symmetric matrices A [B, n, n] (lower triangle read),
// n <= 128, pivots floored at floor_ (0: plain LDL'); one matrix per wavefront, factor held in registers.
hipError_t wreg_launch_ldl_solve(int n, long B, const double* A, const double* rhs, double* out, double floor_,
                                 int num_cu, hipStream_t st);
int wreg_lds_bytes(const WregPlan* p);
int wreg_block_threads(const WregPlan* p);   // 64 x waves per workgroup
int wreg_variant(const WregPlan* p);         // 1 = term tables, 2 = dense image
#endif
