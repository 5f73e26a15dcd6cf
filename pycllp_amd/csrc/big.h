// big.h -- internal interface between ipm_dense.hip (C ABI, handles) and ipm_big.hip (the workgroup-per-LP kernel for LPs
// beyond the register-resident kernels: 128 < m <= 256 rows or 512 < n <= 1280 columns).  Not part of the public ABI.
#ifndef PYCLLP_BIG_H
#define PYCLLP_BIG_H
#include "wave_common.h"

constexpr int BIG_MAX_M = 256;    // rows (16 x 16 blocks: MB <= 16)
constexpr int BIG_MAX_N = 1280;   // columns of the equality form (5 per thread)

struct BigPlan;   // host tables + device copies for one shared constraint matrix

// Builds the plan from a host CSR copy of A (m rows, n columns, equality form).  0 and *out on success, 1 when the problem is
// outside the kernel's limits; a positive hipError_t is returned as (1000 + error).
int big_plan_create(int m, int n, int nnz, const double* val, const int* ptr, const int* col, int max_lds, hipStream_t st,
                    BigPlan** out);
void big_plan_free(BigPlan* p);

// Solve B LPs (argument meaning of pycllp_hip_sparse_solve).
hipError_t big_launch_solve(BigPlan* p, long B, const double* b, const double* c, double* x, double* y, double* z,
                            double* pobj, double* dobj, int* status, int* iters, int* qhead, DevOpts o, int num_cu,
                            hipStream_t st, int* grid_out);
// One Newton step for B states (semantics of pycllp_hip_dense_newton).
hipError_t big_launch_newton(BigPlan* p, long B, const double* x, const double* z, const double* y, const double* b,
                             const double* c, double mu, double* dy, int* nref, int* qhead, DevOpts o, int num_cu,
                             hipStream_t st);
int big_lds_bytes(const BigPlan* p);
int big_dense_mode(const BigPlan* p);   // 1: Gram product on the matrix cores from a dense image; 0: term list
#endif
