// wreg.h -- internal interface between ipm_dense.hip (C ABI, handles) and ipm_wreg.hip (the register-resident
// one-LP-per-wavefront kernel of the sparse shared-A path).  Not part of the public ABI.
#ifndef PYCLLP_WREG_H
#define PYCLLP_WREG_H
#include "wave_common.h"

// ---- shared between the translation units the wave kernels are compiled in (ipm_wreg.hip, twice: WREG_PART 0 / 1) ----
constexpr int MAX_NQ = 8;
constexpr int META_COFF = MAX_NQ, META_SEG = 2 * MAX_NQ, META_N = META_SEG + 16;


// Device view of the tables of one constraint matrix (built by wreg_plan_create).
struct WregTab {
    int m, n, nnz;
    int rmax, n_lev, n_term;
    int meta[META_N];     // [0..8) ELL depth of column register q, [META_COFF..) its first ELL slot, [META_SEG..) first level
                          // (index into lev) of Gram group g: the NCHUNK staging chunks of off-diagonal blocks, then the
                          // diagonal blocks; NCHUNK + 2 used -- copied to LDS
    const double* csr_val; const unsigned short* csr_col; const unsigned short* csr_ptr; const unsigned short* csr_len;
    // A by columns in ELL form over column POSITIONS: the columns are dealt to the (lane, register) positions of the
    // N-vectors sorted by length, so that each register's 64 columns are about equally long (JDS); colmap[pos] = 8 x column
    const double* ec_val; const unsigned short* ec_row; const unsigned* colmap; int ctot;   // (a byte offset; PAD_OFF for pos >= n)
    // Gram terms a_ij a_kj d_j of the strictly lower triangle of M, one record per term: weight a_ij a_kj, column position
    // of j, destination offset inside the group's staging area.  Inside a group the terms are ordered by LEVEL = rank of
    // the term inside its entry (i, k): level 0 holds the first term of every entry, level 1 the second term of the
    // entries that have one, ...; lev[] holds the item boundaries, level l of the table = items [lev[l], lev[l + 1]).
    // Destinations are distinct inside a level, so a level is one flat pass with no inner loop; level 0 stores, the
    // later levels accumulate in the same order a per-entry loop would.
    const double* t_w; const unsigned* t_cd; const int* lev;     // t_cd = column position | destination << 16
    int o_csr_val, o_ec_val, o_t_w, o_wave, o_lev, o_meta, o_csr_col, o_csr_ptr, o_csr_len, o_ec_row, o_colmap,
        o_t_cd;                                           // LDS byte offsets
    int wave_doubles, lds_bytes;
    // per-problem values of A (PA variants; SparseMatrix.data[nproblems, nnz], pycllp/lp.py:16-54): the tables above hold
    // the STRUCTURE only -- csr_val, ec_val and t_w are absent; every wavefront keeps the values of ITS LP (CSR order,
    // nnzp = nnz + 1 rounded up to even doubles, entry [nnz] = 0 for the padded slots) behind its wave area, and finds an
    // ELL slot's value through ec_src (CSR index of the slot) and a Gram term's weight as the product of the two entries
    // t_ab names (CSR index of a_ij | CSR index of a_kj << 16)
    int pa, nnzp, o_ec_src, o_t_ab;
    const unsigned short* ec_src; const unsigned* t_ab;
    // dense variant (DA): no tables, A as a row-major image [img_rows][as] of its first nd columns (the remaining n - nd
    // columns are the identity, column nd + i = e_i, or there are none), as = nd rounded up to 8, + 1
    int nd, as, img_rows, o_img, wpb;
    const double* img;
};


typedef hipError_t (*wsolve_fn)(const WregTab&, long, const double*, const double*, const double*, double*, double*, double*,
                                double*, double*, int*, int*, int*, int*, DevOpts, int, hipStream_t);
typedef hipError_t (*wnewton_fn)(const WregTab&, long, const double*, const double*, const double*, const double*,
                                 const double*, double, double*, int*, int*, DevOpts, int, hipStream_t);

struct WVariant { int mb, nq; bool da, pa; wsolve_fn solve, solve_hsd; wnewton_fn newton; };
// the dense-image variants live in the second translation unit (same source, -DWREG_PART=1), the per-problem-A variants
// in the third (-DWREG_PART=2), compiled in parallel
extern const WVariant kWVariantsDA[];
extern const int kNumWVariantsDA;
extern const WVariant kWVariantsPA[];
extern const int kNumWVariantsPA;
extern const WVariant kWVariantsPC[];      // predictor-corrector kernels of the table variants (-DWREG_PART=3)
extern const int kNumWVariantsPC;
extern const WVariant kWVariantsPCDA[];    // ... and of the dense-image variants (-DWREG_PART=4)
extern const int kNumWVariantsPCDA;
extern const WVariant kWVariantsPCPA[];    // ... and of the per-problem-A variants (-DWREG_PART=5)
extern const int kNumWVariantsPCPA;

struct WregPlan;   // host tables + device copies for one shared constraint matrix

// Builds the plan from a host CSR copy of A (m rows, n columns, equality form).  Returns 0 and *out on success,
// 1 when the register-resident kernel does not cover the problem (too many rows/columns, tables larger than LDS):
// the caller then stays on ipm_block_kernel.  A positive hipError_t is returned as (1000 + error).
// pa != 0: the plan of the per-problem-A variants (structure tables only; `val` is not read).
int wreg_plan_create(int m, int n, int nnz, const double* val, const int* ptr, const int* col, int max_lds, int pa,
                     hipStream_t st, WregPlan** out);
void wreg_plan_free(WregPlan* p);

// Solve B LPs (same argument meaning as pycllp_hip_sparse_solve).  LPs whose factorisation would have needed the
// Nocedal-Wright guard are NOT solved: their indices are appended to defer[1..] (defer[0] = count, zeroed here) and
// their status is left at -1; the caller runs them through the guarded kernel afterwards.
// a_batch: [B, nnz] values of every LP in the CSR order of the arrays the plan was built from (PA plans only, else null).
hipError_t wreg_launch_solve(WregPlan* p, long B, const double* a_batch, const double* b, const double* c, double* x, double* y, double* z,
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
int wreg_has_predcorr(const WregPlan* p);   // 1 when the plan's kernels have a PYCLLP_FLAG_PREDCORR variant (every plan of the wave kernel)
#endif
