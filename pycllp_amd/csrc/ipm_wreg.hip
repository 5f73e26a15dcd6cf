// ipm_wreg.hip -- sparse shared-A path, third generation: ONE LP PER WAVEFRONT, the normal-equations matrix and its
// LDL' factor held in REGISTERS for the whole Newton step, no workgroup barrier anywhere in the solve.
//
// Replaces (as ipm_block.inc does, which stays as the general/guarded path) the reference's sparse twins: host
// ClSparsePrimalNormalSolver (pycllp/solvers/cl.py:127-278) and the kernels sparse_standard_primal_normal
// (pycllp/cl/primal_normal.cl:287-375), sparse_primal_normal_step (:158-198), sparse_*_infeasibility (:50-74, :96-120),
// sparse_AXZAt_ij/_ii (pycllp/cl/ldl.cl:140-196), sparse_primal_normal_rhs_i (:221-257), sparse_factor_primal_normal
// (:381-502), sparse_forward_backward_primal_normal (:540-574), sparse_solve_primal_normal (:656-712).
//
// Why: with M (m = 128: 66 KB packed) in LDS only two LPs fit a CU, so ipm_block_kernel has to spread ONE LP over
// four wavefronts and pays for it in barriers, in waves idling during the serial pivot chain and the triangular solves
// (44 % + 26 % of its run time), and in SIMDs idling.  The register file of a CU is 512 KB -- three times its LDS.  Here
// each of the 4 SIMDs of a CU runs one wavefront with the full 512-register budget that owns one LP:
//   * the factor is kept as U = L' in 16 x 16 blocks U[K][I] (K < I) in the ACCUMULATOR layout of
//     v_mfma_f64_16x16x4_f64 (register r of lane l holds element [4r + (l >> 4)][l & 15]).  That layout is, unchanged, the
//     B operand of the block and the A operand of its transpose, so both the panel solve  Y_KI = L_KK^-1 M_KI  and the
//     trailing update  U_JI -= Y_KJ' U_KI  are MFMAs straight on the resident registers -- no operand ever moves;
//   * only the OFF-DIAGONAL blocks live in registers (m = 128: 28 blocks = 224 of the 256 accumulator registers).  A
//     diagonal block is formed when its turn comes (left-looking): its Schur update on the matrix cores into a 2 KB LDS
//     tile, plus the original block, which the Gram pass left in the block's W slot; it is read in "lane = row" form and
//     factored by a 16-step chain of fused 64-bit DPP FMAs (v_fmac_f64_dpp row_newbcast) -- every 16-lane row of the wave
//     redundantly, so nothing is broadcast across rows -- and its inverse W_K = L_KK^-1 is formed directly in the MFMA
//     A-operand layout (quad q owns columns q, q+4, ...) for the panel; W_K is also what the triangular solves use, from a
//     packed copy in LDS (the slot of the original block);
//   * M = A diag(x/z) A' is assembled from flat term records built once at init (deterministic, atomic-free, no inner
//     loop: first terms of all entries, then triples of further terms), scattered through a 16 KB staging area 8 blocks
//     at a time and loaded in the accumulator layout;
//   * A x and A'u use compact-CSR / JDS-ELL copies of A in LDS; N-vectors live in registers (lane = column) while they
//     are worked on and in LDS across the factorisation and the loop's back edge, m-vectors in a per-wave LDS area;
//   * LDS reads come in inline-asm batches (N reads, one s_waitcnt) and every lane-dependent address is derived from
//     three pinned values where it is used (WReg::pin): no scratch traffic inside the iteration loop;
//   * the triangular solves are 16-row block steps: 4 FMAs per off-diagonal block, quad/row reductions by
//     v_permlane swaps and DPP.
// Variants (MB 16-row blocks, NQ 64-column N-vector registers; term tables or dense image): see kWVariantsTab / kWVariantsDA.
// The Nocedal-Wright guard (ldl.cl:487) is not applied here: the sweep records whether it WOULD have bitten and such an
// LP (never seen on a positive definite M) is deferred to ipm_block_kernel, which applies it exactly.
// Semantics = oracle/ipm_dense_ref.c (ipm_one_path / hsd_one_raw), like every other kernel of this library.
// tools/wreg_sim.py is a lane-level numpy model of the layouts used below.
#ifndef WREG_PART
#define WREG_PART 0     // 0: table variants + host code; 1: the dense-image variants only (second translation unit)
#endif
#include "wreg.h"

namespace {

// An inline-asm operand of the accumulator register class: with one in the kernel the compiler keeps the AGPR form of
// the MFMAs (C/D -- the resident U blocks -- in a0..a255, A/B read from either file).
#define USE_AGPR_FORM() do { int agpr_hint_; asm volatile("; accumulator file in use" : "=a"(agpr_hint_)); } while (0)

typedef double double2_t __attribute__((ext_vector_type(2)));

// A finished panel block.  Rounds 1-2 PARKED it in eight accumulator registers through inline asm (v_accvgpr_write with an
// accumulator-class output): the allocator of the first versions of this kernel kept panel results in architectural VGPRs and,
// out of those, spilled them to scratch -- with one wavefront per SIMD every reload a fully exposed memory round trip (345 k
// cycles per iteration, 42 % of them in the pivot chains waiting for reloads).  Round 3, with the rest of the kernel no longer
// under that pressure: the block is simply the MFMA's own result.  It stays where the matrix pipe wrote it, later MFMAs take it
// as their B operand from the accumulator file directly, and the ~700 v_accvgpr_read / _write per iteration that moved every
// block out of the accumulators and back are gone (PYCLLP_PARK_ASM = 1 restores the asm form): 378.5 -> 391.5 k LPs/s.
#ifndef PYCLLP_WINV_FUSED
#define PYCLLP_WINV_FUSED 1
#endif
#ifndef PYCLLP_PARK_ASM
#define PYCLLP_PARK_ASM 0
#endif
#if PYCLLP_PARK_ASM
struct PBlk { int h[8]; };
__device__ __forceinline__ void park(PBlk& p, const double4_t& v) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int lo = __double2loint(v[r]), hi = __double2hiint(v[r]);
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(p.h[2 * r]) : "v"(lo));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(p.h[2 * r + 1]) : "v"(hi));
    }
}
__device__ __forceinline__ double unpark(const PBlk& p, int r) { return __hiloint2double(p.h[2 * r + 1], p.h[2 * r]); }
#else
struct PBlk { double4_t d; };
__device__ __forceinline__ void park(PBlk& p, const double4_t& v) { p.d = v; }
__device__ __forceinline__ double unpark(const PBlk& p, int r) { return p.d[r]; }
#endif

// stage proper: N-vector staging; during factor/solve t, x and z parked at 0, NP, 2 NP and the stride-17 tile of the current
// diagonal block behind them
constexpr int TILE_D = 272;
__host__ __device__ constexpr int tile_off(int NQ) { return 192 * NQ > 768 ? 192 * NQ : 768; }
__host__ __device__ constexpr int stage_d(int NQ) { return tile_off(NQ) + TILE_D; }
constexpr int HB = 8;            // 16 x 16 blocks per Gram staging chunk: the stage and, behind it, the still unused W area (16 KB)
constexpr int WL = 144;          // doubles per diagonal-block slot: first the ORIGINAL diagonal block of M (lower triangle with
                                 // diagonal, row i at i(i+1)/2: 136), from stage K on W_K (strictly lower triangle, row i at i(i-1)/2)
template <int MB>
struct WGeo {
    static constexpr int MP = 16 * MB;
    static constexpr int MR = (MP + 63) / 64;    // m-vector registers per lane in "lane = row" form
    static constexpr int MPL = 64 * MR;
    static constexpr int NBLK = MB * (MB - 1) / 2;
    // off-diagonal block (K, I), K < I, of U = L'
    __host__ __device__ static constexpr int bix(int K, int I) { return K * MB - K * (K + 1) / 2 + (I - K - 1); }
    // Gram staging chunks: the off-diagonal blocks in bix order, HB at a time; chunk NCHUNK = the diagonal blocks
    static constexpr int NCHUNK = (NBLK + HB - 1) / HB;
    // the diagonal blocks' entries ride with the last chunk when its blocks end in front of the W area (where they go)
    __host__ __device__ static constexpr bool MERGE_DIAG(int NQ) { return NBLK > 0 && (NBLK - HB * (NCHUNK - 1)) * 256 <= stage_d(NQ); }
    static constexpr int WAVE_D(int NQ) { return stage_d(NQ) + 64 * NQ + 5 * MP + MB * WL; }   // per-wave LDS doubles
};

// ---- per-LP vectors in global memory through buffer descriptors ------------------------------------------------
// descriptor (4 SGPRs) of one LP's row of a [B][len] array + a 32-bit byte offset per lane: no 64-bit per-lane pointers
// (which the compiler hoists out of the iteration loop and spills), and offsets past the row read 0 / drop the store, so
// the padded positions of the N-vectors (offset PAD_OFF) need neither a branch nor a select
typedef int int2_t __attribute__((ext_vector_type(2)));
constexpr unsigned PAD_OFF = 0x7ffffff0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const double* row, int len) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, row ? 8 * len : 0, 0x00020000);
}
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const int2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    return __hiloint2double(v.y, v.x);
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, unsigned off, double d) {
    int2_t v; v.x = __double2loint(d); v.y = __double2hiint(d);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, 0);
}

// ---- batched LDS reads -----------------------------------------------------------------------------------------
// One wavefront alone on its SIMD hides no latency by itself, and in this kernel's register-starved regions the compiler
// schedules every LDS read right in front of its use with its own s_waitcnt (the sched_group_barrier hints are not
// honoured there): a run of N reads then costs N round trips.  These helpers issue the whole run and wait ONCE.
#define LDS_RD2_(i) "ds_read2_b64 %" #i ", %24 offset0:2*" #i " offset1:2*" #i "+1\n\t"
#define LDS_RD1_(i, k) "ds_read_b64 %" #i ", %25 offset:%26+8*" #k "\n\t"
// one round trip for a row of the diagonal-block tile and of the original block: t[0..8) <- 16 consecutive doubles at at
// (8-byte aligned), r[0..16) <- 16 consecutive doubles at ar + OFF (the slot offset folded into the instruction)
template <int OFF>
__device__ __forceinline__ void lds_tile_and_raw(unsigned at, unsigned ar, double2_t (&t)[8], double (&r)[16]) {
    asm volatile(LDS_RD2_(0) LDS_RD2_(1) LDS_RD2_(2) LDS_RD2_(3) LDS_RD2_(4) LDS_RD2_(5) LDS_RD2_(6) LDS_RD2_(7)
                 LDS_RD1_(8, 0) LDS_RD1_(9, 1) LDS_RD1_(10, 2) LDS_RD1_(11, 3) LDS_RD1_(12, 4) LDS_RD1_(13, 5) LDS_RD1_(14, 6) LDS_RD1_(15, 7)
                 LDS_RD1_(16, 8) LDS_RD1_(17, 9) LDS_RD1_(18, 10) LDS_RD1_(19, 11) LDS_RD1_(20, 12) LDS_RD1_(21, 13) LDS_RD1_(22, 14) LDS_RD1_(23, 15)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
                   "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
                   "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
                 : "v"(at), "v"(ar), "n"(OFF) : "memory");
}
#undef LDS_RD2_
#undef LDS_RD1_
// eight ELL slots: o[k] = *(double*)(ad + 512 k), r[k] = *(unsigned short*)(au + 128 k)
__device__ __forceinline__ void lds_ell8(unsigned ad, unsigned au, double (&o)[8], unsigned (&r)[8]) {
    asm volatile("ds_read_b64 %0, %16\n\tds_read_b64 %1, %16 offset:512\n\tds_read_b64 %2, %16 offset:1024\n\tds_read_b64 %3, %16 offset:1536\n\t"
                 "ds_read_b64 %4, %16 offset:2048\n\tds_read_b64 %5, %16 offset:2560\n\tds_read_b64 %6, %16 offset:3072\n\tds_read_b64 %7, %16 offset:3584\n\t"
                 "ds_read_u16 %8, %17\n\tds_read_u16 %9, %17 offset:128\n\tds_read_u16 %10, %17 offset:256\n\tds_read_u16 %11, %17 offset:384\n\t"
                 "ds_read_u16 %12, %17 offset:512\n\tds_read_u16 %13, %17 offset:640\n\tds_read_u16 %14, %17 offset:768\n\tds_read_u16 %15, %17 offset:896\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                   "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                 : "v"(ad), "v"(au) : "memory");
}
// four consecutive CSR slots of two rows: v[0..2) <- four doubles at av0, v[2..4) <- at av1; c[0..4) <- four u16 at ac0, c[4..8) <- at ac1
__device__ __forceinline__ void lds_rows4x2(unsigned av0, unsigned ac0, unsigned av1, unsigned ac1, double2_t (&v)[4], unsigned (&c)[8]) {
    asm volatile("ds_read2_b64 %0, %12 offset1:1\n\tds_read2_b64 %1, %12 offset0:2 offset1:3\n\t"
                 "ds_read2_b64 %2, %14 offset1:1\n\tds_read2_b64 %3, %14 offset0:2 offset1:3\n\t"
                 "ds_read_u16 %4, %13\n\tds_read_u16 %5, %13 offset:2\n\tds_read_u16 %6, %13 offset:4\n\tds_read_u16 %7, %13 offset:6\n\t"
                 "ds_read_u16 %8, %15\n\tds_read_u16 %9, %15 offset:2\n\tds_read_u16 %10, %15 offset:4\n\tds_read_u16 %11, %15 offset:6\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]),
                   "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]), "=&v"(c[7])
                 : "v"(av0), "v"(ac0), "v"(av1), "v"(ac1) : "memory");
}
__device__ __forceinline__ void lds_gather4_u16(const unsigned (&a)[4], unsigned (&o)[4]) {
    asm volatile("ds_read_u16 %0, %4\n\tds_read_u16 %1, %5\n\tds_read_u16 %2, %6\n\tds_read_u16 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
}
__device__ __forceinline__ void lds_gather8(const unsigned (&a)[8], double (&o)[8]) {
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %9\n\tds_read_b64 %2, %10\n\tds_read_b64 %3, %11\n\t"
                 "ds_read_b64 %4, %12\n\tds_read_b64 %5, %13\n\tds_read_b64 %6, %14\n\tds_read_b64 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
}
// o[k] = *(double*)a[k], u[k] = *(double*)(a[k] + DELTA)
template <int DELTA>
__device__ __forceinline__ void lds_gather8_pair(const unsigned (&a)[8], double (&o)[8], double (&u)[8]) {
    asm volatile("ds_read_b64 %0, %16\n\tds_read_b64 %1, %17\n\tds_read_b64 %2, %18\n\tds_read_b64 %3, %19\n\t"
                 "ds_read_b64 %4, %20\n\tds_read_b64 %5, %21\n\tds_read_b64 %6, %22\n\tds_read_b64 %7, %23\n\t"
                 "ds_read_b64 %8, %16 offset:%24\n\tds_read_b64 %9, %17 offset:%24\n\tds_read_b64 %10, %18 offset:%24\n\tds_read_b64 %11, %19 offset:%24\n\t"
                 "ds_read_b64 %12, %20 offset:%24\n\tds_read_b64 %13, %21 offset:%24\n\tds_read_b64 %14, %22 offset:%24\n\tds_read_b64 %15, %23 offset:%24\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                   "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "n"(DELTA) : "memory");
}
// Gram term records: o[k] = *(double*)a[k], c[k] = *(unsigned*)b[k], k < 6 / 4
__device__ __forceinline__ void lds_gather6_d_u(const unsigned (&a)[6], const unsigned (&b)[6], double (&o)[6], unsigned (&c)[6]) {
    asm volatile("ds_read_b64 %0, %12\n\tds_read_b64 %1, %13\n\tds_read_b64 %2, %14\n\tds_read_b64 %3, %15\n\tds_read_b64 %4, %16\n\tds_read_b64 %5, %17\n\t"
                 "ds_read_b32 %6, %18\n\tds_read_b32 %7, %19\n\tds_read_b32 %8, %20\n\tds_read_b32 %9, %21\n\tds_read_b32 %10, %22\n\tds_read_b32 %11, %23\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]),
                   "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]),
                   "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]) : "memory");
}
__device__ __forceinline__ void lds_gather6(const unsigned (&a)[6], double (&o)[6]) {
    asm volatile("ds_read_b64 %0, %6\n\tds_read_b64 %1, %7\n\tds_read_b64 %2, %8\n\tds_read_b64 %3, %9\n\tds_read_b64 %4, %10\n\tds_read_b64 %5, %11\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]) : "memory");
}
__device__ __forceinline__ void lds_gather4_d_u(const unsigned (&a)[4], const unsigned (&b)[4], double (&o)[4], unsigned (&c)[4]) {
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %9\n\tds_read_b64 %2, %10\n\tds_read_b64 %3, %11\n\t"
                 "ds_read_b32 %4, %12\n\tds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "memory");
}
__device__ __forceinline__ void lds_gather4(const unsigned (&a)[4], double (&o)[4]) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
}
// three consecutive term records: w[k] = ((double*)aw)[k], c[k] = ((unsigned*)ac)[k]
__device__ __forceinline__ void lds_terms3(unsigned aw, unsigned ac, double (&w)[3], unsigned (&c)[3]) {
    asm volatile("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:8\n\tds_read_b64 %2, %6 offset:16\n\t"
                 "ds_read_b32 %3, %7\n\tds_read_b32 %4, %7 offset:4\n\tds_read_b32 %5, %7 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]) : "v"(aw), "v"(ac) : "memory");
}

// ---- per-problem values of A (PA variants): one more level of indirection, same batching ---------------------------
// eight ELL slots of the structure tables: s[k] = *(unsigned short*)(as + 128 k) (CSR index of the slot's value),
// r[k] = *(unsigned short*)(ar + 128 k) (its row)
__device__ __forceinline__ void lds_ell8_uu(unsigned as, unsigned ar, unsigned (&s)[8], unsigned (&r)[8]) {
    asm volatile("ds_read_u16 %0, %16\n\tds_read_u16 %1, %16 offset:128\n\tds_read_u16 %2, %16 offset:256\n\tds_read_u16 %3, %16 offset:384\n\t"
                 "ds_read_u16 %4, %16 offset:512\n\tds_read_u16 %5, %16 offset:640\n\tds_read_u16 %6, %16 offset:768\n\tds_read_u16 %7, %16 offset:896\n\t"
                 "ds_read_u16 %8, %17\n\tds_read_u16 %9, %17 offset:128\n\tds_read_u16 %10, %17 offset:256\n\tds_read_u16 %11, %17 offset:384\n\t"
                 "ds_read_u16 %12, %17 offset:512\n\tds_read_u16 %13, %17 offset:640\n\tds_read_u16 %14, %17 offset:768\n\tds_read_u16 %15, %17 offset:896\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(s[4]), "=&v"(s[5]), "=&v"(s[6]), "=&v"(s[7]),
                   "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                 : "v"(as), "v"(ar) : "memory");
}
// twelve gathers, one wait: o[k] = *(double*)a[k]
__device__ __forceinline__ void lds_gather12(const unsigned (&a)[12], double (&o)[12]) {
    asm volatile("ds_read_b64 %0, %12\n\tds_read_b64 %1, %13\n\tds_read_b64 %2, %14\n\tds_read_b64 %3, %15\n\t"
                 "ds_read_b64 %4, %16\n\tds_read_b64 %5, %17\n\tds_read_b64 %6, %18\n\tds_read_b64 %7, %19\n\t"
                 "ds_read_b64 %8, %20\n\tds_read_b64 %9, %21\n\tds_read_b64 %10, %22\n\tds_read_b64 %11, %23\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                   "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
                   "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]) : "memory");
}
__device__ __forceinline__ void lds_gather10(const unsigned (&a)[10], double (&o)[10]) {
    asm volatile("ds_read_b64 %0, %10\n\tds_read_b64 %1, %11\n\tds_read_b64 %2, %12\n\tds_read_b64 %3, %13\n\t"
                 "ds_read_b64 %4, %14\n\tds_read_b64 %5, %15\n\tds_read_b64 %6, %16\n\tds_read_b64 %7, %17\n\t"
                 "ds_read_b64 %8, %18\n\tds_read_b64 %9, %19\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                   "=&v"(o[8]), "=&v"(o[9])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
                   "v"(a[8]), "v"(a[9]) : "memory");
}
// term records of the structure tables: c[k] = *(unsigned*)a[k], d[k] = *(unsigned*)b[k], k < 4
__device__ __forceinline__ void lds_gather4_u_u(const unsigned (&a)[4], const unsigned (&b)[4], unsigned (&c)[4], unsigned (&d)[4]) {
    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %9\n\tds_read_b32 %2, %10\n\tds_read_b32 %3, %11\n\t"
                 "ds_read_b32 %4, %12\n\tds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "memory");
}
// three consecutive records of two u32 tables: c[k] = ((unsigned*)aa)[k], d[k] = ((unsigned*)ab)[k]
__device__ __forceinline__ void lds_terms3_uu(unsigned aa, unsigned ab, unsigned (&c)[3], unsigned (&d)[3]) {
    asm volatile("ds_read_b32 %0, %6\n\tds_read_b32 %1, %6 offset:4\n\tds_read_b32 %2, %6 offset:8\n\t"
                 "ds_read_b32 %3, %7\n\tds_read_b32 %4, %7 offset:4\n\tds_read_b32 %5, %7 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]) : "v"(aa), "v"(ab) : "memory");
}

// ---- the per-wave machinery ------------------------------------------------------------------------------------
template <int MB, int NQ, bool DA = false, bool PA = false>
struct WReg {
    static_assert(!(DA && PA), "per-problem values exist on the table variants only");
    using G = WGeo<MB>;
    static constexpr int MP = G::MP, MR = G::MR, MPL = G::MPL, NP = 64 * NQ, STAGE_D = stage_d(NQ), TILE_OFF = tile_off(NQ);

    // Off-diagonal blocks [bix(K, I)], K < I.  Life of a block: gram() parks the original M_KI in P; the trailing update
    // of stage_() 0 takes it out as an MFMA accumulator (U) where it stays through the following stages' updates; panel K
    // turns it into Y_KI = D_K L_IK' and parks that in P for the rest of the Newton step.
    double4_t U[G::NBLK > 0 ? G::NBLK : 1];
    PBlk P[G::NBLK > 0 ? G::NBLK : 1];
    // LDS: shared tables (A by rows and by columns in compact form, Gram entries/terms)
    const double* csr_val; const unsigned short* csr_col; const unsigned short* csr_ptr; const unsigned short* csr_len;
    const double* ec_val; const unsigned short* ec_row; const unsigned* colmap;
    const double* t_w; const unsigned* t_cd; const int* lev;
    const int* meta;
    // PA: structure-only tables (see WregTab); csr_val then points at THIS WAVE's copy of its LP's values, cvl_()
    const unsigned short* ec_src; const unsigned* t_ab;
    // DA: the dense image (LDS), its row stride and the number of dense columns; n_sl = n - nd identity columns behind them
    const double* img; int nd, AS, imgR;
    // LDS: this wave's area, every array at a COMPILE-TIME offset from the one base pointer W0 -- so that the address
    // arithmetic of all of them folds into a handful of lane-dependent bases plus immediate offsets (as separate
    // run-time pointers every (array, index pattern) pair costs a VGPR for the whole kernel)
    double* W0;
    __device__ __forceinline__ double* stage_() const { return W0; }                            // [STAGE_D] N-vector staging (vx = stage_()[0..NP)), parked x / z, tile; with wl_(): the Gram staging area
    __device__ __forceinline__ double* wl_() const { return W0 + STAGE_D; }                     // [MB][WL] diagonal-block slots (see WL); with the stage in front of it: the Gram staging area
    __device__ __forceinline__ double* vd_() const { return W0 + STAGE_D + MB * WL; }           // [NP] d = x/z
    __device__ __forceinline__ double* ys_() const { return W0 + STAGE_D + MB * WL + NP; }                // [MP] y
    __device__ __forceinline__ double* bs_() const { return W0 + STAGE_D + MB * WL + NP + MP; }           // [MP] b
    __device__ __forceinline__ double* um_() const { return W0 + STAGE_D + MB * WL + NP + 2 * MP; }       // [MP] solve vector in/out
    __device__ __forceinline__ double* rdv_() const { return W0 + STAGE_D + MB * WL + NP + 3 * MP; }      // [MP] 1/D
    __device__ __forceinline__ double* flr_() const { return W0 + STAGE_D + MB * WL + NP + 4 * MP; }      // [MP] per-column pivot floors (HSD)
    __device__ __forceinline__ double* cvl_() const { return W0 + G::WAVE_D(NQ); }                         // PA: [nnzp] this LP's values of A, CSR order
    mutable int lane, q, c16;
    int m, n, rmax;
    // Every lane-dependent LDS address in this kernel is `lane`, `q` or `c16` times something plus a constant.  Left alone the
    // compiler computes each of them once, outside the iteration loop, and then has dozens of kernel-lifetime address
    // registers to spill; pin() makes the three values opaque at the point of the call, so that what is derived from them
    // below is recomputed there (a VALU instruction or two) and dies after its use.
    // byte offset inside an LP's row of the column at position lane + 64 qq of the N-vectors (PAD_OFF: padded position)
    __device__ __forceinline__ unsigned coff(int qq) const {
        if constexpr (DA) { const int p = lane + 64 * qq; return p < n ? 8u * (unsigned)p : PAD_OFF; }
        else return colmap[lane + 64 * qq];
    }
    __device__ __forceinline__ void pin() const { asm volatile("" : "+v"(lane), "+v"(q), "+v"(c16)); }

    // out_q = (A'u)_j for the column at position lane + 64 q (see colmap), u in LDS.  ELL: slot t of register q sits at
    // (coff_q + t) 64 + lane -- an immediate offset from one lane-dependent base; padded slots hold value 0, row 0.
    __device__ __forceinline__ void At(const double* u, double (&out)[NQ]) const {
        pin();
        if constexpr (DA) {
            // dense image: column p of the dense part is one image column (lanes read consecutive entries of a row: no bank
            // conflict), u comes as broadcast reads; the identity columns behind them pick their own u_i
            const unsigned ub = lds_addr(u), ib = lds_addr(img);
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                const int p = lane + 64 * qq;
                double a0 = 0.0, a1 = 0.0;
                if (64 * qq < nd) {
                    const unsigned cb = ib + 8 * (unsigned)((p < nd) ? p : 0);
                    for (int i0 = 0; i0 < imgR; i0 += 8) {
                        unsigned ga[8]; double av[8], uv[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) ga[k] = cb + 8 * (unsigned)((i0 + k) * AS);
                        lds_gather8(ga, av);
                        lds_run8<0, 8>(ub + 8 * i0, uv);
#pragma unroll
                        for (int k = 0; k < 8; k += 2) { a0 = fma(av[k], uv[k], a0); a1 = fma(av[k + 1], uv[k + 1], a1); }
                    }
                }
                const bool sl = p >= nd && p < n;
                const double us = u[sl ? p - nd : 0];
                out[qq] = (p < nd) ? a0 + a1 : (sl ? us : 0.0);
            }
            return;
        }
        if constexpr (PA) {
            // the slot's value through its CSR index into this wave's copy of the LP's values: one more gather per round
            const unsigned ub = lds_addr(u), sb = lds_addr(ec_src) + 2 * lane, rb = lds_addr(ec_row) + 2 * lane, cvb = lds_addr(csr_val);
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                const int cm = __builtin_amdgcn_readfirstlane(meta[qq]);
                const int cof = __builtin_amdgcn_readfirstlane(meta[META_COFF + qq]);
                double a0 = 0.0, a1 = 0.0;
                for (int t0 = 0; t0 < cm; t0 += 8) {
                    double av[8], uv[8]; unsigned sr[8], rw[8], va[8], ua[8];
                    lds_ell8_uu(sb + 128 * (cof + t0), rb + 128 * (cof + t0), sr, rw);
#pragma unroll
                    for (int k = 0; k < 8; k++) { va[k] = cvb + 8 * sr[k]; ua[k] = ub + 8 * ((t0 + k < cm) ? rw[k] : 0u); }
                    lds_gather8(va, av);
                    lds_gather8(ua, uv);
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        a0 = fma((t0 + k < cm) ? av[k] : 0.0, uv[k], a0);
                        a1 = fma((t0 + k + 1 < cm) ? av[k + 1] : 0.0, uv[k + 1], a1);
                    }
                }
                out[qq] = a0 + a1;
            }
            return;
        }
        const unsigned ub = lds_addr(u), vb = lds_addr(ec_val) + 8 * lane, rb = lds_addr(ec_row) + 2 * lane;
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) {
            const int cm = __builtin_amdgcn_readfirstlane(meta[qq]);
            const int cof = __builtin_amdgcn_readfirstlane(meta[META_COFF + qq]);
            double a0 = 0.0, a1 = 0.0;
            for (int t0 = 0; t0 < cm; t0 += 8) {          // eight slots per round trip; slots >= cm belong to the next register: masked
                double av[8], uv[8]; unsigned rw[8], ua[8];
                lds_ell8(vb + 512 * (cof + t0), rb + 128 * (cof + t0), av, rw);
#pragma unroll
                for (int k = 0; k < 8; k++) ua[k] = ub + 8 * ((t0 + k < cm) ? rw[k] : 0u);
                lds_gather8(ua, uv);
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    a0 = fma((t0 + k < cm) ? av[k] : 0.0, uv[k], a0);
                    a1 = fma((t0 + k + 1 < cm) ? av[k + 1] : 0.0, uv[k + 1], a1);
                }
            }
            out[qq] = a0 + a1;
        }
    }
    // (A v)_i for the rows i = lane + 64 r2 of this lane, v staged in LDS; with DIAG also diag(A diag(d) A')_i (d in vd_();
    // padded rows get 1: identity rows of M) from the same pass over the row.  Four slots of both rows per round trip.
    template <bool DIAG>
    __device__ __forceinline__ void Arow(const double* v, double (&out)[MR], double (&md)[MR]) const {
        static_assert(MR == 1 || MR == 2, "one or two rows per lane");
        pin();
        if constexpr (DA) {
            // row i of the image (odd stride: conflict free across the lanes) against broadcast reads of v (and d); the
            // identity column of row i adds v[nd + i] (and d[nd + i] to the diagonal)
            const unsigned vb = lds_addr(v), ib = lds_addr(img);
            const bool has_sl = n > nd;
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int i = lane + 64 * r2;
                const bool rok = i < m;
                const unsigned rb = ib + 8 * (unsigned)((rok ? i : 0) * AS);
                double o0 = 0.0, o1 = 0.0, m0 = 0.0, m1 = 0.0;
                for (int j0 = 0; j0 + 1 < AS; j0 += 8) {
                    double a[8], xv[8], dv[8];
                    lds_run8<0, 8>(rb + 8 * j0, a);
                    lds_run8<0, 8>(vb + 8 * j0, xv);
                    if (DIAG) lds_run8<8 * (STAGE_D + MB * WL), 8>(vb + 8 * j0, dv);
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        o0 = fma(a[k], xv[k], o0); o1 = fma(a[k + 1], xv[k + 1], o1);
                        if (DIAG) { m0 = fma(a[k] * a[k], dv[k], m0); m1 = fma(a[k + 1] * a[k + 1], dv[k + 1], m1); }
                    }
                }
                const int js = (rok && has_sl) ? nd + i : 0;
                const double vs = v[js], ds = DIAG ? vd_()[js] : 0.0;
                out[r2] = rok ? (o0 + o1) + (has_sl ? vs : 0.0) : 0.0;
                md[r2] = DIAG ? (rok ? (m0 + m1) + (has_sl ? ds : 0.0) : 1.0) : 0.0;
            }
            return;
        }
        unsigned pl[4];
        {
            // (MR == 1: the second row slot is a copy of the first with length 0 -- its loads are masked)
            const int l2 = lane + (MR == 2 ? 64 : 0);
            const unsigned pa[4] = {lds_addr(csr_ptr + lane), lds_addr(csr_ptr + l2), lds_addr(csr_len + lane), lds_addr(csr_len + l2)};
            lds_gather4_u16(pa, pl);
            if (MR == 1) pl[3] = 0;
        }
        const unsigned vb = lds_addr(v), cvb = lds_addr(csr_val), ccb = lds_addr(csr_col);
        double o0[2] = {0.0, 0.0}, o1[2] = {0.0, 0.0}, m0[2] = {0.0, 0.0}, m1[2] = {0.0, 0.0};
        for (int t0 = 0; t0 < rmax; t0 += 4) {
            double2_t av[4]; unsigned cc[8], ga[8]; double a[8], xv[8], dv[8];
            lds_rows4x2(cvb + 8 * (pl[0] + t0), ccb + 2 * (pl[0] + t0), cvb + 8 * (pl[1] + t0), ccb + 2 * (pl[1] + t0), av, cc);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool on = t0 + (k & 3) < (int)pl[2 + (k >> 2)];
                a[k] = on ? av[k >> 1][k & 1] : 0.0;
                ga[k] = vb + 8 * (on ? cc[k] : 0u);
            }
            if (DIAG) lds_gather8_pair<8 * (STAGE_D + MB * WL)>(ga, xv, dv);     // vd_() sits STAGE_D + MB WL doubles behind the stage
            else lds_gather8(ga, xv);
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                o0[r2] = fma(a[4 * r2], xv[4 * r2], o0[r2]); o1[r2] = fma(a[4 * r2 + 1], xv[4 * r2 + 1], o1[r2]);
                o0[r2] = fma(a[4 * r2 + 2], xv[4 * r2 + 2], o0[r2]); o1[r2] = fma(a[4 * r2 + 3], xv[4 * r2 + 3], o1[r2]);
                if (DIAG) {
                    m0[r2] = fma(a[4 * r2] * a[4 * r2], dv[4 * r2], m0[r2]); m1[r2] = fma(a[4 * r2 + 1] * a[4 * r2 + 1], dv[4 * r2 + 1], m1[r2]);
                    m0[r2] = fma(a[4 * r2 + 2] * a[4 * r2 + 2], dv[4 * r2 + 2], m0[r2]); m1[r2] = fma(a[4 * r2 + 3] * a[4 * r2 + 3], dv[4 * r2 + 3], m1[r2]);
                }
            }
        }
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            out[r2] = o0[r2] + o1[r2];
            md[r2] = DIAG ? ((lane + 64 * r2 < m) ? m0[r2] + m1[r2] : 1.0) : 0.0;
        }
    }

    // First terms of the Gram entries of a group, items [i0, i1): dstbuf[dst] = w d[col].  IPL items per lane per trip, every
    // table read of the trip in ONE round trip, every d in a second
    template <int IPL>
    __device__ __forceinline__ void scatter_first(double* dstbuf, int i0, int i1) const {
        const unsigned wb = lds_addr(t_w), cb = lds_addr(t_cd), db = lds_addr(vd_());
        for (int base = i0; base < i1; base += 64 * IPL) {
            unsigned aw[IPL], ac[IPL], cd[IPL], ad[IPL]; bool on[IPL]; double wv[IPL], dv[IPL];
#pragma unroll
            for (int k = 0; k < IPL; k++) {
                const int ik = base + lane + 64 * k;
                on[k] = ik < i1;
                const int ic = on[k] ? ik : i0;
                aw[k] = wb + 8 * ic; ac[k] = cb + 4 * ic;
            }
            if constexpr (IPL == 6) lds_gather6_d_u(aw, ac, wv, cd); else lds_gather4_d_u(aw, ac, wv, cd);
#pragma unroll
            for (int k = 0; k < IPL; k++) ad[k] = db + 8 * (cd[k] & 0xffffu);
            if constexpr (IPL == 6) lds_gather6(ad, dv); else lds_gather4(ad, dv);
#pragma unroll
            for (int k = 0; k < IPL; k++) if (on[k]) dstbuf[cd[k] >> 16] = wv[k] * dv[k];
        }
    }
    // Further terms of the entries that have more than one: records in threes (the 2nd..4th term of an entry, then its
    // 5th..7th in the next round, ...; short triples padded with weight 0), one entry per lane, added to the entry in term order
    __device__ __forceinline__ void scatter_more(double* dstbuf, int i0, int i1) const {
        const unsigned wb = lds_addr(t_w), cb = lds_addr(t_cd), db = lds_addr(vd_()), ob = lds_addr(dstbuf);
        for (int base = i0; base < i1; base += 192) {
            const int ik = base + 3 * lane;
            const bool on = ik < i1;
            const int ic = on ? ik : i0;
            double wv[3], dv[4]; unsigned cd[3], ad[4];
            lds_terms3(wb + 8 * ic, cb + 4 * ic, wv, cd);
#pragma unroll
            for (int k = 0; k < 3; k++) ad[k] = db + 8 * (cd[k] & 0xffffu);
            ad[3] = ob + 8 * (cd[0] >> 16);
            lds_gather4(ad, dv);
            const double acc = fma(wv[2], dv[2], fma(wv[1], dv[1], fma(wv[0], dv[0], dv[3])));
            if (on) dstbuf[cd[0] >> 16] = acc;
        }
    }
    // PA forms of the two passes: the weight of a term is the product of the two entries of A that t_ab names, taken
    // from this wave's copy of its LP's values (padded records name the zero entry behind the values)
    __device__ __forceinline__ void scatter_first_pa(double* dstbuf, int i0, int i1) const {
        const unsigned abb = lds_addr(t_ab), cb = lds_addr(t_cd), db = lds_addr(vd_()), cvb = lds_addr(csr_val);
        for (int base = i0; base < i1; base += 256) {
            unsigned aa[4], ac[4], ab[4], cd[4], ga[12]; bool on[4]; double gv[12];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int ik = base + lane + 64 * k;
                on[k] = ik < i1;
                const int ic = on[k] ? ik : i0;
                aa[k] = abb + 4 * ic; ac[k] = cb + 4 * ic;
            }
            lds_gather4_u_u(aa, ac, ab, cd);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                ga[3 * k] = cvb + 8 * (ab[k] & 0xffffu); ga[3 * k + 1] = cvb + 8 * (ab[k] >> 16); ga[3 * k + 2] = db + 8 * (cd[k] & 0xffffu);
            }
            lds_gather12(ga, gv);
#pragma unroll
            for (int k = 0; k < 4; k++) if (on[k]) dstbuf[cd[k] >> 16] = (gv[3 * k] * gv[3 * k + 1]) * gv[3 * k + 2];
        }
    }
    __device__ __forceinline__ void scatter_more_pa(double* dstbuf, int i0, int i1) const {
        const unsigned abb = lds_addr(t_ab), cb = lds_addr(t_cd), db = lds_addr(vd_()), ob = lds_addr(dstbuf), cvb = lds_addr(csr_val);
        for (int base = i0; base < i1; base += 192) {
            const int ik = base + 3 * lane;
            const bool on = ik < i1;
            const int ic = on ? ik : i0;
            unsigned ab[3], cd[3], ga[10]; double gv[10];
            lds_terms3_uu(abb + 4 * ic, cb + 4 * ic, ab, cd);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                ga[3 * k] = cvb + 8 * (ab[k] & 0xffffu); ga[3 * k + 1] = cvb + 8 * (ab[k] >> 16); ga[3 * k + 2] = db + 8 * (cd[k] & 0xffffu);
            }
            ga[9] = ob + 8 * (cd[0] >> 16);
            lds_gather10(ga, gv);
            const double acc = fma(gv[6] * gv[7], gv[8], fma(gv[3] * gv[4], gv[5], fma(gv[0] * gv[1], gv[2], gv[9])));
            if (on) dstbuf[cd[0] >> 16] = acc;
        }
    }
    // all terms of Gram group g into dstbuf (zeroed by the caller: entries the structure does not have stay 0)
    __device__ __forceinline__ void scatter_group(double* dstbuf, int g) const {
        const int l0 = __builtin_amdgcn_readfirstlane(meta[META_SEG + g]), l1 = __builtin_amdgcn_readfirstlane(meta[META_SEG + g + 1]);
        if (l0 < l1) {
            int i0 = __builtin_amdgcn_readfirstlane(lev[l0]), i1 = __builtin_amdgcn_readfirstlane(lev[l0 + 1]);
            if constexpr (PA) scatter_first_pa(dstbuf, i0, i1);
            else { if (i1 - i0 > 256) scatter_first<6>(dstbuf, i0, i1); else scatter_first<4>(dstbuf, i0, i1); }
            for (int l = l0 + 1; l < l1; l++) {
                i0 = i1; i1 = __builtin_amdgcn_readfirstlane(lev[l + 1]);
                if constexpr (PA) scatter_more_pa(dstbuf, i0, i1); else scatter_more(dstbuf, i0, i1);
            }
        }
    }

    // M = A diag(d) A' (d in vd_(), diagonal in Md): the off-diagonal blocks go through the staging area HB at a time and are
    // parked in the accumulator file; the diagonal blocks (lower triangle with diagonal, packed by rows) are left in their
    // slots of the W area, where factor() picks block K up when its turn comes and then overwrites it with W_K.
    __device__ __forceinline__ void gram(const double (&Md)[MR]) {
        if constexpr (DA) { gram_dense(Md); return; }
        static_assert(DA || (G::NBLK < HB ? G::NBLK : HB) * 256 <= STAGE_D + MB * WL, "staging area too small");
        const double2_t zero = {0.0, 0.0};
        // zero the diagonal-block slots, scatter group g's diagonal-block entries (dsts relative to `base`), set the diagonal
        // from Md (row 16K + i lives in lane (16K + i) % 64 of register (16K + i) / 64)
        auto zero_slots = [&]() {
#pragma unroll
            for (int w = 0; w < (MB * WL + 127) / 128; w++)
                if ((w + 1) * 128 <= MB * WL || 2 * (w * 64 + lane) < MB * WL) ((double2_t*)wl_())[w * 64 + lane] = zero;
        };
        auto set_diag = [&]() {
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int row = lane + 64 * r2, il = row & 15;
                if (row < MP) wl_()[(row >> 4) * WL + il * (il + 1) / 2 + il] = Md[r2];
            }
        };
        static_for<0, G::NCHUNK>([&](auto cc) {
            constexpr int ci = decltype(cc)::value;
            constexpr int b0 = HB * ci;
            constexpr int nb = (G::NBLK - b0 < HB) ? G::NBLK - b0 : HB;
            constexpr bool with_diag = G::MERGE_DIAG(NQ) && ci == G::NCHUNK - 1;     // the last chunk leaves the W area alone
            pin();
#pragma unroll
            for (int w = 0; w < 2 * nb; w++) ((double2_t*)stage_())[w * 64 + lane] = zero;
            if constexpr (with_diag) zero_slots();
            wave_lds_sync();
            scatter_group(stage_(), ci);
            if constexpr (with_diag) set_diag();
            wave_lds_sync();
            // four blocks (16 registers) per round trip; a last group of fewer reads on into whatever follows in LDS and drops it
#pragma unroll
            for (int b4 = 0; b4 < nb; b4 += 4) {
                double v[16];
                lds_run16<0, 512>(lds_addr(stage_() + b4 * 256 + lane), v);
#pragma unroll
                for (int bi = 0; bi < 4; bi++) {
                    if (b4 + bi < nb) {
                        const double4_t blk = {v[4 * bi], v[4 * bi + 1], v[4 * bi + 2], v[4 * bi + 3]};
                        park(P[b0 + b4 + bi], blk);
                    }
                }
            }
            wave_lds_sync();
        });
        if constexpr (!G::MERGE_DIAG(NQ)) {       // diagonal blocks as a group of their own (dsts relative to the stage as well)
            pin();
            zero_slots();
            wave_lds_sync();
            scatter_group(stage_(), G::NCHUNK);
            set_diag();
            wave_lds_sync();
        }
    }

    // Dense image variant of gram(): M = (A diag(d)) A' block by block on the matrix cores, k = the dense columns four at a
    // time.  Operands straight from the image: A-operand lane (m = c16, k = q) = A[16K + c16][4s + q] d[4s + q], B-operand
    // lane (k = q, n = c16) = A[16I + c16][4s + q]; the accumulator of block (K, I) IS the block in the layout it is kept
    // in.  All off-diagonal accumulators are live through one pass over the columns (NBLK x 8 accumulator registers); the
    // diagonal blocks take a second pass and go, lower triangle packed by rows, to their W slots like in gram().
    __device__ __forceinline__ void gram_dense(const double (&Md)[MR]) {
        pin();
        const unsigned ib = lds_addr(img), db = lds_addr(vd_()) + 8 * q;
        unsigned ra[MB]; bool rok[MB];
#pragma unroll
        for (int J = 0; J < MB; J++) {
            const int r = 16 * J + c16;
            rok[J] = r < imgR;
            ra[J] = ib + 8 * (unsigned)((rok[J] ? r : 0) * AS + q);
        }
        const int ks = (AS - 1) / 4;
        // operands of k-step s: a[J] = A[16J + c16][4s + q] (0 in the padded rows), dk = d[4s + q]
        auto load_step = [&](int s, double (&a)[MB], double& dk) {
            dk = *(const __attribute__((address_space(3))) double*)(size_t)(db + 32 * s);
#pragma unroll
            for (int J = 0; J < MB; J++) {
                const double v = *(const __attribute__((address_space(3))) double*)(size_t)(ra[J] + 32 * s);
                a[J] = rok[J] ? v : 0.0;
            }
        };
        // (at most 14 off-diagonal accumulators per pass over the columns: with all 28 of m = 128 live next to the blocks
        // already parked the register file overflows).  The loop is software-pipelined by hand: the operands of step s + 1
        // are requested before the MFMAs of step s are issued, so their LDS round trip runs under the matrix pipe's time.
        constexpr int PB = (G::NBLK <= 16) ? (G::NBLK > 0 ? G::NBLK : 1) : 14, NPASS = (G::NBLK + PB - 1) / PB;
        static_for<0, NPASS>([&](auto Hc) {
            constexpr int b0 = PB * decltype(Hc)::value, b1 = (b0 + PB < G::NBLK) ? b0 + PB : G::NBLK;
            double4_t acc[PB];
#pragma unroll
            for (int b = 0; b < PB; b++) acc[b] = (double4_t){0.0, 0.0, 0.0, 0.0};
            double a[MB], dk;
            load_step(0, a, dk);
            // (four k-steps per trip: the allocator keeps the loop-carried accumulators in VGPRs and copies them to the
            // accumulator file and back around every trip -- 16 moves per MFMA with one step per trip, 170 instead of 64 cycles)
#pragma unroll 1
            for (int s0 = 0; s0 < ks; s0 += 4)         // (the image is padded to a multiple of 16 columns: ks % 4 == 0)
#pragma unroll
            for (int su = 0; su < 4; su++) {
                const int s = s0 + su;
                double an[MB], dkn, ad[MB];
                load_step((s + 1 < ks) ? s + 1 : s, an, dkn);
#pragma unroll
                for (int J = 0; J < MB; J++) ad[J] = a[J] * dk;
                static_for<0, MB>([&](auto Kc) {
                    constexpr int K = decltype(Kc)::value;
                    static_for<K + 1, MB>([&](auto Ic) {
                        constexpr int I = decltype(Ic)::value;
                        constexpr int bx = G::bix(K, I);
                        if constexpr (bx >= b0 && bx < b1)
                            acc[bx - b0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[K], a[I], acc[bx - b0], 0, 0, 0);
                    });
                });
#pragma unroll
                for (int J = 0; J < MB; J++) a[J] = an[J];
                dk = dkn;
            }
#pragma unroll
            for (int b = b0; b < b1; b++) park(P[b], acc[b - b0]);
        });
        double4_t dacc[MB];
#pragma unroll
        for (int K = 0; K < MB; K++) dacc[K] = (double4_t){0.0, 0.0, 0.0, 0.0};
        {
            double a[MB], dk;
            load_step(0, a, dk);
#pragma unroll 1
            for (int s0 = 0; s0 < ks; s0 += 4)
#pragma unroll
            for (int su = 0; su < 4; su++) {
                const int s = s0 + su;
                double an[MB], dkn;
                load_step((s + 1 < ks) ? s + 1 : s, an, dkn);
#pragma unroll
                for (int J = 0; J < MB; J++) dacc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[J] * dk, a[J], dacc[J], 0, 0, 0);
#pragma unroll
                for (int J = 0; J < MB; J++) a[J] = an[J];
                dk = dkn;
            }
        }
        // accumulator register r of lane (q, c16) = element [row 4r + q][column c16] of the block: lower triangle (with the
        // diagonal, which set below from Md) to offset row (row + 1) / 2 + column of slot K, the rest to the slot's spare doubles
#pragma unroll
        for (int K = 0; K < MB; K++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * r + q;
                wl_()[K * WL + ((c16 <= row) ? row * (row + 1) / 2 + c16 : 138 + r)] = dacc[K][r];
            }
        wave_lds_sync();
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            const int row = lane + 64 * r2, il = row & 15;
            if (row < MP) wl_()[(row >> 4) * WL + il * (il + 1) / 2 + il] = Md[r2];
        }
        wave_lds_sync();
    }

    // W_K element [row 4s + q][column c16] -- the TRANSPOSED operand layout -- from the packed copy in LDS
    // (the strictly-lower entries come from the slot; the slot's spare doubles [136, 144) hold what factor() stored for the
    // positions on and above the diagonal: 1 on it, 0 above -- so the read is one unconditional load at offset woff[s])
    template <int K>
    __device__ __forceinline__ double w_elemT(int s, const int (&woff)[4]) const { return wl_()[K * WL + woff[s]]; }
    __device__ __forceinline__ void w_offsets(int (&woff)[4]) const {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int row = 4 * s + q;
            woff[s] = (c16 < row) ? row * (row - 1) / 2 + c16 : ((c16 == row) ? 136 : 137);
            asm volatile("" : "+v"(woff[s]));
        }
    }

    // Blocked LDL' of the matrix whose off-diagonal blocks are parked in P and whose diagonal blocks sit in the slots of
    // the W area (gram(), or the caller, put them there).
    // RELF: pivot floor of column j is flr_()[j] (LDS) instead of floor_.
    // Returns (wave-uniform) whether the Nocedal-Wright guard would have bitten anywhere.
    template <bool RELF>
    __device__ __forceinline__ bool factor(double beta2, double floor_ STAMP_ARGS) {
        double ymax = 0.0, ymaxc = 0.0;     // running maxima of Y^2 / D (panels) and u^2 / D (pivot chains): the guard bites iff > beta^2
        double* tile = stage_() + TILE_OFF;
        static_for<0, MB>([&](auto Kc) {
            constexpr int K = decltype(Kc)::value;
            pin();
            // ---- diagonal block K, left-looking: Schur update -sum_{K'<K} (D U_K'K)' U_K'K on the matrix cores ----
            //      1/D of the pivots 4t + q, t < 4K, in one or two round trips; two accumulators, so that consecutive MFMAs do
            //      not wait for each other
            double4_t sch = {0.0, 0.0, 0.0, 0.0}, sch1 = {0.0, 0.0, 0.0, 0.0};
            if constexpr (K > 0) {
                static_for<0, (K + 3) / 4>([&](auto Hc) {
                    constexpr int H = decltype(Hc)::value;          // block rows 4H .. 4H + 3 (< K)
                    double rdk[16];
                    lds_run16<512 * H, 32>(lds_addr(rdv_() + q), rdk);
                    static_for<4 * H, (4 * H + 4 < K ? 4 * H + 4 : K)>([&](auto Kp) {
                        constexpr int K2 = decltype(Kp)::value;
#pragma unroll
                        for (int s = 0; s < 4; s++) {
                            const double y = unpark(P[G::bix(K2, K)], s);
                            const double ny = -(y * rdk[4 * (K2 - 4 * H) + s]);
                            if (s & 1) sch1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ny, y, sch1, 0, 0, 0);
                            else sch = __builtin_amdgcn_mfma_f64_16x16x4f64(ny, y, sch, 0, 0, 0);
                        }
                    });
                });
#pragma unroll
                for (int r = 0; r < 4; r++) sch[r] += sch1[r];
            }
            // accumulator layout -> tile, + original block -> lane = row (each 16-lane row of the wave a redundant copy)
#pragma unroll
            for (int r = 0; r < 4; r++) tile[(4 * r + q) * 17 + c16] = sch[r];
            wave_lds_sync();
            STAMP(2)
            double Wd[16], Ws[4];
            [[maybe_unused]] double Ld[16];
#pragma unroll
            for (int s = 0; s < 4; s++) Ws[s] = (c16 == 4 * s + q) ? 1.0 : 0.0;
            {
                // row c16 of the tile and of the original block (slot K, row offset c16 (c16 + 1) / 2); columns > c16: whatever
                // follows in the slot (finite, in bounds, never used).  The two lane-dependent addresses are the same for all K.
                double2_t tl[8]; double rw[16];
                lds_tile_and_raw<8 * K * WL>(lds_addr(tile + c16 * 17), lds_addr(wl_() + c16 * (c16 + 1) / 2), tl, rw);
#pragma unroll
                for (int k = 0; k < 8; k++) { Wd[2 * k] = tl[k][0] + rw[2 * k]; Wd[2 * k + 1] = tl[k][1] + rw[2 * k + 1]; }
            }
            STAMP(3)
            const double myf = RELF ? flr_()[16 * K + c16] : floor_;
            wave_lds_sync();
            double rdiag = 1.0, rD;
            {
                const double piv = bcast64<0>(Wd[0]);
                rD = fast_rcp(fmax(fabs(piv), RELF ? row_bcast<0>(myf) : floor_));
            }
            static_for<0, 16>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const double u = Wd[j];
                // lanes below the pivot / the pivot's lane, in every 16-lane DPP row: compile-time EXEC masks (chain_head_exec)
                constexpr unsigned m16 = ((0xFFFFu << (j + 1)) & 0xFFFFu) * 0x10001u, one16 = (1u << j) * 0x10001u;
                double nli;
                chain_head_exec<m16, one16>(u, rD, nli, ymaxc, rdiag);
                if constexpr (j < 15) {      // Wd[k] -= l_i u_k, k > j, with the next pivot's reciprocal chain in between (chain_asm.inc)
                    double aDn, rDn;
                    if constexpr (RELF) chain_step_pipe_relf<j>(Wd, u, nli, floor_, myf, aDn, rDn);
                    else chain_step_pipe<j>(Wd, u, nli, floor_, aDn, rDn);
                    rD = rDn;
                    // PYCLLP_WINV_FUSED: step j of W = L_KK^-1 (A-operand layout: Ws[s] = W[row c16][column 4s + q]; nli = -L[.][j])
                    // rides along with the sweep instead of running as 15 steps after it: its one to four FMAs fill the tail of the
                    // reciprocal chain that the late columns' few trailing updates leave exposed (+0.5 %, 389.4 -> 391.5 k LPs/s)
                    if constexpr (PYCLLP_WINV_FUSED) winv_step<j>(Ws, nli); else Ld[j] = nli;
                }
            });
            if (q == 0) rdv_()[16 * K + c16] = rdiag;
            STAMP(4)
            pin();
            // ---- W = L_KK^-1 in the A-operand layout: Ws[s] = W[row c16][column 4s + q]; packed copy to LDS (Ld holds -L) ----
            if constexpr (!PYCLLP_WINV_FUSED) {
                static_for<0, 15>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    winv_step<j>(Ws, Ld[j]);
                });
            }
            // (entries on and above the diagonal go to spare doubles of the slot: one store each, no branch; [136] and [137]
            // get the constants 1 and 0 that solve() reads for the diagonal and the upper triangle of W)
#pragma unroll
            for (int s = 0; s < 4; s++) wl_()[K * WL + ((4 * s + q < c16) ? c16 * (c16 - 1) / 2 + 4 * s + q : 138 + s)] = Ws[s];
            if (lane < 2) wl_()[K * WL + 136 + lane] = (lane == 0) ? 1.0 : 0.0;
            // 1/D in the row form of the accumulator layout (register r <-> pivot 4r + q), back from LDS
            wave_lds_sync();
            double rDr[4];
#pragma unroll
            for (int r = 0; r < 4; r++) rDr[r] = rdv_()[16 * K + 4 * r + q];
            // ---- panel: Y_KI = W M_KI = D_K L_IK' on the matrix cores.  The block stays UNSCALED in its accumulator
            //      registers (every use below is an MFMA operand or folds 1/D into a vector): nothing ever writes a
            //      resident block from the VALU side.  Guard test: Y^2 > beta^2 D. ----
            static_for<K + 1, MB>([&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; s++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[s], (K == 0) ? unpark(P[G::bix(K, I)], s) : U[G::bix(K, I)][s], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) ymax = fmax(ymax, acc[r] * acc[r] * rDr[r]);      // Y^2 / D, compared with beta^2 once, at the end
                asm volatile("" : "+v"(ymax));
                park(P[G::bix(K, I)], acc);
            });
            STAMP(5)
            // ---- trailing update of the off-diagonal blocks: M_JI -= Y_KJ' D_K^-1 Y_KI, J < I; the A operand
            //      -D^-1 Y_KJ is formed per block row J ----
            static_for<K + 1, MB>([&](auto Jc) {
                constexpr int J = decltype(Jc)::value;
                double yn[4];
#pragma unroll
                for (int r = 0; r < 4; r++) yn[r] = -(unpark(P[G::bix(K, J)], r) * rDr[r]);
                static_for<J + 1, MB>([&](auto Ic) {
                    constexpr int I = decltype(Ic)::value;
                    double4_t acc;
                    if constexpr (K == 0) {
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[r] = unpark(P[G::bix(J, I)], r);
                    } else {
                        acc = U[G::bix(J, I)];
                    }
#pragma unroll
                    for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(yn[s], unpark(P[G::bix(K, I)], s), acc, 0, 0, 0);
                    U[G::bix(J, I)] = acc;
                });
            });
            STAMP(6)
            __builtin_amdgcn_sched_barrier(0);   // one panel at a time: nothing of panel K+1 is hoisted above this line
        });
        wave_lds_sync();
        return __any(ymax > beta2 || ymaxc > beta2);
    }

    // um <- (L D L')^-1 um.  Vectors of a 16-row block appear in two forms: "column form" (lane (c16, q) holds element
    // c16, identical in the four quads) and "row form" (register r of lane (c16, q) holds element 4r + q, identical in the
    // 16 lanes of a quad).  With W_K read in the TRANSPOSED operand layout (element [4s+q][c16]) every product maps one
    // form onto the other with a DPP row reduction or a quad reduction and NO layout conversion through LDS:
    //   forward   t_I = W_I r_I:   r column form -> products -> row_sum  -> t in row form  (what the Y blocks multiply)
    //   backward  x_K = W_K' v_K:  v row form    -> products -> quad_sum -> x in column form (what the Y blocks multiply)
    // Forward substitution is column oriented (t_K, once known, is folded into the partial sums of all later block rows
    // and dropped), backward substitution row oriented: at most 8 + 4 doubles of vector state live.
    __device__ __forceinline__ void solve() {
        pin();
        // Everything the two sweeps read from LDS -- W_I in the transposed operand layout, 1/D in row form, the right-hand side in
        // column form -- is fetched UP FRONT in a few batched round trips: the factor's blocks sit in the accumulator file, so the
        // vector file is all but empty here, and a wavefront alone on its SIMD pays every LDS round trip inside the serial chain
        // in full (rounds 1-2: nine waits per block row).  The four row sums of a block row run interleaved (row_sum4), and the
        // partial sums p[J] are pinned per block column so that the compiler keeps the column-oriented order written here (left
        // alone it re-associated each p[I] into one chain of up to 28 dependent FMAs in front of its use).
        unsigned wa[4];
        {
            int woff[4];
            w_offsets(woff);
#pragma unroll
            for (int s = 0; s < 4; s++) { wa[s] = lds_addr(wl_() + woff[s]); asm volatile("" : "+v"(wa[s])); }
        }
        double Wel[MB][4], rdR[MB][4], umC[MB];
        static_for<0, MB / 4>([&](auto bc) { constexpr int b = decltype(bc)::value; lds_gather4xN<8 * WL * 4 * b, 8 * WL, 4>(wa, &Wel[4 * b][0]); });
        if constexpr (MB % 4 >= 2) lds_gather4xN<8 * WL * (MB / 4 * 4), 8 * WL, 2>(wa, &Wel[MB / 4 * 4][0]);
        if constexpr (MB % 2 == 1) lds_gather4xN<8 * WL * (MB - 1), 8 * WL, 1>(wa, &Wel[MB - 1][0]);
        lds_run<0, 32, 4 * MB>(lds_addr(rdv_() + q), &rdR[0][0]);         // rdR[I][r] = 1 / D[16 I + 4 r + q]
        lds_run<0, 128, MB>(lds_addr(um_() + c16), &umC[0]);              // umC[I] = s[16 I + c16]
        double p[MB];
#pragma unroll
        for (int I = 0; I < MB; I++) p[I] = 0.0;
        // forward: t_I = W_I (s_I - sum_{K<I} L_IK t_K), L_IK t_K = Y_KI' (D_K^-1 t_K)
        static_for<0, MB>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            double rC = umC[I];
            if constexpr (I > 0) rC -= quad_sum(p[I]);
            double tR[4];
#pragma unroll
            for (int s = 0; s < 4; s++) tR[s] = Wel[I][s] * rC;
            row_sum4(tR);
#pragma unroll
            for (int s = 0; s < 4; s++) if (c16 == 0) um_()[16 * I + 4 * s + q] = tR[s];
            if constexpr (I + 1 < MB) {
#pragma unroll
                for (int r = 0; r < 4; r++) tR[r] *= rdR[I][r];     // D_I^-1 t_I: the resident blocks are Y = D L'
                static_for<I + 1, MB>([&](auto Jc) {
                    constexpr int J = decltype(Jc)::value;
#pragma unroll
                    for (int r = 0; r < 4; r++) p[J] = fma(unpark(P[G::bix(I, J)], r), tR[r], p[J]);
                });
                // (tried: pinning only p[I + 1] here and the others one step later, so that their FMAs may fill the next step's
                // row sums -- 389.5 k against 392.3 k LPs/s)
#pragma unroll
                for (int J = I + 1; J < MB; J++) asm volatile("" : "+v"(p[J]));
            }
        });
        wave_lds_sync();
        // backward: x_K = W_K' D_K^-1 (t_K - sum_{I>K} Y_KI x_I)
        double tB[MB][4];
        lds_run<0, 32, 4 * MB>(lds_addr(um_() + q), &tB[0][0]);           // tB[K][r] = t[16 K + 4 r + q]
        double xCL[MB];
        static_for<0, MB>([&](auto Kr) {
            constexpr int K = MB - 1 - decltype(Kr)::value;
            double pr[4] = {0.0, 0.0, 0.0, 0.0};
            static_for<K + 1, MB>([&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
#pragma unroll
                for (int r = 0; r < 4; r++) pr[r] = fma(unpark(P[G::bix(K, I)], r), xCL[I], pr[r]);
            });
            if constexpr (K < MB - 1) row_sum4(pr);
            double px = 0.0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                double v = tB[K][r];
                if constexpr (K < MB - 1) v -= pr[r];
                px = fma(Wel[K][r], v * rdR[K][r], px);
            }
            xCL[K] = quad_sum(px);
        });
        wave_lds_sync();
#pragma unroll
        for (int K = 0; K < MB; K++) if (q == 0) um_()[16 * K + c16] = xCL[K];
        wave_lds_sync();
    }
};

template <int MB, int NQ, bool DA, bool PA>
__device__ __forceinline__ void wreg_carve(WReg<MB, NQ, DA, PA>& w, double* W0, int tid) {
    w.W0 = W0;
    w.lane = tid & 63; w.q = w.lane >> 4; w.c16 = w.lane & 15;
}

template <int MB, int NQ, bool DA, bool PA>
__device__ __forceinline__ void wreg_setup(WReg<MB, NQ, DA, PA>& w, const WregTab& T, unsigned char* lraw, int tid) {
    using G = WGeo<MB>;
    if constexpr (PA) {
        // structure tables only; the values of a wave's LP go behind its wave area when it takes the LP
        int* s_lev = (int*)(lraw + T.o_lev);
        int* s_meta = (int*)(lraw + T.o_meta);
        unsigned short* s_csr_col = (unsigned short*)(lraw + T.o_csr_col);
        unsigned short* s_csr_ptr = (unsigned short*)(lraw + T.o_csr_ptr);
        unsigned short* s_csr_len = (unsigned short*)(lraw + T.o_csr_len);
        unsigned short* s_ec_row = (unsigned short*)(lraw + T.o_ec_row);
        unsigned short* s_ec_src = (unsigned short*)(lraw + T.o_ec_src);
        unsigned* s_colmap = (unsigned*)(lraw + T.o_colmap);
        unsigned* s_t_cd = (unsigned*)(lraw + T.o_t_cd);
        unsigned* s_t_ab = (unsigned*)(lraw + T.o_t_ab);
        const int nth = blockDim.x;
        for (int i = tid; i < T.nnz; i += nth) s_csr_col[i] = T.csr_col[i];
        for (int i = tid; i < G::MPL; i += nth) { s_csr_ptr[i] = T.csr_ptr[i]; s_csr_len[i] = T.csr_len[i]; }
        for (int i = tid; i < T.ctot * 64; i += nth) { s_ec_row[i] = T.ec_row[i]; s_ec_src[i] = T.ec_src[i]; }
        for (int i = tid; i < 64 * NQ; i += nth) s_colmap[i] = T.colmap[i];
        for (int i = tid; i < T.n_term; i += nth) { s_t_cd[i] = T.t_cd[i]; s_t_ab[i] = T.t_ab[i]; }
        for (int i = tid; i <= T.n_lev; i += nth) s_lev[i] = T.lev[i];
        for (int i = tid; i < META_N; i += nth) s_meta[i] = T.meta[i];
        wreg_carve(w, (double*)(lraw + T.o_wave) + (size_t)(tid >> 6) * T.wave_doubles, tid);
        for (int i = T.nnz + (tid & 63); i < T.nnzp; i += 64) w.cvl_()[i] = 0.0;       // the zero entry padded records name
        __syncthreads();
        w.csr_val = w.cvl_(); w.csr_col = s_csr_col; w.csr_ptr = s_csr_ptr; w.csr_len = s_csr_len;
        w.ec_val = nullptr; w.ec_row = s_ec_row; w.ec_src = s_ec_src; w.colmap = s_colmap;
        w.t_w = nullptr; w.t_cd = s_t_cd; w.t_ab = s_t_ab; w.lev = s_lev; w.meta = s_meta;
        w.m = T.m; w.n = T.n; w.rmax = T.rmax;
        return;
    }
    if constexpr (DA) {
        double* s_img = (double*)(lraw + T.o_img);
        const int cnt = T.img_rows * T.as;
        for (int i = tid; i < cnt; i += (int)blockDim.x) s_img[i] = T.img[i];
        __syncthreads();
        w.img = s_img; w.nd = T.nd; w.AS = T.as; w.imgR = T.img_rows;
        wreg_carve(w, (double*)(lraw + T.o_wave) + (size_t)(tid >> 6) * T.wave_doubles, tid);
        w.m = T.m; w.n = T.n; w.rmax = 0;
        return;
    }
    double* s_csr_val = (double*)(lraw + T.o_csr_val);
    double* s_ec_val = (double*)(lraw + T.o_ec_val);
    double* s_t_w = (double*)(lraw + T.o_t_w);
    int* s_lev = (int*)(lraw + T.o_lev);
    int* s_meta = (int*)(lraw + T.o_meta);
    unsigned short* s_csr_col = (unsigned short*)(lraw + T.o_csr_col);
    unsigned short* s_csr_ptr = (unsigned short*)(lraw + T.o_csr_ptr);
    unsigned short* s_csr_len = (unsigned short*)(lraw + T.o_csr_len);
    unsigned short* s_ec_row = (unsigned short*)(lraw + T.o_ec_row);
    unsigned* s_colmap = (unsigned*)(lraw + T.o_colmap);
    unsigned* s_t_cd = (unsigned*)(lraw + T.o_t_cd);
    const int nth = blockDim.x;
    for (int i = tid; i < T.nnz; i += nth) {
        s_csr_val[i] = T.csr_val[i]; s_csr_col[i] = T.csr_col[i];
    }
    for (int i = tid; i < G::MPL; i += nth) { s_csr_ptr[i] = T.csr_ptr[i]; s_csr_len[i] = T.csr_len[i]; }
    for (int i = tid; i < T.ctot * 64; i += nth) { s_ec_val[i] = T.ec_val[i]; s_ec_row[i] = T.ec_row[i]; }
    for (int i = tid; i < 64 * NQ; i += nth) s_colmap[i] = T.colmap[i];
    for (int i = tid; i < T.n_term; i += nth) { s_t_w[i] = T.t_w[i]; s_t_cd[i] = T.t_cd[i]; }
    for (int i = tid; i <= T.n_lev; i += nth) s_lev[i] = T.lev[i];
    for (int i = tid; i < META_N; i += nth) s_meta[i] = T.meta[i];
    __syncthreads();
    w.csr_val = s_csr_val; w.csr_col = s_csr_col; w.csr_ptr = s_csr_ptr; w.csr_len = s_csr_len;
    w.ec_val = s_ec_val; w.ec_row = s_ec_row; w.colmap = s_colmap;
    w.t_w = s_t_w; w.t_cd = s_t_cd; w.lev = s_lev; w.meta = s_meta;
    wreg_carve(w, (double*)(lraw + T.o_wave) + (size_t)(tid >> 6) * T.wave_doubles, tid);
    w.m = T.m; w.n = T.n; w.rmax = T.rmax;
}

// Newton step of the primal normal equations for the point (x, z, y) of this wave's LP (ldl.cl:656-712 with the x-space
// refinement of oracle newton_dy):  M dy = A(d t) - rho,  dx = d (t - A'dy),  then  e = rho - A dx;  M eta = e;
// dx += d A'eta;  dy -= eta  while max|e| > etol, at most max_refine times.  The first solve is written as pass 0 of that
// loop so that the kernel holds ONE copy of the (fully unrolled) block substitution.
// In: t (per column, parked in the stage), d in vd_(), rho (per row), um = A(d t) - rho in LDS, the factor in w.P / w.wl_().
// TCV: the caller has parked x and z in the stage (at NP, 2 NP) and cv = c - A'y in vd_() in place of d; t = cv + mu / x and
// d = x / z are formed here (the same expressions the caller used for the right-hand side).
// Out: dy (per row), dx, wv = A'dy, e = rho - A dx.  Returns the refinement passes used; `bad` reports a non-finite dy.
// COR (predictor-corrector): the complementarity target of column j is cor[j] (= mu - dx_a dz_a) instead of the scalar mu.
template <bool TCV, bool COR = false, int MB, int NQ, bool DA, bool PA>
__device__ __forceinline__ int newton_solve(WReg<MB, NQ, DA, PA>& w, const bool (&okc)[NQ], const bool (&okr)[WGeo<MB>::MR],
                                            const double (&rho)[WGeo<MB>::MR], double etol, int max_refine, double mu,
                                            double (&dy)[WGeo<MB>::MR], double (&dx)[NQ], double (&wv)[NQ],
                                            double (&e)[WGeo<MB>::MR], bool& bad, const double* cor STAMP_ARGS) {
    constexpr int MR = WGeo<MB>::MR, MP = WGeo<MB>::MP;
    const int& lane = w.lane;
    double* vx = w.stage_();
    int pass = 0;
    bad = false;
    for (;;) {
        w.solve();
        STAMP(7)
        double w2[NQ], d[NQ];
        w.At(w.um_(), w2);
        double xq[NQ];
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) {
            if (TCV) {
                xq[qq] = w.stage_()[64 * NQ + lane + 64 * qq];
                d[qq] = okc[qq] ? xq[qq] * fast_rcp(w.stage_()[128 * NQ + lane + 64 * qq]) : 0.0;
            } else {
                d[qq] = w.vd_()[lane + 64 * qq];      // d = x/z (0 in padded positions), still there from gram()
            }
        }
        if (pass == 0) {
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                double tq;
                if (TCV) tq = okc[qq] ? w.vd_()[lane + 64 * qq] + (COR ? cor[qq] : mu) * fast_rcp(xq[qq]) : 0.0;
                else tq = w.stage_()[lane + 64 * qq];       // t, parked there by the caller
                wv[qq] = w2[qq];
                dx[qq] = (tq - w2[qq]) * d[qq];
            }
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                dy[r2] = (lane + 64 * r2 < MP) ? w.um_()[lane + 64 * r2] : 0.0;
                bad = bad | !isfinite(dy[r2]);
            }
        } else {
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) { dx[qq] = fma(d[qq], w2[qq], dx[qq]); wv[qq] -= w2[qq]; }
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) dy[r2] -= (lane + 64 * r2 < MP) ? w.um_()[lane + 64 * r2] : 0.0;
        }
        wave_lds_sync();
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) vx[lane + 64 * qq] = okc[qq] ? dx[qq] : 0.0;
        wave_lds_sync();
        double Adx[MR], dummy[MR], me = 0.0;
        w.template Arow<false>(vx, Adx, dummy);
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            e[r2] = okr[r2] ? rho[r2] - Adx[r2] : 0.0;
            me = fmax(me, fabs(e[r2]));
        }
        const double maxe = wmax(me);
        STAMP(8)
        if (!(maxe > etol) || pass >= max_refine) break;
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) if (lane + 64 * r2 < MP) w.um_()[lane + 64 * r2] = e[r2];
        wave_lds_sync();
        pass++;
    }
    bad = __any(bad);
    return pass;
}

// PA: the values of LP `lp` (a row of a_batch [B, nnz], CSR order of the plan) into this wave's copy -- HBM -> LDS once per
// LP, 8 nnz bytes next to the 16 (m + n) + 24 of b, c, x, y; through a buffer descriptor (see row_rsrc), all loads of the
// wave in flight before the first store
template <int MB, int NQ, bool DA, bool PA>
__device__ __forceinline__ void load_lp_values(WReg<MB, NQ, DA, PA>& w, const double* ag, long lp, int nnz) {
    const __amdgpu_buffer_rsrc_t ra = row_rsrc(ag + lp * nnz, nnz);
    double* cv = w.cvl_();
    for (int e0 = 0; e0 < nnz; e0 += 512) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = buf_ld(ra, 8u * (unsigned)(e0 + 64 * k + w.lane));      // past the row: 0
#pragma unroll
        for (int k = 0; k < 8; k++) if (e0 + 64 * k + w.lane < nnz) cv[e0 + 64 * k + w.lane] = v[k];
    }
    wave_lds_sync();
}

// ------------------------------------------------------------------------------------------------------------------
// solve kernel: sparse_standard_primal_normal (primal_normal.cl:287-375), one LP per wavefront
// ------------------------------------------------------------------------------------------------------------------
// PC (PYCLLP_FLAG_PREDCORR): Mehrotra's predictor-corrector, oracle ipm_one_pc -- after the iteration's one factorisation a
// predictor solve with mu = 0, the centering parameter from how far it gets, then the corrector solve (newton_solve with the
// per-column target cor = mu - dx_a dz_a): one more block substitution, A'u and A v per iteration, about half the iterations
// on config 5's structure (52.7 -> 25.5)
template <int MB, int NQ, bool DA, bool PA, bool PC = false>
__global__ void __launch_bounds__(256, 1)
ipm_wreg_kernel(WregTab T, long B, const double* __restrict__ ag, const double* __restrict__ bg, const double* __restrict__ cg,
                double* __restrict__ xg, double* __restrict__ yg, double* __restrict__ zg, double* __restrict__ pobj,
                double* __restrict__ dobj, int* __restrict__ status, int* __restrict__ iters, int* __restrict__ queue,
                int* __restrict__ defer, DevOpts o) {
    using G = WGeo<MB>;
    constexpr int MR = G::MR, MP = G::MP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lraw[];
    WReg<MB, NQ, DA, PA> w;
    USE_AGPR_FORM();
    wreg_setup(w, T, lraw, threadIdx.x);
    const int& lane = w.lane;
    const int m = w.m, n = w.n;
    const bool warm = (o.flags & PYCLLP_FLAG_WARM_START) != 0;
    const bool autoscale = (o.flags & PYCLLP_FLAG_AUTOSCALE) != 0;
    const double nm = (double)(n + m);
    double* vx = w.stage_();
    bool okc[NQ], okr[MR];
#pragma unroll
    for (int qq = 0; qq < NQ; qq++) okc[qq] = lane + 64 * qq < n;
#pragma unroll
    for (int r2 = 0; r2 < MR; r2++) okr[r2] = lane + 64 * r2 < m;

    long lp;
    {
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        lp = __builtin_amdgcn_readfirstlane(nxt);
    }
    STAMP_DECL
    while (lp < B) {
        if constexpr (PA) load_lp_values(w, ag, lp, T.nnz);
        double x[NQ], z[NQ];
        double c2 = 0.0;
        const __amdgpu_buffer_rsrc_t rc = row_rsrc(cg + lp * n, n), rx = row_rsrc(xg + lp * n, n), rz = row_rsrc(zg ? zg + lp * n : nullptr, n);
        // PYCLLP_FLAG_AUTOSCALE: the LP is solved with b / max|b| and c / max|c| (the same divisions as ipm_block_kernel and
        // the oracle), undone when storing
        double sb = 1.0, sc = 1.0;
        if (autoscale) {
            double cm = 0.0, bm = 0.0;
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) cm = fmax(cm, fabs(buf_ld(rc, w.coff(qq))));
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) bm = fmax(bm, okr[r2] ? fabs(bg[lp * m + lane + 64 * r2]) : 0.0);
            sb = wmax(bm); sc = wmax(cm);
            sb = uni((sb > 0.0) ? sb : 1.0); sc = uni((sc > 0.0) ? sc : 1.0);
        }
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) {
            const unsigned jo = w.coff(qq);
            double cj = buf_ld(rc, jo);
            if (autoscale) cj = cj / sc;
            c2 = fma(cj, cj, c2);
            x[qq] = (warm && okc[qq]) ? buf_ld(rx, jo) : 1.0;
            z[qq] = (warm && okc[qq]) ? buf_ld(rz, jo) : 1.0;
            if (autoscale && warm) { x[qq] = x[qq] / sb; z[qq] = z[qq] / sc; }
        }
        double b2 = 0.0;
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            const int i = lane + 64 * r2;
            double bi = okr[r2] ? bg[lp * m + i] : 0.0;
            if (autoscale) bi = bi / sb;
            b2 = fma(bi, bi, b2);
            if (i < MP) {
                double yi = okr[r2] ? ((warm && yg) ? yg[lp * m + i] : 1.0) : 0.0;
                if (autoscale && warm && yg) yi = yi / sc;
                w.bs_()[i] = bi;
                w.ys_()[i] = yi;
            }
        }
        wave_lds_sync();
        const double nb2 = wsum(b2), nc2 = wsum(c2);
        const double tol_r = uni(o.eps * (1.0 + sqrt(nb2))), tol_s = uni(o.eps * (1.0 + sqrt(nc2)));
        const double etol = uni(o.refine_tol * (1.0 + sqrt(nb2)));
        double normr0 = 1e300, norms0 = 1e300, po = 0.0, du = 0.0;
        int stat = PYCLLP_STATUS_ITERATION_LIMIT, it = 0;
        bool running = true;
        // The residuals cv = c - A'y and rho = b - A x are CARRIED from iteration to iteration (cv -= theta A'dy,
        // rho -= theta A dx: both products exist anyway, from the Newton step and its refinement test) and recomputed from
        // the point itself only at the start and when the carried values pass the optimality test -- the verdict is then
        // taken again on the exact ones (as the dense group kernel does for rho), and the iteration goes on if it fails.
        // x, z and cv cross the loop's back edge IN LDS (the slots where they wait during factor and solve anyway), not in
        // registers: 36 loop-carried registers through this loop's control flow end up in scratch
        // (rho lives in the floor vector's place, which this kernel does not use)
        bool refresh = true, fresh = false;
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) {
            w.stage_()[64 * NQ + lane + 64 * qq] = x[qq];
            w.stage_()[128 * NQ + lane + 64 * qq] = z[qq];
        }
        wave_lds_sync();

        while (running) {
            double x[NQ], z[NQ], cv[NQ], rho[MR];
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) rho[r2] = (lane + 64 * r2 < MP) ? w.flr_()[lane + 64 * r2] : 0.0;
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                x[qq] = w.stage_()[64 * NQ + lane + 64 * qq];
                z[qq] = w.stage_()[128 * NQ + lane + 64 * qq];
                cv[qq] = w.vd_()[lane + 64 * qq];       // (not yet there in the first pass: refresh sets it)
            }
            wave_lds_sync();
            if (refresh) {
                double v[NQ], cq[NQ], Ax[MR], dm[MR];
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) cq[qq] = buf_ld(rc, w.coff(qq));   // in flight (vmcnt) while A'y runs on LDS; 0 in the padded positions
                w.At(w.ys_(), v);
                if (autoscale) {
#pragma unroll
                    for (int qq = 0; qq < NQ; qq++) cq[qq] = cq[qq] / sc;
                }
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) {
                    cv[qq] = okc[qq] ? cq[qq] - v[qq] : 0.0;
                    vx[lane + 64 * qq] = okc[qq] ? x[qq] : 0.0;
                }
                wave_lds_sync();
                w.template Arow<false>(vx, Ax, dm);
#pragma unroll
                for (int r2 = 0; r2 < MR; r2++) {
                    const int i = lane + 64 * r2;
                    rho[r2] = okr[r2] ? w.bs_()[i] - Ax[r2] : 0.0;
                    if (i < MP) w.flr_()[i] = rho[r2];
                }
                wave_lds_sync();
                refresh = false; fresh = true;
            }
            // ---- sigma, gamma, objectives (primal_normal.cl:96-120, 245-248); c'x = cv'x + y'(b - rho) ----
            double s2 = 0.0, gam = 0.0, pp = 0.0;
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                const double sg = okc[qq] ? cv[qq] + z[qq] : 0.0;
                s2 = fma(sg, sg, s2);
                gam += okc[qq] ? x[qq] * z[qq] : 0.0;
                pp += okc[qq] ? cv[qq] * x[qq] : 0.0;
            }
            double dd = 0.0, r2s = 0.0;
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int i = lane + 64 * r2;
                const double bi = (i < MP) ? w.bs_()[i] : 0.0, yi = (i < MP) ? w.ys_()[i] : 0.0;
                dd = fma(bi, yi, dd);
                pp = fma(yi, bi - rho[r2], pp);
                r2s = fma(rho[r2], rho[r2], r2s);
            }
            s2 = wsum(s2); gam = wsum(gam); po = wsum(pp); du = wsum(dd);
            const double norms = uni(sqrt(s2));
            const double normr = uni(sqrt(wsum(r2s)));
            double mu = PC ? 0.0 : uni(o.delta * gam / nm);      // PC: 0 for the predictor, set from its outcome below
            STAMP(10)
            // ---- stop tests (primal_normal.cl:256-269; oracle ipm_one_path) ----
            if (!(isfinite(normr) && isfinite(norms) && isfinite(gam))) { stat = PYCLLP_STATUS_NUMERICAL; running = false; }
            else if (normr <= tol_r && norms <= tol_s && gam <= o.eps * (1.0 + fabs(po))) {
                if (fresh) { stat = PYCLLP_STATUS_OPTIMAL; running = false; } else refresh = true;
            }
            else if (normr > 10.0 * normr0 && normr > PYCLLP_GROWTH_FLOOR * tol_r) { stat = PYCLLP_STATUS_PRIMAL_INFEASIBLE; running = false; }
            else if (norms > 10.0 * norms0 && norms > PYCLLP_GROWTH_FLOOR * tol_s) { stat = PYCLLP_STATUS_DUAL_INFEASIBLE; running = false; }
            STAMP(11)
            if (running && !refresh) {
                // ---- d, t (primal_normal.cl:50-74); rhs = A (d t) - rho, diag(M) ----
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) {
                    const int j = lane + 64 * qq;
                    const double dq = okc[qq] ? x[qq] * fast_rcp(z[qq]) : 0.0;      // v_rcp_f64 + 2 Newton steps (<= 2 ulp), as
                    const double tq = okc[qq] ? cv[qq] + mu * fast_rcp(x[qq]) : 0.0;   // the dense group kernel
                    vx[j] = dq * tq;
                    w.vd_()[j] = dq;
                }
                wave_lds_sync();
                double Adt[MR], Md[MR];
                w.template Arow<true>(vx, Adt, Md);
                double bmax = 0.0;
#pragma unroll
                for (int r2 = 0; r2 < MR; r2++) {
                    const int i = lane + 64 * r2;
                    if (i < MP) w.um_()[i] = okr[r2] ? Adt[r2] - rho[r2] : 0.0;
                    bmax = fmax(bmax, okr[r2] ? fabs(Md[r2]) : 0.0);
                }
                const double beta2 = wmax(bmax);     // ldl.cl:296-311
                wave_lds_sync();
                STAMP(0)
                // ---- M = A diag(d) A' into registers; factor ----
                w.gram(Md);
                STAMP(1)
                // cv (in d's place, which is not needed any more: the Newton step forms it again), x and z wait in LDS while
                // factor and solve have the registers
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) {
                    w.vd_()[lane + 64 * qq] = cv[qq];
                    w.stage_()[64 * NQ + lane + 64 * qq] = x[qq];
                    w.stage_()[128 * NQ + lane + 64 * qq] = z[qq];
                }
                const bool viol = w.template factor<false>(beta2, o.pivot_floor STAMP_PASS);
                if (viol || (o.flags & PYCLLP_FLAG_FORCE_GUARD_PATH)) { stat = -1; running = false; }
                else {
                    double dy[MR], wv[NQ], dx[NQ], e[MR], rhn[MR];
                    double cor[NQ];
                    bool bad;
#pragma unroll
                    for (int r2 = 0; r2 < MR; r2++) rhn[r2] = (lane + 64 * r2 < MP) ? w.flr_()[lane + 64 * r2] : 0.0;
                    if constexpr (PC) {
                        // ---- predictor: um holds A(d t_a) - rho with t_a = cv (mu = 0) ----
                        w.solve();
                        double w2[NQ], dxa[NQ], dza[NQ], tha = 0.0, ga = 0.0;
                        w.At(w.um_(), w2);
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) {
                            const double xq = w.stage_()[64 * NQ + lane + 64 * qq], zq = w.stage_()[128 * NQ + lane + 64 * qq];
                            const double rx = fast_rcp(xq), rz = fast_rcp(zq);
                            const double dq = okc[qq] ? xq * rz : 0.0;
                            const double ta = okc[qq] ? w.vd_()[lane + 64 * qq] : 0.0;
                            dxa[qq] = (ta - w2[qq]) * dq;
                            dza[qq] = okc[qq] ? (-zq * dxa[qq]) * rx - zq : 0.0;
                            if (okc[qq]) tha = fmax(tha, fmax(-dza[qq] * rz, -dxa[qq] * rx));
                        }
                        const double theta_a = uni(fmin(1.0 / wmax(tha), 1.0));
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) {
                            const double xq = w.stage_()[64 * NQ + lane + 64 * qq], zq = w.stage_()[128 * NQ + lane + 64 * qq];
                            ga += okc[qq] ? fma(theta_a, dxa[qq], xq) * fma(theta_a, dza[qq], zq) : 0.0;
                        }
                        const double sgm = wsum(ga) / gam;
                        mu = uni(sgm * sgm * sgm * gam / (double)n);
                        // ---- corrector right-hand side: A(d t_c) - rho, t_c = cv + cor / x ----
                        wave_lds_sync();
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) {
                            const double xq = w.stage_()[64 * NQ + lane + 64 * qq], zq = w.stage_()[128 * NQ + lane + 64 * qq];
                            cor[qq] = okc[qq] ? mu - dxa[qq] * dza[qq] : 0.0;
                            const double tq = okc[qq] ? w.vd_()[lane + 64 * qq] + cor[qq] * fast_rcp(xq) : 0.0;
                            vx[lane + 64 * qq] = (okc[qq] ? xq * fast_rcp(zq) : 0.0) * tq;
                        }
                        wave_lds_sync();
                        double Adt2[MR], dmy[MR];
                        w.template Arow<false>(vx, Adt2, dmy);
#pragma unroll
                        for (int r2 = 0; r2 < MR; r2++) if (lane + 64 * r2 < MP) w.um_()[lane + 64 * r2] = okr[r2] ? Adt2[r2] - rhn[r2] : 0.0;
                        wave_lds_sync();
                    }
                    (void)newton_solve<true, PC>(w, okc, okr, rhn, etol, o.max_refine, mu, dy, dx, wv, e, bad, cor STAMP_PASS);
#pragma unroll
                    for (int qq = 0; qq < NQ; qq++) {
                        cv[qq] = w.vd_()[lane + 64 * qq];
                        x[qq] = w.stage_()[64 * NQ + lane + 64 * qq];
                        z[qq] = w.stage_()[128 * NQ + lane + 64 * qq];
                    }
                    if (bad) { stat = PYCLLP_STATUS_NUMERICAL; running = false; }
                    else {
                        // ---- step (primal_normal.cl:158-198) ----
                        double dz[NQ], th = 0.0;
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) {
                            const double rx = fast_rcp(x[qq]), rz = fast_rcp(z[qq]);
                            dz[qq] = okc[qq] ? ((PC ? cor[qq] : mu) - z[qq] * dx[qq]) * rx - z[qq] : 0.0;
                            if (okc[qq]) th = fmax(th, fmax(-dz[qq] * rz, -dx[qq] * rx));
                        }
                        th = wmax(th);
                        const double theta = uni(fmin(o.r / th, 1.0));
                        wave_lds_sync();
#pragma unroll
                        for (int r2 = 0; r2 < MR; r2++) {
                            const int i = lane + 64 * r2;
                            if (i < MP) {
                                w.ys_()[i] = fma(theta, dy[r2], w.ys_()[i]);
                                w.flr_()[i] = fma(-theta, rhn[r2] - e[r2], rhn[r2]);       // A dx = rho - e
                            }
                        }
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) {
                            w.stage_()[64 * NQ + lane + 64 * qq] = fma(theta, dx[qq], x[qq]);
                            w.stage_()[128 * NQ + lane + 64 * qq] = fma(theta, dz[qq], z[qq]);
                            w.vd_()[lane + 64 * qq] = okc[qq] ? fma(-theta, wv[qq], cv[qq]) : 0.0;
                        }
                        normr0 = normr; norms0 = norms;
                        fresh = false;
                        wave_lds_sync();
                        it++;
                        if (it >= o.max_iter) running = false;   // status stays ITERATION_LIMIT
                        STAMP(9)
                    }
                }
            }
        }
        wave_lds_sync();
        if (stat == -1) {   // the guard would have bitten: hand the LP to the guarded kernel
            if (lane == 0) { const int k = atomicAdd(defer, 1); defer[1 + k] = (int)lp; status[lp] = -1; }
        } else {
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {      // (padded positions and a null z: dropped)
                const unsigned jo = w.coff(qq);
                buf_st(rx, jo, w.stage_()[64 * NQ + lane + 64 * qq] * sb); buf_st(rz, jo, w.stage_()[128 * NQ + lane + 64 * qq] * sc);
            }
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int i = lane + 64 * r2;
                if (yg && okr[r2]) yg[lp * m + i] = w.ys_()[i] * sc;
            }
            if (lane == 0) {
                if (pobj) pobj[lp] = po * (sb * sc);
                if (dobj) dobj[lp] = du * (sb * sc);
                status[lp] = stat;
                if (iters) iters[lp] = it;
            }
        }
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        lp = __builtin_amdgcn_readfirstlane(nxt);
        STAMP(9)
    }
    STAMP_FLUSH(o, blockIdx.x * 4 + (threadIdx.x >> 6))
}

// ------------------------------------------------------------------------------------------------------------------
// the same solve on the homogeneous self-dual embedding (PYCLLP_FLAG_HSD; oracle hsd_one_raw, ipm_block_kernel's run-time
// branch, csrc/ipm_group_hsd.inc): tau and kappa are wave-uniform scalars, one factorisation serves the two right-hand
// sides  M p = A(d c) - b  and  M q = A(d r1) - eta rho, the pivot floor of column j is pivot_floor^2 |M_jj|
// ------------------------------------------------------------------------------------------------------------------
template <int MB, int NQ, bool DA, bool PA>
__global__ void __launch_bounds__(256, 1)
hsd_wreg_kernel(WregTab T, long B, const double* __restrict__ ag, const double* __restrict__ bg, const double* __restrict__ cg,
                double* __restrict__ xg, double* __restrict__ yg, double* __restrict__ zg, double* __restrict__ pobj,
                double* __restrict__ dobj, int* __restrict__ status, int* __restrict__ iters, int* __restrict__ queue,
                int* __restrict__ defer, DevOpts o) {
    using G = WGeo<MB>;
    constexpr int MR = G::MR, MP = G::MP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lraw[];
    WReg<MB, NQ, DA, PA> w;
    USE_AGPR_FORM();
    wreg_setup(w, T, lraw, threadIdx.x);
    const int& lane = w.lane;
    const int m = w.m, n = w.n;
    const bool warm = (o.flags & PYCLLP_FLAG_WARM_START) != 0;
    const bool autoscale = (o.flags & PYCLLP_FLAG_AUTOSCALE) != 0;
    const double eta = 1.0 - o.delta, einf = 100.0 * o.eps;
    double* vx = w.stage_();
    double* pv = w.flr_();          // p = M^-1 (A(d c) - b): the floor vector is dead once the factor exists
    bool okc[NQ], okr[MR];
#pragma unroll
    for (int qq = 0; qq < NQ; qq++) okc[qq] = lane + 64 * qq < n;
#pragma unroll
    for (int r2 = 0; r2 < MR; r2++) okr[r2] = lane + 64 * r2 < m;

    long lp;
    {
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        lp = __builtin_amdgcn_readfirstlane(nxt);
    }
    STAMP_DECL
    while (lp < B) {
        if constexpr (PA) load_lp_values(w, ag, lp, T.nnz);
        double x[NQ], z[NQ];
        double c2 = 0.0, g0 = 0.0;
        const __amdgpu_buffer_rsrc_t rc = row_rsrc(cg + lp * n, n), rx = row_rsrc(xg + lp * n, n), rz = row_rsrc(zg ? zg + lp * n : nullptr, n);
        // PYCLLP_FLAG_AUTOSCALE: as in ipm_wreg_kernel -- the LP is solved with b / max|b| and c / max|c| (the same divisions as
        // ipm_block_kernel and the oracle), undone when storing
        double sb = 1.0, sc = 1.0;
        if (autoscale) {
            double cm = 0.0, bm = 0.0;
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) cm = fmax(cm, fabs(buf_ld(rc, w.coff(qq))));
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) bm = fmax(bm, okr[r2] ? fabs(bg[lp * m + lane + 64 * r2]) : 0.0);
            sb = wmax(bm); sc = wmax(cm);
            sb = uni((sb > 0.0) ? sb : 1.0); sc = uni((sc > 0.0) ? sc : 1.0);
        }
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) {
            const unsigned jo = w.coff(qq);
            double cj = buf_ld(rc, jo);
            if (autoscale) cj = cj / sc;
            c2 = fma(cj, cj, c2);
            x[qq] = (warm && okc[qq]) ? buf_ld(rx, jo) : 1.0;
            z[qq] = (warm && okc[qq]) ? buf_ld(rz, jo) : 1.0;
            if (autoscale && warm) { x[qq] = x[qq] / sb; z[qq] = z[qq] / sc; }
            g0 += okc[qq] ? x[qq] * z[qq] : 0.0;
        }
        double b2 = 0.0;
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            const int i = lane + 64 * r2;
            double bi = okr[r2] ? bg[lp * m + i] : 0.0;
            if (autoscale) bi = bi / sb;
            b2 = fma(bi, bi, b2);
            if (i < MP) {
                double yi = (okr[r2] && warm && yg) ? yg[lp * m + i] : 0.0;
                if (autoscale && warm && yg) yi = yi / sc;
                w.bs_()[i] = bi;
                w.ys_()[i] = yi;
            }
        }
        wave_lds_sync();
        const double nbn = uni(sqrt(wsum(b2))), ncn = uni(sqrt(wsum(c2)));
        const double tol_r = uni(o.eps * (1.0 + nbn)), tol_s = uni(o.eps * (1.0 + ncn));
        double tau = 1.0, kap = 1.0;
        if (warm) kap = uni(wsum(g0) / (double)n);
        double po = 0.0, du = 0.0;
        int stat = PYCLLP_STATUS_ITERATION_LIMIT, it = 0;
        bool running = true;
        // x and z cross the loop's back edge in LDS (the slots where they wait during factor and solves), as in ipm_wreg_kernel
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) {
            w.stage_()[64 * NQ + lane + 64 * qq] = x[qq];
            w.stage_()[128 * NQ + lane + 64 * qq] = z[qq];
        }
        wave_lds_sync();

        while (running) {
            double x[NQ], z[NQ];
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                x[qq] = w.stage_()[64 * NQ + lane + 64 * qq];
                z[qq] = w.stage_()[128 * NQ + lane + 64 * qq];
            }
            wave_lds_sync();
            // ---- sigma = c tau - A'y + z, gamma, objectives ----
            double v[NQ], cq[NQ], sg[NQ];
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) cq[qq] = buf_ld(rc, w.coff(qq));   // in flight (vmcnt) while A'y runs on LDS; 0 in the padded positions
            w.At(w.ys_(), v);
            if (autoscale) {
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) cq[qq] = cq[qq] / sc;
            }
            double s2 = 0.0, gam = 0.0, pp = 0.0;
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                sg[qq] = okc[qq] ? cq[qq] * tau - v[qq] + z[qq] : 0.0;
                s2 = fma(sg[qq], sg[qq], s2);
                gam += okc[qq] ? x[qq] * z[qq] : 0.0;
                pp += cq[qq] * (okc[qq] ? x[qq] : 0.0);
            }
            double dd = 0.0;
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int i = lane + 64 * r2;
                dd += (i < MP) ? w.bs_()[i] * w.ys_()[i] : 0.0;
            }
            s2 = wsum(s2); gam = wsum(gam); po = wsum(pp); du = wsum(dd);
            const double norms = uni(sqrt(s2));
            const double mu = uni(o.delta * (gam + tau * kap) / (double)(n + 1));
            const double phi = uni(du - po + kap);
            // ---- d, t = r1 = mu/x - z + eta sigma; rho = b tau - A x ----
            double t[NQ];
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {
                const int j = lane + 64 * qq;
                const double dq = okc[qq] ? x[qq] * fast_rcp(z[qq]) : 0.0;
                t[qq] = okc[qq] ? fma(eta, sg[qq], mu * fast_rcp(x[qq]) - z[qq]) : 0.0;
                vx[j] = okc[qq] ? x[qq] : 0.0;
                w.vd_()[j] = dq;
            }
            wave_lds_sync();
            double rho[MR], Ax[MR], Md[MR];
            w.template Arow<false>(vx, Ax, Md);
            double r2s = 0.0;
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int i = lane + 64 * r2;
                rho[r2] = okr[r2] ? w.bs_()[i] * tau - Ax[r2] : 0.0;
                r2s = fma(rho[r2], rho[r2], r2s);
            }
            const double normr = uni(sqrt(wsum(r2s)));
            // ---- stop tests (oracle hsd_one_raw): optimal, or a primal / dual ray ----
            const bool p_ray = po > 0.0 && fma(nbn, tau, normr) <= einf * po;
            const bool d_ray = du < 0.0 && fma(ncn, tau, norms) <= einf * -du;
            if (!(isfinite(normr) && isfinite(norms) && isfinite(gam) && isfinite(tau) && isfinite(kap))) { stat = PYCLLP_STATUS_NUMERICAL; running = false; }
            else if (normr <= tol_r * tau && norms <= tol_s * tau && gam <= o.eps * tau * (tau + fabs(po))) { stat = PYCLLP_STATUS_OPTIMAL; running = false; }
            else if (p_ray || d_ray) {
                stat = (p_ray && d_ray) ? ((-du > po) ? PYCLLP_STATUS_PRIMAL_INFEASIBLE : PYCLLP_STATUS_DUAL_INFEASIBLE)
                                        : (p_ray ? PYCLLP_STATUS_DUAL_INFEASIBLE : PYCLLP_STATUS_PRIMAL_INFEASIBLE);
                running = false;
            }
            if (running) {
                // ---- A(d r1), diag(M), A(d c) ----
                wave_lds_sync();
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) vx[lane + 64 * qq] = w.vd_()[lane + 64 * qq] * t[qq];
                wave_lds_sync();
                double Adt[MR], Adc[MR], dummy[MR];
                w.template Arow<true>(vx, Adt, Md);
                wave_lds_sync();
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) vx[lane + 64 * qq] = w.vd_()[lane + 64 * qq] * cq[qq];
                wave_lds_sync();
                w.template Arow<false>(vx, Adc, dummy);
                double rq[MR], bmax = 0.0;
#pragma unroll
                for (int r2 = 0; r2 < MR; r2++) {
                    const int i = lane + 64 * r2;
                    rq[r2] = okr[r2] ? fma(-eta, rho[r2], Adt[r2]) : 0.0;
                    if (i < MP) {
                        w.um_()[i] = okr[r2] ? Adc[r2] - w.bs_()[i] : 0.0;                   // right-hand side of p
                        w.flr_()[i] = o.pivot_floor * o.pivot_floor * fabs(Md[r2]);         // floor of column i
                    }
                    bmax = fmax(bmax, okr[r2] ? fabs(Md[r2]) : 0.0);
                }
                const double beta2 = uni(wmax(bmax));
                wave_lds_sync();
                STAMP(0)
                w.gram(Md);
                STAMP(1)
                // t = r1, x and z wait in the stage while factor and the solves have the registers
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) {
                    w.stage_()[lane + 64 * qq] = t[qq];
                    w.stage_()[64 * NQ + lane + 64 * qq] = x[qq];
                    w.stage_()[128 * NQ + lane + 64 * qq] = z[qq];
                }
                const bool viol = w.template factor<true>(beta2, 0.0 STAMP_PASS);
                if (viol || (o.flags & PYCLLP_FLAG_FORCE_GUARD_PATH)) { stat = -1; running = false; }
                else {
                    // ---- one loop around the ONE copy of the block substitution: pass 0 solves for p, pass 1 for q and
                    //      combines them through dtau, the following passes are the x-space refinement ----
                    // (u = c - A'p waits in d's place between the two solves: d = x / z is formed again from the parked x, z
                    // where a pass needs it; c comes back from memory, in flight while the substitution runs)
                    double dx[NQ], dy[MR], rhot[MR];
                    double dtau = 0.0, etol_it = 0.0;
                    bool bad = false;
                    int pass = 0;
                    for (;;) {
                        double c2q[NQ];
                        if (pass < 2) {
#pragma unroll
                            for (int qq = 0; qq < NQ; qq++) c2q[qq] = buf_ld(rc, w.coff(qq));
                        }
                        w.solve();
                        STAMP(7)
                        double w2[NQ], d[NQ];
                        w.At(w.um_(), w2);
                        if (autoscale && pass < 2) {
#pragma unroll
                            for (int qq = 0; qq < NQ; qq++) c2q[qq] = c2q[qq] / sc;
                        }
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++)
                            d[qq] = okc[qq] ? w.stage_()[64 * NQ + lane + 64 * qq] * fast_rcp(w.stage_()[128 * NQ + lane + 64 * qq]) : 0.0;
                        bool more = true;
                        if (pass == 0) {
                            // c - A'p is kept (in d's place); p moves to pv, q's right-hand side into um
#pragma unroll
                            for (int qq = 0; qq < NQ; qq++) w.vd_()[lane + 64 * qq] = c2q[qq] - w2[qq];
                            wave_lds_sync();
#pragma unroll
                            for (int r2 = 0; r2 < MR; r2++) {
                                const int i = lane + 64 * r2;
                                if (i < MP) { pv[i] = w.um_()[i]; w.um_()[i] = rq[r2]; }
                            }
                            wave_lds_sync();
                        } else {
                            if (pass == 1) {
                                double dsum = 0.0, nsum = 0.0, bq = 0.0;
                                double u[NQ];
#pragma unroll
                                for (int qq = 0; qq < NQ; qq++) {
                                    const double tq = w.stage_()[lane + 64 * qq];
                                    u[qq] = w.vd_()[lane + 64 * qq];
                                    dx[qq] = d[qq] * (tq - w2[qq]);                       // v = d (r1 - A'q)
                                    dsum = fma(d[qq] * u[qq], u[qq], dsum);               // |sqrt(d)(c - A'p)|^2
                                    nsum = fma(c2q[qq], dx[qq], nsum);                    // c'v
                                }
#pragma unroll
                                for (int r2 = 0; r2 < MR; r2++) {
                                    const int i = lane + 64 * r2;
                                    bq += (i < MP) ? w.bs_()[i] * w.um_()[i] : 0.0;       // b'q
                                }
                                const double den = wsum(dsum) + kap / tau;
                                const double num = fma(eta, phi, mu / tau - kap) + wsum(bq) - wsum(nsum);
                                dtau = uni(num / den);
#pragma unroll
                                for (int r2 = 0; r2 < MR; r2++) {
                                    const int i = lane + 64 * r2;
                                    dy[r2] = (i < MP) ? fma(pv[i], dtau, w.um_()[i]) : 0.0;
                                    rhot[r2] = okr[r2] ? fma(w.bs_()[i], dtau, eta * rho[r2]) : 0.0;   // A dx - b dtau = eta rho
                                    bad = bad | !isfinite(dy[r2]);
                                }
#pragma unroll
                                for (int qq = 0; qq < NQ; qq++) dx[qq] = fma(d[qq] * u[qq], dtau, dx[qq]);   // dx = u dtau + v, u = d (c - A'p)
                                etol_it = uni(o.refine_tol * (1.0 + nbn) * fmax(tau, kap));
                            } else {
#pragma unroll
                                for (int qq = 0; qq < NQ; qq++) dx[qq] = fma(d[qq], w2[qq], dx[qq]);
#pragma unroll
                                for (int r2 = 0; r2 < MR; r2++) dy[r2] -= (lane + 64 * r2 < MP) ? w.um_()[lane + 64 * r2] : 0.0;
                            }
                            wave_lds_sync();
#pragma unroll
                            for (int qq = 0; qq < NQ; qq++) vx[lane + 64 * qq] = okc[qq] ? dx[qq] : 0.0;
                            wave_lds_sync();
                            double Adx[MR], e[MR], dm[MR], me = 0.0;
                            w.template Arow<false>(vx, Adx, dm);
#pragma unroll
                            for (int r2 = 0; r2 < MR; r2++) {
                                e[r2] = okr[r2] ? rhot[r2] - Adx[r2] : 0.0;
                                me = fmax(me, fabs(e[r2]));
                            }
                            const double maxe = wmax(me);
                            STAMP(8)
                            if (!(maxe > etol_it) || pass - 1 >= o.max_refine) more = false;
                            else {
#pragma unroll
                                for (int r2 = 0; r2 < MR; r2++) if (lane + 64 * r2 < MP) w.um_()[lane + 64 * r2] = e[r2];
                                wave_lds_sync();
                            }
                        }
                        if (!more) break;
                        pass++;
                    }
                    if (__any(bad) || !isfinite(dtau)) { stat = PYCLLP_STATUS_NUMERICAL; running = false; }
                    else {
                        // ---- step: ratio test over x, z, tau, kappa ----
                        const double dkap = mu / tau - kap - kap / tau * dtau;
                        double dz[NQ], xs[NQ], zs[NQ], th = fmax(fmax(-dtau / tau, -dkap / kap), 0.0);
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) {
                            xs[qq] = w.stage_()[64 * NQ + lane + 64 * qq];
                            zs[qq] = w.stage_()[128 * NQ + lane + 64 * qq];
                        }
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) {
                            const double rx = fast_rcp(xs[qq]), rz = fast_rcp(zs[qq]);
                            dz[qq] = okc[qq] ? (mu - zs[qq] * dx[qq]) * rx - zs[qq] : 0.0;
                            if (okc[qq]) th = fmax(th, fmax(-dz[qq] * rz, -dx[qq] * rx));
                        }
                        th = wmax(th);
                        const double theta = uni(fmin(o.r / th, 1.0));
                        wave_lds_sync();
#pragma unroll
                        for (int r2 = 0; r2 < MR; r2++) {
                            const int i = lane + 64 * r2;
                            if (i < MP) w.ys_()[i] = fma(theta, dy[r2], w.ys_()[i]);
                        }
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) {
                            w.stage_()[64 * NQ + lane + 64 * qq] = fma(theta, dx[qq], xs[qq]);
                            w.stage_()[128 * NQ + lane + 64 * qq] = fma(theta, dz[qq], zs[qq]);
                        }
                        tau = uni(fma(theta, dtau, tau)); kap = uni(fma(theta, dkap, kap));
                        wave_lds_sync();
                        it++;
                        if (it >= o.max_iter) running = false;
                        STAMP(9)
                    }
                }
            }
        }
        wave_lds_sync();
        if (stat == -1) {
            if (lane == 0) { const int k = atomicAdd(defer, 1); defer[1 + k] = (int)lp; status[lp] = -1; }
        } else {
            // optimal (and iteration-limit) points leave the homogeneous scaling (hsd.c:266-273); certificates stay
            const double rt = (stat == PYCLLP_STATUS_OPTIMAL || stat == PYCLLP_STATUS_ITERATION_LIMIT) ? 1.0 / tau : 1.0;
#pragma unroll
            for (int qq = 0; qq < NQ; qq++) {      // (padded positions, null z: dropped)
                const unsigned jo = w.coff(qq);
                buf_st(rx, jo, w.stage_()[64 * NQ + lane + 64 * qq] * rt * sb); buf_st(rz, jo, w.stage_()[128 * NQ + lane + 64 * qq] * rt * sc);
            }
#pragma unroll
            for (int r2 = 0; r2 < MR; r2++) {
                const int i = lane + 64 * r2;
                if (yg && okr[r2]) yg[lp * m + i] = w.ys_()[i] * rt * sc;
            }
            if (lane == 0) {
                if (pobj) pobj[lp] = po * rt * (sb * sc);
                if (dobj) dobj[lp] = du * rt * (sb * sc);
                status[lp] = stat;
                if (iters) iters[lp] = it;
            }
        }
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        lp = __builtin_amdgcn_readfirstlane(nxt);
        STAMP(9)
    }
    STAMP_FLUSH(o, blockIdx.x * 4 + (threadIdx.x >> 6))
}

// ------------------------------------------------------------------------------------------------------------------
// stand-alone Newton step: sparse_solve_primal_normal (ldl.cl:656-712) as launched by the reference's
// tests/test_ldl.py:276-361, one state per wavefront
// ------------------------------------------------------------------------------------------------------------------
template <int MB, int NQ, bool DA>
__global__ void __launch_bounds__(256, 1)
newton_wreg_kernel(WregTab T, long B, const double* __restrict__ xg, const double* __restrict__ zg,
                   const double* __restrict__ yg, const double* __restrict__ bg, const double* __restrict__ cg, double mu,
                   double* __restrict__ dyg, int* __restrict__ nrefg, int* __restrict__ queue, DevOpts o) {
    using G = WGeo<MB>;
    constexpr int MR = G::MR, MP = G::MP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lraw[];
    WReg<MB, NQ, DA> w;
    USE_AGPR_FORM();
    wreg_setup(w, T, lraw, threadIdx.x);
    const int& lane = w.lane; const int m = w.m, n = w.n;
    double* vx = w.stage_();
    bool okc[NQ], okr[MR];
#pragma unroll
    for (int qq = 0; qq < NQ; qq++) okc[qq] = lane + 64 * qq < n;
#pragma unroll
    for (int r2 = 0; r2 < MR; r2++) okr[r2] = lane + 64 * r2 < m;
    long lp;
    {
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        lp = __builtin_amdgcn_readfirstlane(nxt);
    }
    while (lp < B) {
        double x[NQ], z[NQ], t[NQ], v[NQ];
        double b2 = 0.0;
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            const int i = lane + 64 * r2;
            const double bi = okr[r2] ? bg[lp * m + i] : 0.0;
            b2 = fma(bi, bi, b2);
            if (i < MP) { w.bs_()[i] = bi; w.ys_()[i] = okr[r2] ? yg[lp * m + i] : 0.0; }
        }
        wave_lds_sync();
        const double etol = o.refine_tol * (1.0 + sqrt(wsum(b2)));
        w.At(w.ys_(), v);
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) {
            const int j = lane + 64 * qq;
            const unsigned jo = w.coff(qq);
            x[qq] = okc[qq] ? buf_ld(row_rsrc(xg + lp * n, n), jo) : 1.0;
            z[qq] = okc[qq] ? buf_ld(row_rsrc(zg + lp * n, n), jo) : 1.0;
            const double cj = buf_ld(row_rsrc(cg + lp * n, n), jo);
            t[qq] = okc[qq] ? cj - v[qq] + mu * fast_rcp(x[qq]) : 0.0;
            vx[j] = okc[qq] ? x[qq] : 0.0;
            w.vd_()[j] = okc[qq] ? x[qq] * fast_rcp(z[qq]) : 0.0;
        }
        wave_lds_sync();
        double rho[MR], Ax[MR], Adt[MR], Md[MR];
        w.template Arow<false>(vx, Ax, Md);
        wave_lds_sync();
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) vx[lane + 64 * qq] = w.vd_()[lane + 64 * qq] * t[qq];
        wave_lds_sync();
        w.template Arow<true>(vx, Adt, Md);
        double bmax = 0.0;
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) {
            const int i = lane + 64 * r2;
            rho[r2] = okr[r2] ? w.bs_()[i] - Ax[r2] : 0.0;
            if (i < MP) w.um_()[i] = okr[r2] ? Adt[r2] - rho[r2] : 0.0;
            bmax = fmax(bmax, okr[r2] ? fabs(Md[r2]) : 0.0);
        }
        const double beta2 = wmax(bmax);
        wave_lds_sync();
        w.gram(Md);
#pragma unroll
        for (int qq = 0; qq < NQ; qq++) w.stage_()[lane + 64 * qq] = t[qq];
#ifdef PYCLLP_PROFILE
        unsigned long long t_prev_ = 0, t_acc_[NPHASE] = {0};
#endif
        (void)w.template factor<false>(beta2, o.pivot_floor STAMP_PASS);
        double dy[MR], wv[NQ], dx[NQ];
        bool bad;
        double ed[MR];
        const int nref = newton_solve<false>(w, okc, okr, rho, etol, o.max_refine, mu, dy, dx, wv, ed, bad, nullptr STAMP_PASS);
#pragma unroll
        for (int r2 = 0; r2 < MR; r2++) if (okr[r2]) dyg[lp * m + lane + 64 * r2] = dy[r2];
        if (nrefg && lane == 0) nrefg[lp] = nref;
        wave_lds_sync();
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        lp = __builtin_amdgcn_readfirstlane(nxt);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// stand-alone LDL' solve of explicit dense symmetric matrices: the register factor + block substitution on their own
// (pycllp/ldl.py:202-239 solve_ldl; also the bring-up check of factor()/solve())
// ------------------------------------------------------------------------------------------------------------------
template <int MB>
__global__ void __launch_bounds__(256, 1)
ldl_solve_wreg_kernel(int n, long B, const double* __restrict__ Ag, const double* __restrict__ rhs, double* __restrict__ out,
                      double floor_, int* __restrict__ queue) {
    using G = WGeo<MB>;
    constexpr int MP = G::MP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lraw[];
    WReg<MB, 1> w;
    USE_AGPR_FORM();
    const int tid = threadIdx.x;
    wreg_carve(w, (double*)lraw + (size_t)(tid >> 6) * G::WAVE_D(1), tid);
    const int &lane = w.lane, &q = w.q, &c16 = w.c16;
    long mat;
    {
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        mat = __builtin_amdgcn_readfirstlane(nxt);
    }
    while (mat < B) {
        const double* A = Ag + mat * (long)n * n;
        // off-diagonal blocks: U[K][I] register r = M[i = 16I + c16][k = 16K + 4r + q] (i > k); padded rows are zero
        static_for<0, MB>([&](auto Kc) {
            constexpr int K = decltype(Kc)::value;
            static_for<K + 1, MB>([&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
                double4_t blk;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = 16 * I + c16, k = 16 * K + 4 * r + q;
                    blk[r] = (i < n) ? A[(long)i * n + k] : 0.0;
                }
                park(w.P[G::bix(K, I)], blk);
            });
        });
        for (int i = lane; i < MP; i += 64) w.um_()[i] = (i < n) ? rhs[mat * n + i] : 0.0;
        // diagonal blocks into their slots: element [row][col], col <= row, of block K at K WL + row (row + 1) / 2 + col
        static_for<0, MB>([&](auto Kc) {
            constexpr int K = decltype(Kc)::value;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int il = 4 * r + q, row = 16 * K + il, col = 16 * K + c16;
                if (c16 <= il) w.wl_()[K * WL + il * (il + 1) / 2 + c16] = (row < n) ? A[(long)row * n + col] : ((row == col) ? 1.0 : 0.0);
            }
        });
        wave_lds_sync();
#ifdef PYCLLP_PROFILE
        unsigned long long t_prev_ = 0, t_acc_[NPHASE] = {0};
#endif
        (void)w.template factor<false>(1e300, floor_ STAMP_PASS);
        w.solve();
        for (int i = lane; i < n; i += 64) out[mat * n + i] = w.um_()[i];
        wave_lds_sync();
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1);
        mat = __builtin_amdgcn_readfirstlane(nxt);
    }
}

// ---- selftest of the cross-lane primitives (bring-up aid) --------------------------------------------------------
__global__ void wreg_selftest_kernel(double* out) {
    const int lane = threadIdx.x & 63;
    const double v = 1.0 + lane;
    out[lane] = quad_sum(v);                 // expect sum over l' = l mod 16 + 16 k
    out[64 + lane] = row_sum(v);             // expect sum over the 16-lane row
    out[128 + lane] = wsum(v);               // expect 2080
    out[192 + lane] = row_bcast<5>(v);       // expect 1 + (lane & ~15) + 5
    // MFMA layout: A[m][k] = 100 m + k (k = 0..3), B[k][n] = (k == 1) ? n + 1 : 0  ->  C[m][n] = (100 m + 1)(n + 1)
    const int c16 = lane & 15, qd = lane >> 4;
    const double a = 100.0 * c16 + qd, b = (qd == 1) ? c16 + 1.0 : 0.0;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[256 + 64 * r + lane] = acc[r];   // expect (100 (4r + q) + 1)(c16 + 1)
    out[512 + lane] = wmax(v);               // expect 64
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
#ifndef WREG_PART
#error "WREG_PART must be defined before this point"
#endif
struct WregPlan {
    WregTab tab;
    int mb, nq;
    bool da = false, pa = false;
    void* dev_blob;
};

namespace {

template <typename T>
size_t put(std::vector<char>& host, const std::vector<T>& v) {
    size_t off = (host.size() + 15) & ~(size_t)15;
    host.resize(off + v.size() * sizeof(T));
    if (!v.empty()) memcpy(host.data() + off, v.data(), v.size() * sizeof(T));
    return off;
}

template <int MB, int NQ, bool DA, bool PA, bool PC = false>
hipError_t do_solve(const WregTab& T, long B, const double* a, const double* b, const double* c, double* x, double* y, double* z,
                    double* pobj, double* dobj, int* status, int* iters, int* qhead, int* defer, DevOpts o, int grid,
                    hipStream_t st) {
    hipError_t e = set_dyn_lds((const void*)ipm_wreg_kernel<MB, NQ, DA, PA, PC>, T.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ipm_wreg_kernel<MB, NQ, DA, PA, PC>), dim3(grid), dim3(64 * T.wpb), T.lds_bytes, st, T, B, a, b, c, x, y, z, pobj, dobj,
                       status, iters, qhead, defer, o);
    return hipGetLastError();
}
template <int MB, int NQ, bool DA, bool PA>
hipError_t do_solve_hsd(const WregTab& T, long B, const double* a, const double* b, const double* c, double* x, double* y, double* z,
                        double* pobj, double* dobj, int* status, int* iters, int* qhead, int* defer, DevOpts o, int grid,
                        hipStream_t st) {
    hipError_t e = set_dyn_lds((const void*)hsd_wreg_kernel<MB, NQ, DA, PA>, T.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((hsd_wreg_kernel<MB, NQ, DA, PA>), dim3(grid), dim3(64 * T.wpb), T.lds_bytes, st, T, B, a, b, c, x, y, z, pobj, dobj,
                       status, iters, qhead, defer, o);
    return hipGetLastError();
}
template <int MB, int NQ, bool DA>
hipError_t do_newton(const WregTab& T, long B, const double* x, const double* z, const double* y, const double* b,
                     const double* c, double mu, double* dy, int* nref, int* qhead, DevOpts o, int grid, hipStream_t st) {
    hipError_t e = set_dyn_lds((const void*)newton_wreg_kernel<MB, NQ, DA>, T.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((newton_wreg_kernel<MB, NQ, DA>), dim3(grid), dim3(64 * T.wpb), T.lds_bytes, st, T, B, x, z, y, b, c, mu, dy,
                       nref, qhead, o);
    return hipGetLastError();
}

#define WVARIANT(MB, NQ, DA) { MB, NQ, DA, false, do_solve<MB, NQ, DA, false>, do_solve_hsd<MB, NQ, DA, false>, do_newton<MB, NQ, DA> }
#define WVARIANT_PA(MB, NQ) { MB, NQ, false, true, do_solve<MB, NQ, false, true>, do_solve_hsd<MB, NQ, false, true>, nullptr }
// predictor-corrector kernels of the table variants (fourth translation unit): only `solve` is meaningful
#define WVARIANT_PC(MB, NQ, DA) { MB, NQ, DA, false, do_solve<MB, NQ, DA, false, true>, nullptr, nullptr }
#define WVARIANT_PCPA(MB, NQ) { MB, NQ, false, true, do_solve<MB, NQ, false, true, true>, nullptr, nullptr }
// ordered by cost; the first variant of the wanted kind (tables / dense image) with 16 mb >= m and 64 nq >= n is used
#if WREG_PART == 0
#ifdef PYCLLP_DEV_ONLY_W86   // development builds: only the (8, 6) table variant (BASELINE config 5), compiles in a fraction of the time
const WVariant kWVariantsTab[] = { WVARIANT(8, 6, false) };
#else
const WVariant kWVariantsTab[] = { WVARIANT(1, 4, false), WVARIANT(2, 4, false), WVARIANT(3, 4, false), WVARIANT(4, 2, false), WVARIANT(4, 4, false), WVARIANT(5, 6, false), WVARIANT(6, 6, false),
                                   WVARIANT(7, 6, false), WVARIANT(8, 4, false), WVARIANT(8, 6, false), WVARIANT(8, 8, false) };
#endif
const int kNumWVariantsTab = sizeof(kWVariantsTab) / sizeof(kWVariantsTab[0]);
const int kNumWVariants = kNumWVariantsTab + kNumWVariantsDA + kNumWVariantsPA;
struct VariantList {
    const WVariant& operator[](int i) const {
        return i < kNumWVariantsTab ? kWVariantsTab[i]
             : (i < kNumWVariantsTab + kNumWVariantsDA ? kWVariantsDA[i - kNumWVariantsTab] : kWVariantsPA[i - kNumWVariantsTab - kNumWVariantsDA]);
    }
};
const VariantList kWVariants{};
#endif

}  // namespace

#if WREG_PART == 1
#define WVARIANTS_DA { WVARIANT(1, 4, true), WVARIANT(2, 4, true), WVARIANT(3, 4, true), WVARIANT(4, 2, true), WVARIANT(4, 4, true), WVARIANT(5, 4, true), WVARIANT(6, 4, true), \
                       WVARIANT(7, 4, true), WVARIANT(8, 4, true), WVARIANT(8, 6, true) }
#ifdef __HIP_DEVICE_COMPILE__
// device pass: a file-local copy of the table -- it is never emitted, but referencing the launchers is what makes the kernels
// they launch get instantiated (an external table of host function pointers would be emitted into the device object and
// fail to link there)
namespace { [[maybe_unused]] const WVariant kWVariantsDA_instantiate[] = WVARIANTS_DA; }
#else
extern const WVariant kWVariantsDA[] = WVARIANTS_DA;
extern const int kNumWVariantsDA = sizeof(kWVariantsDA) / sizeof(kWVariantsDA[0]);
#endif
#endif
#if WREG_PART == 2
// per-problem values of A: every (MB, NQ) of the table variants, solve + HSD kernels (third translation unit)
#define WVARIANTS_PA { WVARIANT_PA(1, 4), WVARIANT_PA(2, 4), WVARIANT_PA(3, 4), WVARIANT_PA(4, 2), WVARIANT_PA(4, 4), WVARIANT_PA(5, 6), \
                       WVARIANT_PA(6, 6), WVARIANT_PA(7, 6), WVARIANT_PA(8, 4), WVARIANT_PA(8, 6), WVARIANT_PA(8, 8) }
#ifdef __HIP_DEVICE_COMPILE__
namespace { [[maybe_unused]] const WVariant kWVariantsPA_instantiate[] = WVARIANTS_PA; }
#else
extern const WVariant kWVariantsPA[] = WVARIANTS_PA;
extern const int kNumWVariantsPA = sizeof(kWVariantsPA) / sizeof(kWVariantsPA[0]);
#endif
#endif
#if WREG_PART == 3
#define WVARIANTS_PC { WVARIANT_PC(1, 4, false), WVARIANT_PC(2, 4, false), WVARIANT_PC(3, 4, false), WVARIANT_PC(4, 2, false), WVARIANT_PC(4, 4, false), \
                       WVARIANT_PC(5, 6, false), WVARIANT_PC(6, 6, false), WVARIANT_PC(7, 6, false), WVARIANT_PC(8, 4, false), WVARIANT_PC(8, 6, false), \
                       WVARIANT_PC(8, 8, false) }
#ifdef __HIP_DEVICE_COMPILE__
namespace { [[maybe_unused]] const WVariant kWVariantsPC_instantiate[] = WVARIANTS_PC; }
#else
extern const WVariant kWVariantsPC[] = WVARIANTS_PC;
extern const int kNumWVariantsPC = sizeof(kWVariantsPC) / sizeof(kWVariantsPC[0]);
#endif
#endif
#if WREG_PART == 4
// predictor-corrector kernels of the dense-image variants (fifth translation unit)
#define WVARIANTS_PCDA { WVARIANT_PC(1, 4, true), WVARIANT_PC(2, 4, true), WVARIANT_PC(3, 4, true), WVARIANT_PC(4, 2, true), WVARIANT_PC(4, 4, true), \
                         WVARIANT_PC(5, 4, true), WVARIANT_PC(6, 4, true), WVARIANT_PC(7, 4, true), WVARIANT_PC(8, 4, true), WVARIANT_PC(8, 6, true) }
#ifdef __HIP_DEVICE_COMPILE__
namespace { [[maybe_unused]] const WVariant kWVariantsPCDA_instantiate[] = WVARIANTS_PCDA; }
#else
extern const WVariant kWVariantsPCDA[] = WVARIANTS_PCDA;
extern const int kNumWVariantsPCDA = sizeof(kWVariantsPCDA) / sizeof(kWVariantsPCDA[0]);
#endif
#endif
#if WREG_PART == 5
// predictor-corrector kernels of the per-problem-A variants (sixth translation unit)
#define WVARIANTS_PCPA { WVARIANT_PCPA(1, 4), WVARIANT_PCPA(2, 4), WVARIANT_PCPA(3, 4), WVARIANT_PCPA(4, 2), WVARIANT_PCPA(4, 4), WVARIANT_PCPA(5, 6), \
                         WVARIANT_PCPA(6, 6), WVARIANT_PCPA(7, 6), WVARIANT_PCPA(8, 4), WVARIANT_PCPA(8, 6), WVARIANT_PCPA(8, 8) }
#ifdef __HIP_DEVICE_COMPILE__
namespace { [[maybe_unused]] const WVariant kWVariantsPCPA_instantiate[] = WVARIANTS_PCPA; }
#else
extern const WVariant kWVariantsPCPA[] = WVARIANTS_PCPA;
extern const int kNumWVariantsPCPA = sizeof(kWVariantsPCPA) / sizeof(kWVariantsPCPA[0]);
#endif
#endif
#if WREG_PART == 0

// Plan with A as a dense image in LDS (no tables): for matrices whose Gram term list does not fit -- dense A's, e.g. the LPs
// hip_dense_primal_normal hands over beyond m = 32.  The last m columns are kept out of the image when they are the
// identity (equality form of a StandardLP).  Fewer than four waves per workgroup when that is what lets the image fit.
static int wreg_plan_create_dense(int m, int n, int nnz, const double* val, const int* ptr, const int* col, int max_lds,
                                  hipStream_t st, WregPlan** out) {
    int vi = -1;
    for (int i = 0; i < kNumWVariants; i++)
        if (kWVariants[i].da && !kWVariants[i].pa && m <= 16 * kWVariants[i].mb && n <= 64 * kWVariants[i].nq) { vi = i; break; }
    if (vi < 0) return 1;
    const int MB = kWVariants[vi].mb, NQ = kWVariants[vi].nq, MP = 16 * MB;
    // identity tail?
    bool sl = n > m;
    std::vector<int> tail_cnt(sl ? m : 0, 0);
    for (int i = 0; i < m && sl; i++)
        for (int e = ptr[i]; e < ptr[i + 1]; e++)
            if (col[e] >= n - m) { if (col[e] != n - m + i || val[e] != 1.0) sl = false; else tail_cnt[i]++; }
    for (int i = 0; i < m && sl; i++) if (tail_cnt[i] != 1) sl = false;
    const int nd = sl ? n - m : n, ndp = ((std::max(nd, 1) + 15) / 16) * 16, AS = ndp + 1, R = ((m + 7) / 8) * 8;   // (16: gram_dense takes four k-steps per trip)
    WregPlan* P = new WregPlan();
    WregTab& T = P->tab;
    memset(&T, 0, sizeof(T));
    T.m = m; T.n = n; T.nnz = nnz; T.nd = nd; T.as = AS; T.img_rows = R;
    T.wave_doubles = stage_d(NQ) + 64 * NQ + 5 * MP + MB * WL;
    const size_t img_bytes = sizeof(double) * (size_t)R * AS;
    int wpb = 0;
    for (int w = 4; w >= 1; w--)
        if (img_bytes + 16 + sizeof(double) * (size_t)w * T.wave_doubles <= (size_t)max_lds) { wpb = w; break; }
    if (!wpb) { delete P; return 1; }
    T.wpb = wpb; T.o_img = 0; T.o_wave = (int)((img_bytes + 15) & ~(size_t)15);
    T.lds_bytes = T.o_wave + (int)(sizeof(double) * (size_t)wpb * T.wave_doubles);
    std::vector<double> img((size_t)R * AS, 0.0);
    for (int i = 0; i < m; i++)
        for (int e = ptr[i]; e < ptr[i + 1]; e++)
            if (col[e] < nd) img[(size_t)i * AS + col[e]] = val[e];
    hipError_t e = hipMalloc(&P->dev_blob, img.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(P->dev_blob, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { if (P->dev_blob) (void)hipFree(P->dev_blob); delete P; return 1000 + (int)e; }
    T.img = (const double*)P->dev_blob;
    P->mb = MB; P->nq = NQ; P->da = true;
    *out = P;
    return 0;
}

static int wreg_plan_create_tables(int m, int n, int nnz, const double* val, const int* ptr, const int* col, int max_lds,
                                   bool pa, hipStream_t st, WregPlan** out);

int wreg_plan_create(int m, int n, int nnz, const double* val, const int* ptr, const int* col, int max_lds, int pa,
                     hipStream_t st, WregPlan** out) {
    if (pa) return wreg_plan_create_tables(m, n, nnz, val, ptr, col, max_lds, true, st, out);
    const int rc = wreg_plan_create_tables(m, n, nnz, val, ptr, col, max_lds, false, st, out);
    if (rc != 1) return rc;
    return wreg_plan_create_dense(m, n, nnz, val, ptr, col, max_lds, st, out);
}

// pa: the structure-only tables of the per-problem-A variants (the values `val` only stand in where a table still wants
// one; no kernel of those variants reads them)
static int wreg_plan_create_tables(int m, int n, int nnz, const double* val, const int* ptr, const int* col, int max_lds,
                                   bool pa, hipStream_t st, WregPlan** out) {
    int vi = -1;
    for (int i = 0; i < kNumWVariants; i++)
        if (!kWVariants[i].da && kWVariants[i].pa == pa && m <= 16 * kWVariants[i].mb && n <= 64 * kWVariants[i].nq) { vi = i; break; }
    if (vi < 0 || nnz >= 65535) return 1;
    const int MB = kWVariants[vi].mb, NQ = kWVariants[vi].nq;
    const int MP = 16 * MB, MPL = 64 * ((MP + 63) / 64), NP = 64 * NQ;
    WregPlan* P = new WregPlan();
    WregTab& T = P->tab;
    memset(&T, 0, sizeof(T));
    T.m = m; T.n = n; T.nnz = nnz;
    // ---- column positions: the columns sorted by length (longest first) are dealt to positions 0, 1, ...; position p
    //      is element p % 64 of N-vector register p / 64, so every register holds 64 columns of similar length ----
    std::vector<int> cptr(n + 1, 0), crow(nnz), csc_src(nnz);
    std::vector<double> csc_val(nnz);
    for (int e = 0; e < nnz; e++) cptr[col[e] + 1]++;
    for (int j = 0; j < n; j++) cptr[j + 1] += cptr[j];
    {
        std::vector<int> fill(cptr.begin(), cptr.end() - 1);
        for (int i = 0; i < m; i++)
            for (int e = ptr[i]; e < ptr[i + 1]; e++) { const int p = fill[col[e]]++; crow[p] = i; csc_val[p] = val[e]; csc_src[p] = e; }
    }
    std::vector<int> order(n), posof(n);
    for (int j = 0; j < n; j++) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cptr[a + 1] - cptr[a] > cptr[b + 1] - cptr[b]; });
    std::vector<unsigned> colmap(NP, PAD_OFF);
    for (int p = 0; p < n; p++) { colmap[p] = 8u * (unsigned)order[p]; posof[order[p]] = p; }
    // ---- A by rows (column indices = positions), compact ----
    std::vector<double> csr_val(val, val + nnz);
    std::vector<unsigned short> csr_col(nnz), csr_ptr(MPL, 0), csr_len(MPL, 0);
    int rmax = 0;
    for (int i = 0; i < m; i++) {
        csr_ptr[i] = (unsigned short)ptr[i]; csr_len[i] = (unsigned short)(ptr[i + 1] - ptr[i]);
        rmax = std::max(rmax, ptr[i + 1] - ptr[i]);
    }
    for (int e = 0; e < nnz; e++) csr_col[e] = (unsigned short)posof[col[e]];
    T.rmax = rmax;
    // ---- A by columns, ELL over positions ----
    int ctot = 0;
    for (int q = 0; q < NQ; q++) {
        int cm = 0;
        for (int p = 64 * q; p < std::min(n, 64 * q + 64); p++) cm = std::max(cm, cptr[order[p] + 1] - cptr[order[p]]);
        T.meta[q] = cm; T.meta[META_COFF + q] = ctot; ctot += cm;
    }
    T.ctot = ctot;
    std::vector<double> ec_val((size_t)std::max(ctot, 1) * 64, 0.0);
    std::vector<unsigned short> ec_row((size_t)std::max(ctot, 1) * 64, 0);
    std::vector<unsigned short> ec_src((size_t)std::max(ctot, 1) * 64, (unsigned short)nnz);      // padded slots: the zero entry
    for (int p = 0; p < n; p++) {
        const int j = order[p], q = p / 64, l = p % 64;
        for (int e = cptr[j], t = 0; e < cptr[j + 1]; e++, t++) {
            ec_val[(size_t)(T.meta[META_COFF + q] + t) * 64 + l] = csc_val[e];
            ec_row[(size_t)(T.meta[META_COFF + q] + t) * 64 + l] = (unsigned short)crow[e];
            ec_src[(size_t)(T.meta[META_COFF + q] + t) * 64 + l] = (unsigned short)csc_src[e];
        }
    }
    // ---- Gram entries (strictly lower triangle of M): off-diagonal blocks grouped by staging chunk (HB blocks of the
    //      linear block order each); the entries inside the diagonal blocks go to their slots in the W area, as part of
    //      the last chunk's group when that chunk ends in front of the W area, else as a group of their own.  All
    //      destinations are offsets from the start of the stage. ----
    struct Term { int group, dst, colj; double w; int ia, ib; };
    std::vector<Term> terms;
    const int nblk = MB * (MB - 1) / 2, nchunk = (nblk + HB - 1) / HB;
    const bool merge_diag = nblk > 0 && (nblk - HB * (nchunk - 1)) * 256 <= stage_d(NQ);      // = WGeo<MB>::MERGE_DIAG(NQ)
    const int diag_group = merge_diag ? nchunk - 1 : nchunk, ngroup = merge_diag ? nchunk : nchunk + 1;
    if (ngroup + 1 > META_N - META_SEG || stage_d(NQ) + MB * WL > 65535) { delete P; return 1; }
    for (int j = 0; j < n; j++)
        for (int a = cptr[j]; a < cptr[j + 1]; a++)
            for (int b2 = cptr[j]; b2 < a; b2++) {
                const int i = crow[a], k = crow[b2];   // rows ascend inside a column: i > k
                const int K = k / 16, I = i / 16, il = i % 16, kl = k % 16;
                if (I == K) {          // diagonal block K: element [il][kl] of the packed lower triangle in slot K
                    terms.push_back({diag_group, stage_d(NQ) + K * WL + il * (il + 1) / 2 + kl, posof[j], csc_val[a] * csc_val[b2], csc_src[a], csc_src[b2]});
                } else {               // block (K, I) of U: element [kl][il] of block bix % HB of chunk bix / HB
                    const int bx = K * MB - K * (K + 1) / 2 + (I - K - 1);
                    terms.push_back({bx / HB, (bx % HB) * 256 + kl * 16 + il, posof[j], csc_val[a] * csc_val[b2], csc_src[a], csc_src[b2]});
                }
                if (terms.size() > ((size_t)1 << 22)) { delete P; return 1; }
            }
    // Records of a group: first the FIRST term of every entry (one flat pass), then rounds of triples -- terms 2..4 of the
    // entries that have them, then terms 5..7, ... -- in column order (the order a per-entry loop would add them in), short
    // triples padded with weight 0.  lev[] holds the record boundaries, meta[META_SEG + g] the first boundary of group g.
    std::stable_sort(terms.begin(), terms.end(), [](const Term& a, const Term& b) {
        return a.group != b.group ? a.group < b.group : a.dst < b.dst; });
    std::vector<double> t_w;
    std::vector<unsigned> t_cd, t_ab;
    const unsigned ab_zero = (unsigned)nnz | ((unsigned)nnz << 16);       // padded records: 0 x 0
    std::vector<int> lev(1, 0), gl(ngroup + 1, 0);
    {
        size_t t0 = 0;
        for (int g = 0; g < ngroup; g++) {
            size_t t1 = t0;
            while (t1 < terms.size() && terms[t1].group == g) t1++;
            gl[g] = (int)lev.size() - 1;
            // entries of the group: [e0, e1) term ranges
            std::vector<std::pair<size_t, size_t>> ent;
            for (size_t t = t0; t < t1;) {
                size_t u = t + 1;
                while (u < t1 && terms[u].dst == terms[t].dst) u++;
                ent.push_back({t, u}); t = u;
            }
            if (!ent.empty()) {
                for (auto& e : ent) {
                    t_w.push_back(terms[e.first].w); t_cd.push_back((unsigned)terms[e.first].colj | ((unsigned)terms[e.first].dst << 16));
                    t_ab.push_back((unsigned)terms[e.first].ia | ((unsigned)terms[e.first].ib << 16));
                }
                lev.push_back((int)t_w.size());
                for (size_t r = 0;; r++) {
                    bool any = false;
                    for (auto& e : ent) {
                        const size_t f = e.first + 1 + 3 * r;
                        if (f >= e.second) continue;
                        any = true;
                        for (size_t k = 0; k < 3; k++) {
                            const bool have = f + k < e.second;
                            t_w.push_back(have ? terms[f + k].w : 0.0);
                            t_cd.push_back((have ? (unsigned)terms[f + k].colj : 0u) | ((unsigned)terms[e.first].dst << 16));
                            t_ab.push_back(have ? ((unsigned)terms[f + k].ia | ((unsigned)terms[f + k].ib << 16)) : ab_zero);
                        }
                    }
                    if (!any) break;
                    lev.push_back((int)t_w.size());
                }
            }
            t0 = t1;
        }
        gl[ngroup] = (int)lev.size() - 1;
    }
    for (int g = 0; g <= ngroup; g++) T.meta[META_SEG + g] = gl[g];
    T.n_lev = (int)lev.size() - 1; T.n_term = (int)t_w.size();
    if (t_w.empty()) { t_w.push_back(0.0); t_cd.push_back(0); t_ab.push_back(ab_zero); }
    // ---- LDS plan ----
    // PA: no value tables; every wave carries nnzp doubles of its LP's values behind its area, and as many waves share a
    // workgroup as the LDS takes (three at config 5's structure: 3 x 37.6 KB + 22 KB of structure tables)
    T.pa = pa ? 1 : 0;
    T.nnzp = pa ? ((nnz + 1 + 1) & ~1) : 0;
    T.wave_doubles = stage_d(NQ) + 64 * NQ + 5 * MP + MB * WL + T.nnzp;
    int wpb = 4;
    for (;;) {
        size_t off = 0;
        auto take = [&](size_t bytes) { off = (off + 15) & ~(size_t)15; const size_t o_ = off; off += bytes; return (int)o_; };
        if (!pa) {
            T.o_csr_val = take(sizeof(double) * csr_val.size());
            T.o_ec_val = take(sizeof(double) * ec_val.size());
            T.o_t_w = take(sizeof(double) * t_w.size());
        }
        T.o_wave = take(sizeof(double) * (size_t)wpb * (size_t)T.wave_doubles);
        T.o_lev = take(sizeof(int) * lev.size());
        T.o_t_cd = take(sizeof(unsigned) * t_cd.size());
        if (pa) T.o_t_ab = take(sizeof(unsigned) * t_ab.size());
        T.o_colmap = take(sizeof(unsigned) * colmap.size());
        T.o_meta = take(sizeof(int) * META_N);
        T.o_csr_col = take(sizeof(unsigned short) * csr_col.size());
        T.o_csr_ptr = take(sizeof(unsigned short) * csr_ptr.size());
        T.o_csr_len = take(sizeof(unsigned short) * csr_len.size());
        T.o_ec_row = take(sizeof(unsigned short) * ec_row.size());
        if (pa) T.o_ec_src = take(sizeof(unsigned short) * ec_src.size());
        T.lds_bytes = (int)((off + 15) & ~(size_t)15);
        if (T.lds_bytes <= max_lds) break;
        if (!pa || wpb == 1) { delete P; return 1; }
        wpb--;
    }
    // ---- device copies ----
    std::vector<char> host;
    const size_t a1 = put(host, csr_val), a2 = put(host, ec_val), a3 = put(host, t_w), a4 = put(host, lev),
                 a5 = put(host, csr_col), a6 = put(host, csr_ptr), a7 = put(host, csr_len), a8 = put(host, ec_row),
                 a9 = put(host, colmap), a11 = put(host, t_cd), a12 = put(host, ec_src), a13 = put(host, t_ab);
    hipError_t e = hipMalloc(&P->dev_blob, host.size());
    if (e == hipSuccess) e = hipMemcpyAsync(P->dev_blob, host.data(), host.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { if (P->dev_blob) (void)hipFree(P->dev_blob); delete P; return 1000 + (int)e; }
    char* db = (char*)P->dev_blob;
    T.csr_val = (const double*)(db + a1); T.ec_val = (const double*)(db + a2); T.t_w = (const double*)(db + a3);
    T.lev = (const int*)(db + a4);
    T.csr_col = (const unsigned short*)(db + a5); T.csr_ptr = (const unsigned short*)(db + a6);
    T.csr_len = (const unsigned short*)(db + a7); T.ec_row = (const unsigned short*)(db + a8);
    T.colmap = (const unsigned*)(db + a9);
    T.t_cd = (const unsigned*)(db + a11);
    T.ec_src = (const unsigned short*)(db + a12); T.t_ab = (const unsigned*)(db + a13);
    P->mb = MB; P->nq = NQ; P->pa = pa; T.wpb = wpb;
    *out = P;
    return 0;
}

void wreg_plan_free(WregPlan* p) {
    if (!p) return;
    if (p->dev_blob) (void)hipFree(p->dev_blob);
    delete p;
}

int wreg_lds_bytes(const WregPlan* p) { return p ? p->tab.lds_bytes : 0; }
int wreg_block_threads(const WregPlan* p) { return p ? 64 * p->tab.wpb : 0; }
int wreg_variant(const WregPlan* p) { return p ? (p->da ? 2 : 1) : 0; }
int wreg_has_predcorr(const WregPlan* p) { return p ? 1 : 0; }

static const WVariant* find_variant(const WregPlan* p) {
    for (int i = 0; i < kNumWVariants; i++)
        if (kWVariants[i].mb == p->mb && kWVariants[i].nq == p->nq && kWVariants[i].da == p->da && kWVariants[i].pa == p->pa) return &kWVariants[i];
    return nullptr;
}

hipError_t wreg_launch_solve(WregPlan* p, long B, const double* a_batch, const double* b, const double* c, double* x, double* y, double* z,
                             double* pobj, double* dobj, int* status, int* iters, int* qhead, int* defer, DevOpts o,
                             int num_cu, hipStream_t st, int* grid_out) {
    const WVariant* v = find_variant(p);
    if (!v || (p->pa != (a_batch != nullptr))) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(defer, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    long cus = (long)num_cu - o.reserve_cus > 0 ? (long)num_cu - o.reserve_cus : 1;
    // the m <= 64 variants need fewer than half the registers (234 of 512 per lane): two workgroups per CU -- two waves per
    // SIMD -- where the LDS allows it
    if (p->mb <= 4 && 2 * (long)p->tab.lds_bytes <= 160 * 1024 && p->tab.wpb == 4) cus *= 2;
    long grid = std::min(cus, (B + p->tab.wpb - 1) / p->tab.wpb);
    if (grid < 1) grid = 1;
    if (grid_out) *grid_out = (int)grid;
    wsolve_fn fn = (o.flags & PYCLLP_FLAG_HSD) ? v->solve_hsd : v->solve;
    if ((o.flags & PYCLLP_FLAG_PREDCORR) && !(o.flags & PYCLLP_FLAG_HSD)) {
        fn = nullptr;
        const WVariant* list = p->pa ? kWVariantsPCPA : (p->da ? kWVariantsPCDA : kWVariantsPC);
        const int nlist = p->pa ? kNumWVariantsPCPA : (p->da ? kNumWVariantsPCDA : kNumWVariantsPC);
        for (int i = 0; i < nlist; i++)
            if (list[i].mb == p->mb && list[i].nq == p->nq) fn = list[i].solve;
        if (!fn) return hipErrorNotSupported;
    }
    return fn(p->tab, B, a_batch, b, c, x, y, z, pobj, dobj, status, iters, qhead, defer, o, (int)grid, st);
}

hipError_t wreg_launch_newton(WregPlan* p, long B, const double* x, const double* z, const double* y, const double* b,
                              const double* c, double mu, double* dy, int* nref, DevOpts o, int num_cu, hipStream_t st) {
    const WVariant* v = find_variant(p);
    if (!v || !v->newton) return hipErrorInvalidValue;
    int* qhead = nullptr;
    hipError_t e = hipMallocAsync((void**)&qhead, sizeof(int), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(qhead, 0, sizeof(int), st);
    long grid = std::min((long)num_cu, (B + p->tab.wpb - 1) / p->tab.wpb);
    if (grid < 1) grid = 1;
    if (e == hipSuccess) e = v->newton(p->tab, B, x, z, y, b, c, mu, dy, nref, qhead, o, (int)grid, st);
    hipError_t e2 = hipFreeAsync(qhead, st);
    return e != hipSuccess ? e : e2;
}

hipError_t wreg_launch_ldl_solve(int n, long B, const double* A, const double* rhs, double* out, double floor_,
                                 int num_cu, hipStream_t st) {
    if (n > 128) return hipErrorInvalidValue;
    int* qhead = nullptr;
    hipError_t e = hipMallocAsync((void**)&qhead, sizeof(int), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(qhead, 0, sizeof(int), st);
    long grid = std::min((long)num_cu, (B + 3) / 4);
    if (grid < 1) grid = 1;
    const int lds = (int)(sizeof(double) * 4 * WGeo<8>::WAVE_D(1));
    if (e == hipSuccess) e = set_dyn_lds((const void*)ldl_solve_wreg_kernel<8>, lds);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((ldl_solve_wreg_kernel<8>), dim3((unsigned)grid), dim3(256), lds, st, n, B, A, rhs, out, floor_, qhead);
        e = hipGetLastError();
    }
    hipError_t e2 = hipFreeAsync(qhead, st);
    return e != hipSuccess ? e : e2;
}

extern "C" int pycllp_hip_debug_wreg_selftest(double* out_dev, void* stream) {
    hipLaunchKernelGGL(wreg_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev);
    return (int)hipGetLastError();
}
#endif  // WREG_PART == 0
