// wave_common.h -- device helpers shared by the translation units of libpycllp_hip.so (ipm_dense.hip, ipm_wreg.hip).
#ifndef PYCLLP_WAVE_COMMON_H
#define PYCLLP_WAVE_COMMON_H
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <type_traits>
#include <vector>

#include "../../include/pycllp_hip.h"

// The 10x-growth exits (primal_normal.cl:261-269) keep the reference's own floor -- EPS = 1e-7f absolute, which is
// 1e3 x the relative stopping tolerance used here (for |b|, |c| ~ 1) -- instead of the stopping tolerance itself:
// within a factor 1000 of convergence a residual is rounding noise, and a 10x bump of noise is not divergence.
#define PYCLLP_GROWTH_FLOOR 1e3

typedef double double4_t __attribute__((ext_vector_type(4)));

// hipFuncAttributeMaxDynamicSharedMemorySize, set once per (device, kernel, size): the attribute call costs several
// microseconds of host time, which a small-batch "repeat solve" caller would pay on every launch
inline hipError_t set_dyn_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::unordered_map<unsigned long long, int> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long key = (unsigned long long)(size_t)fn ^ ((unsigned long long)dev << 56);
    std::lock_guard<std::mutex> g(mu);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done[key] = bytes;
    return e;
}

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_d(double v, int srclane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

// 1/a for a normal, positive a: v_rcp_f64 seed + two Newton steps (<= 1 ulp); skips the scaling/fix-up of an
// IEEE division, which the LDL' pivots (floored at pivot_floor) never need
// opaque copies of wave-uniform values that live in SGPRs (kernel arguments): what is derived from them is re-derived at
// every use instead of being hoisted out of a persistent loop into long-lived VGPRs
__device__ __forceinline__ double sopaque(double v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ int iopaque(int v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ double fast_rcp(double a) {
    double r = __builtin_amdgcn_rcp(a);
    r = fma(r, fma(-a, r, 1.0), r);
    r = fma(r, fma(-a, r, 1.0), r);
    return r;
}

__device__ __forceinline__ double wave_sum(double v) {
    asm volatile("" : "+v"(v));   // no fma contraction of the first stage with v's producer: see grp_sum
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
    return v;
}

// Wave-private LDS hand-off: DS operations of one wave execute in order, so only the compiler has
// to be kept from moving a read above the write it depends on.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct DevOpts {
    double eps, delta, r, pivot_floor, refine_tol;
    int max_iter, max_refine, flags;
    int reserve_cus;           // host side only: CUs left idle by the launch plan
    unsigned long long* prof;  // diagnostic build only (-DPYCLLP_PROFILE): per-wave phase cycle sums
};

// In-kernel phase stamps (diagnostic build only; the shipped library has no stamp executing).
#ifdef PYCLLP_PROFILE
#define NPHASE 12
#ifdef PYCLLP_PROFILE_LITE   // fewer live counters: the full set costs registers and distorts a kernel at the VGPR limit
#define PHASE_MAP(i) ((i) <= 1 ? 0 : (i) == 2 ? 2 : (i) <= 4 ? 4 : (i) == 5 ? 5 : (i) <= 7 ? 6 : 8)
#else
#define PHASE_MAP(i) (i)
#endif
#define STAMP_DECL unsigned long long t_prev_ = 0, t_acc_[NPHASE] = {0}; \
    { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev_) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define STAMP(i) { unsigned long long t_now_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    t_acc_[PHASE_MAP(i)] += t_now_ - t_prev_; t_prev_ = t_now_; }
#define STAMP_ARGS , unsigned long long& t_prev_, unsigned long long (&t_acc_)[NPHASE]
#define STAMP_PASS , t_prev_, t_acc_
#define STAMP_FLUSH(o, wid) if ((o).prof && lane == 0) { for (int i_ = 0; i_ < NPHASE; i_++) (o).prof[(size_t)(wid) * NPHASE + i_] = t_acc_[i_]; }
#define STAMP_FLUSH_BLOCK(o, wid) for (int i_ = 0; i_ < NPHASE; i_++) (o).prof[(size_t)(wid) * NPHASE + i_] = t_acc_[i_];
#else
#define STAMP_FLUSH_BLOCK(o, wid)
#define STAMP_DECL
#define STAMP(i)
#define STAMP_ARGS
#define STAMP_PASS
#define STAMP_FLUSH(o, wid)
#endif

// ---- DPP / swizzle helpers -------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swz_xor16_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F);
    hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
    return __hiloint2double(hi, lo);
}
// compile-time loop (DPP controls must be immediates)
template <int K, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (K < N) {
        f(std::integral_constant<int, K>{});
        static_for<K + 1, N>(f);
    }
}
// value of lane K of the caller's own 16-lane row (row_newbcast)
template <int K>
__device__ __forceinline__ double row_bcast(double v) { return dpp_d<0x150 + K>(v); }

// ---- 64-bit DPP (gfx90a+: the DP ALU takes row_newbcast) ---------------------------------------------------------
// The compiler builds a row broadcast of a double from two v_mov_b32_dpp and feeds the copy to the FMA; the hardware can
// broadcast inside the 64-bit instruction itself.  The s_nop covers the VALU-write -> DPP-read hazard (2 wait states),
// which the hazard recogniser does not see through inline asm.
#define DPP_STR_(x) #x
#define DPP_STR(x) DPP_STR_(x)
// v of lane K of each 16-lane row
template <int K>
__device__ __forceinline__ double bcast64(double v) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" DPP_STR(%2) " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
    return r;
}
// Pivot step J of the 16-step chain: Wd[k] += (u of lane k of the row) * nli for k = J+1 .. 15, one fused 64-bit DPP FMA
// each (J = -1: all sixteen, k = 0 .. 15).  ONE asm statement per step, with a leading s_nop: the compiler may have produced u (or moved it between register
// files) in the instruction just before, and the hazard recogniser does not look inside inline asm.
template <int J>
__device__ __forceinline__ void chain_step(double (&Wd)[16], double u, double nli) {
    asm volatile("s_nop 1\n\t"
                 ".if 0 > %18\n\tv_fmac_f64_dpp %0, %16, %17 row_newbcast:0 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 1 > %18\n\tv_fmac_f64_dpp %1, %16, %17 row_newbcast:1 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 2 > %18\n\tv_fmac_f64_dpp %2, %16, %17 row_newbcast:2 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 3 > %18\n\tv_fmac_f64_dpp %3, %16, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 4 > %18\n\tv_fmac_f64_dpp %4, %16, %17 row_newbcast:4 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 5 > %18\n\tv_fmac_f64_dpp %5, %16, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 6 > %18\n\tv_fmac_f64_dpp %6, %16, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 7 > %18\n\tv_fmac_f64_dpp %7, %16, %17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 8 > %18\n\tv_fmac_f64_dpp %8, %16, %17 row_newbcast:8 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 9 > %18\n\tv_fmac_f64_dpp %9, %16, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 10 > %18\n\tv_fmac_f64_dpp %10, %16, %17 row_newbcast:10 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 11 > %18\n\tv_fmac_f64_dpp %11, %16, %17 row_newbcast:11 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 12 > %18\n\tv_fmac_f64_dpp %12, %16, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 13 > %18\n\tv_fmac_f64_dpp %13, %16, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 14 > %18\n\tv_fmac_f64_dpp %14, %16, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 15 > %18\n\tv_fmac_f64_dpp %15, %16, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 : "+v"(Wd[0]), "+v"(Wd[1]), "+v"(Wd[2]), "+v"(Wd[3]), "+v"(Wd[4]), "+v"(Wd[5]), "+v"(Wd[6]), "+v"(Wd[7]), "+v"(Wd[8]), "+v"(Wd[9]), "+v"(Wd[10]), "+v"(Wd[11]), "+v"(Wd[12]), "+v"(Wd[13]), "+v"(Wd[14]), "+v"(Wd[15])
                 : "v"(u), "v"(nli), "n"(J));
}

#include "chain_asm.inc"

// ---- the same steps under an EXEC mask (ipm_group.inc, round 3) ----
// In the lane-group kernel the mask "lanes below the pivot" and the one-hot "the pivot's lane" are compile-time constants per
// column; taking them as EXEC masks replaces, per column, the compare + two selects that zero the multiplier of the lanes
// on and above the pivot, the compare + two selects that pick the pivot's reciprocal, and the two multiplies + compare of the
// guard test (now one multiply + one max into a running maximum) -- five vector instructions per column that are not FP64 work.
// EXEC is saved and restored inside each statement, so the compiler never sees it change.  Source lanes of the broadcasts
// are always active lanes (k > J: below the pivot).
//   l (in: 0 everywhere) <- u rD in the lanes of the mask, u = Wd[J] (read from the operand list: no copy);  ymax <- max(ymax,
//   |l u|) there (= u^2 / D, the guard's test value);  Wd[k] -= (src of lane k) * l for k = J+1 .. 15 (the negation rides on the
//   FMA's source modifier: the caller stores l itself as the column of L, where rounds 2-3 negated it twice).
//   One function per J (the operand number of u is part of the instruction text).
#define CHAIN_COL_FMA(K, J_) ".if " #K " > " #J_ "\n\t.if %25\n\tv_fmac_f64_dpp %" #K ", %" #J_ ", -%16 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n.else\n\t" \
                             "v_fmac_f64_dpp %" #K ", %21, -%16 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n.endif\n.endif\n\t"
#define DEF_CHAIN_COL(J_) \
template <unsigned MLO, unsigned MHI, int SELF, unsigned ONELO, unsigned ONEHI> \
__device__ __forceinline__ void chain_col_##J_(double (&Wd)[16], double src, double rD, double& l, double& ymax, double& rdiag) { \
    double t; \
    unsigned long long save; \
    asm volatile("s_mov_b64 %19, exec\n\ts_mov_b32 exec_lo, %26\n\ts_mov_b32 exec_hi, %27\n\t" \
                 "v_mov_b64 %20, %22\n\t" \
                 "s_mov_b32 exec_lo, %23\n\ts_mov_b32 exec_hi, %24\n\t" \
                 "v_mul_f64 %16, %" #J_ ", %22\n\t" \
                 "v_mul_f64 %18, %16, %" #J_ "\n\t" \
                 "v_max_f64 %17, %17, |%18|\n\t" \
                 CHAIN_COL_FMA(0, J_) \
                 CHAIN_COL_FMA(1, J_) \
                 CHAIN_COL_FMA(2, J_) \
                 CHAIN_COL_FMA(3, J_) \
                 CHAIN_COL_FMA(4, J_) \
                 CHAIN_COL_FMA(5, J_) \
                 CHAIN_COL_FMA(6, J_) \
                 CHAIN_COL_FMA(7, J_) \
                 CHAIN_COL_FMA(8, J_) \
                 CHAIN_COL_FMA(9, J_) \
                 CHAIN_COL_FMA(10, J_) \
                 CHAIN_COL_FMA(11, J_) \
                 CHAIN_COL_FMA(12, J_) \
                 CHAIN_COL_FMA(13, J_) \
                 CHAIN_COL_FMA(14, J_) \
                 CHAIN_COL_FMA(15, J_) \
                 "s_mov_b64 exec, %19" \
                 : "+v"(Wd[0]), "+v"(Wd[1]), "+v"(Wd[2]), "+v"(Wd[3]), "+v"(Wd[4]), "+v"(Wd[5]), "+v"(Wd[6]), "+v"(Wd[7]), "+v"(Wd[8]), "+v"(Wd[9]), "+v"(Wd[10]), "+v"(Wd[11]), "+v"(Wd[12]), "+v"(Wd[13]), "+v"(Wd[14]), "+v"(Wd[15]), \
                   "+v"(l), "+v"(ymax), "=&v"(t), "=&s"(save), "+v"(rdiag) \
                 : "v"(src), "v"(rD), "n"(MLO), "n"(MHI), "n"(SELF), "n"(ONELO), "n"(ONEHI)); \
}
DEF_CHAIN_COL(0) DEF_CHAIN_COL(1) DEF_CHAIN_COL(2) DEF_CHAIN_COL(3) DEF_CHAIN_COL(4) DEF_CHAIN_COL(5) DEF_CHAIN_COL(6) DEF_CHAIN_COL(7)
DEF_CHAIN_COL(8) DEF_CHAIN_COL(9) DEF_CHAIN_COL(10) DEF_CHAIN_COL(11) DEF_CHAIN_COL(12) DEF_CHAIN_COL(13) DEF_CHAIN_COL(14) DEF_CHAIN_COL(15)
#undef DEF_CHAIN_COL
#undef CHAIN_COL_FMA
// SELF = 1: the broadcast source is the column itself, Wd[J] (the pivot's DPP row is the lane's own); `src` is then not read
// rdiag <- rD in the pivot's lane (mask ONELO / ONEHI) rides in the same statement: one save / restore of EXEC per column less
template <int J, unsigned MLO, unsigned MHI, int SELF, unsigned ONELO, unsigned ONEHI>
__device__ __forceinline__ void chain_step_exec(double (&Wd)[16], double src, double rD, double& l, double& ymax, double& rdiag) {
#define CHAIN_COL_CASE(J_) if constexpr (J == J_) chain_col_##J_<MLO, MHI, SELF, ONELO, ONEHI>(Wd, src, rD, l, ymax, rdiag);
    CHAIN_COL_CASE(0) CHAIN_COL_CASE(1) CHAIN_COL_CASE(2) CHAIN_COL_CASE(3) CHAIN_COL_CASE(4) CHAIN_COL_CASE(5) CHAIN_COL_CASE(6) CHAIN_COL_CASE(7)
    CHAIN_COL_CASE(8) CHAIN_COL_CASE(9) CHAIN_COL_CASE(10) CHAIN_COL_CASE(11) CHAIN_COL_CASE(12) CHAIN_COL_CASE(13) CHAIN_COL_CASE(14) CHAIN_COL_CASE(15)
#undef CHAIN_COL_CASE
}
// Wd[k] -= (src of lane k) * l, k = 0 .. 15, in the lanes of the mask (l is already masked; the mask only saves the work)
template <unsigned MLO, unsigned MHI>
__device__ __forceinline__ void chain_all_exec(double (&Wd)[16], double src, double l) {
    unsigned long long save;
    asm volatile("s_mov_b64 %16, exec\n\ts_mov_b32 exec_lo, %19\n\ts_mov_b32 exec_hi, %20\n\ts_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %17, -%18 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %17, -%18 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %17, -%18 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %17, -%18 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %4, %17, -%18 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %5, %17, -%18 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %6, %17, -%18 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %7, %17, -%18 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %8, %17, -%18 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %9, %17, -%18 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %10, %17, -%18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %11, %17, -%18 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %12, %17, -%18 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %13, %17, -%18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %14, %17, -%18 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %15, %17, -%18 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "s_mov_b64 exec, %16"
                 : "+v"(Wd[0]), "+v"(Wd[1]), "+v"(Wd[2]), "+v"(Wd[3]), "+v"(Wd[4]), "+v"(Wd[5]), "+v"(Wd[6]), "+v"(Wd[7]), "+v"(Wd[8]), "+v"(Wd[9]), "+v"(Wd[10]), "+v"(Wd[11]), "+v"(Wd[12]), "+v"(Wd[13]), "+v"(Wd[14]), "+v"(Wd[15]),
                   "=&s"(save)
                 : "v"(src), "v"(l), "n"(MLO), "n"(MHI));
}
// one 64-bit value through LDS, as two statements: the store (no result) and, later, load + wait in ONE statement, so that the
// compiler never sees a register whose load is still in flight.  DS operations of a wavefront execute in order.
__device__ __forceinline__ void lds_put64(unsigned a, double v) { asm volatile("ds_write_b64 %0, %1" : : "v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ double lds_get64(unsigned a) {
    double r;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    return r;
}
// Wd[k] -= (src of lane k) * l, k = 0 .. 15, in ALL lanes (no EXEC mask: for callers whose other lanes hold don't-care values)
__device__ __forceinline__ void chain_all_neg(double (&Wd)[16], double src, double l) {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %16, -%17 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %16, -%17 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %16, -%17 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %16, -%17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %4, %16, -%17 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %5, %16, -%17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %6, %16, -%17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %7, %16, -%17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %8, %16, -%17 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %9, %16, -%17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %10, %16, -%17 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %11, %16, -%17 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %12, %16, -%17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %13, %16, -%17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %14, %16, -%17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %15, %16, -%17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(Wd[0]), "+v"(Wd[1]), "+v"(Wd[2]), "+v"(Wd[3]), "+v"(Wd[4]), "+v"(Wd[5]), "+v"(Wd[6]), "+v"(Wd[7]), "+v"(Wd[8]), "+v"(Wd[9]), "+v"(Wd[10]), "+v"(Wd[11]), "+v"(Wd[12]), "+v"(Wd[13]), "+v"(Wd[14]), "+v"(Wd[15])
                 : "v"(src), "v"(l));
}
// max(|a|, b) as ONE instruction (fmax(fabs(a), b) costs a canonicalising v_max_f64 a, a first); BS: b is wave-uniform (SGPR)
template <bool BS>
__device__ __forceinline__ double max_abs(double a, double b) {
    double r;
    if constexpr (BS) asm volatile("v_max_f64 %0, |%1|, %2" : "=v"(r) : "v"(a), "s"(b));
    else asm volatile("v_max_f64 %0, |%1|, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// Head of a pivot step of the wave kernel's 16-step chain (every 16-lane DPP row holds the same diagonal block, lane = row):
//   nli <- -(u rD) in the lanes below the pivot (mask M16 per DPP row), 0 elsewhere;  ymax <- max(ymax, |nli u|) there (u^2 / D:
//   the guard's test value);  rdiag <- rD in the pivot's lane (mask ONE16).
// Five vector instructions where compares + selects on per-lane predicates took thirteen, and the multiplier no longer passes
// through a select on its way into the trailing FMAs.  EXEC is saved and restored inside the statement.
template <unsigned M16, unsigned ONE16>
__device__ __forceinline__ void chain_head_exec(double u, double rD, double& nli, double& ymax, double& rdiag) {
    double t;
    unsigned long long save;
    asm volatile("v_mov_b64 %0, 0\n\t"
                 "s_mov_b64 %4, exec\n\ts_mov_b32 exec_lo, %7\n\ts_mov_b32 exec_hi, %7\n\t"
                 "v_mul_f64 %0, -%5, %6\n\t"
                 "v_mul_f64 %3, %0, %5\n\t"
                 "v_max_f64 %1, %1, |%3|\n\t"
                 "s_mov_b32 exec_lo, %8\n\ts_mov_b32 exec_hi, %8\n\t"
                 "v_mov_b64 %2, %6\n\t"
                 "s_mov_b64 exec, %4"
                 : "=&v"(nli), "+v"(ymax), "+v"(rdiag), "=&v"(t), "=&s"(save)
                 : "v"(u), "v"(rD), "n"(M16), "n"(ONE16));
}
// dst <- v in the lanes of the mask only
template <unsigned MLO, unsigned MHI>
__device__ __forceinline__ void mov_exec(double& dst, double v) {
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %3\n\ts_mov_b32 exec_hi, %4\n\tv_mov_b64 %0, %2\n\ts_mov_b64 exec, %1"
                 : "+v"(dst), "=&s"(save) : "v"(v), "n"(MLO), "n"(MHI));
}

// Unit-triangular substitution inside a 16-lane DPP row with the broadcast fused into the FMA: 15 dependent steps
//   s -= l[k] * (s of lane k),  k = 0 .. 14   (subst15_up: forward, L)        /  k = 15 .. 1  (subst15_down: backward, L')
// one instruction each (the compiler's form is two v_mov_b32_dpp + v_fma_f64 per step, three dependent issues); the s_nop
// covers the VALU-write -> DPP-read hazard between consecutive steps.  Same FMA, same rounding.
__device__ __forceinline__ void subst15_up(double& s, const double* l) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%6 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%7 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%10 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%11 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%12 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%13 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%14 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%15 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(s) : "v"(l[0]), "v"(l[1]), "v"(l[2]), "v"(l[3]), "v"(l[4]), "v"(l[5]), "v"(l[6]), "v"(l[7]), "v"(l[8]), "v"(l[9]), "v"(l[10]), "v"(l[11]), "v"(l[12]), "v"(l[13]), "v"(l[14]));
}
__device__ __forceinline__ void subst15_down(double& s, const double* l) {      // uses l[15] .. l[1]
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%1 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%2 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%3 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%4 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%5 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%6 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%7 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%8 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%10 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%11 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%13 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%15 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(s) : "v"(l[15]), "v"(l[14]), "v"(l[13]), "v"(l[12]), "v"(l[11]), "v"(l[10]), "v"(l[9]), "v"(l[8]), "v"(l[7]), "v"(l[6]), "v"(l[5]), "v"(l[4]), "v"(l[3]), "v"(l[2]), "v"(l[1]));
}
// the same for two right-hand sides at once, the two dependency chains interleaved (each step of one chain stands between two
// steps of the other: one s_nop 0 more completes the two wait states of the VALU-write -> DPP-read hazard)
__device__ __forceinline__ void subst15_up2(double& s, double& t, const double* l) {
    asm volatile("s_nop 1\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(s), "+v"(t) : "v"(l[0]), "v"(l[1]), "v"(l[2]), "v"(l[3]), "v"(l[4]), "v"(l[5]), "v"(l[6]), "v"(l[7]), "v"(l[8]), "v"(l[9]), "v"(l[10]), "v"(l[11]), "v"(l[12]), "v"(l[13]), "v"(l[14]));
}
__device__ __forceinline__ void subst15_down2(double& s, double& t, const double* l) {      // uses l[15] .. l[1]
    asm volatile("s_nop 1\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%2 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%2 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%3 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%3 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%4 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%4 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%5 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%5 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%6 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%6 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%7 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%7 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%8 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%8 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%14 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%14 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%15 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%15 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\tv_fmac_f64_dpp %0, %0, -%16 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %1, -%16 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(s), "+v"(t) : "v"(l[15]), "v"(l[14]), "v"(l[13]), "v"(l[12]), "v"(l[11]), "v"(l[10]), "v"(l[9]), "v"(l[8]), "v"(l[7]), "v"(l[6]), "v"(l[5]), "v"(l[4]), "v"(l[3]), "v"(l[2]), "v"(l[1]));
}
// sum_k l[k] * (src of lane k), k = 0 .. 15, accumulated in that order (as the scalar loop it replaces): sixteen fused FMAs
__device__ __forceinline__ double dot16_bcast(double src, const double* l) {
    double a = 0.0;
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(a) : "v"(src), "v"(l[0]), "v"(l[1]), "v"(l[2]), "v"(l[3]), "v"(l[4]), "v"(l[5]), "v"(l[6]), "v"(l[7]), "v"(l[8]), "v"(l[9]), "v"(l[10]), "v"(l[11]), "v"(l[12]), "v"(l[13]), "v"(l[14]), "v"(l[15]));
    return a;
}

// ---- cross-lane helpers ----------------------------------------------------------------------------------------
// v[l] + v[l ^ 16], then + the other 32 lanes: the sum over the four 16-lane rows, identical in all of them
__device__ __forceinline__ double quad_sum(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(v); hi = __double2hiint(v);
    auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}
// sum over the 16 lanes of each DPP row (identical inside the row)
__device__ __forceinline__ double row_sum(double v) {
    asm volatile("" : "+v"(v));
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    return v;
}
// four row sums at once, stage by stage: the eight DPP moves of a stage, then its four adds.  One row_sum is a chain of four
// dependent (move, move, add) triples; four of them written one after the other run one after the other (the compiler keeps
// them apart), which a wavefront alone on its SIMD pays in full.  Same operations and order per sum as row_sum.
__device__ __forceinline__ void row_sum4(double (&v)[4]) {
#define ROW_SUM4_STAGE(C)                                                                      \
    {                                                                                          \
        double t_[4];                                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; i++) t_[i] = dpp_d<C>(v[i]);                  \
        _Pragma("unroll") for (int i = 0; i < 4; i++) asm volatile("" : "+v"(t_[i]));          \
        _Pragma("unroll") for (int i = 0; i < 4; i++) v[i] += t_[i];                           \
        _Pragma("unroll") for (int i = 0; i < 4; i++) asm volatile("" : "+v"(v[i]));           \
    }
#pragma unroll
    for (int i = 0; i < 4; i++) asm volatile("" : "+v"(v[i]));
    ROW_SUM4_STAGE(0xB1) ROW_SUM4_STAGE(0x4E) ROW_SUM4_STAGE(0x141) ROW_SUM4_STAGE(0x140)
#undef ROW_SUM4_STAGE
}
__device__ __forceinline__ double row_max(double v) {
    v = fmax(v, dpp_d<0xB1>(v));
    v = fmax(v, dpp_d<0x4E>(v));
    v = fmax(v, dpp_d<0x141>(v));
    v = fmax(v, dpp_d<0x140>(v));
    return v;
}
// a wave-uniform double, moved to a scalar register pair: the f64 arithmetic that produced it left it in VGPRs, where a
// kernel-lifetime scalar (tolerances, mu, theta, ...) costs two registers of every lane -- or a scratch slot
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// whole-wave reductions: DPP inside the rows, then the four row results through SGPRs (wave-uniform result)
__device__ __forceinline__ double wsum(double v) {
    v = row_sum(v);
    return uni((readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48)));
}
__device__ __forceinline__ double wmax(double v) {
    v = row_max(v);
    return uni(fmax(fmax(readlane_d(v, 0), readlane_d(v, 16)), fmax(readlane_d(v, 32), readlane_d(v, 48))));
}

// Step J of the triangular inverse in the A-operand layout: Ws[s] += (Ws[s] of lane J of the row) * nl for the registers
// s <= J / 4 (columns 4s + q <= J)
template <int J>
__device__ __forceinline__ void winv_step(double (&Ws)[4], double nl) {
    asm volatile("s_nop 1\n\t"
                 ".if 0 <= %5\n\tv_fmac_f64_dpp %0, %0, %4 row_newbcast:" DPP_STR(%5) " row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 4 <= %5\n\tv_fmac_f64_dpp %1, %1, %4 row_newbcast:" DPP_STR(%5) " row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 8 <= %5\n\tv_fmac_f64_dpp %2, %2, %4 row_newbcast:" DPP_STR(%5) " row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 ".if 12 <= %5\n\tv_fmac_f64_dpp %3, %3, %4 row_newbcast:" DPP_STR(%5) " row_mask:0xf bank_mask:0xf\n.endif\n\t"
                 : "+v"(Ws[0]), "+v"(Ws[1]), "+v"(Ws[2]), "+v"(Ws[3]) : "v"(nl), "n"(J));
}

// ---- batched LDS reads (see ipm_wreg.hip for the rest of the family) -------------------------------------------
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
#define LDS_RD_(i) "ds_read_b64 %" #i ", %16 offset:%17+%18*" #i "\n\t"
// o[i] = *(double*)(a + OFF0 + STRIDE i), i < 16 (byte address / offsets)
template <int OFF0, int STRIDE>
__device__ __forceinline__ void lds_run16(unsigned a, double* o) {
    asm volatile(LDS_RD_(0) LDS_RD_(1) LDS_RD_(2) LDS_RD_(3) LDS_RD_(4) LDS_RD_(5) LDS_RD_(6) LDS_RD_(7) LDS_RD_(8) LDS_RD_(9)
                 LDS_RD_(10) LDS_RD_(11) LDS_RD_(12) LDS_RD_(13) LDS_RD_(14) LDS_RD_(15) "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                   "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15])
                 : "v"(a), "n"(OFF0), "n"(STRIDE) : "memory");
}
template <int OFF0, int STRIDE>
__device__ __forceinline__ void lds_run8(unsigned a, double* o) {
    asm volatile("ds_read_b64 %0, %8 offset:%9+%10*0\n\tds_read_b64 %1, %8 offset:%9+%10*1\n\tds_read_b64 %2, %8 offset:%9+%10*2\n\t"
                 "ds_read_b64 %3, %8 offset:%9+%10*3\n\tds_read_b64 %4, %8 offset:%9+%10*4\n\tds_read_b64 %5, %8 offset:%9+%10*5\n\t"
                 "ds_read_b64 %6, %8 offset:%9+%10*6\n\tds_read_b64 %7, %8 offset:%9+%10*7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                 : "v"(a), "n"(OFF0), "n"(STRIDE) : "memory");
}
#undef LDS_RD_
// o[4 b + s] = *(double*)(a[s] + OFF0 + STRIDE b), s < 4, b < NB (NB = 1, 2 or 4): the same four lane-dependent addresses in NB
// consecutive slots of a table, one round trip
template <int OFF0, int STRIDE, int NB>
__device__ __forceinline__ void lds_gather4xN(const unsigned (&a)[4], double* o) {
    static_assert(NB == 1 || NB == 2 || NB == 4, "1, 2 or 4 slots");
    if constexpr (NB == 4) {
        asm volatile("ds_read_b64 %0, %16 offset:%20\n\tds_read_b64 %1, %17 offset:%20\n\tds_read_b64 %2, %18 offset:%20\n\tds_read_b64 %3, %19 offset:%20\n\t"
                     "ds_read_b64 %4, %16 offset:%20+%21\n\tds_read_b64 %5, %17 offset:%20+%21\n\tds_read_b64 %6, %18 offset:%20+%21\n\tds_read_b64 %7, %19 offset:%20+%21\n\t"
                     "ds_read_b64 %8, %16 offset:%20+2*%21\n\tds_read_b64 %9, %17 offset:%20+2*%21\n\tds_read_b64 %10, %18 offset:%20+2*%21\n\tds_read_b64 %11, %19 offset:%20+2*%21\n\t"
                     "ds_read_b64 %12, %16 offset:%20+3*%21\n\tds_read_b64 %13, %17 offset:%20+3*%21\n\tds_read_b64 %14, %18 offset:%20+3*%21\n\tds_read_b64 %15, %19 offset:%20+3*%21\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                       "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(OFF0), "n"(STRIDE) : "memory");
    } else if constexpr (NB == 2) {
        asm volatile("ds_read_b64 %0, %8 offset:%12\n\tds_read_b64 %1, %9 offset:%12\n\tds_read_b64 %2, %10 offset:%12\n\tds_read_b64 %3, %11 offset:%12\n\t"
                     "ds_read_b64 %4, %8 offset:%12+%13\n\tds_read_b64 %5, %9 offset:%12+%13\n\tds_read_b64 %6, %10 offset:%12+%13\n\tds_read_b64 %7, %11 offset:%12+%13\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(OFF0), "n"(STRIDE) : "memory");
    } else {
        asm volatile("ds_read_b64 %0, %4 offset:%8\n\tds_read_b64 %1, %5 offset:%8\n\tds_read_b64 %2, %6 offset:%8\n\tds_read_b64 %3, %7 offset:%8\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(OFF0) : "memory");
    }
}
// o[i] = *(double*)(a + OFF0 + STRIDE i), i < N, any N: runs of 16 / 8 and a tail of single reads
template <int OFF0, int STRIDE, int N>
__device__ __forceinline__ void lds_run(unsigned a, double* o) {
    if constexpr (N >= 16) { lds_run16<OFF0, STRIDE>(a, o); lds_run<OFF0 + 16 * STRIDE, STRIDE, N - 16>(a, o + 16); }
    else if constexpr (N >= 8) { lds_run8<OFF0, STRIDE>(a, o); lds_run<OFF0 + 8 * STRIDE, STRIDE, N - 8>(a, o + 8); }
    else if constexpr (N >= 4) {
        asm volatile("ds_read_b64 %0, %4 offset:%5\n\tds_read_b64 %1, %4 offset:%5+%6\n\tds_read_b64 %2, %4 offset:%5+2*%6\n\tds_read_b64 %3, %4 offset:%5+3*%6\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(a), "n"(OFF0), "n"(STRIDE) : "memory");
        lds_run<OFF0 + 4 * STRIDE, STRIDE, N - 4>(a, o + 4);
    } else if constexpr (N >= 1) {
        asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(o[0]) : "v"(a), "n"(OFF0) : "memory");
        lds_run<OFF0 + STRIDE, STRIDE, N - 1>(a, o + 1);
    }
}

#endif  // PYCLLP_WAVE_COMMON_H
