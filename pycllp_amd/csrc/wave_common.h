// wave_common.h -- device helpers shared by the translation units of libpycllp_hip.so (ipm_dense.hip, ipm_wreg.hip).
#ifndef PYCLLP_WAVE_COMMON_H
#define PYCLLP_WAVE_COMMON_H
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <type_traits>
#include <vector>

#include "../../include/pycllp_hip.h"

// The 10x-growth exits (primal_normal.cl:261-269) keep the reference's own floor -- EPS = 1e-7f absolute, which is
// 1e3 x the relative stopping tolerance used here (for |b|, |c| ~ 1) -- instead of the stopping tolerance itself:
// within a factor 1000 of convergence a residual is rounding noise, and a 10x bump of noise is not divergence.
#define PYCLLP_GROWTH_FLOOR 1e3

typedef double double4_t __attribute__((ext_vector_type(4)));

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

#endif  // PYCLLP_WAVE_COMMON_H
