// Do v_mfma_f64_16x16x4_f64 and f64 VALU FMAs of two different wavefronts on one SIMD overlap on gfx950?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_overlap tools/dev/ubench_overlap.hip && /tmp/ubench_overlap
// One workgroup of 8 waves per CU (two per SIMD: waves w and w + 4 share SIMD w % 4).  mode 0: all waves MFMA; 1: all waves
// plain f64 FMAs; 2: waves 0-3 MFMA, waves 4-7 FMAs; 3: all waves DPP f64 FMAs; 4: waves 0-3 MFMA, 4-7 DPP FMAs;
// 5: all waves 32-bit integer VALU; 6: waves 0-3 MFMA, 4-7 integer VALU; 7: waves 0-3 f64 FMA, 4-7 integer VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double run_mfma(int n, double s) {
    double4_t a0 = {s, s, s, s}, a1 = a0, a2 = a0, a3 = a0;
    double x = s, y = s + 1.0;
    for (int i = 0; i < n; i++) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    return a0[0] + a1[1] + a2[2] + a3[3];
}
__device__ __forceinline__ double run_fma(int n, double s) {
    double c[8];
    for (int k = 0; k < 8; k++) c[k] = s + k;
    double x = s * 0.5, y = s + 1.0;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(c[k]) : "v"(x), "v"(y));
    }
    double r = 0; for (int k = 0; k < 8; k++) r += c[k];
    return r;
}
__device__ __forceinline__ double run_dpp(int n, double s) {
    double c[8];
    for (int k = 0; k < 8; k++) c[k] = s + k;
    double x = s * 0.5, y = s + 1.0;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c[k]) : "v"(x), "v"(y));
    }
    double r = 0; for (int k = 0; k < 8; k++) r += c[k];
    return r;
}
__device__ __forceinline__ double run_int(int n, double s) {
    int c[8];
    for (int k = 0; k < 8; k++) c[k] = (int)s + k;
    int x = (int)s + 3;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(c[k]) : "v"(x));
    }
    int r = 0; for (int k = 0; k < 8; k++) r += c[k];
    return (double)r;
}

__global__ void __launch_bounds__(512) bench(int mode, int n, double* out, double s) {
    const int wave = threadIdx.x >> 6;
    const bool first = wave < 4;
    double r = 0;
    switch (mode) {
        case 0: r = run_mfma(n, s); break;                                  // 4 n MFMAs per wave
        case 1: r = run_fma(n, s); break;                                   // 8 n FMAs per wave
        case 2: r = first ? run_mfma(n, s) : run_fma(n, s); break;
        case 3: r = run_dpp(n, s); break;
        case 4: r = first ? run_mfma(n, s) : run_dpp(n, s); break;
        case 5: r = run_int(n, s); break;
        case 6: r = first ? run_mfma(n, s) : run_int(n, s); break;
        case 7: r = first ? run_fma(n, s) : run_int(n, s); break;
        case 8: r = first ? run_dpp(n, s) : run_int(n, s); break;
    }
    if (r == 12345.678) out[threadIdx.x] = r;
}

int main() {
    double* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000, grid = 256;
    const char* names[] = {"all MFMA", "all f64 FMA", "MFMA | f64 FMA", "all DPP f64 FMA", "MFMA | DPP f64 FMA", "all int VALU", "MFMA | int VALU",
                           "f64 FMA | int VALU", "DPP f64 FMA | int VALU"};
    for (int mode = 0; mode < 9; mode++) {
        bench<<<grid, 512>>>(mode, 100, out, 1.0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        bench<<<grid, 512>>>(mode, n, out, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // cycles per loop trip at 2.4 GHz
        printf("mode %d %-24s %8.3f ms   %.1f cycles per loop trip per SIMD (2.4 GHz)\n", mode, names[mode], ms, ms * 1e-3 * 2.4e9 / n);
    }
    printf("# a loop trip = 4 MFMAs (first kind) or 8 VALU instructions (second kind) per wave; two waves per SIMD\n");
    return 0;
}
