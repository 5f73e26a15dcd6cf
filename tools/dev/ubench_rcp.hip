// Accuracy of v_rcp_f64 on gfx950 and of one / two Newton steps on it (what fast_rcp in wave_common.h relies on).
//   hipcc -O3 --offload-arch=gfx950 -o tools/dev/ubench_rcp.bin tools/dev/ubench_rcp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
__global__ void k(const double* a, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i];
    double r = __builtin_amdgcn_rcp(x);
    r0[i] = r;
    r = fma(r, fma(-x, r, 1.0), r);
    r1[i] = r;
    r = fma(r, fma(-x, r, 1.0), r);
    r2[i] = r;
}
int main() {
    const int n = 1 << 24;
    double* h = (double*)malloc(n * 8);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;      // mantissa in [1, 2)
        int e = (int)((s >> 3) % 120) - 60;
        h[i] = ldexp(m, e) * ((s & 1) ? 1.0 : -1.0);
    }
    double *a, *r0, *r1, *r2;
    hipMalloc(&a, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8);
    hipMemcpy(a, h, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(a, r0, r1, r2, n);
    double* o[3];
    for (int j = 0; j < 3; j++) o[j] = (double*)malloc(n * 8);
    hipMemcpy(o[0], r0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(o[1], r1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(o[2], r2, n * 8, hipMemcpyDeviceToHost);
    const char* nm[3] = {"v_rcp_f64", "+ 1 Newton step", "+ 2 Newton steps"};
    for (int j = 0; j < 3; j++) {
        double maxrel = 0; long wrong = 0; double maxulp = 0;
        for (int i = 0; i < n; i++) {
            long double ex = 1.0L / (long double)h[i];
            double exd = (double)ex;
            long double rel = fabsl(((long double)o[j][i] - ex) / ex);
            if (rel > maxrel) maxrel = (double)rel;
            if (o[j][i] != exd) wrong++;
            double ulp = fabs(o[j][i] - exd) / (nextafter(fabs(exd), INFINITY) - fabs(exd));
            if (ulp > maxulp) maxulp = ulp;
        }
        printf("%-18s max relative error %.3e (2^%.1f)   max |result - round(1/x)| %.1f ulp   not correctly rounded: %ld of %d\n", nm[j], maxrel, log2(maxrel), maxulp, wrong, n);
    }
    return 0;
}
