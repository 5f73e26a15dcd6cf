// Where do the waves of co-resident workgroups land?  Launches the sparse kernel's shape (256 threads, 80 KB LDS, 512
// blocks: two per CU) and prints HW_REG_HW_ID / XCC_ID per wave.  Build and run on the GPU box:
//   hipcc -O2 --offload-arch=gfx950 tools/hwid_probe.hip -o build/hwid && ./build/hwid
// Round-1 finding: co-resident blocks are 256 ids apart and their waves 0 sit on different SIMDs (0 of 256 collide).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void __launch_bounds__(256, 2) probe(unsigned* out, int spin) {
    extern __shared__ double lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double a = threadIdx.x;
    for (int i = 0; i < spin; i++) a = a * 1.0000001 + 1e-9;   // keep the block resident for a while
    lds[threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}
int main() {
    const int nb = 512;
    unsigned* d;
    hipMalloc(&d, nb * 8 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80448);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 80448, 0, d, 200000);
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, nb * 8 * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int b : {0, 1, 2, 255, 256, 257, 511}) {
        printf("block %3d:", b);
        for (int w = 0; w < 4; w++) {
            const unsigned hw = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1];
            printf("  w%d simd %u cu %u se %u xcc %u |", w, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7, x & 15);
        }
        printf("\n");
    }
    int same_simd = 0, pairs = 0;
    for (int a = 0; a < nb; a++)
        for (int b = a + 1; b < nb; b++) {
            const unsigned ha = h[a * 8], hb = h[b * 8], xa = h[a * 8 + 1] & 15, xb = h[b * 8 + 1] & 15;
            if (xa == xb && ((ha >> 8) & 0xFF) == ((hb >> 8) & 0xFF)) {
                pairs++;
                if (((ha >> 4) & 3) == ((hb >> 4) & 3)) same_simd++;
                if (pairs <= 4) printf("co-resident blocks %d and %d: wave 0 on simd %u vs %u\n", a, b, (ha >> 4) & 3, (hb >> 4) & 3);
            }
        }
    printf("co-resident pairs %d, wave 0 on the same SIMD in %d\n", pairs, same_simd);
    return 0;
}
