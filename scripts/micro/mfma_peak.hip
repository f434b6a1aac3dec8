// Practical MFMA ceiling on this box: every wave issues back-to-back 16x16x32 bf16 MFMAs on registers only
// (no memory traffic), 8 waves per CU (2 per SIMD) like the GEMM kernels.  Prints TFLOP/s and the shader
// clock implied by 16 cycles per MFMA.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8v;
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* clk) {
    bf8v a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
    long long t0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = wall_clock64(), c1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = c1 - c0; }
}
int main() {
    float* out; long long* clk; hipMalloc(&out, 256 * 512 * 4 * 4); hipMalloc(&clk, 16);
    for (int wgs : {256, 512}) {
        const int iters = 20000;
        k<<<wgs, 512>>>(out, 100, clk);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); k<<<wgs, 512>>>(out, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flops = (double)wgs * 8 * iters * 16 * 16384.0;
        const double wall_us = h[0] / 100.0;
        printf("wgs %d: %.3f ms, %.1f TFLOP/s; in-kernel %.1f us wall, clock64 delta %lld -> %.3f GHz if shader clocks; per-SIMD MFMA cycles at that rate: %.1f\n",
               wgs, ms, flops / ms / 1e9, wall_us, h[1], h[1] / wall_us / 1e3, (h[1] * 1.0) / (iters * 16.0 * (wgs > 256 ? 2 : 2)));
    }
    return 0;
}
