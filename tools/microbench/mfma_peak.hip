// Bare matrix-core rate on this box: every wave issues N x v_mfma_f32_16x16x32_bf16 on register operands (4 independent
// accumulators), one or two waves per SIMD on every CU.  What fraction of the 2.5 PFLOP/s dense bf16 figure does a loop
// with nothing else in it reach (clocks under sustained MFMA load)?     hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(256) void k16(float *out, int iters) {   // 16 independent accumulators per wave
    bf16x8 a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) { a[j][i] = (__bf16)(threadIdx.x * 0.001f + i + j); b[j][i] = (__bf16)(1.0f + i * 0.01f + j); }
    f32x4 c[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c[i][j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], c[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += c[i][j][(i + j) & 3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024}) {
        for (int iters : {2000, 20000, 200000}) {
            k<<<wgs, 256>>>(out, 100); hipDeviceSynchronize();
            hipEventRecord(e0); k<<<wgs, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)wgs * 4 * iters * 4 * 16 * 16 * 32 * 2;
            printf("%4d workgroups x 4 waves, %6d x 4 MFMA per wave: %8.3f ms  %7.1f TFLOP/s (bf16 dense)  %.2f cycles/MFMA at 2.4 GHz if 256 CUs x 4 SIMDs busy\n",
                   wgs, iters, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * 4 * (wgs / 256.0 > 1 ? wgs / 256.0 : 1)));
        }
    }
    for (int wgs : {256, 512}) {
        const int iters = 20000;
        k16<<<wgs, 256>>>(out, 100); hipDeviceSynchronize();
        hipEventRecord(e0); k16<<<wgs, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)wgs * 4 * iters * 16 * 16 * 16 * 32 * 2;
        printf("%4d workgroups x 4 waves, 16 accumulators per wave: %8.3f ms  %7.1f TFLOP/s\n", wgs, ms, flops / ms / 1e9);
    }
    return 0;
}
