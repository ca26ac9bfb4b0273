// Microbenchmark: cost of one launch that reads a 2048 x 4096 f32 batch (32 MiB, cold: 12 batches rotate),
// as a function of the per-wave tile shape.  Every wave issues 16 independent 16-byte loads per lane up front.
//   RI  rows per wave instruction (1, 2, 4, 8 -> 1024, 512, 256, 128 contiguous bytes per row per instruction)
//   CD  column repeats: the wave tile is WR = 16 * RI / CD rows x (1024 / RI) * CD bytes
//   WGD waves of the 4-wave workgroup laid along d (the others along rows)
//   XF  1: blockIdx.x walks the d blocks (fastest), 0: the row blocks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int RI, int CD, int WGD, int XF>
__global__ __launch_bounds__(256) void k(const float *x, float *out, int R, int D) {
    constexpr int LPR = 64 / RI, WR = 16 * RI / CD, WDF = LPR * 4 * CD;   // wave tile: WR rows x WDF floats
    constexpr int WGR = 4 / WGD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bd = XF ? blockIdx.x : blockIdx.y, br = XF ? blockIdx.y : blockIdx.x;
    const int row0 = (br * WGR + wave / WGD) * WR, d0 = (bd * WGD + wave % WGD) * WDF;
    const int rr = lane / LPR, ch = lane % LPR;
    f32x4 v[16];
#pragma unroll
    for (int c = 0; c < CD; ++c)
#pragma unroll
        for (int i = 0; i < 16 / CD; ++i)
            v[c * (16 / CD) + i] = *(const f32x4 *)(x + (size_t)(row0 + RI * i + rr) * D + d0 + c * LPR * 4 + 4 * ch);
    f32x4 s = v[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += v[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];
}
int main() {
    const int R = 2048, D = 4096;
    const size_t nb = 12, bytes = (size_t)R * D * 4;
    float *pool, *out;
    hipMalloc(&pool, nb * bytes); hipMalloc(&out, 64); hipMemset(pool, 0, nb * bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 600;
#define RUN(RI, CD, WGD, XF)                                                                                  \
    {                                                                                                         \
        constexpr int WR = 16 * RI / CD, WDF = (64 / RI) * 4 * CD, WGR = 4 / WGD;                             \
        dim3 g = XF ? dim3(D / (WDF * WGD), R / (WR * WGR)) : dim3(R / (WR * WGR), D / (WDF * WGD));         \
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<RI, CD, WGD, XF>), g, dim3(256), 0, 0, pool + (i % nb) * (bytes / 4), out, R, D); \
        hipDeviceSynchronize(); hipEventRecord(e0);                                                           \
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<RI, CD, WGD, XF>), g, dim3(256), 0, 0, pool + (i % nb) * (bytes / 4), out, R, D); \
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);              \
        printf("instr %d rows x %4d B | wave %3d rows x %4d d | wg %3d rows x %4d d | %s fastest : %6.2f us\n", RI, 1024 / RI, WR, WDF, \
               WR * WGR, WDF * WGD, XF ? "d  " : "row", ms * 1e3 / iters);                                    \
    }
#define RUN4(RI, CD) RUN(RI, CD, 4, 0) RUN(RI, CD, 4, 1) RUN(RI, CD, 1, 0) RUN(RI, CD, 1, 1) RUN(RI, CD, 2, 0)
    RUN4(1, 1) RUN4(1, 2) RUN4(1, 4)
    RUN4(2, 1) RUN4(2, 2) RUN4(2, 4)
    RUN4(4, 1) RUN4(4, 2) RUN4(4, 4)
    RUN4(8, 2) RUN4(8, 4) RUN4(8, 8)
    return 0;
}
