// gemm_gather.h -- correctness-first fp32-MFMA GEMM with gathered operands, shared by the
// convolution building blocks (cfl_conv.hip) and the head input-gradient (cfl_hip.hip).
//
//   C[m][n] (+ split z) = sum_{k in split z} A(m, k) * B(k, n)
//
// 64 x 64 output tile per 256-thread workgroup, K step 16, operands staged through LDS by
// per-element functors (im2col gathers, transposed filter reads, fragment-major reads ...),
// v_mfma_f32_16x16x4_f32 on the staged tiles.  This is the "first correct" form: the gathers
// are scalar; the MFMA-tiled implicit-GEMM with vectorised NHWC loads is round-2 work.
#pragma once
#include <hip/hip_runtime.h>

typedef float gg_f32x4 __attribute__((ext_vector_type(4)));

template <class LoadA, class LoadB, class Store>
__global__ __launch_bounds__(256) void gemm_gather_kernel(int M, int N, int K, int klen, LoadA la,
                                                          LoadB lb, Store st) {
    __shared__ __attribute__((aligned(16))) float As[64][20];  // [m][k], 80-byte rows: conflict-free b128
    __shared__ __attribute__((aligned(16))) float Bs[64][20];  // [n][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int kbeg = blockIdx.z * klen, kend = min(K, kbeg + klen);
    const int lm = tid >> 2, lk = (tid & 3) * 4;  // this thread stages 4 k's of one row / column
    gg_f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = (gg_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + lk + e;
            av[e] = (m0 + lm < M && k < kend) ? la(m0 + lm, k) : 0.f;
            bv[e] = (n0 + lm < N && k < kend) ? lb(k, n0 + lm) : 0.f;
        }
        __syncthreads();
        *(gg_f32x4 *)&As[lm][lk] = (gg_f32x4){av[0], av[1], av[2], av[3]};
        *(gg_f32x4 *)&Bs[lm][lk] = (gg_f32x4){bv[0], bv[1], bv[2], bv[3]};
        __syncthreads();
        const gg_f32x4 a = *(const gg_f32x4 *)&As[wave * 16 + r16][4 * q];
        gg_f32x4 b[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b[nt] = *(const gg_f32x4 *)&Bs[nt * 16 + r16][4 * q];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[nt][e], acc[nt], 0, 0, 0);
    }
    // C layout: col = lane & 15, rows 4*(lane>>4) + e
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + wave * 16 + 4 * q + e, n = n0 + nt * 16 + r16;
            if (m < M && n < N) st(m, n, acc[nt][e], (int)blockIdx.z);
        }
}

// K-range per split (multiple of 16) for about `want` splits; the split count is ceil(K / klen)
static inline int gg_klen(long long K, int want) {
    if (want < 1) want = 1;
    return (int)(((K + want - 1) / want + 15) / 16 * 16);
}
static inline int gg_splits(long long K, int klen) { return (int)((K + klen - 1) / klen); }

template <class LoadA, class LoadB, class Store>
static inline void gemm_gather(int M, int N, int K, int klen, LoadA la, LoadB lb, Store st,
                               hipStream_t stream) {
    const int splits = gg_splits(K, klen);
    dim3 grid((M + 63) / 64, (N + 63) / 64, splits);
    hipLaunchKernelGGL((gemm_gather_kernel<LoadA, LoadB, Store>), grid, dim3(256), 0, stream, M, N, K,
                       klen, la, lb, st);
}
