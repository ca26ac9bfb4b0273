// gemm_gather.h -- correctness-first fp32-MFMA GEMM with gathered operands, shared by the
// convolution building blocks (cfl_conv.hip) and the head input-gradient (cfl_hip.hip).
//
//   C[m][n] (+ split z) = sum_{k in split z} A(m, k) * B(k, n)
//
// 64 x 64 output tile per 256-thread workgroup, K step 16, operands staged through LDS by
// per-element functors (im2col gathers, transposed filter reads, fragment-major reads ...),
// v_mfma_f32_16x16x4_f32 on the staged tiles.  Operands whose inner dimension is a multiple of 4
// are staged with one index computation and one 16-byte load per 4 elements (modes below).
#pragma once
#include <hip/hip_runtime.h>

typedef float gg_f32x4 __attribute__((ext_vector_type(4)));

// Operand staging modes.  A tile is As[m][k] (64 x 16), B tile is Bs[n][k] (64 x 16).
//   GG_SCALAR : one functor call per element, f(m, k) / f(k, n)
//   GG_VEC_K  : functor.v4(m, k) / v4(k, n) returns the 4 elements k .. k+3 (k % 4 == 0): one
//               index computation and one 16-byte global load per 4 elements, one b128 LDS store
//   GG_VEC_MN : functor.v4(m, k) returns the 4 elements m .. m+3 (A) / v4(k, n) the elements n .. n+3 (B)
//               at a fixed k: 16-byte global load, four scalar LDS stores (transposing)
// The vector modes require M (resp. N) and the operand's inner dimension to be multiples of 4; the
// callers fall back to GG_SCALAR otherwise (e.g. 3-channel image inputs).
enum { GG_SCALAR = 0, GG_VEC_K = 1, GG_VEC_MN = 2 };

template <int AMODE, int BMODE, class LoadA, class LoadB, class Store>
__global__ __launch_bounds__(256) void gemm_gather_kernel(int M, int N, int K, int klen, LoadA la,
                                                          LoadB lb, Store st) {
    __shared__ __attribute__((aligned(16))) float As[64][20];  // [m][k], 80-byte rows: conflict-free b128
    __shared__ __attribute__((aligned(16))) float Bs[64][20];  // [n][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int kbeg = blockIdx.z * klen, kend = min(K, kbeg + klen);
    const int lm = tid >> 2, lk = (tid & 3) * 4;    // scalar / vec-k staging: 4 k's of one row / column
    const int tk = tid >> 4, tm = (tid & 15) * 4;   // vec-mn staging: one k, 4 rows / columns
    gg_f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = (gg_f32x4){0.f, 0.f, 0.f, 0.f};
    const gg_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        gg_f32x4 av, bv;
        if constexpr (AMODE == GG_VEC_K) {
            av = (m0 + lm < M && k0 + lk < kend) ? la.v4(m0 + lm, k0 + lk) : zero;
        } else if constexpr (AMODE == GG_VEC_MN) {
            av = (m0 + tm < M && k0 + tk < kend) ? la.v4(m0 + tm, k0 + tk) : zero;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + lk + e;
                av[e] = (m0 + lm < M && k < kend) ? la(m0 + lm, k) : 0.f;
            }
        }
        if constexpr (BMODE == GG_VEC_K) {
            bv = (n0 + lm < N && k0 + lk < kend) ? lb.v4(k0 + lk, n0 + lm) : zero;
        } else if constexpr (BMODE == GG_VEC_MN) {
            bv = (n0 + tm < N && k0 + tk < kend) ? lb.v4(k0 + tk, n0 + tm) : zero;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + lk + e;
                bv[e] = (n0 + lm < N && k < kend) ? lb(k, n0 + lm) : 0.f;
            }
        }
        __syncthreads();
        if constexpr (AMODE == GG_VEC_MN) {
#pragma unroll
            for (int e = 0; e < 4; ++e) As[tm + e][tk] = av[e];
        } else {
            *(gg_f32x4 *)&As[lm][lk] = av;
        }
        if constexpr (BMODE == GG_VEC_MN) {
#pragma unroll
            for (int e = 0; e < 4; ++e) Bs[tm + e][tk] = bv[e];
        } else {
            *(gg_f32x4 *)&Bs[lm][lk] = bv;
        }
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

template <int AMODE, int BMODE, class LoadA, class LoadB, class Store>
static inline void gemm_gather_modes(int M, int N, int K, int klen, LoadA la, LoadB lb, Store st,
                                     hipStream_t stream) {
    const int splits = gg_splits(K, klen);
    dim3 grid((M + 63) / 64, (N + 63) / 64, splits);
    hipLaunchKernelGGL((gemm_gather_kernel<AMODE, BMODE, LoadA, LoadB, Store>), grid, dim3(256), 0, stream, M,
                       N, K, klen, la, lb, st);
}

template <class LoadA, class LoadB, class Store>
static inline void gemm_gather(int M, int N, int K, int klen, LoadA la, LoadB lb, Store st,
                               hipStream_t stream) {
    gemm_gather_modes<GG_SCALAR, GG_SCALAR>(M, N, K, klen, la, lb, st, stream);
}
