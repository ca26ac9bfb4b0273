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
#include <cstdlib>
#include <type_traits>
#include <utility>

typedef float gg_f32x4 __attribute__((ext_vector_type(4)));

// Division by a launch-time constant for the gather index arithmetic: q = floor(n / d) for 0 <= n < 2^31 as
// one v_mul_hi_u32 and a shift (M = ceil(2^(31+l) / d), l = ceil(log2 d)); a plain `/` by a runtime value
// costs ~25 VALU instructions, and an im2col gather needs six of them.
struct gg_div {
    unsigned M;
    int sh, d;
};
static inline gg_div gg_make_div(int d) {
    gg_div r;
    r.d = d;
    if (d <= 1) { r.M = 0; r.sh = 0; return r; }
    int l = 0;
    while ((1ll << l) < d) ++l;
    const int p = 31 + l;
    r.M = (unsigned)(((1ull << p) + (unsigned long long)d - 1) / (unsigned long long)d);
    r.sh = p - 32;
    return r;
}
__device__ __forceinline__ int gg_quot(int n, const gg_div &dv) {
    return dv.d <= 1 ? n : (int)(__umulhi((unsigned)n, dv.M) >> dv.sh);
}
// n = q * d + r
__device__ __forceinline__ void gg_divmod(int n, const gg_div &dv, int &q, int &r) {
    q = gg_quot(n, dv);
    r = n - q * dv.d;
}

// Operand staging modes.  A tile is As[m][k] (64 x 16), B tile is Bs[n][k] (64 x 16).
//   GG_SCALAR : one functor call per element, f(m, k) / f(k, n)
//   GG_VEC_K  : functor.v4(m, k) / v4(k, n) returns the 4 elements k .. k+3 (k % 4 == 0): one
//               index computation and one 16-byte global load per 4 elements, one b128 LDS store
//   GG_VEC_MN : functor.v4(m, k) returns the 4 elements m .. m+3 (A) / v4(k, n) the elements n .. n+3 (B)
//               at a fixed k: 16-byte global load, four scalar LDS stores (transposing)
// The vector modes require M (resp. N) and the operand's inner dimension to be multiples of 4; the
// callers fall back to GG_SCALAR otherwise (e.g. 3-channel image inputs).
enum { GG_SCALAR = 0, GG_VEC_K = 1, GG_VEC_MN = 2 };

// MT = 16-row tiles per wave (1, 2 or 4); the wave's 16 MFMAs per K step are MT x NT with NT = 4 / MT
// 16-column tiles, so the workgroup tile is (64 MT) x (64 / MT): narrow outputs (N <= 16 / <= 32, e.g. the
// 12-channel image layer or the 32-channel discriminator stage) do not pay for 64 columns.
template <int MT, int AMODE, int BMODE, class LoadA, class LoadB, class Store>
__global__ __launch_bounds__(256) void gemm_gather_kernel(int M, int N, int K, int klen, LoadA la,
                                                          LoadB lb, Store st) {
    constexpr int NT = 4 / MT, TM = 64 * MT, TN = 64 / MT;
    __shared__ __attribute__((aligned(16))) float As[TM][20];  // [m][k], 80-byte rows: conflict-free b128
    __shared__ __attribute__((aligned(16))) float Bs[TN][20];  // [n][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    // blockIdx.z = batch * splits + split (batch > 0 only for launches that stack independent products in z: the four
    // parity classes of a stride-2 input gradient, whose functors read the class from blockIdx.z themselves)
    const int zsp = (int)blockIdx.z % ((K + klen - 1) / klen);
    const int kbeg = zsp * klen, kend = min(K, kbeg + klen);
    const int lm = tid >> 2, lk = (tid & 3) * 4;    // scalar / vec-k staging: 4 k's of one row / column
    const int tk = tid >> 4, tm = (tid & 15) * 4;   // vec-mn staging of A: one k, 4 rows (per 64-row pass)
    const int bk = tid / (TN / 4), bn = (tid % (TN / 4)) * 4;   // vec-mn staging of B (threads < 4 TN)
    const bool bact = tid < 4 * TN;                 // B tile = TN x 16 elements = 4 TN float4
    gg_f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (gg_f32x4){0.f, 0.f, 0.f, 0.f};
    const gg_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // software pipeline: the gathers of K step i+1 are issued before the MFMAs of step i
    auto gather = [&](int k0, gg_f32x4 (&av)[MT], gg_f32x4 &bv) {
#pragma unroll
        for (int p = 0; p < MT; ++p) {
            if constexpr (AMODE == GG_VEC_K) {
                const int m = m0 + p * 64 + lm;
                av[p] = (m < M && k0 + lk < kend) ? la.v4(m, k0 + lk) : zero;
            } else if constexpr (AMODE == GG_VEC_MN) {
                const int m = m0 + p * 64 + tm;
                av[p] = (m < M && k0 + tk < kend) ? la.v4(m, k0 + tk) : zero;
            } else {
                const int m = m0 + p * 64 + lm;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + lk + e;
                    av[p][e] = (m < M && k < kend) ? la(m, k) : 0.f;
                }
            }
        }
        bv = zero;
        if (bact) {
            if constexpr (BMODE == GG_VEC_K) {
                if (n0 + lm < N && k0 + lk < kend) bv = lb.v4(k0 + lk, n0 + lm);
            } else if constexpr (BMODE == GG_VEC_MN) {
                if (n0 + bn < N && k0 + bk < kend) bv = lb.v4(k0 + bk, n0 + bn);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + lk + e;
                    bv[e] = (n0 + lm < N && k < kend) ? lb(k, n0 + lm) : 0.f;
                }
            }
        }
    };
    gg_f32x4 av[MT], bv;
    if (kbeg < kend) gather(kbeg, av, bv);
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < MT; ++p) {
            if constexpr (AMODE == GG_VEC_MN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) As[p * 64 + tm + e][tk] = av[p][e];
            } else {
                *(gg_f32x4 *)&As[p * 64 + lm][lk] = av[p];
            }
        }
        if (bact) {
            if constexpr (BMODE == GG_VEC_MN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[bn + e][bk] = bv[e];
            } else {
                *(gg_f32x4 *)&Bs[lm][lk] = bv;
            }
        }
        __syncthreads();
        if (k0 + 16 < kend) gather(k0 + 16, av, bv);
        gg_f32x4 a[MT], b[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *(const gg_f32x4 *)&As[(wave * MT + mt) * 16 + r16][4 * q];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = *(const gg_f32x4 *)&Bs[nt * 16 + r16][4 * q];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][e], b[nt][e], acc[mt][nt], 0, 0, 0);
    }
    // C layout: col = lane & 15, rows 4*(lane>>4) + e
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + (wave * MT + mt) * 16 + 4 * q + e, n = n0 + nt * 16 + r16;
                if (m < M && n < N) st(m, n, acc[mt][nt][e], (int)blockIdx.z);
            }
}

// ---------------------------------------------------------------------------------------------------
// bf16x3 variant (vector staging modes only): the same GEMM on v_mfma_f32_16x16x32_bf16 at fp32-level
// accuracy.  Every staged fp32 value is split exactly into three bf16 values v = h + m + l (truncation:
// h = v & 0xffff0000, m = (v - h) & 0xffff0000, l = v - h - m; both subtractions exact) ONCE, at staging
// time, and parked in LDS as three bf16 planes; a product is accumulated in fp32 from the six partial
// products of weight >= 2^-16 (hh, hm, mh, hl, lh, mm; dropped terms <= 2^-21 |ab| worst case, 2^-24 rms --
// see cfl_hip.hip / tests/test_bf16x3_split.py).  Six bf16 MFMAs over K = 32 cost 96 cycles against 256
// for the eight fp32 MFMAs they replace.  K step 32; tile shapes, split-K and Store functors as above.
//   LDS plane row = 32 bf16 + 8 pad (80 bytes: the b128 fragment reads stay conflict-free).
//   MFMA operand: lane (r = l & 15, q = l >> 4) holds k = 8q .. 8q+7 of row / column r.
// ---------------------------------------------------------------------------------------------------
typedef __bf16 gg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int gg_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gg_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void gg_split3(float v, float &h, float &m, float &l) {
    h = __uint_as_float(__float_as_uint(v) & 0xffff0000u);
    const float r = v - h;
    m = __uint_as_float(__float_as_uint(r) & 0xffff0000u);
    l = r - m;
}
// {bf16(e1) : bf16(e0)} by truncation = the high halves of the two floats
__device__ __forceinline__ unsigned gg_pack(float e0, float e1) {
    return __builtin_amdgcn_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u);
}

// Optional two-phase functor interface (used by the bf16x3 kernel): the index a thread keeps for the whole K loop
// (its row m of A / column n of B) is decomposed once by fix(), the streaming index k once per K step by stream(),
// and get(fix, stream, ok) does what is left (bounds, address, load; `ok` false -> zeros, branch-free: the load goes to a
// safe address and the result is selected) -- instead of v4(m, k) redoing all divisions on every call.
// Functors without fix() are driven through v4().
template <class F, class = void> struct gg_has_fix : std::false_type {};
template <class F> struct gg_has_fix<F, std::void_t<decltype(std::declval<const F &>().fix(0))>> : std::true_type {};
template <class F, bool IS_A, bool H = gg_has_fix<F>::value> struct gg_two_phase;
template <class F, bool IS_A> struct gg_two_phase<F, IS_A, true> {
    using Fix = decltype(std::declval<const F &>().fix(0));
    using Str = decltype(std::declval<const F &>().stream(0));
    static __device__ __forceinline__ Fix fix(const F &f, int i) { return f.fix(i); }
    static __device__ __forceinline__ Str stream(const F &f, int k) { return f.stream(k); }
    static __device__ __forceinline__ gg_f32x4 get(const F &f, const Fix &a, const Str &b, bool ok) {
        return f.get(a, b, ok);
    }
};
template <class F, bool IS_A> struct gg_two_phase<F, IS_A, false> {
    struct Fix { int i; };
    struct Str { int k; };
    static __device__ __forceinline__ Fix fix(const F &, int i) { return Fix{i}; }
    static __device__ __forceinline__ Str stream(const F &, int k) { return Str{k}; }
    static __device__ __forceinline__ gg_f32x4 get(const F &f, const Fix &a, const Str &b, bool ok) {
        if (!ok) return (gg_f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (IS_A) return f.v4(a.i, b.k);
        else return f.v4(b.k, a.i);
    }
};

// Row stride (bf16 elements) of a [k][m] image with `cols` columns: the smallest stride >= cols whose 32-bit word
// count is 16 mod 32.  Together with the chunk swizzle below (16-column chunk index XOR bit 3 of k) the eight
// 4-row x 32-byte blocks that one 32-lane half fetches with ds_read_b64_tr_b16 fall on 64 distinct banks.
constexpr int gg_tr_stride(int cols) {
    int w = cols / 2 < 16 ? 16 : cols / 2;
    while (w % 32 != 16) w += 2;
    return 2 * w;
}
typedef short gg_s16x4 __attribute__((ext_vector_type(4)));
typedef short gg_s16x8 __attribute__((ext_vector_type(8)));

// WM x WN = 4 waves: wave (wm, wn) owns the MT x NT 16x16 tiles at rows (wm MT + mt) 16, columns (wn NT + nt) 16.
// 4 x 1 (the narrow-output shapes) re-reads every B fragment in all four waves; 2 x 2 with MT = NT = 4 is the
// 128 x 128 tile of the wide layers: 96 MFMAs per K step and wave against 24 fragment reads (0.25 LDS reads per
// MFMA instead of 0.625 with the 64 x 64 tile) and every staged / split value feeds 128 outputs instead of 64 --
// the 64 x 64 tile is bound by the LDS pipe and the staging VALU work, not by the matrix cores.
template <int WM, int WN, int MT, int NT, int AMODE, int BMODE, class LoadA, class LoadB, class Store>
__global__ __launch_bounds__(256) void gemm_gather_x3_kernel(int M, int N, int K, int klen, LoadA la,
                                                             LoadB lb, Store st) {
    static_assert(AMODE != GG_SCALAR && BMODE != GG_SCALAR, "bf16x3 path: vector staging modes only");
    static_assert(WM * WN == 4, "four waves");
    constexpr int TM = 16 * WM * MT, TN = 16 * WN * NT;   // workgroup tile
    constexpr int NPA = TM / 64;                           // 64-row staging passes of A
    constexpr int NPB = TN >= 64 ? TN / 64 : 1;            // 64-column staging passes of B (narrow B: one partial pass)
    static_assert(TM % 64 == 0 && (TN < 64 || TN % 64 == 0), "tile shape");
    // LDS images, three bf16 planes each.
    //   VEC_K  operand: [row][k], rows of 32 k + 8 pad (80 bytes); fragment = one ds_read_b128.
    //   VEC_MN operand: [k][row] as it arrives (no transposing stores), 16-column chunks swizzled by bit 3 of k;
    //                   fragment = two ds_read_b64_tr_b16 (hardware transpose: lane i of a 16-lane group receives
    //                   column i of a 4-row block; tools/microbench/trread.hip pins the lane map).
    constexpr bool AT = AMODE == GG_VEC_MN, BT = BMODE == GG_VEC_MN;
    constexpr int RS = 40, SA = gg_tr_stride(TM), SB = gg_tr_stride(TN);
    constexpr int APL = AT ? 32 * SA : TM * RS, BPL = BT ? 32 * SB : TN * RS;   // bf16 per plane
    __shared__ __attribute__((aligned(16))) unsigned short As[3 * APL];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3 * BPL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    // blockIdx.z = batch * splits + split (batch > 0 only for launches that stack independent products in z: the four
    // parity classes of a stride-2 input gradient, whose functors read the class from blockIdx.z themselves)
    const int zsp = (int)blockIdx.z % ((K + klen - 1) / klen);
    const int kbeg = zsp * klen, kend = min(K, kbeg + klen);
    // staging work items (8 values each), 64 rows / columns x 32 k per pass:
    //   VEC_K : one row / column, 8 consecutive k (two 16-byte gathers)          -> one b128 store per plane
    //   VEC_MN: 4 consecutive rows / columns at k and at k+1 (two 16-byte gathers) -> two b64 stores per plane
    const int lm = tid >> 2, lk = (tid & 3) * 8;
    const int tk = (tid >> 4) * 2, tm = (tid & 15) * 4;
    // B narrower than 64 columns: the threads that exist for it cover all 32 k in one pass
    const int bk = TN >= 64 ? tk : (tid / (TN / 4)) * 2, bn = TN >= 64 ? tm : (tid % (TN / 4)) * 4;
    const bool bact = TN >= 64 || tid < 4 * TN;
    gg_f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (gg_f32x4){0.f, 0.f, 0.f, 0.f};
    const gg_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    using PA = gg_two_phase<LoadA, true>;
    using PB = gg_two_phase<LoadB, false>;
    // loop-invariant halves of the gather indices
    typename PA::Fix fa[NPA];
#pragma unroll
    for (int p = 0; p < NPA; ++p) fa[p] = PA::fix(la, m0 + p * 64 + (AT ? tm : lm));
    typename PB::Fix fb[NPB];
#pragma unroll
    for (int p = 0; p < NPB; ++p) fb[p] = PB::fix(lb, n0 + p * 64 + (BT ? bn : lm));
    auto gather = [&](int k0, gg_f32x4 (&av)[NPA][2], gg_f32x4 (&bv)[NPB][2]) {
        const int ka0 = k0 + (AT ? tk : lk), ka1 = ka0 + (AT ? 1 : 4);
        const typename PA::Str sa0 = PA::stream(la, ka0), sa1 = PA::stream(la, ka1);
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
            const int m = m0 + p * 64 + (AT ? tm : lm);
            av[p][0] = PA::get(la, fa[p], sa0, m < M && ka0 < kend);
            av[p][1] = PA::get(la, fa[p], sa1, m < M && ka1 < kend);
        }
        const int kb0 = k0 + (BT ? bk : lk), kb1 = kb0 + (BT ? 1 : 4);
        const typename PB::Str sb0 = PB::stream(lb, kb0), sb1 = PB::stream(lb, kb1);
#pragma unroll
        for (int p = 0; p < NPB; ++p) {
            bv[p][0] = zero; bv[p][1] = zero;
            if (bact) {
                const int n = n0 + p * 64 + (BT ? bn : lm);
                bv[p][0] = PB::get(lb, fb[p], sb0, n < N && kb0 < kend);
                bv[p][1] = PB::get(lb, fb[p], sb1, n < N && kb1 < kend);
            }
        }
    };
    // split the staged values and park the three planes
    auto park_k = [&](unsigned short *img, int plane, int row, int k, const gg_f32x4 (&v)[2]) {
        float h[8], m[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) gg_split3(v[i >> 2][i & 3], h[i], m[i], l[i]);
        gg_u32x4 ph, pm, pq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ph[i] = gg_pack(h[2 * i], h[2 * i + 1]);
            pm[i] = gg_pack(m[2 * i], m[2 * i + 1]);
            pq[i] = gg_pack(l[2 * i], l[2 * i + 1]);
        }
        unsigned short *d = img + row * RS + k;
        *(gg_u32x4 *)d = ph;
        *(gg_u32x4 *)(d + plane) = pm;
        *(gg_u32x4 *)(d + 2 * plane) = pq;
    };
    auto park_mn = [&](unsigned short *img, int plane, int stride, int col, int k, const gg_f32x4 (&v)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // rows k and k+1 of the [k][col] image
            float h[4], m[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) gg_split3(v[i][e], h[e], m[e], l[e]);
            const int kk = k + i;
            unsigned short *d = img + kk * stride + ((((col >> 4) ^ ((kk >> 3) & 1)) << 4) | (col & 15));
            *(gg_u32x2 *)d = (gg_u32x2){gg_pack(h[0], h[1]), gg_pack(h[2], h[3])};
            *(gg_u32x2 *)(d + plane) = (gg_u32x2){gg_pack(m[0], m[1]), gg_pack(m[2], m[3])};
            *(gg_u32x2 *)(d + 2 * plane) = (gg_u32x2){gg_pack(l[0], l[1]), gg_pack(l[2], l[3])};
        }
    };
    // fragment (k = 8q .. 8q+7 of row / column c0 + r16) of plane `lv`
    auto frag_k = [&](const unsigned short *img, int plane, int lv, int c0) -> gg_bf16x8 {
        return *(const gg_bf16x8 *)(img + lv * plane + (c0 + r16) * RS + 8 * q);
    };
    auto frag_mn = [&](const unsigned short *img, int plane, int stride, int lv, int c0) -> gg_bf16x8 {
        // lane 4a+p of the 16-lane group supplies row 8q + a, columns 4p .. 4p+3 of the 16-column block at c0
        const int a = r16 >> 2, p = r16 & 3;
        const unsigned short *s0 = img + lv * plane + (8 * q + a) * stride + ((((c0 >> 4) ^ (q & 1)) << 4) | (4 * p));
        typedef __attribute__((address_space(3))) gg_s16x4 *lds_p;
        const gg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)s0);
        const gg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(s0 + 4 * stride));
        const gg_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(gg_bf16x8, v);
    };
    gg_f32x4 av[NPA][2], bv[NPB][2];
    if (kbeg < kend) gather(kbeg, av, bv);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
            if constexpr (AT) park_mn(As, APL, SA, p * 64 + tm, tk, av[p]);
            else park_k(As, APL, p * 64 + lm, lk, av[p]);
        }
        if (bact) {
#pragma unroll
            for (int p = 0; p < NPB; ++p) {
                if constexpr (BT) park_mn(Bs, BPL, SB, p * 64 + bn, bk, bv[p]);
                else park_k(Bs, BPL, p * 64 + lm, lk, bv[p]);
            }
        }
        __syncthreads();
        if (k0 + 32 < kend) gather(k0 + 32, av, bv);
        gg_bf16x8 a[3][MT], b[3][NT];
#pragma unroll
        for (int lv = 0; lv < 3; ++lv) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (AT) a[lv][mt] = frag_mn(As, APL, SA, lv, (wm * MT + mt) * 16);
                else a[lv][mt] = frag_k(As, APL, lv, (wm * MT + mt) * 16);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (BT) b[lv][nt] = frag_mn(Bs, BPL, SB, lv, (wn * NT + nt) * 16);
                else b[lv][nt] = frag_k(Bs, BPL, lv, (wn * NT + nt) * 16);
            }
        }
        // small terms first; consecutive MFMAs hit different accumulators
#define GG_X3(LA, LB)                                                                                   \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[LA][mt], b[LB][nt], acc[mt][nt], 0, 0, 0);
        GG_X3(1, 1) GG_X3(2, 0) GG_X3(0, 2) GG_X3(1, 0) GG_X3(0, 1) GG_X3(0, 0)
#undef GG_X3
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + (wm * MT + mt) * 16 + 4 * q + e, n = n0 + (wn * NT + nt) * 16 + r16;
                if (m < M && n < N) st(m, n, acc[mt][nt][e], (int)blockIdx.z);
            }
}

// 1: bf16x3 matrix-core arithmetic for the vector-staged GEMMs (default); 0 with CFL_EXACT_FP32=1 in the
// environment (k-ordered fp32 FMA chains of v_mfma_f32_16x16x4_f32, as for the pair step)
static inline bool gg_use_x3() {
    static const int v = [] { const char *e = getenv("CFL_EXACT_FP32"); return (e && atoi(e) > 0) ? 0 : 1; }();
    return v != 0;
}

// 128 x 128 tile (2 x 2 waves of 64 x 64): when both output dimensions fill it and the launch still has enough
// workgroups for the chip (CFL_DEBUG_NOBIGTILE=1 keeps the 64 x 64 tile)
static inline bool gg_big_tile(long long M, long long N, int splits) {
    static const int off = [] { const char *e = getenv("CFL_DEBUG_NOBIGTILE"); return (e && atoi(e) > 0) ? 1 : 0; }();
    if (off || N < 128 || M < 128) return false;
    const long long wgs = ((M + 127) / 128) * ((N + 127) / 128) * splits;
    return wgs >= 512;
}

// K-range per split (multiple of 16) for about `want` splits; the split count is ceil(K / klen)
static inline int gg_klen(long long K, int want) {
    if (want < 1) want = 1;
    return (int)(((K + want - 1) / want + 15) / 16 * 16);
}
static inline int gg_splits(long long K, int klen) { return (int)((K + klen - 1) / klen); }

template <int AMODE, int BMODE, class LoadA, class LoadB, class Store>
static inline void gemm_gather_modes(int M, int N, int K, int klen, LoadA la, LoadB lb, Store st,
                                     hipStream_t stream, int zbatch = 1) {
    const int nsp = gg_splits(K, klen);
    const int splits = nsp * zbatch;      // grid z (tile choice below: as many workgroups)
    if constexpr (AMODE != GG_SCALAR && BMODE != GG_SCALAR) {
        if (gg_use_x3()) {
            if (N <= 16) {
                dim3 grid((M + 255) / 256, (N + 15) / 16, splits);
                hipLaunchKernelGGL((gemm_gather_x3_kernel<4, 1, 4, 1, AMODE, BMODE, LoadA, LoadB, Store>), grid, dim3(256),
                                   0, stream, M, N, K, klen, la, lb, st);
            } else if (N <= 32) {
                dim3 grid((M + 127) / 128, (N + 31) / 32, splits);
                hipLaunchKernelGGL((gemm_gather_x3_kernel<4, 1, 2, 2, AMODE, BMODE, LoadA, LoadB, Store>), grid, dim3(256),
                                   0, stream, M, N, K, klen, la, lb, st);
            } else if (gg_big_tile(M, N, splits)) {
                dim3 grid((M + 127) / 128, (N + 127) / 128, splits);
                hipLaunchKernelGGL((gemm_gather_x3_kernel<2, 2, 4, 4, AMODE, BMODE, LoadA, LoadB, Store>), grid, dim3(256),
                                   0, stream, M, N, K, klen, la, lb, st);
            } else {
                dim3 grid((M + 63) / 64, (N + 63) / 64, splits);
                hipLaunchKernelGGL((gemm_gather_x3_kernel<4, 1, 1, 4, AMODE, BMODE, LoadA, LoadB, Store>), grid, dim3(256),
                                   0, stream, M, N, K, klen, la, lb, st);
            }
            return;
        }
    }
    if (N <= 16) {
        dim3 grid((M + 255) / 256, (N + 15) / 16, splits);
        hipLaunchKernelGGL((gemm_gather_kernel<4, AMODE, BMODE, LoadA, LoadB, Store>), grid, dim3(256), 0, stream,
                           M, N, K, klen, la, lb, st);
    } else if (N <= 32) {
        dim3 grid((M + 127) / 128, (N + 31) / 32, splits);
        hipLaunchKernelGGL((gemm_gather_kernel<2, AMODE, BMODE, LoadA, LoadB, Store>), grid, dim3(256), 0, stream,
                           M, N, K, klen, la, lb, st);
    } else {
        dim3 grid((M + 63) / 64, (N + 63) / 64, splits);
        hipLaunchKernelGGL((gemm_gather_kernel<1, AMODE, BMODE, LoadA, LoadB, Store>), grid, dim3(256), 0, stream,
                           M, N, K, klen, la, lb, st);
    }
}

template <class LoadA, class LoadB, class Store>
static inline void gemm_gather(int M, int N, int K, int klen, LoadA la, LoadB lb, Store st,
                               hipStream_t stream) {
    gemm_gather_modes<GG_SCALAR, GG_SCALAR>(M, N, K, klen, la, lb, st, stream);
}
