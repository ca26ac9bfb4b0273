// conv_stem.h -- the image stem of the discriminator stacks: 4x4 stride-2 'SAME' convolution of a 3-CHANNEL image
// (cfl/models/blocks.py:288-296: conv2d_weight_norm(x, dim, [4, 4], stride 2) on the 64x64x3 input), forward, input
// gradient and weight gradient as three direct exact-fp32 matrix-core kernels (v_mfma_f32_16x16x4_f32).
//
// Why not the gathered GEMM of gemm_gather.h: with Ci = 3 nothing is a multiple of 4, so every operand element is a
// scalar gather with its own index arithmetic (GG_SCALAR) -- measured on the config-5 step (tools/gan_layers_probe.py,
// serial): forward 0.081 ms at B = 500, input gradient 0.265 ms and weight gradient 0.220 ms at B = 300 for 0.9-1.6 GF
// each (3-4 TF/s), and the input gradient is the LAST launch of the chain the generator's backward waits for.  The
// three products move 40-90 MB each: 10-20 us at HBM speed.  Here:
//   forward         out pixel x (4 x 12-float patch rows) : lane (pixel, kh) loads its patch row as 48 contiguous bytes,
//                   the filter lives in 12 NT registers per lane for the whole launch, 12 NT MFMAs per 16 pixels;
//   input gradient  a 2x2 block of input pixels draws from the 3x3 neighbourhood of dy * act'(y): per parity class
//                   (iy & 1, ix & 1) only the taps kh = py + 1 - 2 dy, kw = px + 1 - 2 dx exist, so the product is ONE dense
//                   GEMM  [blocks] x [9 taps x Co] x [12 = (py, px, ci), padded to 16]  with the class-dependent zeros in
//                   the (register-resident) B operand -- no parity classes, no split-K, no reduce launch;
//   weight gradient [48 patch elements] x [pixels] x [Co]: lane (patch pixel, k = output pixel) loads one RGB triple,
//                   lane (co pair, k) one pair of dy * act'(y); the bias gradient rides in the B loader's registers;
//                   split over pixels into <= 256 slabs in the layout the weight-norm finalisation reads.
// Arithmetic: fp32 products, fp32 accumulation (the generic kernel's arithmetic; summation order differs).
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>

typedef float st_f32x4 __attribute__((ext_vector_type(4)));

struct StemArgs {
    const float *x;        // [B, H, W, 3]
    const float *V;        // [4, 4, 3, Co]
    const float *scale;    // [Co]  g / ||V||
    const float *bias;     // forward: nullable
    const float *y;        // backward: post-activation output (slope source), nullable when act == 0
    const float *dy;       // backward: [B, OH, OW, Co]
    float *out;            // forward: y; input gradient: dx [B, H, W, 3]; weight gradient: slabs [splits][(48 + 4) * Co]
    float slope_neg, slope_zero, slope_pos;   // act'(pre) from the sign of y
    int act;               // forward epilogue
    int B, H, W, OH, OW, Co;
    int spr;               // 16-pixel segments per output row (forward, weight gradient) / per block row (input gradient)
    int nseg;              // segments in all
    int seg_per_wg;        // weight gradient: segments per workgroup (= per slab)
    size_t slab_stride;
};

__device__ __forceinline__ float stem_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.2f * v;
    if (act == 2) return fmaxf(v, 0.f);
    return v;
}

// 12 consecutive floats from a 4-byte aligned address
struct __attribute__((packed, aligned(4))) StemRow { float v[12]; };
struct __attribute__((packed, aligned(4))) StemPix { float v[3]; };

// ---- forward --------------------------------------------------------------------------------------------------------
// A wave owns 16 consecutive output pixels of one output row per step.  MFMA k index = kh (lane >> 4), 12 MFMAs walk the
// 12 floats (kw, ci) of the patch row; B[kh][j][co] = V[(12 kh + j) Co + co] stays in registers.
template <int NT>
__global__ __launch_bounds__(256) void conv_stem_fwd_kernel(StemArgs p) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    float bw[12][NT], sc[NT], bi[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 16 + r16;
#pragma unroll
        for (int j = 0; j < 12; ++j) bw[j][nt] = p.V[(size_t)(q * 12 + j) * p.Co + co];
        sc[nt] = p.scale[co];
        bi[nt] = p.bias ? p.bias[co] : 0.f;
    }
    auto load = [&](int seg, StemRow &ld, int &shift, bool &ok) {
        const int sx = seg % p.spr, t = seg / p.spr, oy = t % p.OH, b = t / p.OH;
        const int ox = sx * 16 + r16, iy = 2 * oy + q - 1, ix0 = 2 * ox - 1;
        ok = seg < p.nseg && ox < p.OW && (unsigned)iy < (unsigned)p.H;
        const int base = ix0 < 0 ? 0 : (ix0 > p.W - 4 ? p.W - 4 : ix0);
        shift = ix0 - base;                       // -1 at the left edge, +1 at the right edge, 0 inside
        const size_t o = ok ? (((size_t)b * p.H + iy) * p.W + base) * 3 : (size_t)0;
        ld = *(const StemRow *)(p.x + o);
    };
    StemRow cur, nxt;
    int sh_c, sh_n;
    bool ok_c, ok_n;
    int seg = wid;
    if (seg < p.nseg) load(seg, cur, sh_c, ok_c);
    for (; seg < p.nseg; seg += nw) {
        load(seg + nw, nxt, sh_n, ok_n);          // (past the end: a safe address, never used)
        float v[12];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float same = cur.v[3 * k + c];
                const float left = k >= 1 ? cur.v[3 * (k - 1) + c] : 0.f;    // shift -1: pixel k is loaded pixel k - 1
                const float right = k <= 2 ? cur.v[3 * (k + 1) + c] : 0.f;   // shift +1
                const float s = sh_c == 0 ? same : (sh_c < 0 ? left : right);
                v[3 * k + c] = ok_c ? s : 0.f;
            }
        st_f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (st_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j], bw[j][nt], acc[nt], 0, 0, 0);
        const int sx = seg % p.spr, t = seg / p.spr;     // t = b * OH + oy
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ox = sx * 16 + 4 * q + e;
            if (ox >= p.OW) continue;
            float *row = p.out + ((size_t)t * p.OW + ox) * p.Co;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) row[nt * 16 + r16] = stem_act(acc[nt][e] * sc[nt] + bi[nt], p.act);
        }
        cur = nxt; sh_c = sh_n; ok_c = ok_n;
    }
}

// ---- input gradient ---------------------------------------------------------------------------------------------------
// Block (a, c) = input pixels (2a + py, 2c + px).  M = 16 blocks along c, N = (py, px, ci) = 12 of 16, K = 9 taps x Co with
// lane group kq holding the channels CQ kq .. CQ kq + CQ - 1 of a tap (CQ = Co / 4 consecutive floats per lane).
template <int CQ>
__global__ __launch_bounds__(256) void conv_stem_dx_kernel(StemArgs p) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int H2 = p.H >> 1;
    // B operand: n = r16 -> (py, px, ci); tap (dy, dx) -> kh = py + 1 - 2 dy, kw = px + 1 - 2 dx
    float bw[9][CQ];
    {
        const int py = r16 / 6, px = (r16 / 3) & 1, ci = r16 % 3;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const int kh = py + 1 - 2 * dy, kw = px + 1 - 2 * dx;
            const bool ok = r16 < 12 && (unsigned)kh < 4u && (unsigned)kw < 4u;
#pragma unroll
            for (int j = 0; j < CQ; ++j) {
                const int co = CQ * q + j;
                bw[t][j] = ok ? p.scale[co] * p.V[(size_t)((kh * 4 + kw) * 3 + ci) * p.Co + co] : 0.f;
            }
        }
    }
    for (int seg = wid; seg < p.nseg; seg += nw) {
        const int sx = seg % p.spr, t0 = seg / p.spr, a = t0 % H2, b = t0 / H2;
        const int c = sx * 16 + r16;
        st_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int oy = a + t / 3 - 1, ox = c + t % 3 - 1;
            const bool ok = (unsigned)oy < (unsigned)p.OH && (unsigned)ox < (unsigned)p.OW;
            const size_t o = ok ? (((size_t)b * p.OH + oy) * p.OW + ox) * p.Co + CQ * q : (size_t)0;
            float av[CQ];
#pragma unroll
            for (int j4 = 0; j4 < CQ; j4 += 4) {
                const st_f32x4 d = *(const st_f32x4 *)(p.dy + o + j4);
                const st_f32x4 yy = p.y ? *(const st_f32x4 *)(p.y + o + j4) : (st_f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = yy[e] > 0.f ? p.slope_pos : (yy[e] < 0.f ? p.slope_neg : p.slope_zero);
                    av[j4 + e] = ok ? d[e] * s : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < CQ; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bw[t][j], acc, 0, 0, 0);
        }
        if (r16 < 12) {
            const int py = r16 / 6, rest = r16 - 6 * py;        // rest = 3 px + ci: six consecutive floats of an image row
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cc = sx * 16 + 4 * q + e;
                if (2 * cc >= p.W) continue;
                p.out[(((size_t)b * p.H + 2 * a + py) * p.W + 2 * cc) * 3 + rest] = acc[e];
            }
        }
    }
}

// NT consecutive floats from an NT-float aligned address as one load
template <int NT>
__device__ __forceinline__ void stem_ldn(const float *src, float (&dst)[NT]) {
    if constexpr (NT == 4) {
        const st_f32x4 v = *(const st_f32x4 *)src;
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
    } else if constexpr (NT == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 v = *(const f2 *)src;
        dst[0] = v[0]; dst[1] = v[1];
    } else {
        dst[0] = src[0];
    }
}

// ---- weight gradient ----------------------------------------------------------------------------------------------------
// D[m][n] += sum_k A[m][k] B[k][n] with k = output pixel (4 per MFMA: lane group kq), m = patch element 3 i + mt (lane i =
// patch pixel kh = i / 4, kw = i % 4; mt = channel), n = co = NT j + nt (lane j holds NT consecutive channels).
// Workgroup z walks its segments (16 output pixels each), its four waves' sums are added in wave order and stored as slab z:
// rows 0 .. 47 the filter gradient [(kh, kw, ci)][co], row 48 the bias gradient.
template <int NT>
__global__ __launch_bounds__(256) void conv_stem_dw_kernel(StemArgs p) {
    __shared__ float red[4][(48 + 1) * 16 * NT];
    const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    const int kh = r16 >> 2, kw = r16 & 3;
    st_f32x4 acc[3][NT];
    float bsum[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bsum[nt] = 0.f;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) acc[mt][nt] = (st_f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int s0 = blockIdx.x * p.seg_per_wg, s1 = min(p.nseg, s0 + p.seg_per_wg);
    for (int seg = s0 + wave; seg < s1; seg += 4) {
        const int sx = seg % p.spr, t = seg / p.spr, oy = t % p.OH, b = t / p.OH;
        const int iy = 2 * oy + kh - 1;
        StemPix xa[4];
        float dv[4][NT], yv[4][NT];
        bool okb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ox = sx * 16 + 4 * s + q, ix = 2 * ox + kw - 1;
            const bool oka = ox < p.OW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const StemPix ld = *(const StemPix *)(p.x + (oka ? (((size_t)b * p.H + iy) * p.W + ix) * 3 : (size_t)0));
#pragma unroll
            for (int c = 0; c < 3; ++c) xa[s].v[c] = oka ? ld.v[c] : 0.f;
            okb[s] = ox < p.OW;
            const size_t o = okb[s] ? ((size_t)t * p.OW + ox) * p.Co + NT * r16 : (size_t)0;
            stem_ldn<NT>(p.dy + o, dv[s]);
            if (p.y) stem_ldn<NT>(p.y + o, yv[s]);
            else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) yv[s][nt] = 1.f;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float sl = yv[s][nt] > 0.f ? p.slope_pos : (yv[s][nt] < 0.f ? p.slope_neg : p.slope_zero);
                bv[nt] = okb[s] ? dv[s][nt] * sl : 0.f;
                bsum[nt] += bv[nt];
            }
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s].v[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    // wave tile -> LDS as [row][co]: D row 4 q + e of m-tile mt is patch element 3 (4 q + e) + mt, column r16 is co = NT r16 + nt
    float *mine = red[wave];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) mine[(3 * (4 * q + e) + mt) * (16 * NT) + NT * r16 + nt] = acc[mt][nt][e];
    // bias gradient: the four k groups of a column, summed in group order by group 0
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float s1_ = __shfl(bsum[nt], r16 + 16), s2_ = __shfl(bsum[nt], r16 + 32), s3_ = __shfl(bsum[nt], r16 + 48);
        if (q == 0) mine[48 * (16 * NT) + NT * r16 + nt] = ((bsum[nt] + s1_) + s2_) + s3_;
    }
    __syncthreads();
    float *slab = p.out + (size_t)blockIdx.x * p.slab_stride;
    for (int i = threadIdx.x; i < 49 * 16 * NT; i += 256)
        slab[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static inline bool stem_off() {
    static const int off = [] { const char *e = getenv("CFL_DEBUG_NOSTEM"); return (e && atoi(e) > 0) ? 1 : 0; }();
    return off != 0;
}
// 4x4 stride 2 on a 3-channel image with even sides; Co = 16, 32 or 64
static inline bool stem_shape_ok(int H, int W, int Ci, int Co, int KH, int KW, int S) {
    return !stem_off() && KH == 4 && KW == 4 && S == 2 && Ci == 3 && H % 2 == 0 && W % 2 == 0 && W >= 4 && H >= 2 &&
           (Co == 16 || Co == 32 || Co == 64);
}
// slabs of the weight gradient: ~32 segments (512 output pixels) per workgroup, at most 256
static inline int stem_dw_seg_per_wg(int B, int OH, int OW) {
    const long long nseg = (long long)B * OH * ((OW + 15) / 16);
    long long per = (nseg + 255) / 256;
    if (per < 32) per = 32;
    return (int)per;
}
static inline int stem_dw_splits(int B, int OH, int OW) {
    const long long nseg = (long long)B * OH * ((OW + 15) / 16);
    const int per = stem_dw_seg_per_wg(B, OH, OW);
    return (int)((nseg + per - 1) / per);
}
static inline void stem_slopes(StemArgs &a, int act) {
    a.slope_pos = 1.f;
    a.slope_neg = act == 1 ? 0.2f : (act == 2 ? 0.f : 1.f);
    a.slope_zero = act == 0 ? 1.f : 0.f;
}
static inline unsigned stem_grid(long long nseg) {
    long long wgs = (nseg + 3) / 4;              // one segment per wave at least
    if (wgs > 2048) wgs = 2048;                  // then several per wave: the filter registers are loaded once per wave
    return (unsigned)(wgs < 1 ? 1 : wgs);
}
static inline void stem_fwd(int B, int H, int W, int Co, const float *x, const float *V, const float *scale, const float *bias,
                            int act, float *y, hipStream_t st) {
    StemArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.V = V; a.scale = scale; a.bias = bias; a.out = y; a.act = act;
    a.B = B; a.H = H; a.W = W; a.OH = H / 2; a.OW = W / 2; a.Co = Co;
    a.spr = (a.OW + 15) / 16;
    const long long nseg = (long long)B * a.OH * a.spr;
    a.nseg = (int)nseg;
    const dim3 grid(stem_grid(nseg));
    if (Co == 16) hipLaunchKernelGGL(conv_stem_fwd_kernel<1>, grid, dim3(256), 0, st, a);
    else if (Co == 32) hipLaunchKernelGGL(conv_stem_fwd_kernel<2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv_stem_fwd_kernel<4>, grid, dim3(256), 0, st, a);
}
static inline void stem_dx(int B, int H, int W, int Co, const float *dy, const float *y, int act, const float *V,
                           const float *scale, float *dx, hipStream_t st) {
    StemArgs a;
    memset(&a, 0, sizeof(a));
    a.V = V; a.scale = scale; a.dy = dy; a.y = act != 0 ? y : nullptr; a.out = dx;
    stem_slopes(a, act);
    a.B = B; a.H = H; a.W = W; a.OH = H / 2; a.OW = W / 2; a.Co = Co;
    a.spr = (W / 2 + 15) / 16;
    const long long nseg = (long long)B * (H / 2) * a.spr;
    a.nseg = (int)nseg;
    const dim3 grid(stem_grid(nseg));
    if (Co == 16) hipLaunchKernelGGL(conv_stem_dx_kernel<4>, grid, dim3(256), 0, st, a);
    else if (Co == 32) hipLaunchKernelGGL(conv_stem_dx_kernel<8>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv_stem_dx_kernel<16>, grid, dim3(256), 0, st, a);
}
// slabs [splits][slab_stride]; slab_stride >= 52 * Co
static inline void stem_dw(int B, int H, int W, int Co, const float *x, const float *dy, const float *y, int act, float *slab,
                           size_t slab_stride, hipStream_t st) {
    StemArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.y = act != 0 ? y : nullptr; a.out = slab;
    stem_slopes(a, act);
    a.B = B; a.H = H; a.W = W; a.OH = H / 2; a.OW = W / 2; a.Co = Co;
    a.spr = (a.OW + 15) / 16;
    a.nseg = B * a.OH * a.spr;
    a.seg_per_wg = stem_dw_seg_per_wg(B, a.OH, a.OW);
    a.slab_stride = slab_stride;
    const dim3 grid((unsigned)stem_dw_splits(B, a.OH, a.OW));
    if (Co == 16) hipLaunchKernelGGL(conv_stem_dw_kernel<1>, grid, dim3(256), 0, st, a);
    else if (Co == 32) hipLaunchKernelGGL(conv_stem_dw_kernel<2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv_stem_dw_kernel<4>, grid, dim3(256), 0, st, a);
}
