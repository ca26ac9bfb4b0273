// conv_halo.h -- 3x3 stride-1 'SAME' convolution as a direct (halo-tile) bf16x3 matrix-core kernel.
//
//   out[b, y, x, n] = sum_{kh, kw, k}  A[b, y + kh - 1, x + kw - 1, k] * Wt[kh*3 + kw][k][n]
//
// used for the forward pass of the 3x3 layers (A = x, Wt = g/||V|| * V) and for their input gradient
// (A = dy * act'(y), Wt[t][co][ci] = scale[co] * V[8 - t][ci][co]: the flipped filter), i.e. the layers of
// cfl/layers.py:100-187 that the SR generator / discriminator stacks of cfl/models/blocks.py are made of.
//
// Why not the gathered GEMM of gemm_gather.h: there every K step of every workgroup re-gathers its A tile from
// global memory with per-element index arithmetic and splits it into bf16 planes again -- ~700 VALU instructions
// against 24 MFMAs, and each input value is fetched and split 9 x (N / tile) times.  Here
//   * a workgroup owns 128 output pixels (8 x 16 of one image, or whole 8 x 8 / 4 x 4 images) x TN channels;
//   * per 32-channel chunk of the contraction the input HALO tile (10 x 18 pixels ...) is loaded once with plain
//     16-byte row loads, split once into the three bf16 planes, and parked in LDS; the nine taps then read their
//     A fragments from it at LDS offsets that differ by a per-tap constant -- no address arithmetic in the loop;
//   * the filters are prepared once per call (conv_halo_prep_kernel: weight-norm scale folded in, transposed /
//     flipped, split into planes, [tap][chunk][plane][n][32 k]) so that a tap's B tile is copied global -> LDS as it
//     is, double-buffered: one barrier per tap step;
//   * the next chunk's halo is requested three taps ahead into registers, so the global latency is hidden
//     behind the remaining taps' MFMAs.
// The inner step is then 6 MT NT MFMAs per wave against (MT + NT) x 3 fragment reads and a 16-byte copy or two.
// LDS image (round 4): rows of 64 bytes (32 channels of one plane), no padding, the 16-byte slot of channel octet o of
// row R at o ^ 2 * ((R >> 2) & 1).  `ds_read_b128` serves a wave in four groups of 16 lanes that are NOT contiguous
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: MI355X_MICROARCH.md, LDS) and banks on (address / 4) mod 64: a
// group holds every fragment row once, half of them with octet q and half with q ^ 1.  With rows R = c + r (mod 32) for
// fragment row r -- 16 consecutive halo pixels, or the pixel-to-lane maps of the 8- and 4-wide tiles below -- this
// swizzle puts the 16 lanes of every group on 16 different slots of the 256-byte bank row for every c (the padded
// 80-byte rows it replaces were conflict-free only for contiguous 16-lane groups: measured 2-way, SQ_LDS_BANK_CONFLICT
// = half of SQ_LDS_IDX_ACTIVE).
// Arithmetic: the same exact three-way bf16 split and six partial products as gemm_gather_x3_kernel (fp32-level).
// Split-K over channel chunks (gridDim.z) for launches with few output tiles: partial sums go to slabs and
// conv_halo_reduce_kernel applies the epilogue.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>

#include "gemm_gather.h"

extern int cfl_set_err(int code, const char *fmt, ...);   // cfl_hip.hip

struct HaloArgs {
    const float *a;        // [B, H, W, K] fp32 activations (x, or dy)
    const float *ya;       // dgrad with an activation: y of the layer (slope source), else nullptr
    float slope_neg, slope_zero;   // act'(pre) from the sign of y: y > 0 ? 1 : (y < 0 ? neg : zero)
    const unsigned short *wp;      // prepared filter planes
    float *out;            // [B, H, W, N], or slabs [splits][B*H*W][N] when splits > 1
    const float *bias;     // epilogue (splits == 1): out = act(acc + bias[n] [+ res]); nullptr = no bias
    const float *res;      // residual input [B, H, W, N] added before the activation (round 4: the join of a residual block as
                           // the store epilogue of its second convolution), or nullptr
    int a_cq;              // > 0 (input gradient of a sub-pixel layer): a and ya arrive 2x sub-pixel SHUFFLED, [B, 2H, 2W, a_cq] with
                           // a_cq = K / 4, a multiple of 32: channel k = (2 i + j) a_cq + c of pixel (h, w) is read from channel c of
                           // pixel (2h+i, 2w+j) -- the un-shuffle pass (cfl_subpixel2x_bwd) folded into the halo loader
    int subpixel;          // 1: the result is stored 2x sub-pixel shuffled, out [B, 2H, 2W, N/4] (cfl/layers.py:212-250 is a
                           // pure index remap: channel n = (2 i + j) N/4 + c of pixel (h, w) -> channel c of pixel (2h+i, 2w+j))
    int act;               // epilogue activation (0 none, 1 lrelu, 2 relu)
    int B, H, W, K, N, Npad;
    int tiles_x, tiles_y;  // 128-pixel tiles per image (both 1 for the whole-image tiles)
    int nchunks;           // K / 32
    int chunks_per_split;  // gridDim.z = ceil(nchunks / chunks_per_split)
    size_t slab_stride;    // B*H*W*N when splits > 1
};

__device__ __forceinline__ float halo_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.2f * v;
    if (act == 2) return fmaxf(v, 0.f);
    return v;
}

// filters -> planes.  dgrad == 0: value(t, k = ci, n = co) = V[t][ci][co] * scale[co];
//                     dgrad == 1: value(t, k = co, n = ci) = V[8 - t][ci][co] * scale[co].
// One thread per (t, chunk, n, octet of k): 8 values -> three 16-byte stores.
__global__ __launch_bounds__(256) void conv_halo_prep_kernel(const float *V, const float *scale, int Ci, int Co,
                                                             int dgrad, int K, int N, int Npad,
                                                             unsigned short *wp) {
    const int nchunks = K >> 5;
    const long long total = 9ll * nchunks * Npad * 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int oct = (int)(i & 3);
    long long r = i >> 2;
    const int n = (int)(r % Npad); r /= Npad;
    const int kc = (int)(r % nchunks);
    const int t = (int)(r / nchunks);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = kc * 32 + oct * 8 + e;
        float x = 0.f;
        if (n < N) {
            if (dgrad) x = V[((size_t)(8 - t) * Ci + n) * Co + k] * scale[k];
            else x = V[((size_t)t * Ci + k) * Co + n] * scale[n];
        }
        v[e] = x;
    }
    float h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gg_split3(v[e], h[e], m[e], l[e]);
    gg_u32x4 ph, pm, pl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ph[e] = gg_pack(h[2 * e], h[2 * e + 1]);
        pm[e] = gg_pack(m[2 * e], m[2 * e + 1]);
        pl[e] = gg_pack(l[2 * e], l[2 * e + 1]);
    }
    const size_t blk = (size_t)(t * nchunks + kc) * 3;
    unsigned short *d = wp + ((blk * Npad + n) * 32 + oct * 8);
    const size_t plane = (size_t)Npad * 32;
    *(gg_u32x4 *)d = ph;
    *(gg_u32x4 *)(d + plane) = pm;
    *(gg_u32x4 *)(d + 2 * plane) = pl;
}

// TW = tile width in pixels: 16 (8 rows of one image), 8 (two 8x8 images) or 4 (eight 4x4 images).
// A workgroup's 128 pixels are eight m-tiles of 16 fragment rows; m-tile M, fragment row r:
//   TW = 16: pixel (py = M, px = r)                       TW = 8: image r / 8, pixel (M, r % 8)
//   TW = 4:  image 4 (M / 4) + r / 4, pixel (M % 4, r % 4)
// and an image's halo occupies HPIP LDS rows with HPIP = TW (mod 16) and HPIP = TW (mod 32)'s bit 2 pattern, so that the LDS row
// of fragment row r is c + r (mod 32) in every case (see the header: that is what the slot swizzle needs).
template <int TW>
struct HaloGeom {
    static constexpr int TH = TW == 16 ? 8 : TW, IMGS = 128 / (TW * TH);
    static constexpr int HW = TW + 2, HH = TH + 2, HPI = HH * HW;
    static constexpr int HPIP = TW == 8 ? 104 : HPI;          // 180 (one image), 104 = 96 + 8, 36 = 32 + 4
    static constexpr int HP = IMGS * HPIP;                    // LDS rows of the halo image
    static constexpr int IPM = 16 / TW;                       // images per m-tile
    static_assert(TW == 16 || HPIP % 32 == TW, "halo rows per image");
    __device__ static __forceinline__ void pixel(int M, int r, int &img, int &py, int &px) {
        const int ig = M / TH;
        py = M - ig * TH; img = ig * IPM + r / TW; px = r % TW;
    }
};
constexpr int HALO_RS = 32;                                   // bf16 per LDS row (64 bytes)
// element offset of channel octet `oct` of LDS row R within a plane
__device__ __forceinline__ int halo_sw(int R, int oct) { return R * HALO_RS + ((oct ^ ((R >> 1) & 2)) << 3); }

template <int TN, int WM, int WN, int MT, int NT, int TW, bool SLOPE>
__global__ __launch_bounds__(256, TN <= 64 ? 2 : 1) void conv_halo_x3_kernel(HaloArgs p) {
    static_assert(WM * WN == 4 && WM * MT * 16 == 128 && WN * NT * 16 == TN, "tile shape");
    using Geo = HaloGeom<TW>;
    constexpr int TH = Geo::TH, IMGS = Geo::IMGS;
    constexpr int HW = Geo::HW, HPI = Geo::HPI, HPIP = Geo::HPIP, HP = Geo::HP;   // halo pixels
    constexpr int RS = HALO_RS;
    constexpr int APL = HP * RS, BPL = TN * RS;              // bf16 per plane
    constexpr int AIT = (HP * 4 + 255) / 256;                // staging items (8 channels of one halo pixel) per thread
    constexpr int BIT = (TN * 12 + 255) / 256;               // 16-byte pieces of a tap's B tile per thread
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short *As = smem;                               // [3][HP][RS]
    unsigned short *Bs = smem + 3 * APL;                     // [2][3][TN][RS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, q = lane >> 4;

    // ---- which pixels ----
    int b0, oy0, ox0;
    if (IMGS == 1) {
        const int per = p.tiles_x * p.tiles_y;
        b0 = blockIdx.x / per;
        const int r = blockIdx.x - b0 * per;
        const int ty = r / p.tiles_x;
        oy0 = ty * TH; ox0 = (r - ty * p.tiles_x) * TW;
    } else {
        b0 = blockIdx.x * IMGS; oy0 = 0; ox0 = 0;
    }
    const int n0 = blockIdx.y * TN;
    const int c_beg = blockIdx.z * p.chunks_per_split;
    const int c_end = min(p.nchunks, c_beg + p.chunks_per_split);

    // ---- loop-invariant halves of the staging addresses ----
    unsigned aoff[AIT];     // element offset of the item's 8 channels at chunk 0; 0xffffffff = padding / outside
    int arow[AIT];          // LDS row (halo pixel) of the item, -1 = no item
#pragma unroll
    for (int u = 0; u < AIT; ++u) {
        const int i = tid + 256 * u;
        const int hp = i >> 2, c8 = (i & 3) * 8;
        const int img = hp / HPIP, r = hp - img * HPIP, hy = r / HW, hx = r - hy * HW;
        arow[u] = hp < HP && r < HPI ? hp : -1;              // (rows HPI .. HPIP - 1 of an image are padding)
        const int b = b0 + img, iy = oy0 + hy - 1, ix = ox0 + hx - 1;
        const bool ok = arow[u] >= 0 && b < p.B && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        aoff[u] = !ok ? 0xffffffffu
                  : p.a_cq ? (unsigned)((((size_t)b * 2 * p.H + 2 * iy) * (2 * p.W) + 2 * ix) * p.a_cq + c8)
                           : (unsigned)((((size_t)b * p.H + iy) * p.W + ix) * p.K + c8);
    }
    gg_f32x4 areg[AIT][2], yreg[SLOPE ? AIT : 1][2];
    auto a_request = [&](int kc) {
        // channel chunk kc -> offset from the item's chunk-0 address (uniform).  Shuffled source: the chunk lies in ONE quarter
        // (a_cq % 32 == 0), i.e. one of the four sub-pixels (i, j), channels c0 .. c0 + 31 there
        size_t coff = (size_t)kc * 32;
        if (p.a_cq) {
            const int n0 = kc * 32, qd = n0 / p.a_cq, c0 = n0 - qd * p.a_cq;
            coff = (size_t)((qd >> 1) * 2 * p.W + (qd & 1)) * p.a_cq + c0;
        }
#pragma unroll
        for (int u = 0; u < AIT; ++u) {
            const bool ok = aoff[u] != 0xffffffffu;
            const size_t o = ok ? (size_t)aoff[u] + coff : (size_t)0;   // safe address, selected below
            areg[u][0] = *(const gg_f32x4 *)(p.a + o);
            areg[u][1] = *(const gg_f32x4 *)(p.a + o + 4);
            if (SLOPE) {
                yreg[u][0] = *(const gg_f32x4 *)(p.ya + o);
                yreg[u][1] = *(const gg_f32x4 *)(p.ya + o + 4);
            }
        }
    };
    auto a_park = [&]() {
#pragma unroll
        for (int u = 0; u < AIT; ++u) {
            if (arow[u] < 0) continue;
            const bool ok = aoff[u] != 0xffffffffu;
            float h[8], m[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = areg[u][e >> 2][e & 3];
                if (SLOPE) {
                    const float yy = yreg[u][e >> 2][e & 3];
                    v *= yy > 0.f ? 1.f : (yy < 0.f ? p.slope_neg : p.slope_zero);
                }
                v = ok ? v : 0.f;
                gg_split3(v, h[e], m[e], l[e]);
            }
            gg_u32x4 ph, pm, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ph[e] = gg_pack(h[2 * e], h[2 * e + 1]);
                pm[e] = gg_pack(m[2 * e], m[2 * e + 1]);
                pl[e] = gg_pack(l[2 * e], l[2 * e + 1]);
            }
            unsigned short *d = As + halo_sw(arow[u], (tid + 256 * u) & 3);
            *(gg_u32x4 *)d = ph;
            *(gg_u32x4 *)(d + APL) = pm;
            *(gg_u32x4 *)(d + 2 * APL) = pl;
        }
    };
    // B tile of one (tap, chunk): three planes of TN rows x 64 bytes, contiguous per plane in wp
    gg_u32x4 breg[BIT];
    auto b_request = [&](int t, int kc) {
        const unsigned short *src = p.wp + (size_t)(t * p.nchunks + kc) * 3 * p.Npad * 32;
#pragma unroll
        for (int u = 0; u < BIT; ++u) {
            const int j = tid + 256 * u;
            const int pl = j / (TN * 4), r = j - pl * (TN * 4);      // plane, 16-byte piece within the plane's tile
            if ((TN * 12) % 256 == 0 || j < TN * 12)
                breg[u] = *(const gg_u32x4 *)(src + ((size_t)pl * p.Npad + n0) * 32 + r * 8);
        }
    };
    auto b_park = [&](int buf) {
        unsigned short *dst = Bs + buf * 3 * BPL;
#pragma unroll
        for (int u = 0; u < BIT; ++u) {
            const int j = tid + 256 * u;
            const int pl = j / (TN * 4), r = j - pl * (TN * 4);
            if ((TN * 12) % 256 == 0 || j < TN * 12) *(gg_u32x4 *)(dst + pl * BPL + halo_sw(r >> 2, r & 3)) = breg[u];
        }
    };

    // ---- fragment addresses ----
    int hb[MT];            // halo row of this lane's output pixel in m-tile mt, at tap (0, 0)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int img, py, px;
        Geo::pixel(wm * MT + mt, r16, img, py, px);
        hb[mt] = img * HPIP + py * HW + px;
    }
    int bfo[NT];           // this lane's offset in a plane of the B tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bfo[nt] = halo_sw((wn * NT + nt) * 16 + r16, q);
    gg_f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (gg_f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- main loop: one step = one tap of one 32-channel chunk.  The fragments of step s+1 are read from LDS into a
    // second register set WHILE the MFMAs of step s run (a wave's MFMAs only overlap with its own LDS traffic if that
    // traffic is issued in front of them: with one wave per SIMD nothing else fills the matrix-core time), so
    //   LDS at the start of step s:  halo of chunk(s) [replaced at the start of the chunk's last tap, below],
    //                                B[s+1] in buffer (s+1)&1 (parked during step s-1),
    //   registers:                   F[s] (read during step s-1), breg = B[s+2] (requested during step s-1).
    const int nsteps = (c_end - c_beg) * 9;
    gg_bf16x8 fa0[3][MT], fb0[3][NT], fa1[3][MT], fb1[3][NT];
    auto load_frags = [&](gg_bf16x8 (&fa)[3][MT], gg_bf16x8 (&fb)[3][NT], int tapoff, int buf) {
        // in the order the products consume them
        const int order_a[3] = {1, 2, 0}, order_b[3] = {1, 0, 2};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int la = order_a[i], lb = order_b[i];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                fa[la][mt] = *(const gg_bf16x8 *)(As + la * APL + halo_sw(hb[mt] + tapoff, q));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                fb[lb][nt] = *(const gg_bf16x8 *)(Bs + (buf * 3 + lb) * BPL + bfo[nt]);
        }
    };
    auto products = [&](const gg_bf16x8 (&fa)[3][MT], const gg_bf16x8 (&fb)[3][NT]) {
#define HALO_X3(LA, LB)                                                                                 \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[LA][mt], fb[LB][nt], acc[mt][nt], 0, 0, 0);
        // small terms first; consecutive MFMAs hit different accumulators
        HALO_X3(1, 1) HALO_X3(2, 0) HALO_X3(0, 2) HALO_X3(1, 0) HALO_X3(0, 1) HALO_X3(0, 0)
#undef HALO_X3
    };
    // (tap, chunk) of the running step and of the filter request stream (three steps ahead)
    int t = 0, kc = c_beg, tr = 0, kcr = c_beg;
    auto tapoff_of = [&](int tt) { const int kh = tt / 3; return kh * HW + (tt - 3 * kh); };
    auto request_next_b = [&]() {
        b_request(tr, kcr);
        if (kcr + 1 < c_end || tr < 8) {         // stays on the last tile once the stream is exhausted
            if (++tr == 9) { tr = 0; ++kcr; }
        }
    };
    // PF: prefetch the next step's fragments.  The reads are spread between the MFMAs (one ds_read_b128 per group of
    // MFMAs): issued in one burst after the barrier, the four waves' 96 reads queue up in front of the LDS and every
    // wave waits at issue until its own are accepted -- measured, the reads then cost their full LDS time on top of
    // the MFMAs although they are not consumed until the next step.
    auto step = [&](auto PF, int s, gg_bf16x8 (&fa)[3][MT], gg_bf16x8 (&fb)[3][NT], gg_bf16x8 (&na)[3][MT],
                    gg_bf16x8 (&nb)[3][NT]) {
        constexpr bool pf = decltype(PF)::value;
        if (pf && t == 8) {
            // the next step opens the next chunk: its halo (requested at tap 5) replaces this one.  Every wave
            // holds this step's fragments in registers already (waited for before the last barrier).
            a_park();
            __syncthreads();
        }
        if (t == 5 && kc + 1 < c_end) a_request(kc + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (pf) load_frags(na, nb, tapoff_of(t == 8 ? 0 : t + 1), (s + 1) & 1);
        // B[s+2] over B[s] (whose fragments are in registers), then the request of B[s+3] into the same registers.
        // Unconditional, so that they sit in the MFMAs' basic block and can be spread between them: in the last two
        // steps the park hits a buffer nobody reads again and the request re-reads the last tile.
        b_park(s & 1);
        request_next_b();
        products(fa, fb);
        {
            // schedule: the fragment reads over the first 3/4 of the MFMAs, the filter stores (which the compiler
            // keeps behind the reads: same LDS array) and loads over the last quarter
            constexpr int reads = pf ? 3 * (MT + NT) : 0, mfmas = 6 * MT * NT;
            constexpr int head = reads > 0 ? mfmas * 3 / 4 : 0, per_r = head / (reads > 0 ? reads : 1);
            constexpr int per_w = (mfmas - per_r * reads) / BIT;
#pragma unroll
            for (int i = 0; i < reads; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, per_r, 0);   // MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one LDS read
            }
#pragma unroll
            for (int i = 0; i < BIT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, per_w, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // one LDS store
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // one global load
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own reads landed, own stores done: safe to let others overwrite
        __syncthreads();
        if (++t == 9) { t = 0; ++kc; }
    };
    const std::true_type PF1;
    const std::false_type PF0;
    if (nsteps > 0) {
        // prologue: halo of the first chunk, B[0] and B[1] in LDS, B[2] requested, F[0] in registers
        a_request(c_beg);
        request_next_b();
        a_park();
        b_park(0);
        request_next_b();                      // nsteps >= 9
        __syncthreads();
        load_frags(fa0, fb0, tapoff_of(0), 0);
        b_park(1);
        request_next_b();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        // nsteps is a multiple of 9: odd counts end on (fa0, fb0), even counts on (fa1, fb1)
        int s = 0;
        for (; s + 2 < nsteps; s += 2) {
            step(PF1, s, fa0, fb0, fa1, fb1);
            step(PF1, s + 1, fa1, fb1, fa0, fb0);
        }
        if (s + 1 < nsteps) {
            step(PF1, s, fa0, fb0, fa1, fb1);
            step(PF0, s + 1, fa1, fb1, fa0, fb0);
        } else {
            step(PF0, s, fa0, fb0, fa1, fb1);
        }
    }

    // ---- epilogue: C layout col = lane & 15 (channel), rows 4q + e (pixel) ----
    const bool raw = gridDim.z > 1;
    float *out = p.out + (raw ? (size_t)blockIdx.z * p.slab_stride : (size_t)0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int img, py, px;
            Geo::pixel(wm * MT + mt, 4 * q + e, img, py, px);
            const int b = b0 + img, oy = oy0 + py, ox = ox0 + px;
            if (b >= p.B || oy >= p.H || ox >= p.W) continue;
            const size_t pixoff = (((size_t)b * p.H + oy) * p.W + ox) * p.N;
            float *row = out + pixoff;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + (wn * NT + nt) * 16 + r16;
                if (n >= p.N) continue;
                float v = acc[mt][nt][e];
                if (!raw) {
                    v += p.bias ? p.bias[n] : 0.f;            // (the order of the separate kernels: (conv + b) + residual)
                    if (p.res) v += p.res[pixoff + n];
                    v = halo_act(v, p.act);
                    if (p.subpixel) {
                        const int Cq = p.N >> 2, i = n / Cq, c = n - i * Cq;
                        out[((((size_t)b * 2 * p.H + 2 * oy + (i >> 1)) * (2 * p.W)) + 2 * ox + (i & 1)) * Cq + c] = v;
                        continue;
                    }
                }
                row[n] = v;
            }
        }
}

// out[i] = act(sum_z slab[z][i] + bias[n] [+ res[i]]),  4 elements per thread (N % 4 == 0); subpixel (H, W > 0):
// stored shuffled as in the kernel's own epilogue
__global__ __launch_bounds__(256) void conv_halo_reduce_kernel(const float *slab, int splits, size_t stride, size_t n4,
                                                               int N, const float *bias, int act, float *out,
                                                               const float *res, int subH, int subW) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    gg_f32x4 s = *(const gg_f32x4 *)(slab + 4 * i);
    for (int z = 1; z < splits; ++z) s += *(const gg_f32x4 *)(slab + (size_t)z * stride + 4 * i);
    const int n = (int)((4 * i) % (size_t)N);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += bias ? bias[n + e] : 0.f;
    if (res) s += *(const gg_f32x4 *)(res + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = halo_act(s[e], act);
    if (subH > 0) {
        const size_t pix = (4 * i) / (size_t)N;
        const int ox = (int)(pix % subW), oy = (int)((pix / subW) % subH);
        const size_t b = pix / ((size_t)subW * subH);
        const int Cq = N >> 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {      // (scalar stores: four consecutive channels may straddle two quarters when Cq % 4 != 0)
            const int q = (n + e) / Cq, c = (n + e) - q * Cq;
            out[(((b * 2 * subH + 2 * oy + (q >> 1)) * (2 * subW)) + 2 * ox + (q & 1)) * Cq + c] = s[e];
        }
        return;
    }
    *(gg_f32x4 *)(out + 4 * i) = s;
}

// ---- host side ----------------------------------------------------------------------------------------------
struct HaloPlan {
    bool ok;
    int tw, tiles_x, tiles_y, ptiles, tn, ntiles, Npad, nchunks, splits, chunks_per_split;
    size_t wp_bytes, slab_floats;
};

static inline bool halo_off() {
    static const int off = [] { const char *e = getenv("CFL_DEBUG_NOHALO"); return (e && atoi(e) > 0) ? 1 : 0; }();
    return off != 0;
}

// Fewest output channels the kernel takes (on a 32-column tile, the rest zero planes).  Round 5: 4 instead of 32 -- the
// generator's image layer (128 -> 12 channels, sub-pixel store) ran on the gathered GEMM, whose loop is bound by its gather
// arithmetic whatever N is: 0.26 ms for 34 GF at B = 300; the padded halo tile does 2.7x the matrix work and is still several
// times faster.  CFL_DEBUG_HALO_NMIN=32 restores the old bound.
static inline int halo_nmin() {
    static const int v = [] { const char *e = getenv("CFL_DEBUG_HALO_NMIN"); const int n = e ? atoi(e) : 0; return n > 0 ? n : 4; }();
    return v;
}

// Channel tile.  64 columns x 128 pixels at TWO workgroups per CU (59-80 KiB of LDS, <= 256 registers) beats the 128-column
// tile at one: a single wave per SIMD cannot keep the matrix pipe busy on its own (bare MFMA loop: 59 % of peak with one
// wave, 85 % with two -- tools/microbench/mfma_peak.hip) and two independent workgroups hide each other's barriers.
// Round 4, same box: 300x16x16x256->512 0.925 -> 0.82 ms, 300x8x8x512->1024 0.97 -> 0.94 ms, 500x16x16x64->64 (already a
// 64-column tile, now two per CU) 0.090 -> 0.064 ms.  CFL_DEBUG_HALO_TN=128 restores the wide tile, =32 forces the narrow one.
// (the 4-wide tile -- eight 4x4 images, 288 halo rows, 78 KiB -- takes the 64-column tile too: 500x4x4x256->256 0.076 -> 0.069 ms,
// 300x4x4x64->2048 0.123 -> 0.086 ms; the 32-column tile loses everywhere: tools/halo_probe.py)
static inline int halo_tn_base(int N) { return N >= 128 ? 128 : (N >= 64 ? 64 : 32); }   // also the padding unit of the planes
static inline int halo_tn_of(int N, int tw) {
    static const int cap = [] { const char *e = getenv("CFL_DEBUG_HALO_TN"); const int v = e ? atoi(e) : 0; return v == 32 || v == 128 ? v : 64; }();
    const int tn = halo_tn_base(N);
    (void)tw;
    return tn < cap ? tn : cap;
}

// B images of H x W, contraction over K channels, N output channels
static inline HaloPlan halo_plan(int B, int H, int W, int K, int N) {
    HaloPlan pl;
    memset(&pl, 0, sizeof(pl));
    if (halo_off() || !gg_use_x3()) return pl;
    if (K % 32 != 0 || N % 4 != 0 || N < halo_nmin()) return pl;
    if (W % 16 == 0 && H % 8 == 0) { pl.tw = 16; pl.tiles_x = W / 16; pl.tiles_y = H / 8; pl.ptiles = B * pl.tiles_x * pl.tiles_y; }
    else if (W == 8 && H == 8) { pl.tw = 8; pl.tiles_x = pl.tiles_y = 1; pl.ptiles = (B + 1) / 2; }
    else if (W == 4 && H == 4) { pl.tw = 4; pl.tiles_x = pl.tiles_y = 1; pl.ptiles = (B + 7) / 8; }
    else return pl;
    if ((size_t)B * H * W * (size_t)K >= 0xffffffffull) return pl;     // 32-bit element offsets of A
    pl.tn = halo_tn_of(N, pl.tw);
    pl.ntiles = (N + pl.tn - 1) / pl.tn;
    // the planes are padded to the BASE tile whatever tile runs: their layout is a function of the channel counts alone
    pl.Npad = (N + halo_tn_base(N) - 1) / halo_tn_base(N) * halo_tn_base(N);
    pl.nchunks = K / 32;
    // (the split count is planned on 128-column tiles whatever tile runs: the summation order of every output, and with
    // it every recorded trajectory, stays what it was before the 64-column tile became the default; CFL_DEBUG_HALO_SPLIT_TN=1
    // plans on the tile that runs)
    static const int split_tn = [] { const char *e = getenv("CFL_DEBUG_HALO_SPLIT_TN"); return e ? atoi(e) : 0; }();
    const int tn_plan = split_tn ? pl.tn : (N >= 128 ? 128 : pl.tn);
    const long long tiles = (long long)pl.ptiles * ((N + tn_plan - 1) / tn_plan);
    // split-K only for launches of fewer than ~128 tiles (512 until the end of round 4: with the chains of the MrCGAN step
    // side by side a layer no longer has to fill the chip on its own, and every split costs a slab round trip and a reduce
    // launch -- step 13.7-13.8 -> 13.4-13.5 ms at 128, 64 and 32 alike, 13.7 without any split; CFL_DEBUG_HALO_SPLIT_WGS=<n>)
    static const int wgs = [] { const char *e = getenv("CFL_DEBUG_HALO_SPLIT_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 128; }();
    int want = (int)((wgs + tiles - 1) / tiles);
    if (want > pl.nchunks) want = pl.nchunks;
    if (want > 32) want = 32;
    if (want < 1) want = 1;
    pl.chunks_per_split = (pl.nchunks + want - 1) / want;
    pl.splits = (pl.nchunks + pl.chunks_per_split - 1) / pl.chunks_per_split;
    pl.wp_bytes = (size_t)9 * pl.nchunks * 3 * pl.Npad * 32 * sizeof(unsigned short);
    pl.slab_floats = pl.splits > 1 ? (size_t)pl.splits * B * H * W * N : 0;
    pl.ok = true;
    return pl;
}
// bytes of the prepared filter planes of a (K channels in, N channels out) direction -- a function of the channel counts
// only (the cache of a layer is laid out with it whatever batch / image size a call has); 0 when the direction can never
// run on the halo kernel
static inline size_t halo_planes_bytes(int K, int N) {
    if (halo_off() || !gg_use_x3() || K % 32 != 0 || N % 4 != 0 || N < halo_nmin()) return 0;
    const int tn = halo_tn_base(N);
    const int Npad = (N + tn - 1) / tn * tn;
    return ((size_t)9 * (K / 32) * 3 * Npad * 32 * sizeof(unsigned short) + 15) / 16 * 16;
}
// the filter preparation alone (what halo_conv does first when need_prep): lets a caller build the planes of BOTH
// directions while it is on one stream, before several streams use them
static inline void halo_prep_planes(const HaloPlan &pl, int K, int N, const float *V, const float *scale, int Ci, int Co,
                                    int dgrad, unsigned short *wp, hipStream_t st) {
    const long long prep = 9ll * pl.nchunks * pl.Npad * 4;
    hipLaunchKernelGGL(conv_halo_prep_kernel, dim3((unsigned)((prep + 255) / 256)), dim3(256), 0, st, V, scale, Ci, Co,
                       dgrad, K, N, pl.Npad, wp);
}
static inline size_t halo_scratch_bytes(const HaloPlan &pl) {
    return pl.ok ? ((pl.wp_bytes + 15) / 16 * 16 + pl.slab_floats * sizeof(float)) : 0;
}

template <int TN, int WM, int WN, int MT, int NT, int TW, bool SLOPE>
static inline void halo_launch_one(const HaloArgs &a, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)(3 * HaloGeom<TW>::HP * HALO_RS + 2 * 3 * TN * HALO_RS) * sizeof(unsigned short);
    auto kern = conv_halo_x3_kernel<TN, WM, WN, MT, NT, TW, SLOPE>;
    // > 64 KiB of dynamic LDS needs the opt-in: once per instantiation and device (one process drives one GPU in the
    // data-parallel layout, but nothing here relies on it)
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load() & bit)) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            cfl_set_err(CFL_E_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed for the halo kernel", lds);
            return;
        }
        done.fetch_or(bit);
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
}
template <int TW, bool SLOPE>
static inline void halo_launch_tw(const HaloArgs &a, int tn, dim3 grid, hipStream_t st) {
    if (tn == 128) halo_launch_one<128, 2, 2, 4, 4, TW, SLOPE>(a, grid, st);
    else if (tn == 64) halo_launch_one<64, 2, 2, 4, 2, TW, SLOPE>(a, grid, st);
    else halo_launch_one<32, 4, 1, 2, 2, TW, SLOPE>(a, grid, st);
}

// scratch: [wp planes | slabs].  V: HWIO filter of the LAYER (Ci, Co = the layer's channels).
// planes_cache (nullable): caller-kept prepared filter planes of THIS layer and direction (cfl_conv_cache_bytes); they
// are (re)built when need_prep, otherwise reused -- the planes depend on V, the gains and the direction only, and a
// layer is called 3-6 times per MrCGAN step between two updates of its weights.
static inline void halo_conv(const HaloPlan &pl, int B, int H, int W, int K, int N, const float *a, const float *ya,
                             int slope_act, const float *V, const float *scale, int Ci, int Co, int dgrad,
                             const float *bias, int act, float *out, void *scratch, hipStream_t st,
                             unsigned short *planes_cache = nullptr, bool need_prep = true, const float *res = nullptr,
                             int subpixel = 0, int a_cq = 0) {
    unsigned short *wp = planes_cache ? planes_cache : (unsigned short *)scratch;
    float *slab = (float *)((char *)scratch + (pl.wp_bytes + 15) / 16 * 16);
    const long long prep = 9ll * pl.nchunks * pl.Npad * 4;
    if (need_prep || !planes_cache)
        hipLaunchKernelGGL(conv_halo_prep_kernel, dim3((unsigned)((prep + 255) / 256)), dim3(256), 0, st, V, scale, Ci, Co,
                           dgrad, K, N, pl.Npad, wp);
    HaloArgs h;
    memset(&h, 0, sizeof(h));
    h.a = a; h.ya = (ya && slope_act != 0) ? ya : nullptr;
    h.slope_neg = slope_act == 1 ? 0.2f : 0.f; h.slope_zero = 0.f;
    h.wp = wp; h.out = pl.splits > 1 ? slab : out; h.bias = bias; h.act = act; h.res = res; h.subpixel = subpixel; h.a_cq = a_cq;
    h.B = B; h.H = H; h.W = W; h.K = K; h.N = N; h.Npad = pl.Npad;
    h.tiles_x = pl.tiles_x; h.tiles_y = pl.tiles_y; h.nchunks = pl.nchunks;
    h.chunks_per_split = pl.chunks_per_split; h.slab_stride = (size_t)B * H * W * N;
    const dim3 grid(pl.ptiles, pl.ntiles, pl.splits);
    const bool slope = h.ya != nullptr;
    if (pl.tw == 16) { if (slope) halo_launch_tw<16, true>(h, pl.tn, grid, st); else halo_launch_tw<16, false>(h, pl.tn, grid, st); }
    else if (pl.tw == 8) { if (slope) halo_launch_tw<8, true>(h, pl.tn, grid, st); else halo_launch_tw<8, false>(h, pl.tn, grid, st); }
    else { if (slope) halo_launch_tw<4, true>(h, pl.tn, grid, st); else halo_launch_tw<4, false>(h, pl.tn, grid, st); }
    if (pl.splits > 1) {
        const size_t n4 = h.slab_stride / 4;
        hipLaunchKernelGGL(conv_halo_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, slab, pl.splits,
                           h.slab_stride, n4, N, bias, act, out, res, subpixel ? H : 0, subpixel ? W : 0);
    }
}
