// conv_halo_wgrad.h -- weight gradient of the 3x3 stride-1 'SAME' layers as a direct (halo-tile) bf16x3 matrix-core kernel.
//
//   dW[t = (kh, kw)][ci][co] = sum_{b, y, x}  X[b, y + kh - 1, x + kw - 1, ci] * dYp[b, y, x, co],   dYp = dy * act'(y)
//   db[co]                   = sum_{b, y, x}  dYp[b, y, x, co]
//
// (cfl/layers.py:100-187 of the reference under tf.gradients; the weight-norm chain rule on top of dW stays in the finalize
// kernels of cfl_conv.hip, which read the split-K slabs this kernel writes: slab[z][t * Ci + ci][co], row 9 Ci = db.)
//
// Why not the gathered GEMM (gemm_gather_x3_kernel<.., Im2colXT, DyPre, ..>): as a GEMM the product is (9 Ci) x Co over
// K = pixels, and the gathered form stages -- loads, splits into three bf16 planes, parks -- every x value 9 x (Co / tile) times
// and every dy value (9 Ci / tile) times; measured 130-136 TF/s where the input gradient of the same layer (conv_halo.h) runs
// at 216-224.  Here a workgroup owns a 32 ci x 32 co block of dW for ALL NINE TAPS (9 accumulator tiles per wave) and walks
// pixel tiles of 128 pixels: per tile the x HALO tile (10 x 18 pixels ...) and the dYp tile are loaded once, split once and
// parked in LDS as [pixel][32 channels] rows; the nine taps read their A operand (x^T) from the one halo image at row offsets
// that differ by a per-tap constant.  Both operands arrive pixel-major, i.e. MN-major for this product: the fragments are
// fetched with ds_read_b64_tr_b16 (hardware transpose, lane map pinned by tools/microbench/trread.hip), eight consecutive
// pixels per lane group = four + four consecutive LDS rows.
//   LDS rows are 64 bytes (32 channels of one plane) = two 32-byte blocks of 16 channels; a tr read of a 32-lane half takes
//   32 bytes from each of 8 rows: rows c .. c+3 (four different quarters of the 256-byte bank row) and the same quarters
//   again from the next octet's rows (c + 8 for 16 consecutive pixels; c + 12 for the next image row of the 8-wide tile, whose
//   halo rows are padded to 12 pixels for that purpose, and for the next-but-one row of the 4-wide tile, 2 x 6 rows on).  The block index is XORed with bit 3 (resp. bit 2) of the row number,
//   which those row pairs never share: conflict-free for every tap offset.
//   Two workgroups per CU (59-71 KiB of LDS, <= 256 registers): one stages while the other multiplies.
// Arithmetic: the exact three-way bf16 split and six partial products of gemm_gather_x3_kernel / conv_halo_x3_kernel.
#pragma once
#include "conv_halo.h"

struct HaloWArgs {
    const float *x;        // [B, H, W, Ci]
    const float *dy;       // [B, H, W, Co]
    const float *ya;       // y of the layer (slope source) or nullptr
    float slope_neg, slope_zero;
    int dy_cq;             // > 0: dy and ya arrive 2x sub-pixel shuffled, [B, 2H, 2W, dy_cq], dy_cq = Co / 4 a multiple of 32
    float *slab;           // [splits][9 Ci + 4][Co]
    size_t slab_stride;
    int B, H, W, Ci, Co;
    int tiles_x, tiles_y;  // 128-pixel tiles per image (both 1 for the two-image tiles)
    int ptiles;            // pixel tiles of the whole batch
    int tiles_per_split;   // gridDim.z = ceil(ptiles / tiles_per_split)
};

template <int TW>
struct HaloWGeom {
    static constexpr int TH = TW == 4 ? 4 : 8, IMGS = 128 / (TW * TH);   // 16: 8 rows of one image; 8: two 8x8 images; 4: eight 4x4 images
    static constexpr int HW = TW == 16 ? 18 : (TW == 8 ? 12 : 6);        // halo row length in LDS rows (8-wide: 10 + 2 padding)
    static constexpr int HWV = TW + 2;                          // ... of which carry pixels
    static constexpr int HH = TH + 2, HPI = HH * HW, HP = IMGS * HPI;
    static constexpr int FBIT = TW == 16 ? 3 : 2;               // row-number bit that flips the 16-channel block
    // pixel p of the tile (the K order of the product) -> image, row, column
    __device__ static __forceinline__ void pixel(int p, int &img, int &py, int &px) {
        img = p / (TW * TH);
        const int r = p - img * (TW * TH);
        py = r / TW; px = r - py * TW;
    }
};

// element offset (bf16 units within a plane) of channel c (0 .. 31) of LDS row R; FB = the row-number bit of the swizzle
template <int FB>
__device__ __forceinline__ int halow_off(int R, int c) { return R * 32 + (((c >> 4) ^ ((R >> FB) & 1)) << 4) + (c & 15); }

template <int TW, bool SLOPE>
__global__ __launch_bounds__(256, 2) void conv_halo_wgrad_kernel(HaloWArgs p) {
    using Geo = HaloWGeom<TW>;
    constexpr int HW = Geo::HW, HWV = Geo::HWV, HPI = Geo::HPI, HP = Geo::HP, FB = Geo::FBIT;
    constexpr int XPL = HP * 32, DPL = 128 * 32;               // bf16 per plane
    constexpr int XIT = (HP * 4 + 255) / 256;                  // staging items (8 channels of one halo pixel) per thread
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short *Xs = smem;                                 // [3][HP][32]
    unsigned short *Ds = smem + 3 * XPL;                       // [3][128][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                   // 16-channel block of ci / of co
    const int r16 = lane & 15, q = lane >> 4;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int t_beg = blockIdx.z * p.tiles_per_split;
    const int t_end = min(p.ptiles, t_beg + p.tiles_per_split);

    // ---- staging items: loop-invariant halves ----
    int xrow[XIT], xcode[XIT];                                 // LDS row (-1: none); image << 16 | halo row << 8 | halo column
#pragma unroll
    for (int u = 0; u < XIT; ++u) {
        const int i = tid + 256 * u, hp = i >> 2;
        const int img = hp / HPI, r = hp - img * HPI, hy = r / HW, hx = r - hy * HW;
        xrow[u] = (hp < HP && hx < HWV) ? hp : -1;
        xcode[u] = (img << 16) | (hy << 8) | hx;
    }
    const int xoct = (tid & 3) * 8;
    int dcode[2];                                              // image << 16 | row << 8 | column of this thread's two dy pixels
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int img, py, px;
        Geo::pixel((tid + 256 * u) >> 2, img, py, px);
        dcode[u] = (img << 16) | (py << 8) | px;
    }

    gg_f32x4 xreg[XIT][2], dreg[2][2], yreg[SLOPE ? 2 : 1][2];
    unsigned okmask = 0;      // bit u: x item u is inside the image / batch; bit 8 + u: dy item u
    auto request = [&](int tile) {
        int b0, oy0, ox0;
        if (Geo::IMGS == 1) {
            const int per = p.tiles_x * p.tiles_y;
            b0 = tile / per;
            const int r = tile - b0 * per, ty = r / p.tiles_x;
            oy0 = ty * Geo::TH; ox0 = (r - ty * p.tiles_x) * TW;
        } else {
            b0 = tile * Geo::IMGS; oy0 = 0; ox0 = 0;
        }
#pragma unroll
        for (int u = 0; u < XIT; ++u) {
            const int b = b0 + (xcode[u] >> 16), iy = oy0 + ((xcode[u] >> 8) & 255) - 1, ix = ox0 + (xcode[u] & 255) - 1;
            const bool ok = xrow[u] >= 0 && b < p.B && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const size_t o = ok ? (((size_t)b * p.H + iy) * p.W + ix) * p.Ci + ci0 + xoct : (size_t)0;   // safe address, selected below
            xreg[u][0] = *(const gg_f32x4 *)(p.x + o); xreg[u][1] = *(const gg_f32x4 *)(p.x + o + 4);
            okmask = ok ? (okmask | (1u << u)) : (okmask & ~(1u << u));     // (applied when the values are parked: no wait here)
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int b = b0 + (dcode[u] >> 16), oy = oy0 + ((dcode[u] >> 8) & 255), ox = ox0 + (dcode[u] & 255);
            const bool ok = b < p.B && oy < p.H && ox < p.W;
            size_t o = 0;
            if (p.dy_cq) {   // this workgroup's 32 channels lie in one quarter = one sub-pixel (i, j) of the shuffled gradient
                const int qd = co0 / p.dy_cq, c0 = co0 - qd * p.dy_cq;
                o = ok ? (((size_t)b * 2 * p.H + 2 * oy + (qd >> 1)) * (2 * p.W) + 2 * ox + (qd & 1)) * p.dy_cq + c0 + xoct : (size_t)0;
            } else {
                o = ok ? (((size_t)b * p.H + oy) * p.W + ox) * p.Co + co0 + xoct : (size_t)0;
            }
            dreg[u][0] = *(const gg_f32x4 *)(p.dy + o); dreg[u][1] = *(const gg_f32x4 *)(p.dy + o + 4);
            okmask = ok ? (okmask | (1u << (8 + u))) : (okmask & ~(1u << (8 + u)));
            if (SLOPE) {
                yreg[u][0] = *(const gg_f32x4 *)(p.ya + o);
                yreg[u][1] = *(const gg_f32x4 *)(p.ya + o + 4);
            }
        }
    };
    float bsum[8];         // column sums of dYp over this thread's pixels (both items: the same 8 channels): the bias gradient
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    auto split_store = [&](unsigned short *dst, int plane, const float (&v)[8]) {
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
        *(gg_u32x4 *)dst = ph;
        *(gg_u32x4 *)(dst + plane) = pm;
        *(gg_u32x4 *)(dst + 2 * plane) = pl;
    };
    auto park = [&]() {
#pragma unroll
        for (int u = 0; u < XIT; ++u) {
            if (xrow[u] < 0) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (okmask >> u) & 1 ? xreg[u][e >> 2][e & 3] : 0.f;
            split_store(Xs + halow_off<FB>(xrow[u], xoct), XPL, v);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float d = dreg[u][e >> 2][e & 3];
                if (SLOPE) {
                    const float yy = yreg[u][e >> 2][e & 3];
                    d *= yy > 0.f ? 1.f : (yy < 0.f ? p.slope_neg : p.slope_zero);
                }
                d = (okmask >> (8 + u)) & 1 ? d : 0.f;
                v[e] = d;
                bsum[e] += d;
            }
            split_store(Ds + halow_off<3>((tid + 256 * u) >> 2, xoct), DPL, v);
        }
    };

    // ---- fragment addresses (tr reads): lane 4a + pp of a 16-lane group supplies LDS row (first of its four) + a, channels
    // 4 pp .. 4 pp + 3 of the 16-channel block; the group receives pixels 8 q .. 8 q + 3 (second read: + 4) of the K step
    const int a4 = r16 >> 2, pp = r16 & 3;
    typedef __attribute__((address_space(3))) gg_s16x4 *lds_p;
    auto tr2 = [&](const unsigned short *img, int plane_elems, int lv, int o0, int o1) -> gg_bf16x8 {
        const gg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + lv * plane_elems + o0));
        const gg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + lv * plane_elems + o1));
        const gg_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(gg_bf16x8, v);
    };
    // halo row (tap (0, 0)) of the first pixel of this lane's two 4-pixel runs in each of the four K steps
    int xr0[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int img, py, px;
            Geo::pixel(32 * ks + 8 * q + 4 * h, img, py, px);
            xr0[ks][h] = img * HPI + py * HW + px + a4;
        }

    gg_f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (gg_f32x4){0.f, 0.f, 0.f, 0.f};

    if (t_beg < t_end) request(t_beg);
    for (int tile = t_beg; tile < t_end; ++tile) {
        park();
        __syncthreads();
        {
            // the next tile's operands travel while this one is multiplied (past the end: the last tile again -- a load
            // inside a uniform branch would make hipcc drain the queue at the join)
            const int nx = tile + 1 < t_end ? tile + 1 : tile;
            request(nx);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            gg_bf16x8 bf[3];
            {
                const int r0 = 32 * ks + 8 * q + a4, r1 = r0 + 4;
                const int o0 = halow_off<3>(r0, wn * 16 + 4 * pp), o1 = halow_off<3>(r1, wn * 16 + 4 * pp);
#pragma unroll
                for (int lv = 0; lv < 3; ++lv) bf[lv] = tr2(Ds, DPL, lv, o0, o1);
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                // the three taps of one filter row together: consecutive MFMAs hit different accumulators
                gg_bf16x8 af[3][3];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int toff = kh * HW + kw;
                    const int r0 = xr0[ks][0] + toff, r1 = xr0[ks][1] + toff;
                    const int o0 = halow_off<FB>(r0, wm * 16 + 4 * pp), o1 = halow_off<FB>(r1, wm * 16 + 4 * pp);
#pragma unroll
                    for (int lv = 0; lv < 3; ++lv) af[kw][lv] = tr2(Xs, XPL, lv, o0, o1);
                }
                // small terms first (the order of the other bf16x3 kernels)
#define HALOW_X3(LA, LB)                                                                                              \
    _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) acc[3 * kh + kw] =                                               \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kw][LA], bf[LB], acc[3 * kh + kw], 0, 0, 0);
                HALOW_X3(1, 1) HALOW_X3(2, 0) HALOW_X3(0, 2) HALOW_X3(1, 0) HALOW_X3(0, 1) HALOW_X3(0, 0)
#undef HALOW_X3
            }
        }
        __syncthreads();       // every wave is done with the images before the next park
    }

    // ---- epilogue: C layout col = lane & 15 (co), rows 4 q + e (ci) ----
    float *slab = p.slab + (size_t)blockIdx.z * p.slab_stride;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            slab[((size_t)t * p.Ci + ci0 + wm * 16 + 4 * q + e) * p.Co + co0 + wn * 16 + r16] = acc[t][e];
    if (blockIdx.x == 0) {
        // bias row: this thread's 8 channels (octet tid & 3) summed over its pixels; 64 threads share an octet
        float *red = (float *)smem;                            // [8][256]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[e * 256 + tid] = bsum[e];
        __syncthreads();
        if (tid < 32) {
            const int oct = tid >> 3, e = tid & 7;
            float s = 0.f;
            for (int j = 0; j < 64; ++j) s += red[e * 256 + 4 * j + oct];
            slab[(size_t)9 * p.Ci * p.Co + co0 + oct * 8 + e] = s;
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------
struct HaloWPlan {
    bool ok;
    int tw, tiles_x, tiles_y, ptiles, splits, tiles_per_split;
};

static inline bool halo_wgrad_off() {
    static const int off = [] { const char *e = getenv("CFL_DEBUG_NOHALO_WGRAD"); return (e && atoi(e) > 0) ? 1 : 0; }();
    return off != 0 || halo_off();
}

static inline HaloWPlan halo_wgrad_plan(int B, int H, int W, int Ci, int Co) {
    HaloWPlan pl;
    memset(&pl, 0, sizeof(pl));
    if (halo_wgrad_off() || !gg_use_x3()) return pl;
    if (Ci % 32 != 0 || Co % 32 != 0) return pl;
    if (W % 16 == 0 && H % 8 == 0) { pl.tw = 16; pl.tiles_x = W / 16; pl.tiles_y = H / 8; pl.ptiles = B * pl.tiles_x * pl.tiles_y; }
    else if (W == 8 && H == 8) { pl.tw = 8; pl.tiles_x = pl.tiles_y = 1; pl.ptiles = (B + 1) / 2; }
    else return pl;   // (4x4 images: the eight-image tile was built and measured -- 288 halo rows, 78 KiB of LDS, 5 staging items
                      // per thread spill past 256 registers -- 0.125 / 0.093 ms against the gathered GEMM's 0.129 / 0.095: not kept)
    const size_t big = (size_t)(Ci > Co ? Ci : Co);
    if ((size_t)B * H * W * big >= 0xffffffffull) return pl;
    // about 2048 workgroups (two per CU, four rounds), at least 2 pixel tiles per split, <= 64 slabs (256 for a single 32 x 32 block)
    const long long tiles = (long long)(Ci / 32) * (Co / 32);
    static const int target = [] { const char *e = getenv("CFL_DEBUG_HALO_WGRAD_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 2048; }();
    long long s = (target + tiles - 1) / tiles;
    if (s > pl.ptiles / 2) s = pl.ptiles / 2;
    const long long cap = tiles >= 4 ? 64 : 256 / tiles;        // (each slab is one more pass for the reduction that follows)
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    pl.tiles_per_split = (int)((pl.ptiles + s - 1) / s);
    pl.splits = (pl.ptiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
    pl.ok = true;
    return pl;
}

template <int TW, bool SLOPE>
static inline void halo_wgrad_launch(const HaloWArgs &a, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)(3 * HaloWGeom<TW>::HP * 32 + 3 * 128 * 32) * sizeof(unsigned short);
    static_assert(lds >= 8 * 256 * sizeof(float), "bias reduction scratch");
    auto kern = conv_halo_wgrad_kernel<TW, SLOPE>;
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load() & bit)) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            cfl_set_err(CFL_E_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed for the halo weight-gradient kernel", lds);
            return;
        }
        done.fetch_or(bit);
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
}

// slab: [splits][9 Ci + 4][Co] (rows 9 Ci + 1 .. + 3 are never read)
static inline void halo_wgrad(const HaloWPlan &pl, int B, int H, int W, int Ci, int Co, const float *x, const float *dy,
                              const float *ya, int slope_act, float *slab, size_t slab_stride, hipStream_t st, int dy_cq = 0) {
    HaloWArgs h;
    memset(&h, 0, sizeof(h));
    h.x = x; h.dy = dy; h.ya = (ya && slope_act != 0) ? ya : nullptr;
    h.slope_neg = slope_act == 1 ? 0.2f : 0.f; h.slope_zero = 0.f;
    h.slab = slab; h.slab_stride = slab_stride; h.dy_cq = dy_cq;
    h.B = B; h.H = H; h.W = W; h.Ci = Ci; h.Co = Co;
    h.tiles_x = pl.tiles_x; h.tiles_y = pl.tiles_y; h.ptiles = pl.ptiles; h.tiles_per_split = pl.tiles_per_split;
    const dim3 grid(Ci / 32, Co / 32, pl.splits);
    const bool slope = h.ya != nullptr;
    if (pl.tw == 16) { if (slope) halo_wgrad_launch<16, true>(h, grid, st); else halo_wgrad_launch<16, false>(h, grid, st); }
    else { if (slope) halo_wgrad_launch<8, true>(h, grid, st); else halo_wgrad_launch<8, false>(h, grid, st); }
}
