// cfl_gan.hip -- element-wise, permutation and loss kernels of the MrCGAN post-epoch step
// (generator / discriminator stacks of cfl/models/blocks.py:25-438 and the losses of
// cfl/models/cfl.py:951-1063).  The heavy layers (weight-norm conv / transposed conv / FC) live in
// cfl_conv.hip; everything here is HBM-streaming glue: one pass, float4 where the shape allows,
// fixed-order reductions (single block) so that every scalar is bit-reproducible run to run.
#include <hip/hip_runtime.h>

#include "../../include/cfl_hip.h"

extern int cfl_set_err(int code, const char *fmt, ...);

namespace {

constexpr int EW_THREADS = 256;

__device__ __forceinline__ float ew_act(float x, int act) {
    switch (act) {
        case CFL_EW_LRELU: return x > 0.f ? x : 0.2f * x;
        case CFL_EW_RELU: return fmaxf(x, 0.f);
        case CFL_EW_TANH: return tanhf(x);
        case CFL_EW_SIGMOID: return 1.f / (1.f + __expf(-x));
        default: return x;
    }
}
// derivative from the POST-activation value y
__device__ __forceinline__ float ew_slope(float y, int act) {
    switch (act) {
        case CFL_EW_LRELU: return y > 0.f ? 1.f : (y < 0.f ? 0.2f : 0.f);
        case CFL_EW_RELU: return y > 0.f ? 1.f : 0.f;
        case CFL_EW_TANH: return 1.f - y * y;
        case CFL_EW_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

inline int ew_blocks(int64_t n) {
    int64_t b = (n + EW_THREADS - 1) / EW_THREADS;
    return (int)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

__global__ __launch_bounds__(EW_THREADS) void ew_act_fwd_kernel(const float *x, float *y, int64_t n, int act) {
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS)
        y[i] = ew_act(x[i], act);
}
__global__ __launch_bounds__(EW_THREADS) void ew_act_bwd_kernel(const float *y, const float *dy, float *dx,
                                                                int64_t n, int act) {
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS)
        dx[i] = dy[i] * ew_slope(y[i], act);
}
// acc += dy * act'(y): the backward of a residual join act(r + h) towards h, added onto the gradient that arrives
// through the block's convolutions (one launch instead of act_bwd + axpy)
__global__ __launch_bounds__(EW_THREADS) void ew_act_bwd_add_kernel(const float *y, const float *dy, float *acc,
                                                                    int64_t n, int act) {
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS)
        acc[i] = fmaf(dy[i], ew_slope(y[i], act), acc[i]);
}
__global__ __launch_bounds__(EW_THREADS) void ew_add_act_kernel(const float *a, const float *b, float *y,
                                                                int64_t n, int act) {
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS)
        y[i] = ew_act(a[i] + b[i], act);
}
__global__ __launch_bounds__(EW_THREADS) void ew_axpy_kernel(float alpha, const float *x, float *y, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS)
        y[i] = fmaf(alpha, x[i], y[i]);
}

// y = clip(x * mul + add, lo, hi): the normalisers of cfl/ops.py:66-143 on image / latent batches
__global__ __launch_bounds__(EW_THREADS) void ew_affine_clip_kernel(const float *x, float *y, int64_t n, float mul,
                                                                    float add, float lo, float hi, int has_lo,
                                                                    int has_hi) {
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
        float v = fmaf(x[i], mul, add);
        if (has_lo) v = fmaxf(v, lo);
        if (has_hi) v = fminf(v, hi);
        y[i] = v;
    }
}

// per-channel variant for NHWC images (element i belongs to channel i % C, C <= 4): cfl/ops.py:84-106
struct ChanAffine { float mul[4], add[4]; };
__global__ __launch_bounds__(EW_THREADS) void ew_affine_clip_chan_kernel(const float *x, float *y, int64_t n, int C,
                                                                         ChanAffine p, float lo, float hi,
                                                                         int has_lo, int has_hi) {
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
        const int c = (int)(i % C);
        float v = fmaf(x[i], p.mul[c], p.add[c]);
        if (has_lo) v = fmaxf(v, lo);
        if (has_hi) v = fminf(v, hi);
        y[i] = v;
    }
}

// input transformers of cfl/ops.py:38-63 on NHWC image batches:
//   mode 0: crop / zero-pad window (tf.random_crop with per-sample offsets, or resize_image_with_crop_or_pad when
//           off == NULL: central), then optional per-sample left-right flip (tf.image.random_flip_left_right)
//   mode 1: tf.image.resize_images bilinear, align_corners = False (TF-1: src = dst * in / out), then flip
__global__ __launch_bounds__(EW_THREADS) void image_transform_kernel(const float *x, int H, int W, int C, float *y,
                                                                     int h, int w, int64_t n, const int32_t *off,
                                                                     const int32_t *flip, int mode) {
    for (int64_t o = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; o < n; o += (int64_t)gridDim.x * EW_THREADS) {
        const int c = (int)(o % C);
        int64_t t = o / C;
        int j = (int)(t % w);
        t /= w;
        const int i = (int)(t % h);
        const int64_t b = t / h;
        if (flip && flip[b]) j = w - 1 - j;
        const float *xb = x + b * (int64_t)H * W * C;
        float v;
        if (mode == 0) {
            // central window: crop offset (H - h) / 2 when larger, pad offset (h - H) / 2 when smaller
            const int oy = off ? off[2 * b] : (H >= h ? (H - h) / 2 : -((h - H) / 2));
            const int ox = off ? off[2 * b + 1] : (W >= w ? (W - w) / 2 : -((w - W) / 2));
            const int sy = i + oy, sx = j + ox;
            v = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? xb[((int64_t)sy * W + sx) * C + c] : 0.f;
        } else {
            const float fy = i * ((float)H / (float)h), fx = j * ((float)W / (float)w);
            const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
            const int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
            const float ly = fy - y0, lx = fx - x0;
            const float a0 = xb[((int64_t)y0 * W + x0) * C + c], a1 = xb[((int64_t)y0 * W + x1) * C + c];
            const float b0 = xb[((int64_t)y1 * W + x0) * C + c], b1 = xb[((int64_t)y1 * W + x1) * C + c];
            const float top = a0 + (a1 - a0) * lx, bot = b0 + (b1 - b0) * lx;
            v = top + (bot - top) * ly;
        }
        y[o] = v;
    }
}

// sub-pixel shuffle (cfl/layers.py:212-250): out[b,2h+i,2w+j,c] = in[b,h,w,(2i+j)*Cq+c], Cq = C/4
__global__ __launch_bounds__(EW_THREADS) void subpixel_fwd_kernel(const float *x, float *y, int64_t n, int H,
                                                                  int W, int C, int act) {
    const int Cq = C >> 2;
    for (int64_t o = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; o < n; o += (int64_t)gridDim.x * EW_THREADS) {
        const int c = (int)(o % Cq);
        int64_t t = o / Cq;
        const int ow = (int)(t % (2 * W));
        t /= 2 * W;
        const int oh = (int)(t % (2 * H));
        const int64_t b = t / (2 * H);
        const int64_t src = (((b * H + (oh >> 1)) * W + (ow >> 1)) * C) + ((oh & 1) * 2 + (ow & 1)) * Cq + c;
        y[o] = ew_act(x[src], act);
    }
}
// ... four consecutive channels per thread (Cq % 4 == 0, 16-byte aligned buffers): one float4 load of dy and of y, one float4
// store, 32-bit index arithmetic below 2^31 quads -- the scalar form above spends its time in three 64-bit divisions per
// element (1.7 TB/s on the generator's gradients; this form runs at memory speed)
__global__ __launch_bounds__(EW_THREADS) void subpixel_bwd4_kernel(const float *y, const float *dy, float *dx, int64_t n4,
                                                                   int H, int W, int C, int act) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int Cq4 = C >> 4;      // quads per shuffled pixel
    for (int64_t q = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; q < n4; q += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t pix = q / Cq4;                       // shuffled pixel (b, oh, ow)
        const int c4 = (int)(q - pix * Cq4);
        const int ow = (int)(pix % (2 * W));
        const int64_t t = pix / (2 * W);
        const int oh = (int)(t % (2 * H));
        const int64_t b = t / (2 * H);
        const int64_t src = (((b * H + (oh >> 1)) * W + (ow >> 1)) * C) + ((oh & 1) * 2 + (ow & 1)) * (C >> 2) + 4 * c4;
        const f4 d = *(const f4 *)(dy + 4 * q);
        f4 r = d;
        if (y) {
            const f4 yy = *(const f4 *)(y + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = d[e] * ew_slope(yy[e], act);
        }
        *(f4 *)(dx + src) = r;
    }
}
__global__ __launch_bounds__(EW_THREADS) void subpixel_bwd_kernel(const float *y, const float *dy, float *dx,
                                                                  int64_t n, int H, int W, int C, int act) {
    const int Cq = C >> 2;
    for (int64_t o = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; o < n; o += (int64_t)gridDim.x * EW_THREADS) {
        const int c = (int)(o % Cq);
        int64_t t = o / Cq;
        const int ow = (int)(t % (2 * W));
        t /= 2 * W;
        const int oh = (int)(t % (2 * H));
        const int64_t b = t / (2 * H);
        const int64_t src = (((b * H + (oh >> 1)) * W + (ow >> 1)) * C) + ((oh & 1) * 2 + (ow & 1)) * Cq + c;
        dx[src] = dy[o] * (y ? ew_slope(y[o], act) : 1.f);
    }
}

__global__ __launch_bounds__(EW_THREADS) void concat_cols_kernel(const float *a, int na, const float *b, int nb,
                                                                 int64_t rows, float *out) {
    const int nc = na + nb;
    const int64_t n = rows * nc;
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t r = i / nc;
        const int c = (int)(i % nc);
        out[i] = c < na ? a[r * na + c] : b[r * nb + (c - na)];
    }
}

// conditioning of the cgan discriminator (cfl/models/blocks.py:182-195, 382-395):
// out[p, :C1] = h[p, :], out[p, C1:] = t[p / HW, :]   (tile over the HW pixels of a sample + channel concat)
__global__ __launch_bounds__(EW_THREADS) void tile_concat_kernel(const float *h, int C1, const float *t, int C2,
                                                                 int64_t pixels, int HW, float *out) {
    const int C = C1 + C2;
    const int64_t n = pixels * C;
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t p = i / C;
        const int c = (int)(i % C);
        out[i] = c < C1 ? h[p * C1 + c] : (t ? t[(p / HW) * C2 + (c - C1)] : 0.f);
    }
}
// backward: dh[p, :] = d[p, :C1];  dt[n, c] = sum over the HW pixels of sample n of d[p, C1 + c]
__global__ __launch_bounds__(EW_THREADS) void split_channels_kernel(const float *d, int C1, int C2, int64_t pixels,
                                                                    float *dh) {
    const int C = C1 + C2;
    const int64_t n = pixels * C1;
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS)
        dh[i] = d[(i / C1) * C + (i % C1)];
}
__global__ __launch_bounds__(EW_THREADS) void tile_sum_kernel(const float *d, int C1, int C2, int HW, float *dt) {
    // one block per sample; fixed-order sum over its pixels
    const int C = C1 + C2;
    const int64_t base = (int64_t)blockIdx.x * HW;
    for (int c = threadIdx.x; c < C2; c += EW_THREADS) {
        float acc = 0.f;
        for (int p = 0; p < HW; ++p) acc += d[(base + p) * C + C1 + c];
        dt[(int64_t)blockIdx.x * C2 + c] = acc;
    }
}
// generic strided 2-D copy: dst[r, doff + c] = src[r, soff + c], c < ncols (column slices / concats)
__global__ __launch_bounds__(EW_THREADS) void copy_cols_kernel(const float *src, int sld, int soff, float *dst, int dld,
                                                               int doff, int ncols, int64_t rows) {
    const int64_t n = rows * ncols;
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t r = i / ncols;
        const int c = (int)(i % ncols);
        dst[r * dld + doff + c] = src[r * sld + soff + c];
    }
}

// one_prototype_activations = gather_nd(P, stack([range(B), c])) (cfl/models/base.py, cfl.py:546)
__global__ __launch_bounds__(EW_THREADS) void gather_proto_kernel(const float *P, const int32_t *c, int64_t B,
                                                                  int K, int L, float *out) {
    const int64_t n = B * L;
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t r = i / L;
        int k = c[r];
        k = k < 0 ? 0 : (k >= K ? K - 1 : k);
        out[i] = P[(r * K + k) * L + (i % L)];
    }
}

__device__ __forceinline__ float block_sum(float v, float *red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = EW_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// weight * mean(sigmoid_cross_entropy_with_logits(x, label)); d/dx = weight/n * (sigmoid(x) - label)
__global__ __launch_bounds__(EW_THREADS) void bce_kernel(const float *x, int64_t n, float label, float weight,
                                                         float *loss, float *frac_pos, float *dx, int accumulate) {
    __shared__ float red[EW_THREADS];
    float l = 0.f, cnt = 0.f;
    const float inv = 1.f / (float)n;
    for (int64_t i = threadIdx.x; i < n; i += EW_THREADS) {
        const float v = x[i];
        l += fmaxf(v, 0.f) - v * label + log1pf(__expf(-fabsf(v)));
        cnt += v > 0.f ? 1.f : 0.f;
        if (dx) {
            const float s = 1.f / (1.f + __expf(-v));
            const float d = weight * inv * (s - label);
            dx[i] = accumulate ? dx[i] + d : d;
        }
    }
    l = block_sum(l, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        if (loss) *loss = weight * l * inv;
        if (frac_pos) *frac_pos = cnt * inv;
    }
}

// per-row d = sum_j (a - b)^2 and one of the latent losses of cfl/models/cfl.py:1001-1063
//   mode 0: weight * mean(d)                                 (d_loss_d, g_loss_d without m_enc)
//   mode 1: weight * mean(max(0, sqrt(d + 1e-7) - margin)^2) (g_loss_d with m_enc)
//   mode 2: weight * mean(max(0, margin - sqrt(d + 1e-7))^2) (g_loss_d_neg with m_prj)
// one wave-sized slice of the block per row; da = d loss / d a (overwrite or accumulate)
__global__ __launch_bounds__(EW_THREADS) void rowdist_kernel(const float *a, const float *b, int64_t B, int L,
                                                             int mode, float margin, float weight, float *loss,
                                                             float *da, int accumulate) {
    // one block; wave w owns rows w, w + nwaves, ... (wave-level reductions: no block barrier per row -- the rows are
    // short, L = the latent size, and a barrier per row made this a 120 us kernel); the per-wave totals are added in
    // wave order at the end, so the result does not depend on timing
    __shared__ float tot[EW_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = EW_THREADS / 64;
    float total = 0.f;
    const float invB = 1.f / (float)B;
    for (int64_t r = wave; r < B; r += nw) {
        float d = 0.f;
        for (int j = lane; j < L; j += 64) {
            const float t = a[r * L + j] - b[r * L + j];
            d = fmaf(t, t, d);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
        float li, coef;   // coef = d loss_i / d d
        if (mode == 0) {
            li = d;
            coef = 1.f;
        } else {
            const float s = sqrtf(d + 1e-7f);
            const float h = mode == 1 ? fmaxf(0.f, s - margin) : fmaxf(0.f, margin - s);
            li = h * h;
            coef = (mode == 1 ? 1.f : -1.f) * h / s;   // 2h * (+-1) * 1/(2s)
        }
        total += li;
        if (da)
            for (int j = lane; j < L; j += 64) {
                const float t = a[r * L + j] - b[r * L + j];
                const float gval = weight * invB * coef * 2.f * t;
                da[r * L + j] = accumulate ? da[r * L + j] + gval : gval;
            }
    }
    if (lane == 0) tot[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0 && loss) {
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += tot[w];
        *loss = weight * t * invB;
    }
}

// population mean / variance of all elements, two stages in fixed order
__global__ __launch_bounds__(EW_THREADS) void moments_stage1(const float *x, int64_t n, float *part) {
    __shared__ float red[EW_THREADS];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    float s = 0.f, s2 = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += EW_THREADS) {
        const float v = x[i];
        s += v;
        s2 = fmaf(v, v, s2);
    }
    s = block_sum(s, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = s;
        part[2 * blockIdx.x + 1] = s2;
    }
}
__global__ void moments_stage2(const float *part, int nblocks, int64_t n, float *std_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0, s2 = 0.0;
        for (int i = 0; i < nblocks; ++i) {
            s += part[2 * i];
            s2 += part[2 * i + 1];
        }
        const double mean = s / (double)n;
        double var = s2 / (double)n - mean * mean;
        if (var < 0.0) var = 0.0;
        *std_out = (float)sqrt(var);
    }
}
// X_hat = X + lambda_dra * std(X) * eps[row]  (cfl/models/cfl.py:742-745)
__global__ __launch_bounds__(EW_THREADS) void perturb_kernel(const float *x, const float *eps, int64_t n, int64_t N,
                                                             float lambda_dra, const float *stdp, float *out) {
    const float k = lambda_dra * (*stdp);
    for (int64_t i = blockIdx.x * (int64_t)EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS)
        out[i] = fmaf(k, eps[i / N], x[i]);
}

// gradient penalty (cfl/models/cfl.py:986-991): s_b = ||u_b||, loss = lambda * mean((s_b - 1)^2),
// v_b = d loss / d u_b = lambda * (2/B) (s_b - 1) / s_b * u_b.   One block per row.
__global__ __launch_bounds__(EW_THREADS) void gp_rows_kernel(const float *u, int64_t B, int64_t N, float lambda_gp,
                                                             float *rowloss, float *v) {
    __shared__ float red[EW_THREADS];
    const int64_t r = blockIdx.x;
    float s2 = 0.f;
    for (int64_t j = threadIdx.x; j < N; j += EW_THREADS) {
        const float t = u[r * N + j];
        s2 = fmaf(t, t, s2);
    }
    s2 = block_sum(s2, red);
    const float s = sqrtf(s2);
    const float coef = s > 0.f ? lambda_gp * (2.f / (float)B) * (s - 1.f) / s : 0.f;
    if (v)
        for (int64_t j = threadIdx.x; j < N; j += EW_THREADS) v[r * N + j] = coef * u[r * N + j];
    if (threadIdx.x == 0) rowloss[r] = (s - 1.f) * (s - 1.f);
}
__global__ void gp_final_kernel(const float *rowloss, int64_t B, float lambda_gp, float *loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float t = 0.f;
        for (int64_t i = 0; i < B; ++i) t += rowloss[i];
        *loss = lambda_gp * t / (float)B;
    }
}

inline int done(const char *what) {
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "%s launch failed", what);
}

}  // namespace

extern "C" int cfl_ew_act_fwd(const float *x, float *y, int64_t n, int act, cfl_stream_t stream) {
    if (!x || !y || n < 0 || act < 0 || act > CFL_EW_SIGMOID) return cfl_set_err(CFL_E_SHAPE, "cfl_ew_act_fwd: bad argument");
    if (n == 0) return CFL_OK;
    hipLaunchKernelGGL(ew_act_fwd_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, y, n, act);
    return done("ew_act_fwd");
}

extern "C" int cfl_ew_act_bwd(const float *y, const float *dy, float *dx, int64_t n, int act, cfl_stream_t stream) {
    if (!y || !dy || !dx || n < 0 || act < 0 || act > CFL_EW_SIGMOID) return cfl_set_err(CFL_E_SHAPE, "cfl_ew_act_bwd: bad argument");
    if (n == 0) return CFL_OK;
    hipLaunchKernelGGL(ew_act_bwd_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, y, dy, dx, n, act);
    return done("ew_act_bwd");
}

extern "C" int cfl_ew_act_bwd_add(const float *y, const float *dy, float *acc, int64_t n, int act, cfl_stream_t stream) {
    if (!y || !dy || !acc || n < 0 || act < 0 || act > CFL_EW_SIGMOID) return cfl_set_err(CFL_E_SHAPE, "cfl_ew_act_bwd_add: bad argument");
    if (n == 0) return CFL_OK;
    hipLaunchKernelGGL(ew_act_bwd_add_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, y, dy, acc, n, act);
    return done("ew_act_bwd_add");
}

extern "C" int cfl_ew_add_act(const float *a, const float *b, float *y, int64_t n, int act, cfl_stream_t stream) {
    if (!a || !b || !y || n < 0 || act < 0 || act > CFL_EW_SIGMOID) return cfl_set_err(CFL_E_SHAPE, "cfl_ew_add_act: bad argument");
    if (n == 0) return CFL_OK;
    hipLaunchKernelGGL(ew_add_act_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, a, b, y, n, act);
    return done("ew_add_act");
}

extern "C" int cfl_ew_axpy(float alpha, const float *x, float *y, int64_t n, cfl_stream_t stream) {
    if (!x || !y || n < 0) return cfl_set_err(CFL_E_SHAPE, "cfl_ew_axpy: bad argument");
    if (n == 0) return CFL_OK;
    hipLaunchKernelGGL(ew_axpy_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, alpha, x, y, n);
    return done("ew_axpy");
}

extern "C" int cfl_ew_affine_clip(const float *x, float *y, int64_t n, const CflNorm *norm, cfl_stream_t stream) {
    if (!x || !y || !norm || n < 0) return cfl_set_err(CFL_E_SHAPE, "cfl_ew_affine_clip: bad argument");
    if (n == 0) return CFL_OK;
    hipLaunchKernelGGL(ew_affine_clip_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, y, n,
                       norm->mul, norm->add, norm->lo, norm->hi, norm->has_lo, norm->has_hi);
    return done("ew_affine_clip");
}

extern "C" int cfl_ew_affine_clip_channels(const float *x, float *y, int64_t n, int C, const float *mul,
                                           const float *add, const CflNorm *clip, cfl_stream_t stream) {
    if (!x || !y || !mul || !add || !clip || n < 0 || C < 1 || C > 4 || n % C)
        return cfl_set_err(CFL_E_SHAPE, "cfl_ew_affine_clip_channels: bad argument (1 <= C <= 4)");
    if (n == 0) return CFL_OK;
    ChanAffine p;
    for (int c = 0; c < 4; ++c) { p.mul[c] = c < C ? mul[c] : 1.f; p.add[c] = c < C ? add[c] : 0.f; }
    hipLaunchKernelGGL(ew_affine_clip_chan_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, y, n,
                       C, p, clip->lo, clip->hi, clip->has_lo, clip->has_hi);
    return done("ew_affine_clip_channels");
}

extern "C" int cfl_image_transform(const float *x, int64_t B, int H, int W, int C, float *y, int h, int w,
                                   const int32_t *offsets, const int32_t *flip, int mode, cfl_stream_t stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || h <= 0 || w <= 0 || mode < 0 || mode > 1)
        return cfl_set_err(CFL_E_SHAPE, "cfl_image_transform: bad argument");
    const int64_t n = B * h * w * C;
    hipLaunchKernelGGL(image_transform_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, H, W, C,
                       y, h, w, n, offsets, flip, mode);
    return done("image_transform");
}

extern "C" int cfl_subpixel2x_fwd(const float *x, float *y, int64_t B, int H, int W, int C, int act,
                                  cfl_stream_t stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || act < 0 || act > CFL_EW_SIGMOID)
        return cfl_set_err(CFL_E_SHAPE, "cfl_subpixel2x_fwd: bad argument (C must be a multiple of 4)");
    const int64_t n = B * H * W * C;
    hipLaunchKernelGGL(subpixel_fwd_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, y, n, H, W, C, act);
    return done("subpixel_fwd");
}

extern "C" int cfl_subpixel2x_bwd(const float *y, const float *dy, float *dx, int64_t B, int H, int W, int C,
                                  int act, cfl_stream_t stream) {
    if (!dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || act < 0 || act > CFL_EW_SIGMOID ||
        (!y && act != CFL_EW_NONE))
        return cfl_set_err(CFL_E_SHAPE, "cfl_subpixel2x_bwd: bad argument");
    const int64_t n = B * H * W * C;
    const float *ys = act == CFL_EW_NONE ? nullptr : y;
    if (C % 16 == 0 && !(((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)(ys ? ys : dy)) & 15))
        hipLaunchKernelGGL(subpixel_bwd4_kernel, dim3(ew_blocks(n / 4)), dim3(EW_THREADS), 0, (hipStream_t)stream, ys, dy, dx,
                           n / 4, H, W, C, act);
    else
        hipLaunchKernelGGL(subpixel_bwd_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, ys, dy, dx, n, H, W,
                           C, act);
    return done("subpixel_bwd");
}

extern "C" int cfl_concat_cols(const float *a, int na, const float *b, int nb, int64_t rows, float *out,
                               cfl_stream_t stream) {
    if (!a || !b || !out || na <= 0 || nb <= 0 || rows <= 0) return cfl_set_err(CFL_E_SHAPE, "cfl_concat_cols: bad argument");
    hipLaunchKernelGGL(concat_cols_kernel, dim3(ew_blocks(rows * (na + nb))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       a, na, b, nb, rows, out);
    return done("concat_cols");
}

extern "C" int cfl_tile_concat_channels(const float *h, int C1, const float *t, int C2, int64_t samples, int HW,
                                        float *out, cfl_stream_t stream) {
    if (!h || !out || C1 <= 0 || C2 <= 0 || samples <= 0 || HW <= 0)
        return cfl_set_err(CFL_E_SHAPE, "cfl_tile_concat_channels: bad argument");
    const int64_t pixels = samples * HW;
    hipLaunchKernelGGL(tile_concat_kernel, dim3(ew_blocks(pixels * (C1 + C2))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       h, C1, t, C2, pixels, HW, out);
    return done("tile_concat_channels");
}

extern "C" int cfl_tile_concat_channels_bwd(const float *d, int C1, int C2, int64_t samples, int HW, float *dh,
                                            float *dt, cfl_stream_t stream) {
    if (!d || C1 <= 0 || C2 <= 0 || samples <= 0 || HW <= 0 || (!dh && !dt))
        return cfl_set_err(CFL_E_SHAPE, "cfl_tile_concat_channels_bwd: bad argument");
    const int64_t pixels = samples * HW;
    hipStream_t st = (hipStream_t)stream;
    if (dh) hipLaunchKernelGGL(split_channels_kernel, dim3(ew_blocks(pixels * C1)), dim3(EW_THREADS), 0, st, d, C1, C2, pixels, dh);
    if (dt) hipLaunchKernelGGL(tile_sum_kernel, dim3((unsigned)samples), dim3(EW_THREADS), 0, st, d, C1, C2, HW, dt);
    return done("tile_concat_channels_bwd");
}

extern "C" int cfl_copy_cols(const float *src, int src_ld, int src_off, float *dst, int dst_ld, int dst_off, int ncols,
                             int64_t rows, cfl_stream_t stream) {
    if (!src || !dst || ncols <= 0 || rows <= 0 || src_off < 0 || dst_off < 0 || src_off + ncols > src_ld ||
        dst_off + ncols > dst_ld)
        return cfl_set_err(CFL_E_SHAPE, "cfl_copy_cols: bad argument");
    hipLaunchKernelGGL(copy_cols_kernel, dim3(ew_blocks(rows * ncols)), dim3(EW_THREADS), 0, (hipStream_t)stream, src,
                       src_ld, src_off, dst, dst_ld, dst_off, ncols, rows);
    return done("copy_cols");
}

extern "C" int cfl_gather_prototype(const float *P, const int32_t *c, int64_t B, int K, int L, float *out,
                                    cfl_stream_t stream) {
    if (!P || !c || !out || B <= 0 || K <= 0 || L <= 0) return cfl_set_err(CFL_E_SHAPE, "cfl_gather_prototype: bad argument");
    hipLaunchKernelGGL(gather_proto_kernel, dim3(ew_blocks(B * L)), dim3(EW_THREADS), 0, (hipStream_t)stream, P, c, B, K, L, out);
    return done("gather_prototype");
}

extern "C" int cfl_bce_logits(const float *logits, int64_t n, float label, float weight, float *loss,
                              float *frac_pos, float *dlogits, int accumulate, cfl_stream_t stream) {
    if (!logits || n <= 0) return cfl_set_err(CFL_E_SHAPE, "cfl_bce_logits: bad argument");
    hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(EW_THREADS), 0, (hipStream_t)stream, logits, n, label, weight, loss,
                       frac_pos, dlogits, accumulate);
    return done("bce_logits");
}

extern "C" int cfl_rowdist_loss(const float *a, const float *b, int64_t B, int L, int mode, float margin,
                                float weight, float *loss, float *da, int accumulate, cfl_stream_t stream) {
    if (!a || !b || B <= 0 || L <= 0 || mode < 0 || mode > 2) return cfl_set_err(CFL_E_SHAPE, "cfl_rowdist_loss: bad argument");
    hipLaunchKernelGGL(rowdist_kernel, dim3(1), dim3(EW_THREADS), 0, (hipStream_t)stream, a, b, B, L, mode, margin,
                       weight, loss, da, accumulate);
    return done("rowdist_loss");
}

extern "C" size_t cfl_perturb_workspace_bytes(void) { return (2 * 1024 + 64) * sizeof(float); }

extern "C" int cfl_perturb(const float *x, const float *eps, int64_t B, int64_t N, float lambda_dra, float *out,
                           void *workspace, size_t workspace_bytes, cfl_stream_t stream) {
    if (!x || !eps || !out || !workspace || B <= 0 || N <= 0) return cfl_set_err(CFL_E_SHAPE, "cfl_perturb: bad argument");
    if (workspace_bytes < cfl_perturb_workspace_bytes()) return cfl_set_err(CFL_E_WORKSPACE, "cfl_perturb: workspace too small");
    const int64_t n = B * N;
    int nb = (int)((n + 4095) / 4096);
    if (nb > 1024) nb = 1024;
    float *part = (float *)workspace, *stdp = part + 2 * 1024;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(moments_stage1, dim3(nb), dim3(EW_THREADS), 0, st, x, n, part);
    hipLaunchKernelGGL(moments_stage2, dim3(1), dim3(64), 0, st, part, nb, n, stdp);
    hipLaunchKernelGGL(perturb_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, st, x, eps, n, N, lambda_dra, stdp, out);
    return done("perturb");
}

extern "C" int cfl_grad_penalty(const float *u, int64_t B, int64_t N, float lambda_gp, float *loss, float *v,
                                float *rowloss, cfl_stream_t stream) {
    if (!u || !loss || !rowloss || B <= 0 || N <= 0) return cfl_set_err(CFL_E_SHAPE, "cfl_grad_penalty: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gp_rows_kernel, dim3((unsigned)B), dim3(EW_THREADS), 0, st, u, B, N, lambda_gp, rowloss, v);
    hipLaunchKernelGGL(gp_final_kernel, dim3(1), dim3(64), 0, st, rowloss, B, lambda_gp, loss);
    return done("grad_penalty");
}
