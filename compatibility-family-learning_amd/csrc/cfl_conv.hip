// cfl_conv.hip -- weight-normalised convolution building blocks (NHWC activations, HWIO
// filters, TensorFlow 'SAME' padding) of the cfl conv encoder / MrCGAN stacks:
//   forward   y = act( conv(x, g * V / ||V||) + b )                 cfl/layers.py:100-187
//   backward  dx, dV, dg, db (dV includes the weight-norm correction and the L2 term)
// First-correct form: implicit GEMM on fp32 MFMA through the generic gathered-operand GEMM
// of gemm_gather.h (scalar im2col gathers).  See DESIGN.md section 6 for what remains.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../../include/cfl_hip.h"
#include "gemm_gather.h"
#include "conv_halo.h"
#include "conv_halo_wgrad.h"
#include "conv_stem.h"

extern int cfl_set_err(int code, const char *fmt, ...);

struct ConvGeom {
    int B, H, W, Ci, Co, KH, KW, S, OH, OW, pt, pl, act;
    gg_div dOW, dOH, dW, dH, dCi, dCo, dKW, dS;   // fast division by the geometry constants
};
static void geom_divs(ConvGeom *g) {
    g->dOW = gg_make_div(g->OW); g->dOH = gg_make_div(g->OH); g->dW = gg_make_div(g->W); g->dH = gg_make_div(g->H);
    g->dCi = gg_make_div(g->Ci); g->dCo = gg_make_div(g->Co); g->dKW = gg_make_div(g->KW); g->dS = gg_make_div(g->S);
}

static inline void same_pad(int n, int k, int s, int *out, int *lo) {
    *out = (n + s - 1) / s;
    int total = (*out - 1) * s + k - n;
    if (total < 0) total = 0;
    *lo = total / 2;
}

static int make_geom(const CflConv *c, ConvGeom *g) {
    if (!c) return cfl_set_err(CFL_E_SHAPE, "conv shape is NULL");
    if (c->B <= 0 || c->H <= 0 || c->W <= 0 || c->Ci <= 0 || c->Co <= 0 || c->KH <= 0 || c->KW <= 0 ||
        c->stride <= 0 || c->act < 0 || c->act > 2)
        return cfl_set_err(CFL_E_SHAPE, "bad conv shape");
    g->B = c->B; g->H = c->H; g->W = c->W; g->Ci = c->Ci; g->Co = c->Co; g->KH = c->KH; g->KW = c->KW;
    g->S = c->stride; g->act = c->act;
    same_pad(c->H, c->KH, c->stride, &g->OH, &g->pt);
    same_pad(c->W, c->KW, c->stride, &g->OW, &g->pl);
    geom_divs(g);
    return CFL_OK;
}

__device__ __forceinline__ float act_apply(float y, int act) {
    if (act == 1) return y > 0.f ? y : 0.2f * y;   // lrelu(x) = relu(x) - 0.2 relu(-x), cfl/ops.py:10-12
    if (act == 2) return fmaxf(y, 0.f);
    return y;
}
// d act / d pre-activation, from the POST-activation value (both are monotone through 0)
// (TF: relu'(0) = 0, so lrelu(x) = relu(x) - 0.2 relu(-x) has slope 0 exactly at 0)
__device__ __forceinline__ float act_slope(float y, int act) {
    if (act == 1) return y > 0.f ? 1.f : (y < 0.f ? 0.2f : 0.f);
    if (act == 2) return y > 0.f ? 1.f : 0.f;
    return 1.f;
}

// the same slope from uniform constants (no branch on `act` per element): y > 0 ? 1 : (y < 0 ? neg : zer)
struct ActSlope {
    float neg, zer;
    __device__ __forceinline__ float operator()(float y) const { return y > 0.f ? 1.f : (y < 0.f ? neg : zer); }
};
__device__ __forceinline__ ActSlope act_slope_consts(int act) {
    return act == 1 ? ActSlope{0.2f, 0.f} : (act == 2 ? ActSlope{0.f, 0.f} : ActSlope{1.f, 1.f});
}

// ---- per-output-channel weight-norm scale: g / sqrt(max(sum V^2, 1e-12)) ----------------
__global__ __launch_bounds__(256) void conv_scale_kernel(const float *V, const float *g, int rows, int Co,
                                                         float *scale, float *n2out) {
    const int co = blockIdx.x;
    float acc = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const float v = V[(size_t)r * Co + co];
        acc = fmaf(v, v, acc);
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float n2 = fmaxf(red[0], 1e-12f);
        n2out[co] = n2;
        scale[co] = (g ? g[co] : 1.f) * rsqrtf(n2);
    }
}

// ---- operand functors -------------------------------------------------------------------
struct Im2colX {   // A(m = (b,oh,ow), k = (kh,kw,ci)) = x[b, oh*S+kh-pt, ow*S+kw-pl, ci]
    const float *x; ConvGeom g;
    __device__ float operator()(int m, int k) const {
        int ow, t, oh, b, ci, t2, kw, kh;
        gg_divmod(m, g.dOW, t, ow); gg_divmod(t, g.dOH, b, oh);
        gg_divmod(k, g.dCi, t2, ci); gg_divmod(t2, g.dKW, kh, kw);
        const int ih = oh * g.S + kh - g.pt, iw = ow * g.S + kw - g.pl;
        if (ih < 0 || ih >= g.H || iw < 0 || iw >= g.W) return 0.f;
        return x[(((size_t)b * g.H + ih) * g.W + iw) * g.Ci + ci];
    }
    __device__ gg_f32x4 v4(int m, int k) const {   // k .. k+3 = 4 consecutive ci (Ci % 4 == 0)
        int ow, t, oh, b, ci, t2, kw, kh;
        gg_divmod(m, g.dOW, t, ow); gg_divmod(t, g.dOH, b, oh);
        gg_divmod(k, g.dCi, t2, ci); gg_divmod(t2, g.dKW, kh, kw);
        const int ih = oh * g.S + kh - g.pt, iw = ow * g.S + kw - g.pl;
        if (ih < 0 || ih >= g.H || iw < 0 || iw >= g.W) return (gg_f32x4){0.f, 0.f, 0.f, 0.f};
        return *(const gg_f32x4 *)(x + (((size_t)b * g.H + ih) * g.W + iw) * g.Ci + ci);
    }
    // two-phase form (gemm_gather.h): pixel index decomposed once, tap index once per K step
    struct Fix { int bH, ih0, iw0; };
    struct Str { int kh, kw, ci; };
    __device__ Fix fix(int m) const {
        int ow, t, oh, b;
        gg_divmod(m, g.dOW, t, ow); gg_divmod(t, g.dOH, b, oh);
        return Fix{b * g.H, oh * g.S - g.pt, ow * g.S - g.pl};
    }
    __device__ Str stream(int k) const {
        int ci, t2, kw, kh;
        gg_divmod(k, g.dCi, t2, ci); gg_divmod(t2, g.dKW, kh, kw);
        return Str{kh, kw, ci};
    }
    __device__ gg_f32x4 get(const Fix &a, const Str &s, bool ok) const {
        const int ih = a.ih0 + s.kh, iw = a.iw0 + s.kw;
        ok = ok && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
        const size_t o = ok ? ((size_t)(a.bH + ih) * g.W + iw) * g.Ci + s.ci : (size_t)s.ci;   // safe address
        const gg_f32x4 v = *(const gg_f32x4 *)(x + o);
        return ok ? v : (gg_f32x4){0.f, 0.f, 0.f, 0.f};
    }
};
struct FilterKN {  // B(k, n = co) = V[k][co]
    const float *V; int Co;
    __device__ float operator()(int k, int n) const { return V[(size_t)k * Co + n]; }
    __device__ gg_f32x4 v4(int k, int n) const { return *(const gg_f32x4 *)(V + (size_t)k * Co + n); }  // n .. n+3
    struct Fix { int n; };
    struct Str { size_t o; };
    __device__ Fix fix(int n) const { return Fix{n}; }
    __device__ Str stream(int k) const { return Str{(size_t)k * Co}; }
    __device__ gg_f32x4 get(const Fix &a, const Str &s, bool ok) const {
        const gg_f32x4 v = *(const gg_f32x4 *)(V + (ok ? s.o + a.n : (size_t)0));
        return ok ? v : (gg_f32x4){0.f, 0.f, 0.f, 0.f};
    }
};
struct StoreFwd {
    float *y; const float *scale, *bias; int Co, act;
    const float *res;   // residual [M, Co] added before the activation, or nullptr
    int subH, subW;     // > 0: 2x sub-pixel shuffled store, m = (b, oh, ow) of an subH x subW output (conv_halo.h: HaloArgs::subpixel)
    __device__ void operator()(int m, int n, float v, int) const {
        v = v * scale[n] + (bias ? bias[n] : 0.f);      // (the order of the separate kernels: (conv + b) + residual)
        if (res) v += res[(size_t)m * Co + n];
        v = act_apply(v, act);
        if (subH > 0) {
            const int ow = m % subW, t = m / subW, oh = t % subH, b = t / subH;
            const int Cq = Co >> 2, q = n / Cq, c = n - q * Cq;
            y[((((size_t)b * 2 * subH + 2 * oh + (q >> 1)) * (2 * subW)) + 2 * ow + (q & 1)) * Cq + c] = v;
        } else {
            y[(size_t)m * Co + n] = v;
        }
    }
};
struct DyGather {  // A(m = (b,ih,iw), k = (kh,kw,co)) = dy_pre[b,(ih+pt-kh)/S,(iw+pl-kw)/S,co] * scale[co]
    const float *dy, *y, *scale; ConvGeom g;
    __device__ float operator()(int m, int k) const {
        int iw, t, ih, b, co, t2, kw, kh, oh, ow, rh, rw;
        gg_divmod(m, g.dW, t, iw); gg_divmod(t, g.dH, b, ih);
        gg_divmod(k, g.dCo, t2, co); gg_divmod(t2, g.dKW, kh, kw);
        const int nh = ih + g.pt - kh, nw = iw + g.pl - kw;
        if (nh < 0 || nw < 0) return 0.f;
        gg_divmod(nh, g.dS, oh, rh); gg_divmod(nw, g.dS, ow, rw);
        if (rh || rw) return 0.f;
        if (oh >= g.OH || ow >= g.OW) return 0.f;
        const size_t o = (((size_t)b * g.OH + oh) * g.OW + ow) * g.Co + co;
        return dy[o] * act_slope(y[o], g.act) * scale[co];
    }
    __device__ gg_f32x4 v4(int m, int k) const {   // k .. k+3 = 4 consecutive co (Co % 4 == 0)
        const gg_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        int iw, t, ih, b, co, t2, kw, kh, oh, ow, rh, rw;
        gg_divmod(m, g.dW, t, iw); gg_divmod(t, g.dH, b, ih);
        gg_divmod(k, g.dCo, t2, co); gg_divmod(t2, g.dKW, kh, kw);
        const int nh = ih + g.pt - kh, nw = iw + g.pl - kw;
        if (nh < 0 || nw < 0) return zero;
        gg_divmod(nh, g.dS, oh, rh); gg_divmod(nw, g.dS, ow, rw);
        if (rh || rw) return zero;
        if (oh >= g.OH || ow >= g.OW) return zero;
        const size_t o = (((size_t)b * g.OH + oh) * g.OW + ow) * g.Co + co;
        const gg_f32x4 d = *(const gg_f32x4 *)(dy + o), yy = *(const gg_f32x4 *)(y + o),
                       sc = *(const gg_f32x4 *)(scale + co);
        gg_f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = d[e] * act_slope(yy[e], g.act) * sc[e];
        return r;
    }
    struct Fix { int bOH, ihp, iwp; };
    struct Str { int kh, kw, co; };
    __device__ Fix fix(int m) const {
        int iw, t, ih, b;
        gg_divmod(m, g.dW, t, iw); gg_divmod(t, g.dH, b, ih);
        return Fix{b * g.OH, ih + g.pt, iw + g.pl};
    }
    __device__ Str stream(int k) const {
        int co, t2, kw, kh;
        gg_divmod(k, g.dCo, t2, co); gg_divmod(t2, g.dKW, kh, kw);
        return Str{kh, kw, co};
    }
    __device__ gg_f32x4 get(const Fix &a, const Str &s, bool ok) const {
        const int nh = a.ihp - s.kh, nw = a.iwp - s.kw;
        int oh, ow, rh, rw;
        gg_divmod(nh < 0 ? 0 : nh, g.dS, oh, rh); gg_divmod(nw < 0 ? 0 : nw, g.dS, ow, rw);
        ok = ok && nh >= 0 && nw >= 0 && (rh | rw) == 0 && oh < g.OH && ow < g.OW;
        const size_t o = ok ? ((size_t)(a.bOH + oh) * g.OW + ow) * g.Co + s.co : (size_t)s.co;   // safe address
        const gg_f32x4 d = *(const gg_f32x4 *)(dy + o), yy = *(const gg_f32x4 *)(y + o),
                       sc = *(const gg_f32x4 *)(scale + s.co);
        const ActSlope sl = act_slope_consts(g.act);
        gg_f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = ok ? d[e] * sl(yy[e]) * sc[e] : 0.f;
        return r;
    }
};
struct FilterT {   // B(k = (kh,kw,co), n = ci) = V[kh,kw,ci,co]
    const float *V; ConvGeom g;
    __device__ float operator()(int k, int n) const {
        int co, t;  // t = kh*KW + kw
        gg_divmod(k, g.dCo, t, co);
        return V[((size_t)t * g.Ci + n) * g.Co + co];
    }
    __device__ gg_f32x4 v4(int k, int n) const {   // k .. k+3 = 4 consecutive co
        int co, t;
        gg_divmod(k, g.dCo, t, co);
        return *(const gg_f32x4 *)(V + ((size_t)t * g.Ci + n) * g.Co + co);
    }
    struct Fix { int nCo; };
    struct Str { size_t o; };
    __device__ Fix fix(int n) const { return Fix{n * g.Co}; }
    __device__ Str stream(int k) const {
        int co, t;
        gg_divmod(k, g.dCo, t, co);
        return Str{(size_t)t * g.Ci * g.Co + co};
    }
    __device__ gg_f32x4 get(const Fix &a, const Str &s, bool ok) const {
        const gg_f32x4 v = *(const gg_f32x4 *)(V + (ok ? s.o + a.nCo : (size_t)0));
        return ok ? v : (gg_f32x4){0.f, 0.f, 0.f, 0.f};
    }
};
// ---- stride-2 input gradient by output-pixel parity ------------------------------------------
// With stride 2 an input pixel (ih, iw) only meets the taps kh == (ih + pt) mod 2, kw == (iw + pl) mod 2,
// so the plain gather multiplies 3/4 zeros.  Per parity class (ph, pw) = (ih & 1, iw & 1) the gradient is
// a dense GEMM over the taps kh = 2 kh' + oh0, kw = 2 kw' + ow0 (K = ceil(KH/2) ceil(KW/2) Co).
// parity class of a stride-2 input-gradient product: from the members, or from the grid (four classes stacked in z)
struct S2Class { int ph, pw, oh0, ow0; };
__device__ __forceinline__ S2Class s2_class(int nsp, int ph, int pw, int oh0, int ow0, int pt, int pl) {
    if (nsp <= 0) return S2Class{ph, pw, oh0, ow0};
    const int c = (int)blockIdx.z / nsp, h = c >> 1, w = c & 1;
    return S2Class{h, w, (h + pt) & 1, (w + pl) & 1};
}
struct DyGatherS2 {   // A(m = (b, i, j), k = (kh', kw', co)), input pixel (2i + ph, 2j + pw)
    const float *dy, *y, *scale; ConvGeom g; int ph_, pw_, oh0_, ow0_, KH2, KW2, H2, W2; gg_div dKW2, dH2, dW2;
    int nsp;   // > 0: the four classes are stacked in grid z (class = blockIdx.z / nsp); 0: the class of the members
    __device__ __forceinline__ S2Class cls() const { return s2_class(nsp, ph_, pw_, oh0_, ow0_, g.pt, g.pl); }
    __device__ bool locate(int m, int k, size_t *o, int *co) const {
        int j, t, i, b, t2, a0, a1;
        gg_divmod(m, dW2, t, j); gg_divmod(t, dH2, b, i);
        gg_divmod(k, g.dCo, t2, *co); gg_divmod(t2, dKW2, a1, a0);
        const S2Class c = cls();
        const int kw = 2 * a0 + c.ow0, kh = 2 * a1 + c.oh0;
        if (kh >= g.KH || kw >= g.KW) return false;
        const int nh = 2 * i + c.ph + g.pt - kh, nw = 2 * j + c.pw + g.pl - kw;   // even by construction
        if (nh < 0 || nw < 0) return false;
        const int oh = nh >> 1, ow = nw >> 1;
        if (oh >= g.OH || ow >= g.OW) return false;
        *o = (((size_t)b * g.OH + oh) * g.OW + ow) * g.Co + *co;
        return true;
    }
    __device__ float operator()(int m, int k) const {
        size_t o; int co;
        if (!locate(m, k, &o, &co)) return 0.f;
        return dy[o] * act_slope(y[o], g.act) * scale[co];
    }
    __device__ gg_f32x4 v4(int m, int k) const {
        size_t o; int co;
        if (!locate(m, k, &o, &co)) return (gg_f32x4){0.f, 0.f, 0.f, 0.f};
        const gg_f32x4 d = *(const gg_f32x4 *)(dy + o), yy = *(const gg_f32x4 *)(y + o),
                       sc = *(const gg_f32x4 *)(scale + co);
        gg_f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = d[e] * act_slope(yy[e], g.act) * sc[e];
        return r;
    }
    struct Fix { int bOH, nh0, nw0; };
    struct Str { int kh, kw, co; };   // kh < 0: tap outside the filter
    __device__ Fix fix(int m) const {
        int j, t, i, b;
        gg_divmod(m, dW2, t, j); gg_divmod(t, dH2, b, i);
        const S2Class c = cls();
        return Fix{b * g.OH, 2 * i + c.ph + g.pt, 2 * j + c.pw + g.pl};
    }
    __device__ Str stream(int k) const {
        int co, t2, a0, a1;
        gg_divmod(k, g.dCo, t2, co); gg_divmod(t2, dKW2, a1, a0);
        const S2Class c = cls();
        const int kw = 2 * a0 + c.ow0, kh = 2 * a1 + c.oh0;
        return Str{(kh >= g.KH || kw >= g.KW) ? -1 : kh, kw, co};
    }
    __device__ gg_f32x4 get(const Fix &a, const Str &s, bool ok) const {
        const int nh = a.nh0 - s.kh, nw = a.nw0 - s.kw;   // even by construction
        const int oh = nh >> 1, ow = nw >> 1;
        ok = ok && s.kh >= 0 && nh >= 0 && nw >= 0 && oh < g.OH && ow < g.OW;
        const size_t o = ok ? ((size_t)(a.bOH + oh) * g.OW + ow) * g.Co + s.co : (size_t)s.co;   // safe address
        const gg_f32x4 d = *(const gg_f32x4 *)(dy + o), yy = *(const gg_f32x4 *)(y + o),
                       sc = *(const gg_f32x4 *)(scale + s.co);
        const ActSlope sl = act_slope_consts(g.act);
        gg_f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = ok ? d[e] * sl(yy[e]) * sc[e] : 0.f;
        return r;
    }
};
struct FilterTS2 {    // B(k = (kh', kw', co), n = ci) = V[2kh'+oh0, 2kw'+ow0, ci, co]
    const float *V; ConvGeom g; int oh0_, ow0_, KW2; gg_div dKW2;
    int nsp;   // (as DyGatherS2)
    __device__ __forceinline__ S2Class cls() const { return s2_class(nsp, 0, 0, oh0_, ow0_, g.pt, g.pl); }
    __device__ float operator()(int k, int n) const {
        int co, t2, a0, a1;
        gg_divmod(k, g.dCo, t2, co); gg_divmod(t2, dKW2, a1, a0);
        const S2Class c = cls();
        const int kw = 2 * a0 + c.ow0, kh = 2 * a1 + c.oh0;
        if (kh >= g.KH || kw >= g.KW) return 0.f;
        return V[((size_t)(kh * g.KW + kw) * g.Ci + n) * g.Co + co];
    }
    __device__ gg_f32x4 v4(int k, int n) const {
        int co, t2, a0, a1;
        gg_divmod(k, g.dCo, t2, co); gg_divmod(t2, dKW2, a1, a0);
        const S2Class c = cls();
        const int kw = 2 * a0 + c.ow0, kh = 2 * a1 + c.oh0;
        if (kh >= g.KH || kw >= g.KW) return (gg_f32x4){0.f, 0.f, 0.f, 0.f};
        return *(const gg_f32x4 *)(V + ((size_t)(kh * g.KW + kw) * g.Ci + n) * g.Co + co);
    }
    struct Fix { int nCo; };
    struct Str { size_t o; bool ok; };
    __device__ Fix fix(int n) const { return Fix{n * g.Co}; }
    __device__ Str stream(int k) const {
        int co, t2, a0, a1;
        gg_divmod(k, g.dCo, t2, co); gg_divmod(t2, dKW2, a1, a0);
        const S2Class c = cls();
        const int kw = 2 * a0 + c.ow0, kh = 2 * a1 + c.oh0;
        return Str{(size_t)(kh * g.KW + kw) * g.Ci * g.Co + co, kh < g.KH && kw < g.KW};
    }
    __device__ gg_f32x4 get(const Fix &a, const Str &s, bool ok) const {
        ok = ok && s.ok;
        const gg_f32x4 v = *(const gg_f32x4 *)(V + (ok ? s.o + a.nCo : (size_t)0));
        return ok ? v : (gg_f32x4){0.f, 0.f, 0.f, 0.f};
    }
};
struct StoreS2 {      // class-local pixel m = (b, i, j) -> dx[b, 2i+ph, 2j+pw, n]
    float *out; int ld, H, W, H2, W2, ph, pw; gg_div dH2, dW2;
    int nsp;   // > 0: launched from the stacked GEMM itself (class = z / nsp); 0: the class of the members (the reduce kernel
               // sets them per grid row with `of_class`)
    __device__ void operator()(int m, int n, float v, int z) const {
        int j, t, i, b;
        gg_divmod(m, dW2, t, j); gg_divmod(t, dH2, b, i);
        const int c = nsp > 0 ? z / nsp : 2 * ph + pw;
        out[(((size_t)b * H + 2 * i + (c >> 1)) * W + 2 * j + (c & 1)) * ld + n] = v;
    }
    __device__ StoreS2 of_class(int c) const { StoreS2 s = *this; s.nsp = 0; s.ph = c >> 1; s.pw = c & 1; return s; }
};
struct StorePlain {
    float *out; int ld;
    __device__ void operator()(int m, int n, float v, int) const { out[(size_t)m * ld + n] = v; }
};
struct Im2colXT {  // A(m = (kh,kw,ci), k = (b,oh,ow)) : the same gather with the roles swapped.
    // Row m == rows is all ones (rows rows+1 .. rows+3 zero), so that slab row `rows` of the weight
    // gradient is sum_k dy_pre[k][co] = the bias gradient: no separate pass over dy.
    Im2colX f; int rows;
    __device__ float operator()(int m, int k) const { return m < rows ? f(k, m) : (m == rows ? 1.f : 0.f); }
    __device__ gg_f32x4 v4(int m, int k) const {   // m .. m+3 = 4 consecutive ci
        if (m >= rows) return (gg_f32x4){m == rows ? 1.f : 0.f, 0.f, 0.f, 0.f};
        return f.v4(k, m);
    }
    // roles swapped: the tap index (kh,kw,ci) is the fixed one here, the pixel streams
    struct Fix { Im2colX::Str s; int m; };
    typedef Im2colX::Fix Str;
    __device__ Fix fix(int m) const { return Fix{f.stream(m), m}; }
    __device__ Str stream(int k) const { return f.fix(k); }
    __device__ gg_f32x4 get(const Fix &a, const Str &s, bool ok) const {
        const gg_f32x4 v = f.get(s, a.s, ok && a.m < rows);
        return (ok && a.m == rows) ? (gg_f32x4){1.f, 0.f, 0.f, 0.f} : v;
    }
};
struct DyPre {     // B(k = (b,oh,ow), n = co) = dy * act'(y)
    const float *dy, *y; int Co, act;
    __device__ float operator()(int k, int n) const {
        const size_t o = (size_t)k * Co + n;
        return dy[o] * act_slope(y[o], act);
    }
    __device__ gg_f32x4 v4(int k, int n) const {   // n .. n+3
        const size_t o = (size_t)k * Co + n;
        const gg_f32x4 d = *(const gg_f32x4 *)(dy + o), yy = *(const gg_f32x4 *)(y + o);
        gg_f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = d[e] * act_slope(yy[e], act);
        return r;
    }
    struct Fix { int n; };
    struct Str { int k; };
    __device__ Fix fix(int n) const { return Fix{n}; }
    __device__ Str stream(int k) const { return Str{k}; }
    __device__ gg_f32x4 get(const Fix &a, const Str &s, bool ok) const {
        const size_t o = ok ? (size_t)s.k * Co + a.n : (size_t)0;
        const gg_f32x4 d = *(const gg_f32x4 *)(dy + o), yy = *(const gg_f32x4 *)(y + o);
        const ActSlope sl = act_slope_consts(act);
        gg_f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = ok ? d[e] * sl(yy[e]) : 0.f;
        return r;
    }
};
struct StoreSlab {
    float *slab; size_t stride; int ld;
    __device__ void operator()(int m, int n, float v, int z) const { slab[z * stride + (size_t)m * ld + n] = v; }
};

// split-K partial sums of an input-gradient GEMM -> the Store functor of the unsplit launch (4 columns per thread)
template <class Store>
__global__ __launch_bounds__(256) void slab_reduce_store_kernel(const float *slab, int splits, size_t stride, int M,
                                                                int N, Store st0) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int n4 = N / 4;
    if (i >= (size_t)M * n4) return;
    const int m = (int)(i / n4), n = (int)(i - (size_t)m * n4) * 4;
    // grid row = product of a stacked launch (the parity classes of a stride-2 input gradient): its slabs follow the
    // previous product's
    const Store st = st0.of_class((int)blockIdx.y);
    const float *p = slab + (size_t)blockIdx.y * splits * stride + (size_t)m * N + n;
    gg_f32x4 acc = *(const gg_f32x4 *)p;
    for (int z = 1; z < splits; ++z) acc += *(const gg_f32x4 *)(p + (size_t)z * stride);
#pragma unroll
    for (int e = 0; e < 4; ++e) st(m, n + e, acc[e], 0);
}
// The stride-2 input gradient of the small feature maps (4x4 ... 16x16 images, 256 ... 64 channels) has a long
// contraction (taps/4 x Co) and few output tiles: unsplit, a few dozen workgroups walk hundreds of K steps (measured:
// 0.47 ms for 1.7 GF at the 4x4 stage of config 5).  Splits over K for ~768 workgroups IN THE LAUNCH (its four parity
// classes together), >= 128 k per split.
static int dx_s2_splits(const ConvGeom &g) {
    if (g.Ci % 4 != 0 || g.Co % 4 != 0) return 1;
    const long long M = (long long)g.B * (g.H / 2) * (g.W / 2), N = g.Ci;
    const int K2 = ((g.KH + 1) / 2) * ((g.KW + 1) / 2) * g.Co;
    const long long tiles = ((M + 63) / 64) * ((N + 63) / 64);
    long long s = (768 + 4 * tiles - 1) / (4 * tiles);
    if (s > K2 / 128) s = K2 / 128;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    return (int)s;
}

// ---- finalize: dV = s*dW - (s/n^2)(dW.V) V + reg*V ; dg = (dW.V)/n ; db = slab row `rows` ----------
// one block per output channel (measured: blocking 16 / 64 channels per block for coalescing leaves
// too few blocks on the narrow layers and is 10x slower)
// Split-K slabs -> few slabs, coalesced.  The per-channel kernels below walk a filter column (stride Co floats),
// which is fine for one or a few slabs but not for the ~10^2 slabs of a small-filter layer (tens of MB).  This pass
// sums slab groups element-wise -- every load a full line, up to 16 independent loads in flight per thread, slabs added
// in index order -- into the first slab of each group; the per-channel kernel then sees `groups` slabs `gsz * stride`
// apart.  n = floats of a slab that carry data.
template <int VEC>
__global__ __launch_bounds__(256) void slab_sum_kernel(float *slab, int splits, size_t stride, size_t n, int gsz) {
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (e >= n) return;
    const int z0 = blockIdx.y * gsz, z1 = z0 + gsz < splits ? z0 + gsz : splits;
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    float *base = slab + e;
    vec_t acc = *(const vec_t *)(base + (size_t)z0 * stride);
    for (int z = z0 + 1; z < z1; z += 16) {
        vec_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (z + u < z1) v[u] = *(const vec_t *)(base + (size_t)(z + u) * stride);
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (z + u < z1) acc += v[u];
    }
    *(vec_t *)(base + (size_t)z0 * stride) = acc;
}

// returns the number of slab groups left (their first slabs are gsz * stride apart: *gstride)
static int slab_presum(float *slab, int splits, size_t stride, size_t n, size_t *gstride, hipStream_t st) {
    *gstride = stride;
    if (splits <= 4) return splits;
    const bool v4 = stride % 4 == 0 && n % 4 == 0 && ((uintptr_t)slab & 15) == 0;
    const size_t threads = v4 ? n / 4 : n;
    // enough groups for ~16 k threads, at least 8 slabs per group
    int groups = (int)((16384 + threads - 1) / threads);
    if (groups > splits / 8) groups = splits / 8;
    if (groups > 16) groups = 16;
    if (groups < 1) groups = 1;
    const int gsz = (splits + groups - 1) / groups;
    groups = (splits + gsz - 1) / gsz;
    const dim3 grid((unsigned)((threads + 255) / 256), groups);
    if (v4)
        hipLaunchKernelGGL(slab_sum_kernel<4>, grid, dim3(256), 0, st, slab, splits, stride, n, gsz);
    else
        hipLaunchKernelGGL(slab_sum_kernel<1>, grid, dim3(256), 0, st, slab, splits, stride, n, gsz);
    *gstride = (size_t)gsz * stride;
    return groups;
}

__global__ __launch_bounds__(256) void conv_wfinal_kernel(const float *slab, int splits, size_t stride,
                                                          const float *V, const float *scale,
                                                          const float *n2, int rows, int Co, float reg,
                                                          float *dV, float *dg, float *db) {
    const int co = blockIdx.x;
    float acc = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const size_t o = (size_t)r * Co + co;
        float dw = 0.f;
        for (int z = 0; z < splits; ++z) dw += slab[z * stride + o];
        dV[o] = dw;   // parked; rewritten below
        acc = fmaf(dw, V[o], acc);
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float c = red[0], s = scale[co], nn = n2[co];
    if (threadIdx.x == 0) {
        if (dg) dg[co] = c * rsqrtf(nn);
        if (db) {
            float t = 0.f;
            for (int z = 0; z < splits; ++z) t += slab[z * stride + (size_t)rows * Co + co];
            db[co] = t;
        }
    }
    for (int r = threadIdx.x; r < rows; r += 256) {
        const size_t o = (size_t)r * Co + co;
        dV[o] = s * dV[o] - (s / nn) * c * V[o] + reg * V[o];
    }
}

__global__ __launch_bounds__(256) void conv_bgrad_kernel(const float *dy, const float *y, int rows, int Co,
                                                         int act, float *db) {
    const int co = blockIdx.x;
    float acc = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const size_t o = (size_t)r * Co + co;
        acc += dy[o] * act_slope(y[o], act);
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) db[co] = red[0];
}

// ---- coalesced forms for the large filters ---------------------------------------------------------------------
// The per-channel kernels above walk a filter column (stride Co floats: one 4-byte element per 64-byte sector); fine
// for a 288 x 32 filter, 43 us per call for the generator's 4608 x 1024 one, three calls per layer and step.  These
// read rows: a block owns 64 channels x a chunk of rows (4 row lanes x 64 consecutive channels per pass), per-chunk
// partial sums go through `part` [chunks][Co] and a per-channel pass finishes.  Fixed summation order.
#define CFL_COL_CHUNKS 16
__global__ __launch_bounds__(256) void conv_sumsq_partial_kernel(const float *V, int rows, int Co, int rpc, float *part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rpc, r1 = min(rows, r0 + rpc);
    float acc = 0.f;
    if (c < Co)
        for (int r = r0 + rl; r < r1; r += 4) { const float v = V[(size_t)r * Co + c]; acc = fmaf(v, v, acc); }
    red[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && c < Co) part[(size_t)blockIdx.y * Co + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void conv_scale_final_kernel(const float *part, int chunks, const float *g, int Co,
                                                               float *scale, float *n2out) {
    const int co = blockIdx.x * 256 + threadIdx.x;
    if (co >= Co) return;
    float t = 0.f;
    for (int z = 0; z < chunks; ++z) t += part[(size_t)z * Co + co];
    const float n2 = fmaxf(t, 1e-12f);
    n2out[co] = n2;
    scale[co] = (g ? g[co] : 1.f) * rsqrtf(n2);
}
// dV := sum of the slab groups (parked), part[chunk][co] = sum_r dW V over the chunk's rows
__global__ __launch_bounds__(256) void conv_wdot_partial_kernel(const float *slab, int splits, size_t stride,
                                                                const float *V, int rows, int Co, int rpc, float *dV,
                                                                float *part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rpc, r1 = min(rows, r0 + rpc);
    float acc = 0.f;
    if (c < Co)
        for (int r = r0 + rl; r < r1; r += 4) {
            const size_t o = (size_t)r * Co + c;
            float dw = 0.f;
            for (int z = 0; z < splits; ++z) dw += slab[z * stride + o];
            dV[o] = dw;
            acc = fmaf(dw, V[o], acc);
        }
    red[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && c < Co) part[(size_t)blockIdx.y * Co + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// c = sum of the partials: dg, db (slab row `rows`), and the factor of V in dV, left in part[0][co]
__global__ __launch_bounds__(256) void conv_wdot_final_kernel(float *part, int chunks, const float *slab, int splits,
                                                              size_t stride, const float *scale, const float *n2,
                                                              int rows, int Co, float reg, float *dg, float *db) {
    const int co = blockIdx.x * 256 + threadIdx.x;
    if (co >= Co) return;
    float c = 0.f;
    for (int z = 0; z < chunks; ++z) c += part[(size_t)z * Co + co];
    const float s = scale[co], nn = n2[co];
    if (dg) dg[co] = c * rsqrtf(nn);
    if (db) {
        float t = 0.f;
        for (int z = 0; z < splits; ++z) t += slab[z * stride + (size_t)rows * Co + co];
        db[co] = t;
    }
    part[co] = reg - (s / nn) * c;
}
// dV = s dW + (reg - (s/n^2) c) V, 4 elements per thread (Co % 4 == 0)
__global__ __launch_bounds__(256) void conv_wapply_kernel(const float *V, const float *scale, const float *vfac,
                                                          size_t n4, int Co, float *dV) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int co = (int)((4 * i) % (size_t)Co);
    const gg_f32x4 dw = *(const gg_f32x4 *)(dV + 4 * i), v = *(const gg_f32x4 *)(V + 4 * i);
    const gg_f32x4 s = *(const gg_f32x4 *)(scale + co), f = *(const gg_f32x4 *)(vfac + co);
    gg_f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = s[e] * dw[e] + f[e] * v[e];
    *(gg_f32x4 *)(dV + 4 * i) = r;
}
// (measured with rocprofv3 on the config-5 step: below ~10^6 elements the one-launch per-channel kernels win)
static bool conv_big_filter(int rows, int Co) { return (long long)rows * Co >= (1 << 20) && Co % 4 == 0; }
static int conv_col_rpc(int rows) { return (rows + CFL_COL_CHUNKS - 1) / CFL_COL_CHUNKS; }
static void conv_scale(const float *V, const float *gain, int rows, int Co, float *scale, float *n2, float *part,
                       hipStream_t st) {
    if (!conv_big_filter(rows, Co)) {
        hipLaunchKernelGGL(conv_scale_kernel, dim3(Co), dim3(256), 0, st, V, gain, rows, Co, scale, n2);
        return;
    }
    const int rpc = conv_col_rpc(rows), chunks = (rows + rpc - 1) / rpc;
    hipLaunchKernelGGL(conv_sumsq_partial_kernel, dim3((Co + 63) / 64, chunks), dim3(256), 0, st, V, rows, Co, rpc, part);
    hipLaunchKernelGGL(conv_scale_final_kernel, dim3((Co + 255) / 256), dim3(256), 0, st, part, chunks, gain, Co, scale, n2);
}

// ---- C ABI --------------------------------------------------------------------------------
// split-K of the weight-gradient GEMM (M = taps*Ci + 4, N = Co, K = pixels): enough splits for ~3072
// workgroups given the output tile the GEMM will pick for this N, at least 512 pixels per split, <= 256 slabs
static int wgrad_split_count(long long K, long long M, int N) {
    const int tm = N <= 16 ? 256 : (N <= 32 ? 128 : 64), tn = 64 * 64 / tm;
    const long long tiles = ((M + tm - 1) / tm) * ((N + tn - 1) / tn);
    long long s = (3072 + tiles - 1) / tiles;
    if (s > K / 512) s = K / 512;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    return (int)s;
}
static int wgrad_klen(const ConvGeom &g) {
    const long long K = (long long)g.B * g.OH * g.OW;
    return gg_klen(K, wgrad_split_count(K, (long long)g.KH * g.KW * g.Ci + 4, g.Co));
}
// 3x3 stride 1 with 32-channel multiples on both sides: the direct halo-tile kernel (conv_halo_wgrad.h) and ITS split count
static HaloWPlan halo_wgrad_plan_of(const ConvGeom &g) {
    if (!(g.KH == 3 && g.KW == 3 && g.S == 1)) return HaloWPlan{};
    return halo_wgrad_plan(g.B, g.H, g.W, g.Ci, g.Co);
}
// the 3-channel image stem (4x4 stride 2): direct exact-fp32 kernels for all three products (conv_stem.h)
static bool stem_shape(const ConvGeom &g) { return stem_shape_ok(g.H, g.W, g.Ci, g.Co, g.KH, g.KW, g.S); }
static int wgrad_splits(const ConvGeom &g) {
    const HaloWPlan hw = halo_wgrad_plan_of(g);
    if (hw.ok) return hw.splits;
    if (stem_shape(g)) return stem_dw_splits(g.B, g.OH, g.OW);
    return gg_splits((long long)g.B * g.OH * g.OW, wgrad_klen(g));
}

// workspace = [scale Co | n2 Co | pad] + one scratch region shared by the products of a call (they run one after the
// other on the stream): the split-K slabs of the weight gradient, or the prepared filter planes + split slabs of
// the halo kernel (conv_halo.h) for the forward pass / the input gradient
static size_t conv_ws_header_floats(const ConvGeom &g) { return ((2 + CFL_COL_CHUNKS) * (size_t)g.Co + 64 + 3) / 4 * 4; }
static bool halo_shape(const ConvGeom &g) { return g.KH == 3 && g.KW == 3 && g.S == 1; }
static HaloPlan halo_fwd_plan(const ConvGeom &g) {
    return halo_shape(g) ? halo_plan(g.B, g.H, g.W, g.Ci, g.Co) : HaloPlan{};
}
static HaloPlan halo_dx_plan(const ConvGeom &g) {
    return halo_shape(g) ? halo_plan(g.B, g.H, g.W, g.Co, g.Ci) : HaloPlan{};
}
// Plans for PREPARING a layer's cached filter planes: taken at B = 1, i.e. from the image size and the channel counts only.
// A plan's `ok` also depends on the batch (32-bit element offsets), and the forward of a step runs at another batch than the
// backward slices that later read the planes on several streams at once: whether the planes get built must not depend on
// the batch of the call that happens to build them (the planes' layout never did: nchunks / Npad are channel counts).
static HaloPlan halo_fwd_prep_plan(const ConvGeom &g) {
    return halo_shape(g) ? halo_plan(1, g.H, g.W, g.Ci, g.Co) : HaloPlan{};
}
static HaloPlan halo_dx_prep_plan(const ConvGeom &g) {
    return halo_shape(g) ? halo_plan(1, g.H, g.W, g.Co, g.Ci) : HaloPlan{};
}

extern "C" size_t cfl_conv_workspace_bytes(const CflConv *c) {
    ConvGeom g;
    if (make_geom(c, &g)) return 0;
    const size_t rows = (size_t)g.KH * g.KW * g.Ci;
    size_t region = (size_t)wgrad_splits(g) * (rows + 4) * g.Co * sizeof(float);
    const size_t hf = halo_scratch_bytes(halo_fwd_plan(g)), hd = halo_scratch_bytes(halo_dx_plan(g));
    if (hf > region) region = hf;
    if (hd > region) region = hd;
    if (g.S == 2 && g.H % 2 == 0 && g.W % 2 == 0) {
        const int sp = dx_s2_splits(g);
        const size_t s2 = sp > 1 ? (size_t)4 * sp * g.B * (g.H / 2) * (g.W / 2) * g.Ci * sizeof(float) : 0;   // four stacked classes
        if (s2 > region) region = s2;
    }
    return conv_ws_header_floats(g) * sizeof(float) + region;
}

// 1x1 layer on a 1x1 image with one or two outputs (the discriminator's real/fake logit): a matrix-vector product.
// One wave per row; the generic GEMM would put all of K = 2048 on the two workgroups that cover 500 rows.
__global__ __launch_bounds__(256) void fc_narrow_fwd_kernel(const float *x, const float *V, const float *scale,
                                                            const float *bias, int M, int K, int N, int act, float *y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float *xr = x + (size_t)row * K;
    for (int n = 0; n < N; ++n) {
        float acc = 0.f;
        for (int k = lane; k < K; k += 64) acc = fmaf(xr[k], V[(size_t)k * N + n], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) y[(size_t)row * N + n] = act_apply(acc * scale[n] + (bias ? bias[n] : 0.f), act);
    }
}

extern "C" int cfl_conv_uses_direct_kernel(const CflConv *c, int product) {
    ConvGeom g;
    if (make_geom(c, &g)) return -1;
    if (product == 3) return stem_shape(g) ? 1 : 0;   // the image stem's own kernels (all three products)
    if (product == 2) return halo_wgrad_plan_of(g).ok ? 1 : 0;
    return (product == 0 ? halo_fwd_plan(g) : halo_dx_plan(g)).ok ? 1 : 0;
}

// ---- per-layer cache of what depends on the weights only ----------------------------------------------------
// [scale Co | n2 Co | column-chunk scratch | pad] [halo planes, forward] [halo planes, input gradient].  The caller owns
// the buffer and the validity bits (CFL_CONV_CACHE_*): it clears them whenever V or the gains change.
// Cache layout: [header: scale, n2, scratch | forward planes | input-gradient planes].  The plane regions are sized and
// placed from the channel counts alone (halo_planes_bytes), never from the batch / image size of the call at hand: a
// layer is called with several batch sizes between two updates of its weights (round 4: the offsets used to follow
// HaloPlan::ok of the call, which a different batch size could flip while the validity bits stayed set).
static size_t conv_cache_planes_off(const ConvGeom &g) { return conv_ws_header_floats(g) * sizeof(float); }
static size_t conv_cache_fwd_bytes(const ConvGeom &g) { return halo_shape(g) ? halo_planes_bytes(g.Ci, g.Co) : 0; }
static size_t conv_cache_dx_bytes(const ConvGeom &g) { return halo_shape(g) ? halo_planes_bytes(g.Co, g.Ci) : 0; }
extern "C" size_t cfl_conv_cache_bytes(const CflConv *c) {
    ConvGeom g;
    if (make_geom(c, &g)) return 0;
    return conv_cache_planes_off(g) + conv_cache_fwd_bytes(g) + conv_cache_dx_bytes(g);
}
// scale / n2 / scratch pointers of a call: in the cache (computed once per weight version) or in the workspace header
static void conv_scale_of(const ConvGeom &g, const float *V, const float *gain, void *workspace, void *cache,
                          int32_t *flags, float **scale, float **n2, hipStream_t st) {
    const int rows = g.KH * g.KW * g.Ci;
    *scale = cache ? (float *)cache : (float *)workspace;
    *n2 = *scale + g.Co;
    if (!cache || !(*flags & CFL_CONV_CACHE_SCALE)) {
        conv_scale(V, gain, rows, g.Co, *scale, *n2, *n2 + g.Co, st);
        if (cache) *flags |= CFL_CONV_CACHE_SCALE;
    }
}

extern "C" int cfl_conv2d_wn_fwd_fused(const CflConv *c, const float *x, const float *V, const float *gain,
                                       const float *bias, const float *residual, int32_t subpixel, float *y, void *workspace,
                                       size_t workspace_bytes, void *cache, size_t cache_bytes, int32_t *cache_flags,
                                       cfl_stream_t stream) {
    ConvGeom g;
    int rc = make_geom(c, &g);
    if (rc) return rc;
    if (!x || !V || !y || !workspace) return cfl_set_err(CFL_E_SHAPE, "NULL pointer");
    if (subpixel && (g.Co % 4 != 0)) return cfl_set_err(CFL_E_SHAPE, "sub-pixel store needs Co %% 4 == 0");
    if ((residual || subpixel) && g.KH == 1 && g.KW == 1 && g.H == 1 && g.W == 1 && g.S == 1 && g.Co <= 2)
        return cfl_set_err(CFL_E_UNSUPPORTED, "residual / sub-pixel epilogue on the narrow fully connected layer");
    if (workspace_bytes < cfl_conv_workspace_bytes(c)) return cfl_set_err(CFL_E_WORKSPACE, "conv workspace too small");
    if (cache && (!cache_flags || cache_bytes < cfl_conv_cache_bytes(c) || ((uintptr_t)cache & 15)))
        return cfl_set_err(CFL_E_WORKSPACE, "conv cache too small / misaligned / without flags");
    hipStream_t st = (hipStream_t)stream;
    float *scale, *n2;
    conv_scale_of(g, V, gain, workspace, cache, cache_flags, &scale, &n2, st);
    const int rows = g.KH * g.KW * g.Ci;
    const bool vec = (g.Ci % 4 == 0) && (g.Co % 4 == 0);   // 16-byte gathers along the channel dimension
    const HaloPlan hp = halo_fwd_plan(g);
    if (g.KH == 1 && g.KW == 1 && g.H == 1 && g.W == 1 && g.S == 1 && g.Co <= 2)
        hipLaunchKernelGGL(fc_narrow_fwd_kernel, dim3((g.B + 3) / 4), dim3(256), 0, st, x, V, scale, bias, g.B, g.Ci,
                           g.Co, g.act, y);
    else if (hp.ok) {   // 3x3 stride 1: direct halo-tile kernel (weight-norm scale folded into the prepared filters)
        unsigned short *planes = cache ? (unsigned short *)((char *)cache + conv_cache_planes_off(g)) : nullptr;
        const bool prep = !cache || !(*cache_flags & CFL_CONV_CACHE_PLANES_FWD);
        halo_conv(hp, g.B, g.H, g.W, g.Ci, g.Co, x, nullptr, 0, V, scale, g.Ci, g.Co, 0, bias, g.act, y,
                  (float *)workspace + conv_ws_header_floats(g), st, planes, prep, residual, subpixel ? 1 : 0);
        if (cache) *cache_flags |= CFL_CONV_CACHE_PLANES_FWD;
    }
    else if (stem_shape(g) && !residual && !subpixel)
        stem_fwd(g.B, g.H, g.W, g.Co, x, V, scale, bias, g.act, y, st);
    else if (vec)
        gemm_gather_modes<GG_VEC_K, GG_VEC_MN>(g.B * g.OH * g.OW, g.Co, rows, gg_klen(rows, 1), Im2colX{x, g},
                                               FilterKN{V, g.Co},
                                               StoreFwd{y, scale, bias, g.Co, g.act, residual, subpixel ? g.OH : 0, subpixel ? g.OW : 0}, st);
    else
        gemm_gather(g.B * g.OH * g.OW, g.Co, rows, gg_klen(rows, 1), Im2colX{x, g}, FilterKN{V, g.Co},
                    StoreFwd{y, scale, bias, g.Co, g.act, residual, subpixel ? g.OH : 0, subpixel ? g.OW : 0}, st);
    // The input-gradient planes of the layer are built here too, on the forward's stream, the first time the layer is
    // touched after an update: the backward passes of a step may run on several streams at once (MrCGAN: three chains
    // share every discriminator layer), and a plane set must not be built on one of them while another reads it.
    if (cache && !(*cache_flags & CFL_CONV_CACHE_PLANES_DX)) {
        const HaloPlan hd = halo_dx_prep_plan(g);
        if (hd.ok) {
            halo_prep_planes(hd, g.Co, g.Ci, V, scale, g.Ci, g.Co, 1,
                             (unsigned short *)((char *)cache + conv_cache_planes_off(g) + conv_cache_fwd_bytes(g)), st);
            *cache_flags |= CFL_CONV_CACHE_PLANES_DX;
        }
    }
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "conv fwd launch failed");
}

// Everything of a layer's cache that is missing, WITHOUT running the layer: the weight-norm scale and the filter planes of
// the forward / input-gradient halo kernels (for the products that take them at this call shape).  Lets a caller rebuild the
// caches of all layers right behind the optimizer step, on a side stream, instead of in front of every layer's first
// convolution of the next step (MrCGAN: ~70 launches of 5 us on the forward chains).
extern "C" int cfl_conv_prepare_cached(const CflConv *c, const float *V, const float *gain, void *cache, size_t cache_bytes,
                                       int32_t *cache_flags, cfl_stream_t stream) {
    ConvGeom g;
    int rc = make_geom(c, &g);
    if (rc) return rc;
    if (!V || !cache || !cache_flags) return cfl_set_err(CFL_E_SHAPE, "NULL pointer");
    if (cache_bytes < cfl_conv_cache_bytes(c) || ((uintptr_t)cache & 15))
        return cfl_set_err(CFL_E_WORKSPACE, "conv cache too small / misaligned");
    hipStream_t st = (hipStream_t)stream;
    float *scale, *n2;
    conv_scale_of(g, V, gain, cache, cache, cache_flags, &scale, &n2, st);
    if (!(*cache_flags & CFL_CONV_CACHE_PLANES_FWD)) {
        const HaloPlan hp = halo_fwd_prep_plan(g);
        if (hp.ok) {
            halo_prep_planes(hp, g.Ci, g.Co, V, scale, g.Ci, g.Co, 0, (unsigned short *)((char *)cache + conv_cache_planes_off(g)), st);
            *cache_flags |= CFL_CONV_CACHE_PLANES_FWD;
        }
    }
    if (!(*cache_flags & CFL_CONV_CACHE_PLANES_DX)) {
        const HaloPlan hd = halo_dx_prep_plan(g);
        if (hd.ok) {
            halo_prep_planes(hd, g.Co, g.Ci, V, scale, g.Ci, g.Co, 1,
                             (unsigned short *)((char *)cache + conv_cache_planes_off(g) + conv_cache_fwd_bytes(g)), st);
            *cache_flags |= CFL_CONV_CACHE_PLANES_DX;
        }
    }
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "conv prepare launch failed");
}

// ... for ALL the layers of a network in a handful of launches (round 6): behind the optimizer step of the MrCGAN post epochs the
// per-layer form above is ~65 launches of ~6 us in a row on the preparation stream -- 0.4 ms that the next step's first layers wait
// for.  Here the weight-norm scales of all layers are ONE launch (one workgroup per output channel of every layer, job table in the
// kernel arguments) and the filter planes of all layers another; the per-element code is that of conv_scale_kernel /
// conv_halo_prep_kernel (bit-identical caches).  The few filters of >= 2^20 elements keep their coalesced two-launch scale.
#define CFL_PREP_MAX_JOBS 48
struct ScaleJob { const float *V, *g; float *scale, *n2; int rows, Co, block0; };
struct ScaleManyArgs { ScaleJob job[CFL_PREP_MAX_JOBS]; int njobs; };
struct PlanesJob { const float *V, *scale; unsigned short *wp; int Ci, Co, dgrad, K, N, Npad, block0, nblocks; };
struct PlanesManyArgs { PlanesJob job[CFL_PREP_MAX_JOBS]; int njobs; };

__global__ __launch_bounds__(256) void conv_scale_many_kernel(ScaleManyArgs a) {
    int j = 0;
    while (j < a.njobs - 1 && (int)blockIdx.x >= a.job[j + 1].block0) ++j;
    const ScaleJob &jb = a.job[j];
    const int co = (int)blockIdx.x - jb.block0, rows = jb.rows, Co = jb.Co;
    const float *V = jb.V;
    float acc = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const float v = V[(size_t)r * Co + co];
        acc = fmaf(v, v, acc);
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float n2 = fmaxf(red[0], 1e-12f);
        jb.n2[co] = n2;
        jb.scale[co] = (jb.g ? jb.g[co] : 1.f) * rsqrtf(n2);
    }
}

__global__ __launch_bounds__(256) void conv_halo_prep_many_kernel(PlanesManyArgs a) {
    int j = 0;
    while (j < a.njobs - 1 && (int)blockIdx.x >= a.job[j + 1].block0) ++j;
    const PlanesJob &jb = a.job[j];
    const int K = jb.K, N = jb.N, Npad = jb.Npad, Ci = jb.Ci, Co = jb.Co, dgrad = jb.dgrad;
    const float *V = jb.V, *scale = jb.scale;
    const int nchunks = K >> 5;
    const long long total = 9ll * nchunks * Npad * 4;
    const long long i = (long long)((int)blockIdx.x - jb.block0) * 256 + threadIdx.x;
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
    unsigned short *d = jb.wp + ((blk * Npad + n) * 32 + oct * 8);
    const size_t plane = (size_t)Npad * 32;
    *(gg_u32x4 *)d = ph;
    *(gg_u32x4 *)(d + plane) = pm;
    *(gg_u32x4 *)(d + 2 * plane) = pl;
}

extern "C" int cfl_conv_prepare_cached_many(int32_t n, const CflConv *convs, const float *const *V, const float *const *gain,
                                            void *const *caches, const size_t *cache_bytes, int32_t *const *cache_flags,
                                            cfl_stream_t stream) {
    if (n <= 0 || !convs || !V || !gain || !caches || !cache_bytes || !cache_flags) return cfl_set_err(CFL_E_SHAPE, "cfl_conv_prepare_cached_many: NULL pointer / no layers");
    hipStream_t st = (hipStream_t)stream;
    ScaleManyArgs sa;
    PlanesManyArgs pa;
    memset(&sa, 0, sizeof(sa));
    memset(&pa, 0, sizeof(pa));
    int sblocks = 0, pblocks = 0;
    // pass 1: the scales (the planes fold them in: every scale launch goes in front of every plane launch on the stream)
    for (int i = 0; i < n; ++i) {
        ConvGeom g;
        int rc = make_geom(&convs[i], &g);
        if (rc) return rc;
        if (!V[i] || !caches[i] || !cache_flags[i]) return cfl_set_err(CFL_E_SHAPE, "cfl_conv_prepare_cached_many: layer %d has a NULL pointer", i);
        if (cache_bytes[i] < cfl_conv_cache_bytes(&convs[i]) || ((uintptr_t)caches[i] & 15))
            return cfl_set_err(CFL_E_WORKSPACE, "conv cache of layer %d too small / misaligned", i);
        if (*cache_flags[i] & CFL_CONV_CACHE_SCALE) continue;
        const int rows = g.KH * g.KW * g.Ci;
        float *scale = (float *)caches[i], *n2 = scale + g.Co;
        if (conv_big_filter(rows, g.Co)) {
            conv_scale(V[i], gain[i], rows, g.Co, scale, n2, n2 + g.Co, st);
        } else {
            if (sa.njobs == CFL_PREP_MAX_JOBS) {
                hipLaunchKernelGGL(conv_scale_many_kernel, dim3(sblocks), dim3(256), 0, st, sa);
                sa.njobs = 0; sblocks = 0;
            }
            ScaleJob &sj = sa.job[sa.njobs++];
            sj.V = V[i]; sj.g = gain[i]; sj.scale = scale; sj.n2 = n2; sj.rows = rows; sj.Co = g.Co; sj.block0 = sblocks;
            sblocks += g.Co;
        }
        *cache_flags[i] |= CFL_CONV_CACHE_SCALE;
    }
    if (sa.njobs) hipLaunchKernelGGL(conv_scale_many_kernel, dim3(sblocks), dim3(256), 0, st, sa);
    // pass 2: the filter planes of the halo kernels (forward, input gradient)
    for (int i = 0; i < n; ++i) {
        ConvGeom g;
        (void)make_geom(&convs[i], &g);
        const float *scale = (const float *)caches[i];
        for (int dgrad = 0; dgrad < 2; ++dgrad) {
            const int bit = dgrad ? CFL_CONV_CACHE_PLANES_DX : CFL_CONV_CACHE_PLANES_FWD;
            if (*cache_flags[i] & bit) continue;
            const HaloPlan hp = dgrad ? halo_dx_prep_plan(g) : halo_fwd_prep_plan(g);
            if (!hp.ok) continue;
            if (pa.njobs == CFL_PREP_MAX_JOBS) {
                hipLaunchKernelGGL(conv_halo_prep_many_kernel, dim3(pblocks), dim3(256), 0, st, pa);
                pa.njobs = 0; pblocks = 0;
            }
            PlanesJob &pj = pa.job[pa.njobs++];
            pj.V = V[i]; pj.scale = scale; pj.Ci = g.Ci; pj.Co = g.Co; pj.dgrad = dgrad;
            pj.K = dgrad ? g.Co : g.Ci; pj.N = dgrad ? g.Ci : g.Co; pj.Npad = hp.Npad;
            pj.wp = (unsigned short *)((char *)caches[i] + conv_cache_planes_off(g) + (dgrad ? conv_cache_fwd_bytes(g) : 0));
            const long long prep = 9ll * hp.nchunks * hp.Npad * 4;
            pj.block0 = pblocks; pj.nblocks = (int)((prep + 255) / 256);
            pblocks += pj.nblocks;
            *cache_flags[i] |= bit;
        }
    }
    if (pa.njobs) hipLaunchKernelGGL(conv_halo_prep_many_kernel, dim3(pblocks), dim3(256), 0, st, pa);
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "conv prepare launch failed");
}

extern "C" int cfl_conv2d_wn_fwd_cached(const CflConv *c, const float *x, const float *V, const float *gain,
                                        const float *bias, float *y, void *workspace, size_t workspace_bytes,
                                        void *cache, size_t cache_bytes, int32_t *cache_flags, cfl_stream_t stream) {
    return cfl_conv2d_wn_fwd_fused(c, x, V, gain, bias, nullptr, 0, y, workspace, workspace_bytes, cache, cache_bytes, cache_flags,
                                   stream);
}

extern "C" int cfl_conv2d_wn_fwd(const CflConv *c, const float *x, const float *V, const float *gain,
                                 const float *bias, float *y, void *workspace, size_t workspace_bytes,
                                 cfl_stream_t stream) {
    return cfl_conv2d_wn_fwd_cached(c, x, V, gain, bias, y, workspace, workspace_bytes, nullptr, 0, nullptr, stream);
}

extern "C" int cfl_conv2d_wn_bwd(const CflConv *c, const float *x, const float *V, const float *gain,
                                 const float *y, const float *dy, float reg_const, float *dx, float *dV,
                                 float *dg, float *db, void *workspace, size_t workspace_bytes,
                                 cfl_stream_t stream) {
    return cfl_conv2d_wn_bwd_cached(c, x, V, gain, y, dy, reg_const, dx, dV, dg, db, workspace, workspace_bytes, nullptr, 0,
                                    nullptr, stream);
}

extern "C" int cfl_conv2d_wn_bwd_cached(const CflConv *c, const float *x, const float *V, const float *gain,
                                        const float *y, const float *dy, float reg_const, float *dx, float *dV,
                                        float *dg, float *db, void *workspace, size_t workspace_bytes, void *cache,
                                        size_t cache_bytes, int32_t *cache_flags, cfl_stream_t stream) {
    return cfl_conv2d_wn_bwd_fused(c, x, V, gain, y, dy, 0, reg_const, dx, dV, dg, db, workspace, workspace_bytes, cache,
                                   cache_bytes, cache_flags, stream);
}

// 1 when the backward of this shape can take dy (and y) 2x sub-pixel shuffled: both products on the halo-tile kernels and
// whole 32-channel chunks per quarter
extern "C" int cfl_conv_bwd_takes_subpixel(const CflConv *c) {
    ConvGeom g;
    if (make_geom(c, &g)) return -1;
    return (g.Co % 128 == 0 && halo_dx_plan(g).ok && halo_wgrad_plan_of(g).ok) ? 1 : 0;
}

extern "C" int cfl_conv2d_wn_bwd_fused(const CflConv *c, const float *x, const float *V, const float *gain,
                                       const float *y, const float *dy, int32_t dy_subpixel, float reg_const, float *dx,
                                       float *dV, float *dg, float *db, void *workspace, size_t workspace_bytes, void *cache,
                                       size_t cache_bytes, int32_t *cache_flags, cfl_stream_t stream) {
    ConvGeom g;
    int rc = make_geom(c, &g);
    if (rc) return rc;
    if (dy_subpixel) {
        if (cfl_conv_bwd_takes_subpixel(c) != 1 || (db && !dV))
            return cfl_set_err(CFL_E_SHAPE, "conv bwd: a shuffled dy needs the halo-tile kernels for both products and Co %% 128 == 0 "
                                            "(cfl_conv_bwd_takes_subpixel); un-shuffle with cfl_subpixel2x_bwd instead");
    }
    const int dy_cq = dy_subpixel ? g.Co / 4 : 0;
    if (!V || !dy || !workspace || (!x && dV) || (!dV && !dx)) return cfl_set_err(CFL_E_SHAPE, "NULL pointer");
    if (!y) {
        if (g.act != 0) return cfl_set_err(CFL_E_SHAPE, "conv bwd: y is required when act != 0");
        y = dy;  // never dereferenced for act == 0 slopes, but keeps the functors simple
    }
    if (workspace_bytes < cfl_conv_workspace_bytes(c)) return cfl_set_err(CFL_E_WORKSPACE, "conv workspace too small");
    if (cache && (!cache_flags || cache_bytes < cfl_conv_cache_bytes(c) || ((uintptr_t)cache & 15)))
        return cfl_set_err(CFL_E_WORKSPACE, "conv cache too small / misaligned / without flags");
    hipStream_t st = (hipStream_t)stream;
    float *scale, *n2;
    conv_scale_of(g, V, gain, workspace, cache, cache_flags, &scale, &n2, st);
    float *slab = (float *)workspace + conv_ws_header_floats(g);
    const int rows = g.KH * g.KW * g.Ci;
    const int npix = g.B * g.OH * g.OW;
    const bool vec = (g.Ci % 4 == 0) && (g.Co % 4 == 0);
    if (dx) {
        const HaloPlan hp = halo_dx_plan(g);
        if (hp.ok) {
            // 3x3 stride 1: the input gradient is the same direct convolution over dy * act'(y) with the flipped,
            // scale-weighted filter (conv_halo.h)
            unsigned short *planes = cache ? (unsigned short *)((char *)cache + conv_cache_planes_off(g) + conv_cache_fwd_bytes(g))
                                           : nullptr;
            const bool prep = !cache || !(*cache_flags & CFL_CONV_CACHE_PLANES_DX);
            halo_conv(hp, g.B, g.H, g.W, g.Co, g.Ci, dy, y, g.act, V, scale, g.Ci, g.Co, 1, nullptr, 0, dx, slab, st, planes, prep,
                      nullptr, 0, dy_cq);
            if (cache) *cache_flags |= CFL_CONV_CACHE_PLANES_DX;
        } else if (stem_shape(g)) {
            stem_dx(g.B, g.H, g.W, g.Co, dy, y, g.act, V, scale, dx, st);
        } else if (g.S == 2 && g.H % 2 == 0 && g.W % 2 == 0) {
            // four dense sub-problems, one per parity class of the input pixel
            const int KH2 = (g.KH + 1) / 2, KW2 = (g.KW + 1) / 2, H2 = g.H / 2, W2 = g.W / 2;
            const int K2 = KH2 * KW2 * g.Co;
            const gg_div dKW2 = gg_make_div(KW2), dH2 = gg_make_div(H2), dW2 = gg_make_div(W2);
            if (g.Co % 4 == 0) {
                // ... stacked in grid z of ONE launch (round 4: four launches + four reduce launches per call before; the
                // functors take the class from blockIdx.z): 4x the workgroups per launch for products that are far too small
                // to fill the chip one at a time
                const int sp = dx_s2_splits(g);
                const int M2 = g.B * H2 * W2, klen = gg_klen(K2, sp), nsp = gg_splits(K2, klen);
                DyGatherS2 fa{dy, y, scale, g, 0, 0, 0, 0, KH2, KW2, H2, W2, dKW2, dH2, dW2, nsp};
                FilterTS2 fb{V, g, 0, 0, KW2, dKW2, nsp};
                StoreS2 fs{dx, g.Ci, g.H, g.W, H2, W2, 0, 0, dH2, dW2, nsp};
                if (nsp > 1) {
                    const size_t sstride = (size_t)M2 * g.Ci;
                    gemm_gather_modes<GG_VEC_K, GG_VEC_K>(M2, g.Ci, K2, klen, fa, fb, StoreSlab{slab, sstride, g.Ci}, st, 4);
                    const size_t items = (size_t)M2 * (g.Ci / 4);
                    hipLaunchKernelGGL(slab_reduce_store_kernel<StoreS2>, dim3((unsigned)((items + 255) / 256), 4),
                                       dim3(256), 0, st, slab, nsp, sstride, M2, g.Ci, fs);
                } else
                    gemm_gather_modes<GG_VEC_K, GG_VEC_K>(M2, g.Ci, K2, klen, fa, fb, fs, st, 4);
            } else {
                for (int ph = 0; ph < 2; ++ph)
                    for (int pw = 0; pw < 2; ++pw) {
                        const int oh0 = (ph + g.pt) & 1, ow0 = (pw + g.pl) & 1;
                        DyGatherS2 fa{dy, y, scale, g, ph, pw, oh0, ow0, KH2, KW2, H2, W2, dKW2, dH2, dW2, 0};
                        FilterTS2 fb{V, g, oh0, ow0, KW2, dKW2, 0};
                        StoreS2 fs{dx, g.Ci, g.H, g.W, H2, W2, ph, pw, dH2, dW2, 0};
                        gemm_gather(g.B * H2 * W2, g.Ci, K2, gg_klen(K2, 1), fa, fb, fs, st);
                    }
            }
        } else if (g.Co % 4 == 0)
            gemm_gather_modes<GG_VEC_K, GG_VEC_K>(g.B * g.H * g.W, g.Ci, g.KH * g.KW * g.Co,
                                                  gg_klen(g.KH * g.KW * g.Co, 1), DyGather{dy, y, scale, g},
                                                  FilterT{V, g}, StorePlain{dx, g.Ci}, st);
        else
            gemm_gather(g.B * g.H * g.W, g.Ci, g.KH * g.KW * g.Co, gg_klen(g.KH * g.KW * g.Co, 1),
                        DyGather{dy, y, scale, g}, FilterT{V, g}, StorePlain{dx, g.Ci}, st);
    }
    if (dV) {
        // M = rows + 4: the extra row block carries the bias gradient (see Im2colXT)
        const int splits = wgrad_splits(g);
        const size_t sstride = (size_t)(rows + 4) * g.Co;
        const HaloWPlan hw = halo_wgrad_plan_of(g);
        if (hw.ok)
            halo_wgrad(hw, g.B, g.H, g.W, g.Ci, g.Co, x, dy, y, g.act, slab, sstride, st, dy_cq);
        else if (stem_shape(g))
            stem_dw(g.B, g.H, g.W, g.Co, x, dy, y, g.act, slab, sstride, st);
        else if (vec)
            gemm_gather_modes<GG_VEC_MN, GG_VEC_MN>(rows + 4, g.Co, npix, wgrad_klen(g), Im2colXT{Im2colX{x, g}, rows},
                                                    DyPre{dy, y, g.Co, g.act}, StoreSlab{slab, sstride, g.Co}, st);
        else
            gemm_gather(rows + 4, g.Co, npix, wgrad_klen(g), Im2colXT{Im2colX{x, g}, rows}, DyPre{dy, y, g.Co, g.act},
                        StoreSlab{slab, sstride, g.Co}, st);
        size_t gstride;
        const int groups = slab_presum(slab, splits, sstride, (size_t)(rows + 1) * g.Co, &gstride, st);
        if (conv_big_filter(rows, g.Co)) {
            float *part = n2 + g.Co;
            const int rpc = conv_col_rpc(rows), chunks = (rows + rpc - 1) / rpc;
            hipLaunchKernelGGL(conv_wdot_partial_kernel, dim3((g.Co + 63) / 64, chunks), dim3(256), 0, st, slab, groups,
                               gstride, V, rows, g.Co, rpc, dV, part);
            hipLaunchKernelGGL(conv_wdot_final_kernel, dim3((g.Co + 255) / 256), dim3(256), 0, st, part, chunks, slab,
                               groups, gstride, scale, n2, rows, g.Co, reg_const, gain ? dg : nullptr, db);
            const size_t n4 = (size_t)rows * g.Co / 4;
            hipLaunchKernelGGL(conv_wapply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, V, scale, part, n4,
                               g.Co, dV);
        } else
            hipLaunchKernelGGL(conv_wfinal_kernel, dim3(g.Co), dim3(256), 0, st, slab, groups, gstride, V, scale, n2, rows,
                               g.Co, reg_const, dV, gain ? dg : nullptr, db);
    } else if (db) {
        hipLaunchKernelGGL(conv_bgrad_kernel, dim3(g.Co), dim3(256), 0, st, dy, y, npix, g.Co, g.act, db);
    }
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "conv bwd launch failed");
}

// ---- deferred weight-gradient finalisation (round 6) -----------------------------------------------------------------------------
// cfl_conv2d_wn_bwd_* ends every weight gradient with its own small launches: slab sums (slab_sum_kernel) and the per-channel
// weight-norm finalisation (conv_wfinal_kernel) -- ~80 launches of 5-16 us per MrCGAN step, one pair per layer and backward chain.
// The two entry points below split that work: cfl_conv2d_wn_wgrad_slabs runs ONLY the contraction of a layer into slabs the
// caller keeps (one region per layer), and cfl_conv_wfinal_many finishes the layers of a whole backward chain at once -- one
// slab-sum launch and one finalisation launch for all of them (job tables in the kernel arguments; the same per-element code in
// the same order as the per-layer kernels: bit-identical), the few large filters on their coalesced three-launch form as before.
extern "C" size_t cfl_conv_wgrad_slab_bytes(const CflConv *c) {
    ConvGeom g;
    if (make_geom(c, &g)) return 0;
    return (size_t)wgrad_splits(g) * ((size_t)g.KH * g.KW * g.Ci + 4) * g.Co * sizeof(float);
}

extern "C" int cfl_conv2d_wn_wgrad_slabs(const CflConv *c, const float *x, const float *y, const float *dy, int32_t dy_subpixel,
                                         float *slab, size_t slab_bytes, cfl_stream_t stream) {
    ConvGeom g;
    int rc = make_geom(c, &g);
    if (rc) return rc;
    if (dy_subpixel && cfl_conv_bwd_takes_subpixel(c) != 1)
        return cfl_set_err(CFL_E_SHAPE, "conv wgrad: a shuffled dy needs the halo-tile kernels (cfl_conv_bwd_takes_subpixel)");
    if (!x || !dy || !slab) return cfl_set_err(CFL_E_SHAPE, "NULL pointer");
    if (!y) {
        if (g.act != 0) return cfl_set_err(CFL_E_SHAPE, "conv wgrad: y is required when act != 0");
        y = dy;
    }
    if (slab_bytes < cfl_conv_wgrad_slab_bytes(c) || ((uintptr_t)slab & 15)) return cfl_set_err(CFL_E_WORKSPACE, "conv wgrad slabs too small / misaligned");
    hipStream_t st = (hipStream_t)stream;
    const int dy_cq = dy_subpixel ? g.Co / 4 : 0;
    const int rows = g.KH * g.KW * g.Ci;
    const int npix = g.B * g.OH * g.OW;
    const bool vec = (g.Ci % 4 == 0) && (g.Co % 4 == 0);
    const size_t sstride = (size_t)(rows + 4) * g.Co;
    const HaloWPlan hw = halo_wgrad_plan_of(g);
    // (the dispatch of cfl_conv2d_wn_bwd_fused's weight-gradient branch, slabs in the caller's region)
    if (hw.ok)
        halo_wgrad(hw, g.B, g.H, g.W, g.Ci, g.Co, x, dy, y, g.act, slab, sstride, st, dy_cq);
    else if (stem_shape(g))
        stem_dw(g.B, g.H, g.W, g.Co, x, dy, y, g.act, slab, sstride, st);
    else if (vec)
        gemm_gather_modes<GG_VEC_MN, GG_VEC_MN>(rows + 4, g.Co, npix, wgrad_klen(g), Im2colXT{Im2colX{x, g}, rows},
                                                DyPre{dy, y, g.Co, g.act}, StoreSlab{slab, sstride, g.Co}, st);
    else
        gemm_gather(rows + 4, g.Co, npix, wgrad_klen(g), Im2colXT{Im2colX{x, g}, rows}, DyPre{dy, y, g.Co, g.act},
                    StoreSlab{slab, sstride, g.Co}, st);
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "conv wgrad launch failed");
}

#define CFL_WFINAL_MAX_JOBS 32
struct SlabSumJob { float *slab; size_t stride, n; int splits, gsz, bx, block0, nblocks, vec; };
struct SlabSumManyArgs { SlabSumJob job[CFL_WFINAL_MAX_JOBS]; int njobs; };
struct WFinalJob { const float *slab, *V, *scale, *n2; float *dV, *dg, *db; size_t stride; int splits, rows, Co, block0; };
struct WFinalManyArgs { WFinalJob job[CFL_WFINAL_MAX_JOBS]; int njobs; float reg; };

// slab_sum_kernel for many layers: block b belongs to the job whose [block0, block0 + nblocks) holds it; inside the job the grid
// is (bx, groups) linearised x fastest -- the same element, the same slabs in the same order as the per-layer launch
__global__ __launch_bounds__(256) void slab_sum_many_kernel(SlabSumManyArgs a) {
    int j = 0;
    while (j < a.njobs - 1 && (int)blockIdx.x >= a.job[j].block0 + a.job[j].nblocks) ++j;
    const SlabSumJob &jb = a.job[j];
    const int lb = (int)blockIdx.x - jb.block0, bxi = lb % jb.bx, gy = lb / jb.bx;
    const int z0 = gy * jb.gsz, z1 = z0 + jb.gsz < jb.splits ? z0 + jb.gsz : jb.splits;
    if (jb.vec) {
        const size_t e = ((size_t)bxi * 256 + threadIdx.x) * 4;
        if (e >= jb.n) return;
        float *base = jb.slab + e;
        gg_f32x4 acc = *(const gg_f32x4 *)(base + (size_t)z0 * jb.stride);
        for (int z = z0 + 1; z < z1; z += 16) {
            gg_f32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (z + u < z1) v[u] = *(const gg_f32x4 *)(base + (size_t)(z + u) * jb.stride);
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (z + u < z1) acc += v[u];
        }
        *(gg_f32x4 *)(base + (size_t)z0 * jb.stride) = acc;
    } else {
        const size_t e = (size_t)bxi * 256 + threadIdx.x;
        if (e >= jb.n) return;
        float *base = jb.slab + e;
        float acc = base[(size_t)z0 * jb.stride];
        for (int z = z0 + 1; z < z1; z += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (z + u < z1) v[u] = base[(size_t)(z + u) * jb.stride];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (z + u < z1) acc += v[u];
        }
        base[(size_t)z0 * jb.stride] = acc;
    }
}

// conv_wfinal_kernel for many layers: one block per output channel of every job
__global__ __launch_bounds__(256) void conv_wfinal_many_kernel(WFinalManyArgs a) {
    int j = 0;
    while (j < a.njobs - 1 && (int)blockIdx.x >= a.job[j + 1].block0) ++j;
    const WFinalJob &jb = a.job[j];
    const int co = (int)blockIdx.x - jb.block0, rows = jb.rows, Co = jb.Co, splits = jb.splits;
    const float *slab = jb.slab, *V = jb.V;
    float *dV = jb.dV;
    const size_t stride = jb.stride;
    float acc = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const size_t o = (size_t)r * Co + co;
        float dw = 0.f;
        for (int z = 0; z < splits; ++z) dw += slab[z * stride + o];
        dV[o] = dw;   // parked; rewritten below
        acc = fmaf(dw, V[o], acc);
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float c = red[0], s = jb.scale[co], nn = jb.n2[co];
    if (threadIdx.x == 0) {
        if (jb.dg) jb.dg[co] = c * rsqrtf(nn);
        if (jb.db) {
            float t = 0.f;
            for (int z = 0; z < splits; ++z) t += slab[z * stride + (size_t)rows * Co + co];
            jb.db[co] = t;
        }
    }
    for (int r = threadIdx.x; r < rows; r += 256) {
        const size_t o = (size_t)r * Co + co;
        dV[o] = s * dV[o] - (s / nn) * c * V[o] + a.reg * V[o];
    }
}

extern "C" int cfl_conv_wfinal_many(int32_t n, const CflConv *convs, float *const *slabs, const float *const *V,
                                    const float *const *gain, void *const *caches, float reg_const, float *const *dV,
                                    float *const *dg, float *const *db, cfl_stream_t stream) {
    if (n <= 0 || !convs || !slabs || !V || !gain || !caches || !dV || !dg || !db) return cfl_set_err(CFL_E_SHAPE, "cfl_conv_wfinal_many: NULL pointer / no jobs");
    hipStream_t st = (hipStream_t)stream;
    SlabSumManyArgs sa;
    WFinalManyArgs wa;
    memset(&sa, 0, sizeof(sa));
    memset(&wa, 0, sizeof(wa));
    wa.reg = reg_const;
    int sblocks = 0, wblocks = 0;
    auto flush = [&]() {
        if (sa.njobs) hipLaunchKernelGGL(slab_sum_many_kernel, dim3(sblocks), dim3(256), 0, st, sa);
        if (wa.njobs) hipLaunchKernelGGL(conv_wfinal_many_kernel, dim3(wblocks), dim3(256), 0, st, wa);
        sa.njobs = wa.njobs = 0;
        sblocks = wblocks = 0;
    };
    for (int i = 0; i < n; ++i) {
        ConvGeom g;
        int rc = make_geom(&convs[i], &g);
        if (rc) return rc;
        if (!slabs[i] || !V[i] || !caches[i] || !dV[i]) return cfl_set_err(CFL_E_SHAPE, "cfl_conv_wfinal_many: job %d has a NULL pointer", i);
        const int rows = g.KH * g.KW * g.Ci, splits = wgrad_splits(g);
        const size_t sstride = (size_t)(rows + 4) * g.Co;
        float *scale = (float *)caches[i], *n2 = scale + g.Co;       // (the layer cache's header: valid -- CFL_CONV_CACHE_SCALE -- by contract)
        float *slab = slabs[i];
        if (conv_big_filter(rows, g.Co)) {
            // the large filters: coalesced three-launch form, as cfl_conv2d_wn_bwd_fused
            size_t gstride;
            const int groups = slab_presum(slab, splits, sstride, (size_t)(rows + 1) * g.Co, &gstride, st);
            float *part = n2 + g.Co;
            const int rpc = conv_col_rpc(rows), chunks = (rows + rpc - 1) / rpc;
            hipLaunchKernelGGL(conv_wdot_partial_kernel, dim3((g.Co + 63) / 64, chunks), dim3(256), 0, st, slab, groups, gstride, V[i],
                               rows, g.Co, rpc, dV[i], part);
            hipLaunchKernelGGL(conv_wdot_final_kernel, dim3((g.Co + 255) / 256), dim3(256), 0, st, part, chunks, slab, groups, gstride,
                               scale, n2, rows, g.Co, reg_const, gain[i] ? dg[i] : nullptr, db[i]);
            const size_t n4 = (size_t)rows * g.Co / 4;
            hipLaunchKernelGGL(conv_wapply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, V[i], scale, part, n4, g.Co, dV[i]);
            continue;
        }
        if (wa.njobs == CFL_WFINAL_MAX_JOBS || sa.njobs == CFL_WFINAL_MAX_JOBS) flush();
        // slab_presum's grouping, as a job of the batched slab-sum launch
        int groups = splits;
        size_t gstride = sstride;
        if (splits > 4) {
            const size_t nsum = (size_t)(rows + 1) * g.Co;
            const bool v4 = sstride % 4 == 0 && nsum % 4 == 0 && ((uintptr_t)slab & 15) == 0;
            const size_t threads = v4 ? nsum / 4 : nsum;
            groups = (int)((16384 + threads - 1) / threads);
            if (groups > splits / 8) groups = splits / 8;
            if (groups > 16) groups = 16;
            if (groups < 1) groups = 1;
            const int gsz = (splits + groups - 1) / groups;
            groups = (splits + gsz - 1) / gsz;
            SlabSumJob &sj = sa.job[sa.njobs++];
            sj.slab = slab; sj.stride = sstride; sj.n = nsum; sj.splits = splits; sj.gsz = gsz; sj.vec = v4 ? 1 : 0;
            sj.bx = (int)((threads + 255) / 256); sj.block0 = sblocks; sj.nblocks = sj.bx * groups;
            sblocks += sj.nblocks;
            gstride = (size_t)gsz * sstride;
        }
        WFinalJob &wj = wa.job[wa.njobs++];
        wj.slab = slab; wj.V = V[i]; wj.scale = scale; wj.n2 = n2; wj.dV = dV[i]; wj.dg = gain[i] ? dg[i] : nullptr; wj.db = db[i];
        wj.stride = gstride; wj.splits = groups; wj.rows = rows; wj.Co = g.Co; wj.block0 = wblocks;
        wblocks += g.Co;
    }
    flush();
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "conv wfinal launch failed");
}

// =============================================================================================
// Weight-normalised TRANSPOSED convolution (cfl/layers.py:253-361): x [B,H,W,Ci], V [KH,KW,Co,Ci],
// y [B,H*S,W*S,Co] = act( conv2d_transpose(x, g[co] * V / ||V[:,:,co,:]||) + b ), 'SAME'.
// conv2d_transpose is the adjoint of the 'SAME' stride-S convolution F: [B,H*S,W*S,Co] -> [B,H,W,Ci]
// with the same HWIO filter; `gt` below is the geometry of F.
// =============================================================================================
static int make_geom_t(const CflConv *c, ConvGeom *g) {
    if (!c) return cfl_set_err(CFL_E_SHAPE, "conv shape is NULL");
    if (c->B <= 0 || c->H <= 0 || c->W <= 0 || c->Ci <= 0 || c->Co <= 0 || c->KH <= 0 || c->KW <= 0 ||
        c->stride <= 0 || c->act < 0 || c->act > 2)
        return cfl_set_err(CFL_E_SHAPE, "bad transposed conv shape");
    g->B = c->B; g->H = c->H * c->stride; g->W = c->W * c->stride;
    g->Ci = c->Co;   // F's input channels  = the transposed layer's outputs
    g->Co = c->Ci;   // F's output channels = the transposed layer's inputs
    g->KH = c->KH; g->KW = c->KW; g->S = c->stride; g->act = c->act;
    same_pad(g->H, g->KH, g->S, &g->OH, &g->pt);
    same_pad(g->W, g->KW, g->S, &g->OW, &g->pl);
    geom_divs(g);
    return CFL_OK;
}

// per-OUTPUT-channel scale of the transposed layer: norm over (kh, kw, ci) of V[kh,kw,co,ci]
__global__ __launch_bounds__(256) void convt_scale_kernel(const float *V, const float *g, int taps, int Co,
                                                          int Ci, float *scale, float *n2out) {
    const int co = blockIdx.x;
    float acc = 0.f;
    for (int i = threadIdx.x; i < taps * Ci; i += 256) {
        const float v = V[((size_t)(i / Ci) * Co + co) * Ci + (i % Ci)];
        acc = fmaf(v, v, acc);
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float n2 = fmaxf(red[0], 1e-12f);
        n2out[co] = n2;
        scale[co] = (g ? g[co] : 1.f) * rsqrtf(n2);
    }
}

struct TGatherX {  // A(m = (b,oh,ow) of y, k = (kh,kw,ci)) = x[b,(oh+pt-kh)/S,(ow+pl-kw)/S,ci]
    const float *x; ConvGeom g;   // g = geometry of F (g.Co = channels of x)
    __device__ float operator()(int m, int k) const {
        int iw, t, ih, b, ci, t2, kw, kh, oh, ow, rh, rw;
        gg_divmod(m, g.dW, t, iw); gg_divmod(t, g.dH, b, ih);
        gg_divmod(k, g.dCo, t2, ci); gg_divmod(t2, g.dKW, kh, kw);
        const int nh = ih + g.pt - kh, nw = iw + g.pl - kw;
        if (nh < 0 || nw < 0) return 0.f;
        gg_divmod(nh, g.dS, oh, rh); gg_divmod(nw, g.dS, ow, rw);
        if (rh || rw) return 0.f;
        if (oh >= g.OH || ow >= g.OW) return 0.f;
        return x[(((size_t)b * g.OH + oh) * g.OW + ow) * g.Co + ci];
    }
    __device__ gg_f32x4 v4(int m, int k) const {   // k .. k+3 = 4 consecutive ci
        const gg_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        int iw, t, ih, b, ci, t2, kw, kh, oh, ow, rh, rw;
        gg_divmod(m, g.dW, t, iw); gg_divmod(t, g.dH, b, ih);
        gg_divmod(k, g.dCo, t2, ci); gg_divmod(t2, g.dKW, kh, kw);
        const int nh = ih + g.pt - kh, nw = iw + g.pl - kw;
        if (nh < 0 || nw < 0) return zero;
        gg_divmod(nh, g.dS, oh, rh); gg_divmod(nw, g.dS, ow, rw);
        if (rh || rw) return zero;
        if (oh >= g.OH || ow >= g.OW) return zero;
        return *(const gg_f32x4 *)(x + (((size_t)b * g.OH + oh) * g.OW + ow) * g.Co + ci);
    }
};
struct TFilterFwd {  // B(k = (kh,kw,ci), n = co) = V[kh,kw,co,ci]
    const float *V; int Co, Ci; gg_div dCi;
    __device__ float operator()(int k, int n) const {
        int t, ci;
        gg_divmod(k, dCi, t, ci);
        return V[((size_t)t * Co + n) * Ci + ci];
    }
    __device__ gg_f32x4 v4(int k, int n) const {   // k .. k+3 = 4 consecutive ci
        int t, ci;
        gg_divmod(k, dCi, t, ci);
        return *(const gg_f32x4 *)(V + ((size_t)t * Co + n) * Ci + ci);
    }
};
struct TIm2colDy {  // A(m = (b,p,q) of x, k = (kh,kw,co)) = dy_pre[b,p*S+kh-pt,q*S+kw-pl,co] (* scale[co])
    const float *dy, *y, *scale; ConvGeom g;
    __device__ float operator()(int m, int k) const {
        int ow, t, oh, b, co, t2, kw, kh;
        gg_divmod(m, g.dOW, t, ow); gg_divmod(t, g.dOH, b, oh);
        gg_divmod(k, g.dCi, t2, co); gg_divmod(t2, g.dKW, kh, kw);
        const int ih = oh * g.S + kh - g.pt, iw = ow * g.S + kw - g.pl;
        if (ih < 0 || ih >= g.H || iw < 0 || iw >= g.W) return 0.f;
        const size_t o = (((size_t)b * g.H + ih) * g.W + iw) * g.Ci + co;
        return dy[o] * act_slope(y[o], g.act) * (scale ? scale[co] : 1.f);
    }
    __device__ gg_f32x4 v4(int m, int k) const {   // k .. k+3 = 4 consecutive co
        int ow, t, oh, b, co, t2, kw, kh;
        gg_divmod(m, g.dOW, t, ow); gg_divmod(t, g.dOH, b, oh);
        gg_divmod(k, g.dCi, t2, co); gg_divmod(t2, g.dKW, kh, kw);
        const int ih = oh * g.S + kh - g.pt, iw = ow * g.S + kw - g.pl;
        if (ih < 0 || ih >= g.H || iw < 0 || iw >= g.W) return (gg_f32x4){0.f, 0.f, 0.f, 0.f};
        const size_t o = (((size_t)b * g.H + ih) * g.W + iw) * g.Ci + co;
        const gg_f32x4 d = *(const gg_f32x4 *)(dy + o), yy = *(const gg_f32x4 *)(y + o);
        gg_f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = d[e] * act_slope(yy[e], g.act) * (scale ? scale[co + e] : 1.f);
        return r;
    }
};
struct TIm2colDyT {
    TIm2colDy f;
    __device__ float operator()(int m, int k) const { return f(k, m); }
    __device__ gg_f32x4 v4(int m, int k) const { return f.v4(k, m); }   // m .. m+3 = 4 consecutive co
};
struct PlainKN {
    const float *p; int ld;
    __device__ float operator()(int k, int n) const { return p[(size_t)k * ld + n]; }
    __device__ gg_f32x4 v4(int k, int n) const { return *(const gg_f32x4 *)(p + (size_t)k * ld + n); }  // n .. n+3
};

// dV[t,co,ci] = s dW - (s/n^2)(sum_{t,ci} dW V) V + reg V ; dg[co] = (dW . V)/n
__global__ __launch_bounds__(256) void convt_wfinal_kernel(const float *slab, int splits, size_t stride,
                                                           const float *V, const float *scale, const float *n2,
                                                           int taps, int Co, int Ci, float reg, float *dV,
                                                           float *dg) {
    const int co = blockIdx.x;
    float acc = 0.f;
    for (int i = threadIdx.x; i < taps * Ci; i += 256) {
        const size_t o = ((size_t)(i / Ci) * Co + co) * Ci + (i % Ci);
        float dw = 0.f;
        for (int z = 0; z < splits; ++z) dw += slab[z * stride + o];
        dV[o] = dw;
        acc = fmaf(dw, V[o], acc);
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float c = red[0], s = scale[co], nn = n2[co];
    if (threadIdx.x == 0 && dg) dg[co] = c * rsqrtf(nn);
    for (int i = threadIdx.x; i < taps * Ci; i += 256) {
        const size_t o = ((size_t)(i / Ci) * Co + co) * Ci + (i % Ci);
        dV[o] = s * dV[o] - (s / nn) * c * V[o] + reg * V[o];
    }
}

static int wgrad_t_klen(const ConvGeom &g) {   // transposed layer: M = taps * Co_t (= g.Ci), N = Ci_t (= g.Co)
    const long long K = (long long)g.B * g.OH * g.OW;
    return gg_klen(K, wgrad_split_count(K, (long long)g.KH * g.KW * g.Ci, g.Co));
}

extern "C" size_t cfl_conv_transpose_workspace_bytes(const CflConv *c) {
    ConvGeom g;
    if (make_geom_t(c, &g)) return 0;
    const size_t welems = (size_t)g.KH * g.KW * g.Ci * g.Co;
    return (2 * (size_t)g.Ci + 64 + (size_t)gg_splits((long long)g.B * g.OH * g.OW, wgrad_t_klen(g)) * welems) *
           sizeof(float);
}

extern "C" int cfl_conv2d_transpose_wn_fwd(const CflConv *c, const float *x, const float *V, const float *gain,
                                           const float *bias, float *y, void *workspace, size_t workspace_bytes,
                                           cfl_stream_t stream) {
    ConvGeom g;
    int rc = make_geom_t(c, &g);
    if (rc) return rc;
    if (!x || !V || !y || !workspace) return cfl_set_err(CFL_E_SHAPE, "NULL pointer");
    if (workspace_bytes < cfl_conv_transpose_workspace_bytes(c))
        return cfl_set_err(CFL_E_WORKSPACE, "transposed conv workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int Co = g.Ci, Ci = g.Co, taps = g.KH * g.KW;   // of the transposed layer
    float *scale = (float *)workspace, *n2 = scale + Co;
    hipLaunchKernelGGL(convt_scale_kernel, dim3(Co), dim3(256), 0, st, V, gain, taps, Co, Ci, scale, n2);
    if (Ci % 4 == 0)
        gemm_gather_modes<GG_VEC_K, GG_VEC_K>(g.B * g.H * g.W, Co, taps * Ci, gg_klen(taps * Ci, 1), TGatherX{x, g},
                                              TFilterFwd{V, Co, Ci, gg_make_div(Ci)}, StoreFwd{y, scale, bias, Co, g.act}, st);
    else
        gemm_gather(g.B * g.H * g.W, Co, taps * Ci, gg_klen(taps * Ci, 1), TGatherX{x, g}, TFilterFwd{V, Co, Ci, gg_make_div(Ci)},
                    StoreFwd{y, scale, bias, Co, g.act}, st);
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "transposed conv fwd launch failed");
}

extern "C" int cfl_conv2d_transpose_wn_bwd(const CflConv *c, const float *x, const float *V, const float *gain,
                                           const float *y, const float *dy, float reg_const, float *dx,
                                           float *dV, float *dg, float *db, void *workspace,
                                           size_t workspace_bytes, cfl_stream_t stream) {
    ConvGeom g;
    int rc = make_geom_t(c, &g);
    if (rc) return rc;
    if (!V || !dy || !workspace || (!x && dV) || (!dV && !dx)) return cfl_set_err(CFL_E_SHAPE, "NULL pointer");
    if (!y) {
        if (g.act != 0) return cfl_set_err(CFL_E_SHAPE, "transposed conv bwd: y is required when act != 0");
        y = dy;
    }
    if (workspace_bytes < cfl_conv_transpose_workspace_bytes(c))
        return cfl_set_err(CFL_E_WORKSPACE, "transposed conv workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int Co = g.Ci, Ci = g.Co, taps = g.KH * g.KW;
    float *scale = (float *)workspace, *n2 = scale + Co;
    float *slab = (float *)workspace + 2 * (size_t)Co + 64;
    const int npix_in = g.B * g.OH * g.OW;   // pixels of x
    hipLaunchKernelGGL(convt_scale_kernel, dim3(Co), dim3(256), 0, st, V, gain, taps, Co, Ci, scale, n2);
    const bool vec = (Ci % 4 == 0) && (Co % 4 == 0);
    if (dx) {
        if (vec)
            gemm_gather_modes<GG_VEC_K, GG_VEC_MN>(npix_in, Ci, taps * Co, gg_klen(taps * Co, 1),
                                                   TIm2colDy{dy, y, scale, g}, PlainKN{V, Ci}, StorePlain{dx, Ci}, st);
        else
            gemm_gather(npix_in, Ci, taps * Co, gg_klen(taps * Co, 1), TIm2colDy{dy, y, scale, g}, PlainKN{V, Ci},
                        StorePlain{dx, Ci}, st);
    }
    if (dV) {
        const int klen = wgrad_t_klen(g);
        const int splits = gg_splits(npix_in, klen);
        const size_t welems = (size_t)taps * Co * Ci;
        if (vec)
            gemm_gather_modes<GG_VEC_MN, GG_VEC_MN>(taps * Co, Ci, npix_in, klen,
                                                    TIm2colDyT{TIm2colDy{dy, y, nullptr, g}}, PlainKN{x, Ci},
                                                    StoreSlab{slab, welems, Ci}, st);
        else
            gemm_gather(taps * Co, Ci, npix_in, klen, TIm2colDyT{TIm2colDy{dy, y, nullptr, g}}, PlainKN{x, Ci},
                        StoreSlab{slab, welems, Ci}, st);
        size_t gstride;
        const int groups = slab_presum(slab, splits, welems, welems, &gstride, st);
        hipLaunchKernelGGL(convt_wfinal_kernel, dim3(Co), dim3(256), 0, st, slab, groups, gstride, V, scale, n2,
                           taps, Co, Ci, reg_const, dV, gain ? dg : nullptr);
    }
    if (db) hipLaunchKernelGGL(conv_bgrad_kernel, dim3(Co), dim3(256), 0, st, dy, y, g.B * g.H * g.W, Co, g.act, db);
    return hipGetLastError() == hipSuccess ? CFL_OK : cfl_set_err(CFL_E_HIP, "transposed conv bwd launch failed");
}
