// pair_finalize.h -- finalize kernel (CFL_DEBUG_NOFUSE escape hatch), flat TF-Adam kernels, row gather
// Part of the pair-step translation unit: included by cfl_hip.hip (and nothing else) behind the common device helpers; see the
// header comment of cfl_hip.hip for the launch structure and the fragment-major layouts, DESIGN.md section 4 for what runs when.
#pragma once

// ---------------------------------------------------------------------------
// finalize: weight-gradient slabs + row-reduced column sums -> flat gradient ;
//           last block -> scalars.  Purely element-wise: every reduction over rows
//           was done by grad_red_block, every reduction over theta by mid_reg_block.
// ---------------------------------------------------------------------------
enum { RK_ZERO = 0, RK_W, RK_BIAS, RK_GAIN, RK_THR, RK_MONO_W, RK_MONO_G };

struct Region {
    long long off, cnt;        // floats (64-aligned)
    int kind, reg;
    const float *slab[2];      // weight-gradient slabs [P] x Wf (one per contributing side)
    int cs_dy[2], cs_cw[2];    // colsum offsets of the bias / gain column sums (-1: none)
    int npad, n;               // padded / logical columns of the head
    const float *g, *n2;       // weight-norm (gain snapshot, squared norms)
};

struct FinArgs {
    // compact copy of the region bounds (floats): the region search reads these with three wide scalar loads
    int rbeg[CFL_MAX_REGIONS], rend[CFL_MAX_REGIONS];
    Region reg[CFL_MAX_REGIONS];
    int nregions;
    long long total;           // floats in theta
    const float *theta;
    float *grad;
    const float *colsum;
    int cs_rowq, cs_mono, cs_duc;
    int P, D, L, kpad, weight_norm;
    float in_mul, reg_const;
    int use_threshold;
    float pos_weight, caffe_margin, lambda_m;
    int B;
    const float *regpart;
    int nregblocks;
    float *scalars;
    int nblocks_main;
    // optional fused Adam (theta_out aliases theta)
    float *adam_m, *adam_v, *theta_out;
    float lr_t, b1, b2, eps;
    const float *thr_copy;     // max(thr,1e-6) as seen by the mid kernel of this step
};

// gradient of the 4 parameters at `base`, which lie in region `rg`
__device__ __forceinline__ f32x4 fin_region_grad(const FinArgs &a, const Region &rg, long long base, const f32x4 th) {
    f32x4 gr = {0.f, 0.f, 0.f, 0.f};
    const long long rel = base - rg.off;
    switch (rg.kind) {
        case RK_W: {
            // Wf layout: block = rel/256 -> nt = block / G ; c16 = ((rel%256)/4) % 16
            const int G = a.D >> 4;
            const int c = (int)((rel >> 8) / G) * 16 + (int)((rel >> 2) & 15);
            const long long ps = (long long)rg.npad * a.D;
            f32x4 t[2][8];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pp = 0; pp < 8; ++pp)
                    t[s][pp] = (rg.slab[s] && pp < a.P) ? *(const f32x4 *)(rg.slab[s] + rel + pp * ps)
                                                        : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) gr += t[s][pp];
            // the slabs hold X^T dy with unscaled dy: apply the input / weight-norm scale
            if (a.weight_norm) {
                if (c < rg.n) {
                    const float n2 = rg.n2[c], n = sqrtf(n2);
                    gr *= n2 > 0.f ? a.in_mul * rg.g[c] / n : 0.f;
                    float cw = 0.f;
                    for (int s = 0; s < 2; ++s)
                        if (rg.cs_cw[s] >= 0) cw += a.colsum[rg.cs_cw[s] + c];
                    // (explicit fma: the fused tail of the weight-gradient launch performs the same operations)
                    const float s2 = n2 > 0.f ? rg.g[c] * cw / (n2 * n) : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gr[e] = fmaf(-s2, th[e], gr[e]);
                } else {
                    gr *= 0.f;
                }
            } else {
                gr *= a.in_mul;
            }
            break;
        }
        case RK_BIAS: {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (int)rel + e;
                if (c < rg.n)
                    for (int s = 0; s < 2; ++s)
                        if (rg.cs_dy[s] >= 0) gr[e] += a.colsum[rg.cs_dy[s] + c];
            }
            break;
        }
        case RK_GAIN: {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (int)rel + e;
                if (c < rg.n) {
                    float cw = 0.f;
                    for (int s = 0; s < 2; ++s)
                        if (rg.cs_cw[s] >= 0) cw += a.colsum[rg.cs_cw[s] + c];
                    const float n2 = rg.n2[c];
                    gr[e] = n2 > 0.f ? cw / sqrtf(n2) : 0.f;
                }
            }
            break;
        }
        case RK_THR: {
            if (rel == 0) gr[0] = th[0] >= CFL_THR_FLOOR ? a.colsum[a.cs_rowq + P_DTHR] : 0.f;
            break;
        }
        case RK_MONO_W: {  // V[L][kpad]; cs_dy[0] >= 0 marks the encoder whose gate is used
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int l = (int)((rel + e) / a.kpad), kk = (int)((rel + e) % a.kpad);
                if (l < a.L && kk < rg.n && rg.cs_dy[0] >= 0) {
                    float g1 = a.colsum[a.cs_mono + l * a.kpad + kk];
                    if (a.weight_norm) {
                        const float cw = a.colsum[a.cs_duc + kk];
                        const float n2 = rg.n2[kk], n = sqrtf(n2);
                        if (n2 > 0.f) g1 = fmaf(-(rg.g[kk] * cw / (n2 * n)), th[e], g1);   // (explicit: the fused tail performs the same operations)
                    }
                    gr[e] = g1;
                }
            }
            break;
        }
        case RK_MONO_G: {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = (int)rel + e;
                if (rg.cs_dy[0] >= 0 && kk < rg.n) {
                    const float n2 = rg.n2[kk];
                    gr[e] = n2 > 0.f ? a.colsum[a.cs_duc + kk] / sqrtf(n2) : 0.f;
                }
            }
            break;
        }
        default: break;
    }
    if (rg.reg) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gr[e] = fmaf(a.reg_const, th[e], gr[e]);
    }
    return gr;
}

extern "C" __global__ __launch_bounds__(256) void cfl_finalize_kernel(FinArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x == a.nblocks_main) {
        // ---- scalars (cfl/models/cfl.py:868-949) ----------------------------------
        __shared__ float regsum_s;
        if (wave == 0) {
            float s = 0.f;
            for (int b = lane; b < a.nregblocks; b += 64) s += a.regpart[b];
            s = wave_sum(s);
            if (lane == 0) regsum_s = 0.5f * a.reg_const * s;
        }
        __syncthreads();
        if (tid == 0)
            write_scalars(a.scalars, a.colsum + a.cs_rowq, regsum_s, a.B, a.use_threshold, a.pos_weight,
                          a.caffe_margin, a.lambda_m, a.thr_copy[0]);
        return;
    }
    // main blocks: one float4 (4 consecutive parameters, same region) per thread; all
    // loads of a thread are independent and issued together.
    const long long base = ((long long)blockIdx.x * 256 + tid) * 4;
    if (base >= a.total) return;
    // region of this float4: all descriptors' bounds are fetched at once (a search loop with an early
    // exit made every iteration a dependent kernel-argument load)
    int k = a.nregions;
    const int b32 = (int)base;   // theta has < 2^31 floats (make_plan)
#pragma unroll
    for (int i = CFL_MAX_REGIONS - 1; i >= 0; --i)
        k = ((b32 >= a.rbeg[i]) & (b32 < a.rend[i])) ? i : k;   // unused slots are empty ranges (0, 0)
    const f32x4 th = *(const f32x4 *)(a.theta + base);
    // the Adam slots are requested together with theta and the slabs (one memory round trip, not two:
    // behind the gradient store the compiler could not hoist them)
    f32x4 mm = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (a.adam_m) {
        mm = *(const f32x4 *)(a.adam_m + base);
        vv = *(const f32x4 *)(a.adam_v + base);
    }
    f32x4 gr = {0.f, 0.f, 0.f, 0.f};
    if (k < a.nregions) {
        // Almost every wave lies inside one region (regions are 64-float aligned, a wave covers 256 floats): its
        // descriptor is then fetched with scalar loads.  Per-lane descriptors (vector loads from the kernel
        // arguments, one more dependent round trip before the slab loads can be issued) only at region boundaries.
        const int ku = __builtin_amdgcn_readfirstlane(k);
        if (__builtin_amdgcn_ballot_w64(k != ku) == 0)
            gr = fin_region_grad(a, a.reg[ku], base, th);
        else
            gr = fin_region_grad(a, a.reg[k], base, th);
    }
    *(f32x4 *)(a.grad + base) = gr;
    if (a.adam_m) {  // fused TF-Adam apply (single-GPU step)
        f32x4 tn = th;
        adam4(tn, mm, vv, gr, a.lr_t, a.b1, a.b2, a.eps);
        *(f32x4 *)(a.adam_m + base) = mm;
        *(f32x4 *)(a.adam_v + base) = vv;
        *(f32x4 *)(a.theta_out + base) = tn;
    }
}

// ---------------------------------------------------------------------------
// TF-1.x Adam, flat
// ---------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void cfl_adam_kernel(float *theta, float *m, float *v,
                                                                 const float *grad, long long n4,
                                                                 float lr_t, float b1, float b2,
                                                                 float eps, float gscale) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n4; i += stride) {
        f32x4 g = ((const f32x4 *)grad)[i] * gscale;
        f32x4 mm = ((f32x4 *)m)[i], vv = ((f32x4 *)v)[i], th = ((f32x4 *)theta)[i];
        adam4(th, mm, vv, g, lr_t, b1, b2, eps);
        ((f32x4 *)m)[i] = mm;
        ((f32x4 *)v)[i] = vv;
        ((f32x4 *)theta)[i] = th;
    }
}

// ... that also writes the kept bf16 planes of the weights it updates (the update of a data-parallel step: theta_planes.h)
extern "C" __global__ __launch_bounds__(256) void cfl_adam_planes_kernel(float *theta, float *m, float *v,
                                                                        const float *grad, long long n4,
                                                                        float lr_t, float b1, float b2,
                                                                        float eps, float gscale, ThetaPlaneRegions pr,
                                                                        const float *scal_src, float *scal_dst) {
    // (data-parallel loop with the validation fetch inside: the 16 global scalar sums behind the gradient also go to the caller's
    // pinned ring slot -- no copy command on the stream; both NULL everywhere else)
    if (scal_dst && blockIdx.x == 0 && threadIdx.x < 16) scal_dst[threadIdx.x] = scal_src[threadIdx.x];
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n4; i += stride) {
        f32x4 g = ((const f32x4 *)grad)[i] * gscale;
        f32x4 mm = ((f32x4 *)m)[i], vv = ((f32x4 *)v)[i], th = ((f32x4 *)theta)[i];
        adam4(th, mm, vv, g, lr_t, b1, b2, eps);
        ((f32x4 *)m)[i] = mm;
        ((f32x4 *)v)[i] = vv;
        ((f32x4 *)theta)[i] = th;
        theta_planes_store4(pr, i * 4, th);
    }
}

// ---------------------------------------------------------------------------
// row gather: out[i,:] = table[idx[i],:]   (one wave per row, 16 B per lane)
// ---------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void cfl_gather_kernel(const float *table,
                                                                   const long long *idx,
                                                                   long long n, long long D,
                                                                   float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long row = (long long)blockIdx.x * 4 + wave;
    const long long stride = (long long)gridDim.x * 4;
    const long long d4 = D >> 2;
    for (; row < n; row += stride) {
        const f32x4 *src = (const f32x4 *)(table + idx[row] * D);
        f32x4 *dst = (f32x4 *)(out + row * D);
        for (long long k = lane; k < d4; k += 64) dst[k] = src[k];
    }
}

