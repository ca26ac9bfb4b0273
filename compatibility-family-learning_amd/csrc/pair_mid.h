// pair_mid.h -- row math of the pair step: slice sums, heads, distance, loss, dL/dY (register form and wave-per-row form), L2 partial sums
// Part of the pair-step translation unit: included by cfl_hip.hip (and nothing else) behind the common device helpers; see the
// header comment of cfl_hip.hip for the launch structure and the fragment-major layouts, DESIGN.md section 4 for what runs when.
#pragma once

// ---------------------------------------------------------------------------
// mid: per pair row: slice-sum, head epilogue, distance, loss, dL/dY.
//   workgroup = 64 threads = 4 rows x 16 column parts; lane (p = tid&15, j = tid>>4)
//   owns the latent coordinates l == p (mod 16) of row blockIdx.x*4 + j, for every
//   prototype k.
//   phase 1: the block's 4 rows of every slice slab are read with coalesced
//            16-byte loads (all slices in flight at once), summed in slice order
//            and parked in LDS as Y[row][col]; biases / weight-norm scales / gate
//            weights are staged in LDS in the same round of loads;
//   phase 2: per-lane math; sums over l are completed with xor-shuffles inside
//            the 16-lane row group.  Per-lane runtime-indexed state lives in LDS
//            as [slot][64].
//   dL/dy is written UNSCALED in fragment-major order (dYf) -- the weight-norm /
//   input scale is applied to the finished weight gradient by finalize -- so its
//   column sums are the bias gradients.  Everything else that needs a sum over
//   rows is written as one more fragment-major tile; the reductions ride in the
//   grad launch (grad_red_block).
// ---------------------------------------------------------------------------
#define MID_RB 4

struct MidSide {
    const float *ypart;   // [S][Rpad][npad]
    long long sstride;
    const float *b;       // biases or null
    const float *g;       // wn gains or null
    const float *n2;      // wn squared column norms or null
    float *dyf;           // fragment-major dL/dy (unscaled)
    float *cwf;           // fragment-major dL/dy * (x_hat.V) (weight-norm gain rows) or null
    int n, npad;
    int is_proto;         // 1: columns are k*L + l ; 0: columns are l
};

struct MidArgs {
    MidSide side[2];      // 0 = src, 1 = dst
    const float *mono_w, *mono_g, *mono_n2;  // monomer gate head V[L][kpad]
    float *mono_ya, *mono_du;                // row-major [Rpad][lpad], [Rpad][kpad]
    float *mono_duc;                         // row-major [Rpad][kpad] (weight-norm)
    int kpad, lpad;
    int S, L, K, Lq, dist_type, act, weight_norm;
    float in_mul;
    const float *thr;
    int B, R, Rpad;
    int train, use_threshold;
    float pos_weight, caffe_margin, lambda_m;
    float *scores, *dists;
    float *rowqf;         // fragment-major tile of the per-row loss quantities
    float *thr_copy;      // max(thr, 1e-6) of this step (read by finalize's scalar block)
    int *zero_i;          // hand-off tickets / flags of the fused weight-gradient launch: cleared here, every step
    int nzero;
    int nrb, ys;          // row blocks; LDS row stride of Y (floats)
    // extra scoring rows of a training call (RowExtra): rows [xrow0, xrow0 + xn) of the partial slabs, forward only, scores to
    // xscores[0 .. xn); nxb = their row blocks (wave-per-row kernels only), dispatched behind the training rows' blocks
    int xrow0, xn, nxb;
    float *xscores;
    // regulariser blocks
    const float *theta;
    float *regpart;
    int nreg_ranges;
    long long reg_off[CFL_MAX_REGIONS], reg_cnt[CFL_MAX_REGIONS];
    long long reg_total_groups;  // number of 64-float groups over all ranges
};

__device__ __forceinline__ float act_fn(float y, int act) {
    switch (act) {
        case CFL_ACT_SIGMOID: return 1.f / (1.f + expf(-y));
        case CFL_ACT_TANH: return tanhf(y);
        case CFL_ACT_RELU: return fmaxf(y, 0.f);
        default: return y;
    }
}
__device__ __forceinline__ float act_grad(float a, int act) {
    switch (act) {
        case CFL_ACT_SIGMOID: return a * (1.f - a);
        case CFL_ACT_TANH: return 1.f - a * a;
        case CFL_ACT_RELU: return a > 0.f ? 1.f : 0.f;
        default: return 1.f;
    }
}
// all-reduce over the 16 column parts of a row (one DPP row): pure VALU, no LDS crossbar.
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror form a butterfly.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float sum_p(float x) {
    x = dpp_add<0xB1>(x);   // lane ^ 1
    x = dpp_add<0x4E>(x);   // lane ^ 2
    x = dpp_add<0x141>(x);  // other quad of the 8-lane half
    x = dpp_add<0x140>(x);  // other half of the 16-lane row
    return x;
}
// sum over the 64 lanes, the same value in every lane: four DPP adds inside the 16-lane rows, then the four row sums
// through v_readlane (SGPRs).  The wave-per-row mid kernel IS its latency chain (tools/mid_stamp_probe.py); wave_sum's six
// ds_bpermute round trips through the LDS crossbar were ~450-750 cycles of it.
__device__ __forceinline__ float wave_sum_dpp(float x) {
    const int r = __float_as_int(sum_p(x));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(r, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(r, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(r, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(r, 48));
    return (r0 + r1) + (r2 + r3);
}
// 1-ulp hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sqrt_f32): the
// per-row math is one wave per block, so its instruction count is its latency.
__device__ __forceinline__ float fexp(float x) { return __expf(x); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float flog1pexp(float negabs) { return __logf(1.f + __expf(negabs)); }

__device__ void mid_reg_block(const MidArgs &a, int blk) {
    // 64 threads; block handles 64 groups of 64 floats of the regularised ranges
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int gi = 0; gi < 64; ++gi) {
        long long g = (long long)blk * 64 + gi;
        if (g >= a.reg_total_groups) break;
        long long rem = g;
        for (int k = 0; k < a.nreg_ranges; ++k) {
            long long ng = a.reg_cnt[k] >> 6;
            if (rem < ng) {
                float v = a.theta[a.reg_off[k] + rem * 64 + tid];
                acc = fmaf(v, v, acc);
                break;
            }
            rem -= ng;
        }
    }
    acc = wave_sum(acc);
    // (written through: in the mid-in-grad launch the consumer is a reduction block of the same launch)
    if (tid == 0) __hip_atomic_store(a.regpart + blk, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int S>
__device__ __forceinline__ f32x4 slab_sum(const float *src, long long sstride) {
    f32x4 t[S];
#pragma unroll
    for (int s = 0; s < S; ++s) t[s] = *(const f32x4 *)(src + (size_t)s * sstride);
    f32x4 acc = t[0];
#pragma unroll
    for (int s = 1; s < S; ++s) acc += t[s];
    return acc;
}

// float offset of (row r, column c) inside a fragment-major buffer with RG row groups
__device__ __forceinline__ size_t frag_off(int r, int c, int RG) {
    return ((size_t)(c >> 4) * RG + (r >> 4)) * 256 + (((r >> 2) & 3) * 16 + (c & 15)) * 4 + (r & 3);
}


// ---------------------------------------------------------------------------
// Register-resident version of the per-row math of the mid kernel for small
// (K <= KM, ceil(L/16) <= LQ): the same arithmetic as the generic LDS-array path
// below, with every per-lane array in VGPRs and fully unrolled loops (entries
// beyond the lane's own coordinates are zero-filled and contribute nothing), so
// that the kernel is a short straight-line ALU sequence instead of a chain of
// dependent LDS round trips.  Returns through the same row buffers.
// ---------------------------------------------------------------------------
template <int KM, int LQ>
__device__ __forceinline__ void mid_math_reg(const MidArgs &a, const float *Y, const float *SC,
                                             const float *BI, const float *MW, float thr_raw) {
    const int tid = threadIdx.x, p = tid & 15, j = tid >> 4;
    const int r = blockIdx.x * MID_RB + j;
    const bool valid = r < a.R;
    const int L = a.L, K = a.K, RG = a.Rpad >> 4;
    const int myL = p < L ? (L - p + 15) >> 4 : 0;
    const MidSide &ss = a.side[0], &sd = a.side[1];
    const int ks = ss.is_proto ? K : 1, kd = sd.is_proto ? K : 1;
    const int offd = ss.npad;

    float As[KM][LQ], Ad[KM][LQ], Xs[KM][LQ], Xd[KM][LQ], Rl[LQ];
    // ---- head epilogue ----
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int li = 0; li < LQ; ++li) {
            const int l = p + 16 * li;
            {
                const bool in = valid && li < myL && k < ks;
                const int c = in ? k * L + l : 0;
                const float xv = in ? Y[j * a.ys + c] * a.in_mul : 0.f;
                const float yy = xv * SC[c] + BI[c];
                Xs[k][li] = xv;
                if (a.dist_type == CFL_DIST_MONOMER && k == 0) Rl[li] = in ? yy : 0.f;
                As[k][li] = in ? act_fn(yy, a.act) : 0.f;
            }
            {
                const bool in = valid && li < myL && k < kd;
                const int c = in ? k * L + l : 0;
                const float xv = in ? Y[j * a.ys + offd + c] * a.in_mul : 0.f;
                const float yy = xv * SC[offd + c] + BI[offd + c];
                Xd[k][li] = xv;
                Ad[k][li] = in ? act_fn(yy, a.act) : 0.f;
            }
        }

    // ---- distance ----
    float d = 0.f, sK[KM], qK[KM], eK[KM], uK[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) sK[k] = qK[k] = eK[k] = uK[k] = 0.f;
    if (a.dist_type == CFL_DIST_PCD) {
        if (K > 1) {
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {
                    float e = 0.f;
#pragma unroll
                    for (int li = 0; li < LQ; ++li) { const float df = Ad[0][li] - As[k][li]; e = fmaf(df, df, e); }
                    e = -sum_p(e);
                    sK[k] = e;
                    mx = fmaxf(mx, e);
                }
            float den = 0.f;
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) { sK[k] = fexp(sK[k] - mx); den += sK[k]; }
            const float inv = frcp(den);
#pragma unroll
            for (int k = 0; k < KM; ++k) sK[k] = k < K ? sK[k] * inv : 0.f;
#pragma unroll
            for (int li = 0; li < LQ; ++li) {
                float m = 0.f;
#pragma unroll
                for (int k = 0; k < KM; ++k) m = fmaf(sK[k], As[k][li], m);
                const float rl = Ad[0][li] - m;
                Rl[li] = rl;
                d = fmaf(rl, rl, d);
#pragma unroll
                for (int k = 0; k < KM; ++k) qK[k] = fmaf(rl, As[k][li], qK[k]);
            }
            d = sum_p(d);
        } else {
#pragma unroll
            for (int li = 0; li < LQ; ++li) { const float df = Ad[0][li] - As[0][li]; d = fmaf(df, df, d); }
            d = sum_p(d);
        }
    } else if (a.dist_type == CFL_DIST_MONOMER) {
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) {
                float u = 0.f, e = 0.f;
#pragma unroll
                for (int li = 0; li < LQ; ++li) {
                    const int l = p + 16 * li;
                    const float w = li < myL ? MW[l * a.kpad + k] : 0.f;
                    u = fmaf(Rl[li], w, u);
                    const float df = As[0][li] - Ad[k][li];
                    e = fmaf(df, df, e);
                }
                u = sum_p(u);
                e = sum_p(e);
                uK[k] = u;
                if (a.weight_norm) u *= a.mono_g[k] / sqrtf(a.mono_n2[k]);
                sK[k] = u;
                eK[k] = e;
                mx = fmaxf(mx, u);
            }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) { sK[k] = fexp(sK[k] - mx); den += sK[k]; }
        const float inv = frcp(den);
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            sK[k] = k < K ? sK[k] * inv : 0.f;
            d = fmaf(sK[k], eK[k], d);
        }
    } else {
#pragma unroll
        for (int li = 0; li < LQ; ++li) { const float df = As[0][li] - Ad[0][li]; d = fmaf(df, df, d); }
        d = sum_p(d);
    }

    // ---- threshold, loss, dL/dd ----
    const float thr = fmaxf(thr_raw, CFL_THR_FLOOR);
    const float o = thr - d;
    if (!a.train) {
        if (valid && p == 0) {
            a.scores[r] = o;
            if (a.dists) a.dists[r] = d;
        }
        return;
    }
    const bool is_pos = r < a.B;
    const float invB = 1.f / (float)a.B;
    const float pw = a.pos_weight != 0.f ? a.pos_weight : 1.f;
    const float sp = flog1pexp(-fabsf(o));
    const float bce = fmaxf(o, 0.f) - (is_pos ? o : 0.f) + sp;
    const float eo = fexp(-fabsf(o));
    const float sig = (o >= 0.f ? 1.f : eo) * frcp(1.f + eo);
    const float dlo = is_pos ? (sig - 1.f) * pw * invB : sig * invB;
    float dd = 0.f;
    if (a.use_threshold) dd -= dlo;
    float hinge = 0.f;
    if (a.caffe_margin != 0.f) {
        if (is_pos) dd += 0.5f * pw * invB;
        else {
            hinge = fmaxf(0.f, a.caffe_margin - d);
            if (d < a.caffe_margin) dd -= 0.5f * invB;
        }
    } else if (a.lambda_m != 0.f) {
        if (is_pos) dd += pw * a.lambda_m * invB;
    }
    if (!valid) dd = 0.f;
    if (blockIdx.x == 0 && tid == 0) a.thr_copy[0] = thr;
    if (blockIdx.x == 0 && a.zero_i)
        for (int i = tid; i < a.nzero; i += blockDim.x) a.zero_i[i] = 0;
    {
        const bool pos = valid && is_pos, neg = valid && !is_pos;
        float qv = 0.f;
        switch (p) {
            case P_BCE_POS: qv = pos ? bce : 0.f; break;
            case P_BCE_NEG: qv = neg ? bce : 0.f; break;
            case P_OK_POS: qv = (pos && o > 0.f) ? 1.f : 0.f; break;
            case P_OK_NEG: qv = (neg && o <= 0.f) ? 1.f : 0.f; break;
            case P_D_POS: qv = pos ? d : 0.f; break;
            case P_D_NEG: qv = neg ? d : 0.f; break;
            case P_O_POS: qv = pos ? o : 0.f; break;
            case P_O_NEG: qv = neg ? o : 0.f; break;
            case P_DTHR: qv = valid ? dlo : 0.f; break;
            case P_HINGE_NEG: qv = neg ? hinge : 0.f; break;
            case P_SQRT_POS: qv = pos ? fsqrt(d + 1e-7f) : 0.f; break;
            case P_SQRT_NEG: qv = neg ? fsqrt(d + 1e-7f) : 0.f; break;
            default: break;
        }
        a.rowqf[frag_off(r, p, RG)] = qv;
    }

    // ---- backward ----
    auto emit = [&](const MidSide &sx, float A, float X, int k, int li, float dA, float extra_dy) {
        if (li >= myL) return;
        const int c = k * L + p + 16 * li;
        float dy = dA * act_grad(A, a.act) + extra_dy;
        if (!valid) dy = 0.f;
        const size_t o_ = frag_off(r, c, RG);
        sx.dyf[o_] = dy;
        if (sx.cwf) sx.cwf[o_] = dy * X;
    };
    if (a.dist_type == CFL_DIST_PCD) {
        if (K > 1) {
            float qbar = 0.f;
#pragma unroll
            for (int k = 0; k < KM; ++k) { qK[k] = -2.f * sum_p(qK[k]); qbar = fmaf(sK[k], qK[k], qbar); }
#pragma unroll
            for (int k = 0; k < KM; ++k) qK[k] = sK[k] * (qK[k] - qbar);  // dl_k
#pragma unroll
            for (int li = 0; li < LQ; ++li) {
                const float v = Ad[0][li], rl = Rl[li];
                float dv = 2.f * rl;
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
                        const float vmP = v - As[k][li];
                        dv = fmaf(-2.f * qK[k], vmP, dv);
                        const float dP = -2.f * sK[k] * rl + 2.f * qK[k] * vmP;
                        emit(ss, As[k][li], Xs[k][li], k, li, dP * dd, 0.f);
                    }
                emit(sd, Ad[0][li], Xd[0][li], 0, li, dv * dd, 0.f);
            }
        } else {
#pragma unroll
            for (int li = 0; li < LQ; ++li) {
                const float df = Ad[0][li] - As[0][li];
                emit(ss, As[0][li], Xs[0][li], 0, li, -2.f * df * dd, 0.f);
                emit(sd, Ad[0][li], Xd[0][li], 0, li, 2.f * df * dd, 0.f);
            }
        }
    } else if (a.dist_type == CFL_DIST_MONOMER) {
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) {
                const float du = sK[k] * (eK[k] - d) * dd;
                float scm = 1.f;
                if (a.weight_norm) scm = a.mono_g[k] / sqrtf(a.mono_n2[k]);
                if (p == 0) {
                    a.mono_du[(size_t)r * a.kpad + k] = du * scm;
                    if (a.weight_norm) a.mono_duc[(size_t)r * a.kpad + k] = valid ? du * uK[k] : 0.f;
                }
                qK[k] = du * scm;
            }
#pragma unroll
        for (int li = 0; li < LQ; ++li) {
            const int l = p + 16 * li;
            if (li < myL) a.mono_ya[(size_t)r * a.lpad + l] = valid ? Rl[li] : 0.f;
            float da = 0.f, ex = 0.f;
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {
                    const float amP = As[0][li] - Ad[k][li];
                    da = fmaf(2.f * sK[k], amP, da);
                    emit(sd, Ad[k][li], Xd[k][li], k, li, -2.f * sK[k] * amP * dd, 0.f);
                    ex = fmaf(qK[k], li < myL ? MW[l * a.kpad + k] : 0.f, ex);
                }
            emit(ss, As[0][li], Xs[0][li], 0, li, da * dd, ex);
        }
    } else {
#pragma unroll
        for (int li = 0; li < LQ; ++li) {
            const float df = As[0][li] - Ad[0][li];
            emit(ss, As[0][li], Xs[0][li], 0, li, 2.f * df * dd, 0.f);
            emit(sd, Ad[0][li], Xd[0][li], 0, li, -2.f * df * dd, 0.f);
        }
    }
    for (int side = 0; side < 2; ++side) {
        const MidSide &sx = a.side[side];
        for (int c = sx.n + p; c < sx.npad; c += 16) {
            sx.dyf[frag_off(r, c, RG)] = 0.f;
            if (sx.cwf) sx.cwf[frag_off(r, c, RG)] = 0.f;
        }
    }
}

template <int KM, int LQ>
__global__ __launch_bounds__(64) void cfl_mid_kernel(MidArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = (float *)smem;
    if ((int)blockIdx.x >= a.nrb) {
        mid_reg_block(a, blockIdx.x - a.nrb);
        return;
    }
    const int tid = threadIdx.x, p = tid & 15, j = tid >> 4;
    const int row0 = blockIdx.x * MID_RB;
    const int r = row0 + j;
    const bool valid = r < a.R;
    const int L = a.L, K = a.K, Lq = a.Lq, RG = a.Rpad >> 4;
    const int myL = p < L ? (L - p + 15) >> 4 : 0;  // number of l = p + 16*li < L
    const MidSide &ss = a.side[0], &sd = a.side[1];
    const int ks = ss.is_proto ? K : 1, kd = sd.is_proto ? K : 1;

    // LDS carve: Y[4][ys], per-column scale SC[ys] and bias BI[ys], gate weights
    // MW[L*kpad], then per-lane arrays [slots][64]
    float *Y = lds;
    float *SC = Y + MID_RB * a.ys;
    float *BI = SC + a.ys;
    float *MW = BI + a.ys;
    float *As = MW + (a.dist_type == CFL_DIST_MONOMER ? ((L * a.kpad + 3) & ~3) : 0);  // src activations ks*Lq
    float *Ad = As + ks * Lq * 64;             // dst activations   kd*Lq
    float *Xs = Ad + kd * Lq * 64;             // src raw x_hat.V (weight-norm)  ks*Lq
    float *Xd = Xs + (a.weight_norm ? ks * Lq * 64 : 0);
    float *Rl = Xd + (a.weight_norm ? kd * Lq * 64 : 0);   // Lq : pcd residual / monomer pre-act
    float *Kv = Rl + Lq * 64;                  // 4 x K small vectors
    float *Ks = Kv, *Kq = Kv + K * 64, *Ke = Kv + 2 * K * 64, *Ku = Kv + 3 * K * 64;

    // ---- phase 1: slice sums + parameters -> LDS ----------------------------------
    for (int side = 0; side < 2; ++side) {
        const MidSide &sx = a.side[side];
        const int nq = sx.npad >> 2;
        const int coloff = side ? a.side[0].npad : 0;
        for (int idx = tid; idx < MID_RB * nq; idx += 64) {
            const int jj = idx / nq, c4 = idx - jj * nq;
            const float *src = sx.ypart + (size_t)(row0 + jj) * sx.npad + 4 * c4;
            f32x4 acc;
            switch (a.S) {
                case 1: acc = slab_sum<1>(src, sx.sstride); break;
                case 2: acc = slab_sum<2>(src, sx.sstride); break;
                case 4: acc = slab_sum<4>(src, sx.sstride); break;
                case 8: acc = slab_sum<8>(src, sx.sstride); break;
                default: acc = slab_sum<16>(src, sx.sstride); break;
            }
            *(f32x4 *)(Y + jj * a.ys + coloff + 4 * c4) = acc;
        }
        for (int c = tid; c < sx.n; c += 64) {
            SC[coloff + c] = a.weight_norm ? sx.g[c] * __builtin_amdgcn_rsqf(sx.n2[c]) : 1.f;
            BI[coloff + c] = sx.b ? sx.b[c] : 0.f;
        }
    }
    if (a.dist_type == CFL_DIST_MONOMER)
        for (int i = tid; i < L * a.kpad; i += 64) MW[i] = a.mono_w[i];
    const float thr_raw = *a.thr;
    __syncthreads();
    // small shapes: register-resident math (same arithmetic as the generic path below); one
    // kernel instantiation per shape class -- co-inlined variants made hipcc spill to scratch
    if constexpr (KM > 0) {
        mid_math_reg<KM, LQ>(a, Y, SC, BI, MW, thr_raw);
        return;
    }
#ifdef ABL_MID_P1ONLY
    if (a.train) { if (tid == 0) a.rowqf[blockIdx.x] = Y[0] + thr_raw; return; }
#endif

    // ---- phase 2.1: head epilogue ------------------------------------------------
    for (int side = 0; side < 2; ++side) {
        const MidSide &sx = a.side[side];
        float *A = side ? Ad : As, *X = side ? Xd : Xs;
        const int coloff = side ? a.side[0].npad : 0;
        const int kk = sx.is_proto ? K : 1;
        for (int k = 0; k < kk; ++k)
            for (int li = 0; li < myL; ++li) {
                const int c = k * L + p + 16 * li;
                float y = Y[j * a.ys + coloff + c];
                if (!valid) y = 0.f;  // rows >= R of the scratch slabs are never written
                const float xv = y * a.in_mul;
                const float yy = xv * SC[coloff + c] + BI[coloff + c];
                const int slot = (k * Lq + li) * 64 + tid;
                if (a.weight_norm) X[slot] = xv;
                if (a.dist_type == CFL_DIST_MONOMER && side == 0) Rl[li * 64 + tid] = yy;
                A[slot] = act_fn(yy, a.act);
            }
    }

    // ---- phase 2.2: distance -----------------------------------------------------
    float d = 0.f;
    if (a.dist_type == CFL_DIST_PCD) {
        if (K > 1) {
            float mx = -INFINITY;
            for (int k = 0; k < K; ++k) {
                float e = 0.f;
                for (int li = 0; li < myL; ++li) {
                    float df = Ad[li * 64 + tid] - As[(k * Lq + li) * 64 + tid];
                    e = fmaf(df, df, e);
                }
                e = -sum_p(e);
                Ks[k * 64 + tid] = e;
                mx = fmaxf(mx, e);
            }
            float den = 0.f;
            for (int k = 0; k < K; ++k) {
                float ex = fexp(Ks[k * 64 + tid] - mx);
                Ks[k * 64 + tid] = ex;
                den += ex;
                Kq[k * 64 + tid] = 0.f;
            }
            const float inv = frcp(den);
            for (int k = 0; k < K; ++k) Ks[k * 64 + tid] *= inv;
            for (int li = 0; li < myL; ++li) {
                const float v = Ad[li * 64 + tid];
                float m = 0.f;
                for (int k = 0; k < K; ++k) m = fmaf(Ks[k * 64 + tid], As[(k * Lq + li) * 64 + tid], m);
                const float rl = v - m;
                Rl[li * 64 + tid] = rl;
                d = fmaf(rl, rl, d);
                for (int k = 0; k < K; ++k) Kq[k * 64 + tid] += rl * As[(k * Lq + li) * 64 + tid];
            }
            d = sum_p(d);
        } else {
            for (int li = 0; li < myL; ++li) {
                float df = Ad[li * 64 + tid] - As[li * 64 + tid];
                d = fmaf(df, df, d);
            }
            d = sum_p(d);
        }
    } else if (a.dist_type == CFL_DIST_MONOMER) {
        // gate u_k = (ya . Vm[:,k]) * scale_k from the PRE-activation outputs (base.py:96)
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) {
            float u = 0.f, e = 0.f;
            for (int li = 0; li < myL; ++li) {
                const int l = p + 16 * li;
                u = fmaf(Rl[li * 64 + tid], MW[l * a.kpad + k], u);
                float df = As[li * 64 + tid] - Ad[(k * Lq + li) * 64 + tid];
                e = fmaf(df, df, e);
            }
            u = sum_p(u);
            e = sum_p(e);
            Ku[k * 64 + tid] = u;  // raw ya.Vm (needed for the weight-norm gain grad)
            if (a.weight_norm) u *= a.mono_g[k] / sqrtf(a.mono_n2[k]);
            Ks[k * 64 + tid] = u;
            Ke[k * 64 + tid] = e;
            mx = fmaxf(mx, u);
        }
        float den = 0.f;
        for (int k = 0; k < K; ++k) {
            float ex = fexp(Ks[k * 64 + tid] - mx);
            Ks[k * 64 + tid] = ex;
            den += ex;
        }
        const float inv = frcp(den);
        for (int k = 0; k < K; ++k) {
            const float w = Ks[k * 64 + tid] * inv;
            Ks[k * 64 + tid] = w;
            d = fmaf(w, Ke[k * 64 + tid], d);
        }
    } else {
        for (int li = 0; li < myL; ++li) {
            float df = As[li * 64 + tid] - Ad[li * 64 + tid];
            d = fmaf(df, df, d);
        }
        d = sum_p(d);
    }

    // ---- phase 2.3: threshold, loss, dL/dd ----------------------------------------
    const float thr = fmaxf(thr_raw, CFL_THR_FLOOR);
    const float o = thr - d;
    if (!a.train) {
        if (valid && p == 0) {
            a.scores[r] = o;
            if (a.dists) a.dists[r] = d;
        }
        return;
    }
    const bool is_pos = r < a.B;
    const float invB = 1.f / (float)a.B;
    const float pw = a.pos_weight != 0.f ? a.pos_weight : 1.f;
    const float sp = flog1pexp(-fabsf(o));
    const float bce = fmaxf(o, 0.f) - (is_pos ? o : 0.f) + sp;
    const float eo = fexp(-fabsf(o));
    const float sig = (o >= 0.f ? 1.f : eo) * frcp(1.f + eo);
    const float dlo = is_pos ? (sig - 1.f) * pw * invB : sig * invB;  // dL_thr/do
    float dd = 0.f;
    if (a.use_threshold) dd -= dlo;
    float hinge = 0.f;
    if (a.caffe_margin != 0.f) {
        if (is_pos) dd += 0.5f * pw * invB;
        else {
            hinge = fmaxf(0.f, a.caffe_margin - d);
            if (d < a.caffe_margin) dd -= 0.5f * invB;
        }
    } else if (a.lambda_m != 0.f) {
        if (is_pos) dd += pw * a.lambda_m * invB;
    }
    if (!valid) dd = 0.f;

#ifdef ABL_MID_NOBWD
    if (a.train) { if (tid == 0) a.rowqf[blockIdx.x] = dd; return; }
#endif
    if (blockIdx.x == 0 && tid == 0) a.thr_copy[0] = thr;
    if (blockIdx.x == 0 && a.zero_i)
        for (int i = tid; i < a.nzero; i += blockDim.x) a.zero_i[i] = 0;
    // per-row loss quantities: lane p writes quantity #p of its row (one fragment tile,
    // summed over rows by grad_red_block)
    {
        const bool pos = valid && is_pos, neg = valid && !is_pos;
        float qv = 0.f;
        switch (p) {
            case P_BCE_POS: qv = pos ? bce : 0.f; break;
            case P_BCE_NEG: qv = neg ? bce : 0.f; break;
            case P_OK_POS: qv = (pos && o > 0.f) ? 1.f : 0.f; break;
            case P_OK_NEG: qv = (neg && o <= 0.f) ? 1.f : 0.f; break;
            case P_D_POS: qv = pos ? d : 0.f; break;
            case P_D_NEG: qv = neg ? d : 0.f; break;
            case P_O_POS: qv = pos ? o : 0.f; break;
            case P_O_NEG: qv = neg ? o : 0.f; break;
            case P_DTHR: qv = valid ? dlo : 0.f; break;
            case P_HINGE_NEG: qv = neg ? hinge : 0.f; break;
            case P_SQRT_POS: qv = pos ? fsqrt(d + 1e-7f) : 0.f; break;
            case P_SQRT_NEG: qv = neg ? fsqrt(d + 1e-7f) : 0.f; break;
            default: break;
        }
        a.rowqf[frag_off(r, p, RG)] = qv;
    }

    // ---- phase 2.4: backward to dL/dY (fragment-major, unscaled) -------------------
    auto emit = [&](const MidSide &sx, const float *A, const float *X, int k, int li, float dA,
                    float extra_dy) {
        const int c = k * L + p + 16 * li;
        const int slot = (k * Lq + li) * 64 + tid;
        float dy = dA * act_grad(A[slot], a.act) + extra_dy;
        if (!valid) dy = 0.f;
        const size_t o_ = frag_off(r, c, RG);
        sx.dyf[o_] = dy;
        if (sx.cwf) sx.cwf[o_] = valid ? dy * X[slot] : 0.f;
    };

    if (a.dist_type == CFL_DIST_PCD) {
        if (K > 1) {
            float qbar = 0.f;
            for (int k = 0; k < K; ++k) {
                float qk = -2.f * sum_p(Kq[k * 64 + tid]);
                Kq[k * 64 + tid] = qk;
                qbar = fmaf(Ks[k * 64 + tid], qk, qbar);
            }
            for (int k = 0; k < K; ++k)  // dl_k = s_k (q_k - qbar)
                Kq[k * 64 + tid] = Ks[k * 64 + tid] * (Kq[k * 64 + tid] - qbar);
            for (int li = 0; li < myL; ++li) {
                const float v = Ad[li * 64 + tid], rl = Rl[li * 64 + tid];
                float dv = 2.f * rl;
                for (int k = 0; k < K; ++k) {
                    const float vmP = v - As[(k * Lq + li) * 64 + tid];
                    const float dl = Kq[k * 64 + tid];
                    dv = fmaf(-2.f * dl, vmP, dv);
                    const float dP = -2.f * Ks[k * 64 + tid] * rl + 2.f * dl * vmP;
                    emit(ss, As, Xs, k, li, dP * dd, 0.f);
                }
                emit(sd, Ad, Xd, 0, li, dv * dd, 0.f);
            }
        } else {
            for (int li = 0; li < myL; ++li) {
                const float df = Ad[li * 64 + tid] - As[li * 64 + tid];
                emit(ss, As, Xs, 0, li, -2.f * df * dd, 0.f);
                emit(sd, Ad, Xd, 0, li, 2.f * df * dd, 0.f);
            }
        }
    } else if (a.dist_type == CFL_DIST_MONOMER) {
        // du_k = w_k (e_k - d) dd ; gate-head rows for dVm, dgm
        for (int k = 0; k < K; ++k) {
            float du = Ks[k * 64 + tid] * (Ke[k * 64 + tid] - d) * dd;
            float scm = 1.f;
            if (a.weight_norm) scm = a.mono_g[k] / sqrtf(a.mono_n2[k]);
            if (p == 0) {
                a.mono_du[(size_t)r * a.kpad + k] = du * scm;
                if (a.weight_norm) a.mono_duc[(size_t)r * a.kpad + k] = valid ? du * Ku[k * 64 + tid] : 0.f;
            }
            Kq[k * 64 + tid] = du * scm;  // grad wrt raw ya.Vm
        }
        for (int li = 0; li < myL; ++li) {
            const int l = p + 16 * li;
            const float av = As[li * 64 + tid], ya = Rl[li * 64 + tid];
            a.mono_ya[(size_t)r * a.lpad + l] = valid ? ya : 0.f;
            float da = 0.f, ex = 0.f;
            for (int k = 0; k < K; ++k) {
                const float amP = av - Ad[(k * Lq + li) * 64 + tid];
                const float w = Ks[k * 64 + tid];
                da = fmaf(2.f * w, amP, da);
                emit(sd, Ad, Xd, k, li, -2.f * w * amP * dd, 0.f);
                ex = fmaf(Kq[k * 64 + tid], MW[l * a.kpad + k], ex);
            }
            emit(ss, As, Xs, 0, li, da * dd, ex);
        }
    } else {
        for (int li = 0; li < myL; ++li) {
            const float df = As[li * 64 + tid] - Ad[li * 64 + tid];
            emit(ss, As, Xs, 0, li, 2.f * df * dd, 0.f);
            emit(sd, Ad, Xd, 0, li, -2.f * df * dd, 0.f);
        }
    }
    // zero the padding columns of dYf (read by the grad GEMM and the column sums)
    for (int side = 0; side < 2; ++side) {
        const MidSide &sx = a.side[side];
        for (int c = sx.n + p; c < sx.npad; c += 16) {
            sx.dyf[frag_off(r, c, RG)] = 0.f;
            if (sx.cwf) sx.cwf[frag_off(r, c, RG)] = 0.f;
        }
    }
}


// ---------------------------------------------------------------------------
// mid, wave-per-row form for PCD with at most 64 padded columns per side (the
// Monomer / Polyvore / dyadic pcd shapes): one wave owns one pair row, lane c owns
// source column c = k*L + l (and, for c < L, destination column c).  The per-wave
// instruction count -- which IS the latency of this one-wave-per-row kernel -- drops
// ~4x against the 16-lanes-per-row form: slice sums are one coalesced 256-byte load
// per slice, every per-column quantity is one VALU op, and the few cross-column sums
// go through a 1 KiB wave-private LDS scratch.  Same arithmetic, same outputs.
// ---------------------------------------------------------------------------
// `lead`: the one wave of the launch that also clears the hand-off flags of the weight-gradient launch and snapshots the threshold
template <int J>   // J = columns per lane: sides of up to 64 * J padded columns
__device__ __forceinline__ void mid_row_body(const MidArgs &a, int r, bool lead, float *W) {
    const int lane = threadIdx.x & 63;
    constexpr int CW = 64 * J;
    float *Pl = W, *Vl = W + CW, *Rl = W + 2 * CW, *T = W + 3 * CW, *S = W + 4 * CW, *Q = W + 5 * CW;
    RSTAMP(0);
    const bool extra = a.xn > 0 && r >= a.xrow0;
    const bool valid = extra ? r - a.xrow0 < a.xn : r < a.R;
    const int L = a.L, K = a.K, RG = a.Rpad >> 4;
    const MidSide &ss = a.side[0], &sd = a.side[1];
    const int ns = ss.n;
    int c[J], kk[J], ll[J];
    bool cs[J], cd[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        c[j] = lane + 64 * j;
        cs[j] = c[j] < ns;
        cd[j] = c[j] < L;
        kk[j] = cs[j] ? c[j] / L : 0;
        ll[j] = cs[j] ? c[j] - kk[j] * L : 0;
    }

    // ---- head parameters + slice sums: EVERY load of this phase is unconditional and issued in one batch ---------------
    // This one-wave-per-row kernel IS its latency chain.  Round 4 (tools/mid_stamp_probe.py + the ISA): written with
    // `cond ? ptr[i] : const` / `s < S ? slab[s] : 0` the loads sat inside uniform branches, and hipcc's waitcnt pass
    // drains the queue (s_waitcnt vmcnt(0)) at every such join -- the 16 + 16 slab loads went out one round trip after
    // the other (3.2 us of the 6.7 us wave lifetime at the headline shape: 7700 -> 4100 cycles with this form).  Now:
    // absent arrays point at a dummy word and their values are replaced by selects, and the slab count is a template
    // parameter of the loader (switch on S BEFORE anything is in flight), so the compiler sees straight-line loads.
    const float thr_raw = *a.thr;
    float scs[J], scd[J], bs[J], bd[J], ys[J], yd[J];
    {
        const bool wn = a.weight_norm != 0;
        const float *gsp = wn ? ss.g : a.thr, *nsp = wn ? ss.n2 : a.thr, *gdp = wn ? sd.g : a.thr, *ndp = wn ? sd.n2 : a.thr;
        const float *bsp = ss.b ? ss.b : a.thr, *bdp = sd.b ? sd.b : a.thr;
        const bool hbs = ss.b != nullptr, hbd = sd.b != nullptr;
        float gs[J], ns[J], gd[J], nd[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int ks = cs[j] ? c[j] : 0, kd = cd[j] ? c[j] : 0;   // clamped: unconditional loads
            gs[j] = gsp[wn ? ks : 0];
            ns[j] = nsp[wn ? ks : 0];
            gd[j] = gdp[wn ? kd : 0];
            nd[j] = ndp[wn ? kd : 0];
            bs[j] = bsp[hbs ? ks : 0];
            bd[j] = bdp[hbd ? kd : 0];
        }
        const float *srcs[J], *srcd[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int ccs = c[j] < ss.npad ? c[j] : 0, ccd = c[j] < sd.npad ? c[j] : 0;
            srcs[j] = ss.ypart + (size_t)r * ss.npad + ccs;
            srcd[j] = sd.ypart + (size_t)r * sd.npad + ccd;
        }
        // all slice loads of both sides are independent and in flight together; summed in slice order afterwards
        // (the same order of additions as before: s = 0, 1, ..., S - 1)
        auto slabs = [&](auto ns_c) {
            constexpr int NS = decltype(ns_c)::value;
            float ts[J][NS], td[J][NS];
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    ts[j][sl] = srcs[j][(size_t)sl * ss.sstride];
                    td[j][sl] = srcd[j][(size_t)sl * sd.sstride];
                }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                ys[j] = 0.f;
                yd[j] = 0.f;
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) { ys[j] += ts[j][sl]; yd[j] += td[j][sl]; }
            }
        };
        switch (a.S) {   // (uniform; the d split is a power of two up to 16, anything else from CFL_DEBUG_S: generic tail)
            case 1: slabs(std::integral_constant<int, 1>()); break;
            case 2: slabs(std::integral_constant<int, 2>()); break;
            case 4: slabs(std::integral_constant<int, 4>()); break;
            case 8: slabs(std::integral_constant<int, 8>()); break;
            case 16: slabs(std::integral_constant<int, 16>()); break;
            default: {
                // any other split: clamped slice indices (re-reads of the last slice are masked out), still branch-free
                const int S1 = a.S - 1;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    float accs = 0.f, accd = 0.f, ts[16], td[16];
#pragma unroll
                    for (int sl = 0; sl < 16; ++sl) {
                        const int sc = sl < a.S ? sl : S1;
                        ts[sl] = srcs[j][(size_t)sc * ss.sstride];
                        td[sl] = srcd[j][(size_t)sc * sd.sstride];
                    }
#pragma unroll
                    for (int sl = 0; sl < 16; ++sl) { accs += sl < a.S ? ts[sl] : 0.f; accd += sl < a.S ? td[sl] : 0.f; }
                    ys[j] = accs;
                    yd[j] = accd;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            scs[j] = (wn && cs[j]) ? gs[j] * __builtin_amdgcn_rsqf(ns[j]) : 1.f;
            scd[j] = (wn && cd[j]) ? gd[j] * __builtin_amdgcn_rsqf(nd[j]) : 1.f;
            bs[j] = (hbs && cs[j]) ? bs[j] : 0.f;
            bd[j] = (hbd && cd[j]) ? bd[j] : 0.f;
        }
    }
    RSTAMP(1);   // slabs summed: the loads have landed
    float xvs[J], xvd[J], P[J], v[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        xvs[j] = (valid && cs[j]) ? ys[j] * a.in_mul : 0.f;
        xvd[j] = (valid && cd[j]) ? yd[j] * a.in_mul : 0.f;
        P[j] = (valid && cs[j]) ? act_fn(xvs[j] * scs[j] + bs[j], a.act) : 0.f;
        v[j] = (valid && cd[j]) ? act_fn(xvd[j] * scd[j] + bd[j], a.act) : 0.f;
        Pl[c[j]] = P[j];
        Vl[c[j]] = v[j];
    }
    float diff[J];
#pragma unroll
    for (int j = 0; j < J; ++j) diff[j] = cs[j] ? Vl[ll[j]] - P[j] : 0.f;   // destination coordinate l of this column

    // ---- distance ----------------------------------------------------------------
    float d, sk[J], rl[J], dlk[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { sk[j] = 1.f; rl[j] = 0.f; dlk[j] = 0.f; }
    // K <= MID_KU prototypes (round 5): the per-prototype sums e_k = sum_l (v_l - P_kl)^2 and q_k = -2 sum_l r_l P_kl are MASKED
    // WAVE SUMS (four DPP adds + four v_readlane each, the K of them independent) whose results are wave-uniform: logits, soft-min
    // weights s_k, q-bar and dl_k then live in scalar registers and every per-column use is a select -- no LDS at all in the
    // soft-min.  The form below it (any K) lets lane k walk its segment of L values through LDS: two serial chains of L
    // dependent ds_read + add (2 x ~2000 cycles at L = 20) plus K-long LDS broadcast loops, half of this one-wave-per-row
    // kernel's lifetime (tools/mid_stamp_probe.py: distance phase 4270 of 11870 cycles).  Same formulas; the sums add in
    // butterfly order instead of l = 0, 1, ..., L - 1.
    constexpr int MID_KU = 8;
    float su[MID_KU], dlu[MID_KU];      // uniform: s_k, dl_k
#pragma unroll
    for (int k = 0; k < MID_KU; ++k) { su[k] = 0.f; dlu[k] = 0.f; }
    const bool ku = K > 1 && K <= MID_KU;
    if (ku) {
        float eu[MID_KU];
#pragma unroll
        for (int k = 0; k < MID_KU; ++k) {
            eu[k] = 0.f;
            if (k < K) {
                float part = 0.f;
#pragma unroll
                for (int j = 0; j < J; ++j) part += (cs[j] && kk[j] == k) ? diff[j] * diff[j] : 0.f;
                eu[k] = wave_sum_dpp(part);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < MID_KU; ++k) if (k < K) mx = fmaxf(mx, -eu[k]);
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < MID_KU; ++k) if (k < K) den += fexp(-eu[k] - mx);
        const float inv = frcp(den);
#pragma unroll
        for (int k = 0; k < MID_KU; ++k) if (k < K) su[k] = fexp(-eu[k] - mx) * inv;
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float s0 = su[0];
#pragma unroll
            for (int k = 1; k < MID_KU; ++k) s0 = kk[j] == k ? su[k] : s0;
            sk[j] = s0;
            float m = 0.f;
            if (cd[j]) {
#pragma unroll
                for (int k = 0; k < MID_KU; ++k) if (k < K) m = fmaf(su[k], Pl[k * L + c[j]], m);
            }
            rl[j] = cd[j] ? v[j] - m : 0.f;
            Rl[c[j]] = rl[j];
            dsum = fmaf(rl[j], rl[j], dsum);
        }
        d = wave_sum_dpp(dsum);
        float t2[J];
#pragma unroll
        for (int j = 0; j < J; ++j) t2[j] = cs[j] ? Rl[ll[j]] * P[j] : 0.f;
        float qu[MID_KU], qbar = 0.f;
#pragma unroll
        for (int k = 0; k < MID_KU; ++k) {
            qu[k] = 0.f;
            if (k < K) {
                float part = 0.f;
#pragma unroll
                for (int j = 0; j < J; ++j) part += kk[j] == k ? t2[j] : 0.f;
                qu[k] = -2.f * wave_sum_dpp(part);
            }
        }
#pragma unroll
        for (int k = 0; k < MID_KU; ++k) if (k < K) qbar = fmaf(su[k], qu[k], qbar);
#pragma unroll
        for (int k = 0; k < MID_KU; ++k) if (k < K) dlu[k] = su[k] * (qu[k] - qbar);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float x0 = dlu[0];
#pragma unroll
            for (int k = 1; k < MID_KU; ++k) x0 = kk[j] == k ? dlu[k] : x0;
            dlk[j] = x0;
        }
    } else if (K > 1) {
#pragma unroll
        for (int j = 0; j < J; ++j) T[c[j]] = diff[j] * diff[j];
        // (measured, round 4: the segment sums with LDS reads batched eight at a time -- clamped indices, masked values, all
        // lanes -- and the K-loops on v_readlane instead of LDS broadcasts: this phase 4170 -> 4800 cycles at the headline
        // shape.  Three lanes reading a segment each is cheap; sixty-four reading strided segments is not.  Left as it was.)
        float e = 0.f;
        if (lane < K)
            for (int i = 0; i < L; ++i) e += T[lane * L + i];
        S[lane] = -e;                 // logits (lanes >= K: unused)
        float mx = -INFINITY;
        for (int i = 0; i < K; ++i) mx = fmaxf(mx, S[i]);
        float den = 0.f;
        for (int i = 0; i < K; ++i) den += fexp(S[i] - mx);
        const float inv = frcp(den);
        const float sme = lane < K ? fexp(-e - mx) * inv : 0.f;   // s_k for lanes < K
        Q[lane] = sme;
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            sk[j] = Q[kk[j]];         // softmax weight of this column's prototype
            float m = 0.f;
            if (cd[j])
                for (int i = 0; i < K; ++i) m = fmaf(Q[i], Pl[i * L + c[j]], m);
            rl[j] = cd[j] ? v[j] - m : 0.f;
            Rl[c[j]] = rl[j];
            dsum = fmaf(rl[j], rl[j], dsum);
        }
        d = wave_sum_dpp(dsum);
#pragma unroll
        for (int j = 0; j < J; ++j) T[c[j]] = cs[j] ? Rl[ll[j]] * P[j] : 0.f;
        float q = 0.f;
        if (lane < K)
            for (int i = 0; i < L; ++i) q += T[lane * L + i];
        q *= -2.f;
        S[lane] = q;                  // q_k (lanes < K)
        float qbar = 0.f;
        for (int i = 0; i < K; ++i) qbar = fmaf(Q[i], S[i], qbar);
        const float dl_me = lane < K ? sme * (q - qbar) : 0.f;
        T[lane] = dl_me;              // dl_k (lanes < K); T is free again, q has been reduced
#pragma unroll
        for (int j = 0; j < J; ++j) dlk[j] = T[kk[j]];
    } else {
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) dsum = fmaf(diff[j], diff[j], dsum);
        d = wave_sum_dpp(dsum);
    }

    // ---- threshold, loss, dL/dd ------------------------------------------------------
    RSTAMP(2);   // distance done
    const float thr = fmaxf(thr_raw, CFL_THR_FLOOR);
    const float o = thr - d;
    if (!a.train || extra) {
        if (valid && lane == 0) {
            if (extra) {
                a.xscores[r - a.xrow0] = o;
            } else {
                a.scores[r] = o;
                if (a.dists) a.dists[r] = d;
            }
        }
        return;
    }
    const bool is_pos = r < a.B;
    const float invB = 1.f / (float)a.B;
    const float pw = a.pos_weight != 0.f ? a.pos_weight : 1.f;
    const float eo = fexp(-fabsf(o));
    const float sp = __logf(1.f + eo);
    const float bce = fmaxf(o, 0.f) - (is_pos ? o : 0.f) + sp;
    const float sig = (o >= 0.f ? 1.f : eo) * frcp(1.f + eo);
    const float dlo = is_pos ? (sig - 1.f) * pw * invB : sig * invB;
    float dd = 0.f;
    if (a.use_threshold) dd -= dlo;
    float hinge = 0.f;
    if (a.caffe_margin != 0.f) {
        if (is_pos) dd += 0.5f * pw * invB;
        else {
            hinge = fmaxf(0.f, a.caffe_margin - d);
            if (d < a.caffe_margin) dd -= 0.5f * invB;
        }
    } else if (a.lambda_m != 0.f) {
        if (is_pos) dd += pw * a.lambda_m * invB;
    }
    if (!valid) dd = 0.f;
    auto put = [&](float *q, float val) { *q = val; };
    if (lead && lane == 0) put(a.thr_copy, thr);
    if (lead && a.zero_i)
        for (int i = lane; i < a.nzero; i += 64) a.zero_i[i] = 0;
    if (lane < 16) {
        const bool pos = valid && is_pos, neg = valid && !is_pos;
        float qv = 0.f;
        switch (lane) {
            case P_BCE_POS: qv = pos ? bce : 0.f; break;
            case P_BCE_NEG: qv = neg ? bce : 0.f; break;
            case P_OK_POS: qv = (pos && o > 0.f) ? 1.f : 0.f; break;
            case P_OK_NEG: qv = (neg && o <= 0.f) ? 1.f : 0.f; break;
            case P_D_POS: qv = pos ? d : 0.f; break;
            case P_D_NEG: qv = neg ? d : 0.f; break;
            case P_O_POS: qv = pos ? o : 0.f; break;
            case P_O_NEG: qv = neg ? o : 0.f; break;
            case P_DTHR: qv = valid ? dlo : 0.f; break;
            case P_HINGE_NEG: qv = neg ? hinge : 0.f; break;
            case P_SQRT_POS: qv = pos ? fsqrt(d + 1e-7f) : 0.f; break;
            case P_SQRT_NEG: qv = neg ? fsqrt(d + 1e-7f) : 0.f; break;
            default: break;
        }
        put(a.rowqf + frag_off(r, lane, RG), qv);
    }

    RSTAMP(3);   // loss + row quantities stored
    // ---- backward: J source columns and (columns < L) J destination columns per lane ----
#pragma unroll
    for (int j = 0; j < J; ++j) {
        float dP, dv;
        if (K > 1) {
            const float rls = Rl[ll[j]];
            dP = -2.f * sk[j] * rls + 2.f * dlk[j] * diff[j];
            dv = 2.f * rl[j];
            if (cd[j] && ku) {
#pragma unroll
                for (int k = 0; k < MID_KU; ++k) if (k < K) dv = fmaf(-2.f * dlu[k], v[j] - Pl[k * L + c[j]], dv);
            } else if (cd[j]) {
                for (int i = 0; i < K; ++i) dv = fmaf(-2.f * T[i], v[j] - Pl[i * L + c[j]], dv);
            }
        } else {
            dP = -2.f * diff[j];
            dv = 2.f * (v[j] - Pl[c[j]]);
        }
        if (c[j] < ss.npad) {
            const float dy = cs[j] ? dP * dd * act_grad(P[j], a.act) : 0.f;
            const size_t o_ = frag_off(r, c[j], RG);
            put(ss.dyf + o_, dy);
            if (ss.cwf) put(ss.cwf + o_, dy * xvs[j]);
        }
        if (c[j] < sd.npad) {
            const float dy = cd[j] ? dv * dd * act_grad(v[j], a.act) : 0.f;
            const size_t o_ = frag_off(r, c[j], RG);
            put(sd.dyf + o_, dy);
            if (sd.cwf) put(sd.cwf + o_, dy * xvd[j]);
        }
    }
    RSTAMP(4);   // stores issued
#ifdef CFL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RSTAMP(5);   // stores acknowledged
#endif
}

template <int J>
__global__ __launch_bounds__(256) void cfl_mid_row_kernel(MidArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((int)blockIdx.x >= a.nrb + a.nxb) {
        if (wave == 0) mid_reg_block(a, blockIdx.x - a.nrb - a.nxb);
        return;
    }
    if ((int)blockIdx.x >= a.nrb) {   // extra scoring rows (forward only)
        mid_row_body<J>(a, a.xrow0 + ((int)blockIdx.x - a.nrb) * 4 + wave, false, (float *)smem + wave * 6 * 64 * J);
        return;
    }
    // (Round 5 measured the other placement: row tiles dealt over the XCDs by proj -- all d slices and column jobs of a 32-row
    // tile on ONE XCD -- and the blocks here taking the rows whose slabs their own XCD's L2 still holds.  mid -0.2 us, proj
    // +0.6, grad +0.5: the slab loads are not what this launch waits for.  profiles/r05_mid_xcd_ab.txt; not kept)
    mid_row_body<J>(a, blockIdx.x * 4 + wave, blockIdx.x == 0 && wave == 0, (float *)smem + wave * 6 * 64 * J);
}


