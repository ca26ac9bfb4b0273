/*
 * cfl_hip.h -- C ABI of libcfl_hip.so: the MI355X (gfx950) implementation of the
 * cfl pair-distance ("triplet") training / scoring hot path.
 *
 * The reference (appier/compatibility-family-learning) has no FFI: this path is
 * a TensorFlow-1 graph region executed by `sess.run`.  Every entry point below
 * replaces the graph region named in its comment (paths relative to the
 * reference tree).  A maintainer of the reference binds these with ctypes (see
 * INTEGRATION.md); the in-tree binding is
 * compatibility-family-learning_amd/cfl/hipabi.py.
 *
 * Rules of the boundary
 *   - extern "C", plain pointers and sizes; no torch / TF types.
 *   - every pointer marked `dev` is a DEVICE pointer owned by the caller; the
 *     library never allocates persistent device memory and never frees caller
 *     memory.  Scratch comes from a caller-provided workspace whose size is
 *     given by cfl_workspace_bytes().
 *   - every call is asynchronous on the given HIP stream (pass
 *     torch.cuda.current_stream().cuda_stream); no implicit synchronisation;
 *     safe to capture into a hipGraph.
 *   - return 0 on success; <0 on error (CFL_E_*), message via cfl_last_error()
 *     (thread-local).  Never aborts or throws across the ABI.
 *
 * Device layout of the parameters ("theta"), its Adam slots m / v and the flat
 * gradient: ONE contiguous fp32 array each, laid out by cfl_layout().  Weight
 * matrices are column-padded to Npad = N rounded up to 16 (pad columns zero) and
 * stored FRAGMENT-MAJOR, Wf[nt][g][q][c16][e] = W[d = 16g+4q+e][col = 16nt+c16]:
 * one contiguous 1 KiB block per (16-column tile nt, 16-row group g), in exactly
 * the order the v_mfma_f32_16x16x4_f32 B operand consumes it, so that both the
 * projection and the weight-gradient kernels move whole 1 KiB blocks per wave
 * instruction.  cfl/hipabi.py pack_theta()/unpack_theta() convert to and from
 * the reference's [D, N] arrays.
 */
#ifndef CFL_HIP_H
#define CFL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFL_ABI_VERSION 1

/* error codes */
#define CFL_OK 0
#define CFL_E_SHAPE (-1)       /* bad / inconsistent shape or NULL pointer      */
#define CFL_E_HIP (-2)         /* a HIP runtime call failed                     */
#define CFL_E_UNSUPPORTED (-3) /* valid in the reference, not built yet         */
#define CFL_E_WORKSPACE (-4)   /* workspace too small                           */

/* dist_type: cfl/models/base.py:107-146 (== cfl/models/dist.py:70-89) */
#define CFL_DIST_PCD 0
#define CFL_DIST_MONOMER 1
#define CFL_DIST_SIAMESE 2

/* act_type: cfl/models/cfl.py:579-586 */
#define CFL_ACT_LINEAR 0
#define CFL_ACT_SIGMOID 1
#define CFL_ACT_TANH 2
#define CFL_ACT_RELU 3

typedef void *cfl_stream_t; /* hipStream_t */

/* Shape / flag set of the distance model.
 *   weight_norm = 0, has_bias = 1 : FCEncoder, cfl/models/dist.py:12-68
 *   weight_norm = 1               : FCPCD + DistBase.build_prototypes,
 *                                   cfl/models/blocks.py:477-527,
 *                                   cfl/models/base.py:43-105 (cfl/layers.py:28-97)
 *   directed = 1                  : separate DistEncoderSrc / DistEncoderDst,
 *                                   cfl/models/cfl.py:676-681 */
typedef struct {
    int32_t D;           /* input size; must be a multiple of 64                */
    int32_t L;           /* latent_size (num_outputs)                           */
    int32_t K;           /* num_components                                      */
    int32_t dist_type;   /* CFL_DIST_*                                          */
    int32_t weight_norm; /* 0 / 1                                               */
    int32_t has_bias;    /* 0 / 1                                               */
    int32_t act_type;    /* CFL_ACT_*                                           */
    int32_t directed;    /* 0 / 1                                               */
} CflShape;

/* One linear head inside theta (offsets in floats, -1 = absent).               */
typedef struct {
    int64_t w;     /* Wf, npad*D floats (monomer gate head: V[L][kpad] row-major) */
    int64_t b;     /* biases[npad]                                              */
    int64_t g;     /* weight-norm gains g[npad]                                 */
    int32_t n;     /* logical columns                                           */
    int32_t npad;  /* n rounded up to 16                                        */
} CflHead;

/* Offsets of every variable of the model inside theta / m / v / grad.
 * enc[0] = DistEncoder (or DistEncoderSrc), enc[1] = DistEncoderDst when
 * directed (otherwise a copy of enc[0]).  Names: SURVEY.md App. D.            */
typedef struct {
    struct {
        CflHead outputs; /* 'outputs' / 'latent_outputs' head, N = L            */
        CflHead proto;   /* 'prototype_outputs' / 'pcd_outputs', N = K*L        */
        CflHead mono;    /* 'monomer_outputs', input L, N = K                   */
    } enc[2];
    int64_t thr;         /* Thresholder raw threshold (cfl/models/blocks.py:18) */
    int64_t total;       /* floats in theta (multiple of 64)                    */
} CflLayout;

/* Input normalisation applied to every input vector before the heads.
 *   x_hat = clip(x * mul + add, lo, hi)
 * cfl/ops.py:198-202 (mul = 1/normalize_value, add = shift) and the scalar
 * branch of cfl/ops.py:66-124 (mul = scale/norm, add = -mean/norm).  When
 * add == 0 and no clip is requested the scale is folded into the projection
 * epilogue instead of touching every element. */
typedef struct {
    float mul;
    float add;
    float lo;        /* used iff has_lo */
    float hi;        /* used iff has_hi */
    int32_t has_lo;
    int32_t has_hi;
} CflNorm;

/* Loss of cfl/models/cfl.py:868-949 / cfl/models/dist.py:253-284.             */
typedef struct {
    int32_t use_threshold; /* add the threshold BCE to the encoder loss         */
    float pos_weight;      /* 0 = unset (weight 1)                              */
    float caffe_margin;    /* 0 = off; contrastive hinge, cfl.py:912-921        */
    float lambda_m;        /* 0 = off; pull term, cfl.py:922-929                */
    float reg_const;       /* L2 regulariser scale, cfl/models/base.py:16-19    */
} CflLossCfg;

/* Indices into the `scalars` output of cfl_pair_step_fwd_bwd (device floats).  */
enum {
    CFL_S_TOTAL = 0,   /* s_total_loss                                          */
    CFL_S_REG,         /* s_loss_reg                                            */
    CFL_S_THRES,       /* s_thres_loss  ([pw*]pos + neg)                        */
    CFL_S_LOSS_POS,    /* s_p_loss_pos                                          */
    CFL_S_LOSS_NEG,    /* s_p_loss_neg                                          */
    CFL_S_CD,          /* s_cd_loss                                             */
    CFL_S_ACCURACY,    /* s_accuracy                                            */
    CFL_S_MEAN_D_POS,  /* mean(s_pos_dists)                                     */
    CFL_S_MEAN_D_NEG,  /* mean(s_neg_dists)                                     */
    CFL_S_MEAN_O_POS,  /* mean(s_pos_predicts.outputs)                          */
    CFL_S_MEAN_O_NEG,  /* mean(s_neg_predicts.outputs)                          */
    CFL_S_THRESHOLD,   /* max(raw_threshold, 1e-6)                              */
    CFL_S_DIST_ADAPT_POS, /* mean(sqrt(d_pos + 1e-7)), cfl.py:897-898           */
    CFL_S_DIST_ADAPT_NEG, /* mean(sqrt(d_neg + 1e-7)), cfl.py:899-900           */
    CFL_S_COUNT = 16
};

int cfl_version(void);
const char *cfl_last_error(void);

/* Fill `out` with the offsets of every variable for `shape`.                   */
int cfl_layout(const CflShape *shape, CflLayout *out);

/* Bytes of scratch a call with `rows` pairs per group and `groups` pair groups
 * (2 for a training row batch: pos + neg; 1 for scoring) needs.               */
size_t cfl_workspace_bytes(const CflShape *shape, int64_t rows, int32_t groups);

/* score[i] = max(thr, 1e-6) - dist(src_i, dst_i), i < n.
 * Replaces `sess.run(model.val_s_pos_predicts.outputs, feed_dict=...)` in
 * cfl/utils.py:245,262,293,308 (dist_eval / dist_predict).
 *   xs, xt : dev [n, D] row-major fp32; theta : dev, cfl_layout();
 *   scores : dev [n];  dists : dev [n] or NULL.                                */
int cfl_pair_scores(const CflShape *shape, const CflNorm *norm,
                    const float *xs, const float *xt, int64_t n,
                    const float *theta, float *scores, float *dists,
                    void *workspace, size_t workspace_bytes, cfl_stream_t stream);

/* Forward + backward of one training row batch: the part of
 * `sess.run([summary, [s_optim], s_accuracy, ...])` (cfl/bin/train_dist.py:81-82,
 * cfl/models/cfl.py:1399-1414) that precedes the Adam apply.
 *   x4      : 4 dev pointers [B, D]: pos_src, pos_dst, neg_src, neg_dst
 *             (cfl/input_data.py:585-589)
 *   grad    : dev, same layout as theta; fully overwritten.  The threshold slot
 *             holds d s_total_loss/d thr when use_threshold, else
 *             d s_thres_loss/d thr (the separate th_optim, cfl.py:1076-1079).
 *   scalars : dev [CFL_S_COUNT]                                                */
int cfl_pair_step_fwd_bwd(const CflShape *shape, const CflNorm *norm,
                          const CflLossCfg *loss, const float *const x4[4],
                          int64_t B, const float *theta, float *grad,
                          float *scalars, void *workspace,
                          size_t workspace_bytes, cfl_stream_t stream);

/* Single-GPU fast path: cfl_pair_step_fwd_bwd with the TF-Adam apply fused into
 * the last kernel (theta, m, v updated in place; grad still written).  Equals
 * cfl_pair_step_fwd_bwd followed by cfl_adam_tf(..., grad_scale = 1); replaces the
 * whole `sess.run([s_optim, ...])` of cfl/bin/train_dist.py:81-82.               */
int cfl_pair_train_step(const CflShape *shape, const CflNorm *norm,
                        const CflLossCfg *loss, const float *const x4[4], int64_t B,
                        float *theta, float *m, float *v, float *grad, float *scalars,
                        float lr_t, float beta1, float beta2, float eps,
                        void *workspace, size_t workspace_bytes, cfl_stream_t stream);

/* TF-1.x AdamOptimizer apply over a flat array:
 *   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; theta -= lr_t m / (sqrt(v)+eps)
 * Replaces tf.train.AdamOptimizer(...).minimize at cfl/models/dist.py:291-293,
 * cfl/models/cfl.py:1077-1085.  `grad_scale` multiplies g first (1/world_size
 * after a sum all-reduce).                                                     */
int cfl_adam_tf(float *theta, float *m, float *v, const float *grad, int64_t n,
                float lr_t, float beta1, float beta2, float eps,
                float grad_scale, cfl_stream_t stream);

/* Gather rows of a resident feature table into a dense batch:
 * out[i, :] = table[idx[i], :].  Replaces the per-row seek+read loop of
 * cfl/input_data.py:212-228 when features.b is kept resident in HBM.           */
int cfl_gather_rows(const float *table, const int64_t *idx, int64_t n, int64_t D,
                    float *out, cfl_stream_t stream);

/* Gradient of the step's loss with respect to the (normalised) input rows, for callers
 * whose pair rows are not leaves (the conv encoder ConvPCD, cfl/models/blocks.py:530-590).
 * Must follow cfl_pair_step_fwd_bwd / cfl_pair_train_step on the SAME shape, B and
 * workspace (it reads dL/dY and the weight-norm scales the step left there; with the
 * fused train step theta is already updated, so use cfl_pair_step_fwd_bwd before it).
 *   dx_src, dx_dst : dev [2B, D] row-major: rows [0,B) = positive pairs, [B,2B) = negative */
int cfl_pair_input_grad(const CflShape *shape, const CflNorm *norm, int64_t B, const float *theta,
                        const void *workspace, size_t workspace_bytes, float *dx_src,
                        float *dx_dst, cfl_stream_t stream);

/* Weight-normalised 2-D convolution, NHWC activations, HWIO filters, TensorFlow 'SAME'
 * padding: y = act(conv(x, g * V / sqrt(max(sum_{h,w,i} V^2, 1e-12)), stride) + b).
 * Replaces conv2d_weight_norm, cfl/layers.py:100-187 (act: 0 none, 1 lrelu of
 * cfl/ops.py:10-12, 2 relu).  The backward returns d/dx (nullable), d/dV (with the
 * weight-norm correction and + reg_const * V), d/dg, d/db (nullable) given the layer's
 * POST-activation output y and dL/dy.                                            */
typedef struct {
    int32_t B, H, W, Ci, Co, KH, KW, stride, act;
} CflConv;
size_t cfl_conv_workspace_bytes(const CflConv *conv);
int cfl_conv2d_wn_fwd(const CflConv *conv, const float *x, const float *V, const float *g,
                      const float *b, float *y, void *workspace, size_t workspace_bytes,
                      cfl_stream_t stream);
int cfl_conv2d_wn_bwd(const CflConv *conv, const float *x, const float *V, const float *g,
                      const float *y, const float *dy, float reg_const, float *dx, float *dV,
                      float *dg, float *db, void *workspace, size_t workspace_bytes,
                      cfl_stream_t stream);

/* Optional per-kernel timing (bench.py's roofline object).  While enabled,
 * every kernel the library launches is bracketed by two HIP events recorded on
 * the caller's stream.  cfl_profile_read() synchronises those events, adds the
 * elapsed milliseconds / launch counts per kernel kind into the two arrays of
 * CFL_K_COUNT entries and clears the record list.  Not for production steps.   */
enum {
    CFL_K_COLNORM = 0, CFL_K_PROJ, CFL_K_MID, CFL_K_GRAD, CFL_K_FINALIZE,
    CFL_K_ADAM, CFL_K_GATHER, CFL_K_COUNT = 8
};
int cfl_profile_enable(int on);
int cfl_profile_read(double *ms_sum, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* CFL_HIP_H */
