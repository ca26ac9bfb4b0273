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

#define CFL_ABI_VERSION 6

/* error codes */
#define CFL_OK 0
#define CFL_E_SHAPE (-1)       /* bad / inconsistent shape or NULL pointer      */
#define CFL_E_HIP (-2)         /* a HIP runtime call failed                     */
#define CFL_E_UNSUPPORTED (-3) /* valid in the reference, not built yet         */
#define CFL_E_WORKSPACE (-4)   /* workspace too small                           */
#define CFL_E_HANDOFF (-5)     /* a kernel gave up waiting for a partner inside a launch (cfl_scalars_status) */

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
    int32_t valid_cols; /* > 0: the input rows carry only this many features, zero-padded to CflShape.D (a multiple
                         * of 64); the pad columns stay zero after the map (they must not pick up `add` or a positive
                         * `lo`, or the zero weight rows under them would start to train).  0 = every column is data */
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
    CFL_S_ERROR = 15,  /* STICKY error word, no reference counterpart: 0 = healthy.  A training kernel that gives up
                        * waiting for a partner workgroup inside its launch (bounded spins of the fused weight-gradient
                        * tail) stores 1.0 here and poisons the entries it was finishing with NaN.  The library only ever
                        * SETS this word: the caller zeroes scalars[CFL_S_ERROR] once, before its first training call, and
                        * checks it whenever it reads the scalars back (cfl_scalars_status).  Under data parallelism the
                        * word travels in the [gradient | scalars] sum, so every rank sees a failure of any rank.       */
    CFL_S_COUNT = 16
};

/* HOST-ONLY: status of a scalars array that was read back from the device: CFL_OK, or CFL_E_HANDOFF (message via
 * cfl_last_error()) when scalars[CFL_S_ERROR] != 0 -- the parameters are poisoned with NaN from that step on and must
 * not be checkpointed.  This is how "negative code on error" reaches the caller for failures that happen inside an
 * asynchronous launch (SURVEY 8(b) Errors): at the caller's next read-back, without a synchronisation of the library's own. */
int cfl_scalars_status(const float *host_scalars);

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

/* bf16 planes of theta (ABI 4; no reference counterpart).  At large row counts the projection runs on the bf16 matrix
 * cores at fp32 accuracy from three bf16 "planes" of every weight (w = h + m + l exactly); splitting the weights is a
 * launch of its own per call unless the caller keeps a plane buffer beside theta, as it keeps the Adam slots: the fused
 * training step then WRITES the planes of the updated weights from the same registers that write theta (6 bytes per
 * weight, no extra launch), and the next step's projection starts from current planes.
 *   buf    dev, cfl_theta_planes_bytes(shape) bytes, 16-byte aligned, caller-owned, one per theta
 *   valid  HOST flag: nonzero = buf holds the planes of the CURRENT theta.  The library sets it after a fused training
 *          step that wrote the planes and clears it after a training step that did not; the CALLER clears it whenever
 *          theta is changed by anything else (checkpoint load, cfl_adam_tf, a data-parallel update).
 * Layout: plane p of W[d = 32 tq + 8 (lane >> 4) + j][col = 16 nt + (lane & 15)] of the head whose Wf array starts at
 * theta offset w is the ushort at 3 w + ((nt * (D/32) + tq) * 3 + p) * 512 + lane * 8 + j -- the B operand of
 * v_mfma_f32_16x16x32_bf16, one 1 KiB block per (column tile, 32-d quarter, plane).                                   */
typedef struct {
    void *buf;
    int32_t valid;
} CflThetaPlanes;
size_t cfl_theta_planes_bytes(const CflShape *shape);

/* cfl_pair_step_fwd_bwd with a kept plane buffer (ABI 5): the forward / backward of a step whose update happens
 * elsewhere -- the data-parallel step, where the gradient exchange sits between this call and cfl_adam_tf_planes.  The
 * projection reads the caller's planes (when !planes->valid they are first split from theta into the buffer and valid is
 * set); theta is not changed, so the planes stay valid.  planes may be NULL: identical to cfl_pair_step_fwd_bwd.          */
int cfl_pair_step_fwd_bwd_planes(const CflShape *shape, const CflNorm *norm,
                                 const CflLossCfg *loss, const float *const x4[4],
                                 int64_t B, const float *theta, float *grad,
                                 float *scalars, CflThetaPlanes *planes, void *workspace,
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
/* ... with a kept plane buffer (planes may be NULL: identical to the call above)                                    */
int cfl_pair_train_step_planes(const CflShape *shape, const CflNorm *norm,
                               const CflLossCfg *loss, const float *const x4[4], int64_t B,
                               float *theta, float *m, float *v, float *grad, float *scalars,
                               float lr_t, float beta1, float beta2, float eps, CflThetaPlanes *planes,
                               void *workspace, size_t workspace_bytes, cfl_stream_t stream);

/* The same three entry points fed from a RESIDENT FEATURE TABLE instead of dense batches: row r of input stream k
 * is table[idx[k][r * idx_stride], :].  Replaces the batch assembly of cfl/input_data.py:542-589 + 212-228 (index
 * pairs -> one seek + read per vector) together with the step: features.b stays in HBM, the training loop hands
 * over the index pairs only (idx_stride = 2 walks one column of an int32 [n, 2] pair array in place), and the
 * kernels read each 4*D-byte row where it lies -- no gather pass, no batch copy.
 *   table  : dev [table_rows, D] row-major fp32, 16-byte aligned;  idx[k] : dev int32, 4-byte aligned, entries in
 *            [0, table_rows) (larger values are clamped to the last row);  stream order as x4 / (xs, xt).      */
int cfl_pair_scores_idx(const CflShape *shape, const CflNorm *norm, const float *table, int64_t table_rows,
                        const int32_t *const idx2[2], int64_t idx_stride, int64_t n, const float *theta,
                        float *scores, float *dists, void *workspace, size_t workspace_bytes,
                        cfl_stream_t stream);
/* Scores of the n (idx4[0], idx4[1]) pairs followed by the n (idx4[2], idx4[3]) pairs in ONE projection + row-math
 * launch pair: scores / dists [2 n].  The validation fetch `val_s_accuracy` of cfl/bin/train_dist.py:52-56, 81-82 scores a
 * positive and a negative batch; workspace: cfl_workspace_bytes(shape, n, 2).                                      */
int cfl_pair_scores_idx4(const CflShape *shape, const CflNorm *norm, const float *table, int64_t table_rows,
                         const int32_t *const idx4[4], int64_t idx_stride, int64_t n, const float *theta,
                         float *scores, float *dists, void *workspace, size_t workspace_bytes,
                         cfl_stream_t stream);
int cfl_pair_step_fwd_bwd_idx(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss,
                              const float *table, int64_t table_rows, const int32_t *const idx4[4],
                              int64_t idx_stride, int64_t B, const float *theta, float *grad, float *scalars,
                              void *workspace, size_t workspace_bytes, cfl_stream_t stream);
int cfl_pair_step_fwd_bwd_idx_planes(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss,
                                     const float *table, int64_t table_rows, const int32_t *const idx4[4],
                                     int64_t idx_stride, int64_t B, const float *theta, float *grad, float *scalars,
                                     CflThetaPlanes *planes, void *workspace, size_t workspace_bytes, cfl_stream_t stream);
int cfl_pair_train_step_idx(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss,
                            const float *table, int64_t table_rows, const int32_t *const idx4[4],
                            int64_t idx_stride, int64_t B, float *theta, float *m, float *v, float *grad,
                            float *scalars, float lr_t, float beta1, float beta2, float eps, void *workspace,
                            size_t workspace_bytes, cfl_stream_t stream);
int cfl_pair_train_step_idx_planes(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss,
                                   const float *table, int64_t table_rows, const int32_t *const idx4[4],
                                   int64_t idx_stride, int64_t B, float *theta, float *m, float *v, float *grad,
                                   float *scalars, float lr_t, float beta1, float beta2, float eps,
                                   CflThetaPlanes *planes, void *workspace, size_t workspace_bytes, cfl_stream_t stream);

/* A train of `nsteps` consecutive training steps over windows of the device copies of the (shuffled) pair lists
 * pos_pairs / neg_pairs (int32 [n, 2] = (source, target) positions): step i trains rows
 * [head + i*batch_rows + shard_lo, ... + rows) of both lists (shard_lo / rows select this rank's slice of the global
 * batch; single GPU: 0 / batch_rows).  Replaces the loop `for i in t: sess.run([summary, [s_optim], ...])` of
 * cfl/bin/train_dist.py:77-87 between two read-backs of the display scalars.  lr_t is derived per step from the float32
 * power accumulators *beta1_power / *beta2_power (HOST floats, TF's beta1_power / beta2_power variables), which are
 * advanced by nsteps.  switched: HOST bytes [nsteps] (data_switch coin flips, cfl/input_data.py:575-577) or NULL.
 * grad / scalars hold the last step's values.  n_pos / n_neg: rows of the two pair lists; a window that runs past the
 * end of either (head + nsteps * batch_rows > n) is rejected with CFL_E_SHAPE before anything is launched.         */
int cfl_pair_train_steps_idx(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss,
                             const float *table, int64_t table_rows, const int32_t *pos_pairs, int64_t n_pos,
                             const int32_t *neg_pairs, int64_t n_neg, int64_t pos_head, int64_t neg_head, int64_t batch_rows,
                             int64_t shard_lo, int64_t rows, const uint8_t *switched, int64_t nsteps, float *theta,
                             float *m, float *v, float *grad, float *scalars, float lr, float beta1, float beta2,
                             float eps, float *beta1_power, float *beta2_power, void *workspace,
                             size_t workspace_bytes, cfl_stream_t stream);
int cfl_pair_train_steps_idx_planes(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss,
                                    const float *table, int64_t table_rows, const int32_t *pos_pairs, int64_t n_pos,
                                    const int32_t *neg_pairs, int64_t n_neg, int64_t pos_head, int64_t neg_head,
                                    int64_t batch_rows, int64_t shard_lo, int64_t rows, const uint8_t *switched,
                                    int64_t nsteps, float *theta, float *m, float *v, float *grad, float *scalars,
                                    float lr, float beta1, float beta2, float eps, float *beta1_power,
                                    float *beta2_power, CflThetaPlanes *planes, void *workspace, size_t workspace_bytes,
                                    cfl_stream_t stream);

/* The same train of steps WITH the reference's validation fetch inside it (ABI 5).  The reference's loop fetches
 * `val_s_accuracy` -- the scores of one VALIDATION batch -- and the display scalars in the same sess.run as every training
 * step (cfl/bin/train_dist.py:79-86); scored separately that is a second projection + row-math launch pair and two copies
 * per iteration.  Here step i with val_mask[i] != 0 carries the next validation batch as EXTRA SCORING ROWS of its own
 * projection and row-math launches (forward only: no loss, no gradient), with theta as it stands BEFORE the step's update
 * (the fetch and the update of one sess.run are unordered in TensorFlow), and leaves
 *     ring_slots[k][0 .. CFL_S_COUNT)                     the step's scalars
 *     ring_slots[k][CFL_S_COUNT .. CFL_S_COUNT + 2 vb)     scores of the vb positive, then the vb negative validation pairs
 * in the k-th slot (k counts the validation steps of the call) -- written by the kernels themselves, so the slots may be
 * pinned HOST memory mapped into the device's address space (no copy command on the stream; read them after an event).
 *   val_table / val_pos_pairs / val_neg_pairs / val_*_head / val_batch_rows / val_switched[k]: as table / pos_pairs / ... for
 *   the validation split (the whole validation batch on every rank: no shard); ring_slots: HOST array of >= (number of
 *   nonzero val_mask entries) device-accessible pointers, each CFL_S_COUNT + 2 * val_batch_rows floats.
 * Only where cfl_train_val_fusable(shape, rows, val_batch_rows) returns 1 (chunk-at-a-time projection, wave-per-row math:
 * the batch sizes of the training loops); CFL_E_UNSUPPORTED otherwise -- score separately with cfl_pair_scores_idx4.      */
int cfl_train_val_fusable(const CflShape *shape, int64_t rows, int64_t val_rows);
int cfl_pair_train_val_steps_idx_planes(
    const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss, const float *table, int64_t table_rows,
    const int32_t *pos_pairs, int64_t n_pos, const int32_t *neg_pairs, int64_t n_neg, int64_t pos_head, int64_t neg_head,
    int64_t batch_rows, int64_t shard_lo, int64_t rows, const uint8_t *switched, int64_t nsteps,
    const float *val_table, int64_t val_table_rows, const int32_t *val_pos_pairs, int64_t n_val_pos,
    const int32_t *val_neg_pairs, int64_t n_val_neg, int64_t val_pos_head, int64_t val_neg_head, int64_t val_batch_rows,
    const uint8_t *val_switched, const uint8_t *val_mask, float *const *ring_slots,
    float *theta, float *m, float *v, float *grad, float *scalars, float lr, float beta1, float beta2, float eps,
    float *beta1_power, float *beta2_power, CflThetaPlanes *planes, void *workspace, size_t workspace_bytes,
    cfl_stream_t stream);

/* HOST-ONLY (no GPU, all pointers are host pointers): the per-epoch reshuffle `pairs = pairs[rng.permutation(n)]` of
 * cfl/input_data.py:543-551 in numpy's legacy RandomState (MT19937) stream, bit for bit.  key[624] / *pos are the
 * generator state (numpy: rng.get_state()[1], [2]) and are advanced exactly as rng.permutation(n) would advance them.
 * rows_out[i, :] = rows_in[perm[i], :] (int64, `cols` per row; both NULL: permutation only); perm_out (nullable)
 * receives the permutation; rows_out32 (nullable) a second copy of rows_out narrowed to int32 (the device index
 * format of the *_idx entry points).  Runs without the interpreter lock when called through ctypes.               */
int cfl_mt19937_reshuffle(uint32_t *key, int32_t *pos, int64_t n, const int64_t *rows_in, int64_t cols,
                          int64_t *rows_out, int64_t *perm_out, int32_t *rows_out32);

/* HOST-ONLY introspection (ABI 5; no reference counterpart): the kernels and the launch geometry the library will use
 * for a call of this shape -- produced by the same planner the entry points execute, so that benchmarks, profiles and
 * documentation name kernels from ONE source instead of re-deriving the dispatch.
 *   rows / groups  as cfl_workspace_bytes (groups 2 = a training batch or cfl_pair_scores_idx4; 1 = scoring)
 *   train          nonzero: forward + backward (+ update); planes_kept: the caller passes a CflThetaPlanes buffer        */
typedef struct {
    char proj[40], mid[40], grad[40], tail[40];   /* kernel names ("" = no such launch; tail = a separate finalize launch) */
    int32_t launches;              /* kernel launches of the call, without the stand-alone Adam of a data-parallel step  */
    int32_t per_call_plane_split;  /* 1: + cfl_wplanes_kernel in every call (bf16x3 projection without kept planes)      */
    int32_t S, P;                  /* d slices of the projection, row ranges of the weight gradient                      */
    int32_t rows_padded, column_jobs;
    int32_t proj_tile_rows, proj_workgroups;
    int32_t grad_tile_d, grad_workgroups, grad_waves;
    int32_t fused_tail;            /* gradient (+ Adam, + planes) finished inside the weight-gradient launch             */
    int32_t reads_planes;          /* the projection multiplies bf16 planes of theta                                     */
    int32_t xcd_aligned;           /* blockIdx -> tile mapping deals d slices / d tiles one per XCD                      */
} CflPlanInfo;
int cfl_plan_describe(const CflShape *shape, int64_t rows, int32_t groups, int32_t train, int32_t planes_kept,
                      CflPlanInfo *out);

/* HOST-ONLY: CRC-32C (Castagnoli; reflected, init / xor-out ~0) of n bytes, continuing from `crc` (0 to start).  The
 * checksum of TensorFlow's checkpoint files (tensor bundle + its SSTable index): cfl/tf_bundle.py writes and verifies the
 * files the reference's tf.train.Saver exchanges at cfl/utils.py:465-497 without TensorFlow.                             */
uint32_t cfl_crc32c(const void *data, size_t n, uint32_t crc);

/* The launch plan of a (shape, rows, groups) combination is computed once per process and thread; the tuning /
 * diagnostic overrides it reads from the environment (CFL_EXACT_FP32, CFL_DEBUG_*) are re-read after this call. */
int cfl_reload_env(void);

/* TF-1.x AdamOptimizer apply over a flat array:
 *   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; theta -= lr_t m / (sqrt(v)+eps)
 * Replaces tf.train.AdamOptimizer(...).minimize at cfl/models/dist.py:291-293,
 * cfl/models/cfl.py:1077-1085.  `grad_scale` multiplies g first (1/world_size
 * after a sum all-reduce).                                                     */
int cfl_adam_tf(float *theta, float *m, float *v, const float *grad, int64_t n,
                float lr_t, float beta1, float beta2, float eps,
                float grad_scale, cfl_stream_t stream);

/* cfl_adam_tf over the whole theta of `shape` (n = cfl_layout().total) that also WRITES the kept bf16 planes of the
 * weights it updates and sets planes->valid (ABI 5; planes may be NULL: identical to cfl_adam_tf).  The update of a
 * data-parallel step: cfl_pair_step_fwd_bwd[_idx]_planes -> exchange of [gradient | scalars] -> this call, so that the
 * next step's projection runs on the bf16 matrix cores from current planes exactly as the fused single-GPU step's does
 * (bit-identical planes: the same round-to-nearest split, csrc/theta_planes.h).                                         */
int cfl_adam_tf_planes(const CflShape *shape, float *theta, float *m, float *v, const float *grad,
                       float lr_t, float beta1, float beta2, float eps, float grad_scale,
                       CflThetaPlanes *planes, cfl_stream_t stream);

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
 * POST-activation output y and dL/dy.  In the backward dx, (dV, dg) and db may each be NULL
 * (that product is skipped; x may be NULL when dV is), and y may be NULL when act == 0.     */
typedef struct {
    int32_t B, H, W, Ci, Co, KH, KW, stride, act;
} CflConv;
size_t cfl_conv_workspace_bytes(const CflConv *conv);
/* Introspection (tests, profiling): 1 when `product` (0 = forward, 1 = input gradient, 2 = weight gradient) of this shape
 * runs on a direct 3x3 halo-tile kernel, 0 when it runs on the gathered GEMM; negative on a bad shape.  product 3: 1 when the
 * shape is the 3-channel 4x4 stride-2 image stem, whose three products run on the kernels of csrc/conv_stem.h.  No reference
 * counterpart. */
int cfl_conv_uses_direct_kernel(const CflConv *conv, int product);
int cfl_conv2d_wn_fwd(const CflConv *conv, const float *x, const float *V, const float *g,
                      const float *b, float *y, void *workspace, size_t workspace_bytes,
                      cfl_stream_t stream);
int cfl_conv2d_wn_bwd(const CflConv *conv, const float *x, const float *V, const float *g,
                      const float *y, const float *dy, float reg_const, float *dx, float *dV,
                      float *dg, float *db, void *workspace, size_t workspace_bytes,
                      cfl_stream_t stream);

/* The same two calls with a caller-kept, per-layer cache of everything that depends on the weights only: the
 * per-channel weight-norm scale g / ||V|| and -- for 3x3 stride-1 layers -- the prepared (scaled, split, re-ordered)
 * filter planes of the direct kernel, forward and input-gradient direction.  A layer is called 3-6 times per MrCGAN
 * step between two updates of its weights (batched G / D passes, gradient-penalty double backward); without the cache
 * each call recomputes them (two to three small launches per call).
 *   cache        dev buffer of cfl_conv_cache_bytes(conv) bytes, 16-byte aligned, owned by the caller, one per layer
 *                (it does not depend on B, H, W); NULL = no caching (identical to the plain calls)
 *   cache_flags  HOST int32: CFL_CONV_CACHE_* bits of what the buffer holds; the library sets bits as it fills the
 *                buffer, the CALLER clears them (to 0) whenever V or g change.  Results are bit-identical with and
 *                without the cache.  No reference counterpart (TensorFlow recomputes per op).                       */
#define CFL_CONV_CACHE_SCALE 1
#define CFL_CONV_CACHE_PLANES_FWD 2
#define CFL_CONV_CACHE_PLANES_DX 4
size_t cfl_conv_cache_bytes(const CflConv *conv);
int cfl_conv2d_wn_fwd_cached(const CflConv *conv, const float *x, const float *V, const float *g,
                             const float *b, float *y, void *workspace, size_t workspace_bytes,
                             void *cache, size_t cache_bytes, int32_t *cache_flags, cfl_stream_t stream);
int cfl_conv2d_wn_bwd_cached(const CflConv *conv, const float *x, const float *V, const float *g,
                             const float *y, const float *dy, float reg_const, float *dx, float *dV,
                             float *dg, float *db, void *workspace, size_t workspace_bytes,
                             void *cache, size_t cache_bytes, int32_t *cache_flags, cfl_stream_t stream);
/* Rebuild whatever of a layer's cache is not valid -- weight-norm scale, forward / input-gradient filter planes for the
 * products that take the halo kernel at this call shape -- without running the layer (e.g. for all layers right behind the
 * optimizer step, on a side stream).  The planes are laid out from the channel counts alone, so a cache prepared with one
 * batch size serves calls with another. */
int cfl_conv_prepare_cached(const CflConv *conv, const float *V, const float *g, void *cache, size_t cache_bytes,
                            int32_t *cache_flags, cfl_stream_t stream);
/* ... for n layers at once (round 6): ONE launch for the weight-norm scales of all of them and ONE for their filter planes
 * (up to 48 jobs per launch; filters of >= 2^20 elements keep their coalesced two-launch scale) -- instead of ~65 launches in a row
 * behind every optimizer step of the MrCGAN post epochs.  Arrays of n entries; bit-identical to n calls of the function above.   */
int cfl_conv_prepare_cached_many(int32_t n, const CflConv *convs, const float *const *V, const float *const *g,
                                 void *const *caches, const size_t *cache_bytes, int32_t *const *cache_flags, cfl_stream_t stream);

/* The backward with the sub-pixel un-shuffle folded into its dy loaders (round 4; cache nullable as above):
 *   dy_subpixel != 0: dy -- and y, the layer's activated output, when conv->act != 0 -- arrive 2x sub-pixel shuffled,
 *             [B, 2 OH, 2 OW, Co/4], exactly as cfl_conv2d_wn_fwd_fused(subpixel = 1) stored y; equal bit for bit to
 *             cfl_subpixel2x_bwd followed by the plain backward.  Only where cfl_conv_bwd_takes_subpixel() returns 1 (3x3
 *             stride 1, both products on the halo-tile kernels, Co % 128 == 0); CFL_E_SHAPE otherwise.
 * Reference: tf.gradients through conv2d_subpixel + activation, cfl/layers.py:212-250. */
int cfl_conv_bwd_takes_subpixel(const CflConv *conv);
int cfl_conv2d_wn_bwd_fused(const CflConv *conv, const float *x, const float *V, const float *g, const float *y,
                            const float *dy, int32_t dy_subpixel, float reg_const, float *dx, float *dV, float *dg, float *db,
                            void *workspace, size_t workspace_bytes, void *cache, size_t cache_bytes, int32_t *cache_flags,
                            cfl_stream_t stream);
/* Deferred weight-gradient finalisation (round 6).  cfl_conv2d_wn_bwd_* ends every weight gradient with launches of its own (slab
 * sums + the per-channel weight-norm finalisation): ~80 launches of 5-16 us per MrCGAN step.  These two entry points split the work:
 *   cfl_conv2d_wn_wgrad_slabs   ONLY the contraction dW = x^T (dy * act'(y)) of a layer (+ the bias-gradient row), as split-K slabs in
 *                               the caller's region `slab` (cfl_conv_wgrad_slab_bytes(conv) bytes, 16-byte aligned, one region per
 *                               layer and backward chain; the caller keeps it until the chain is finished)
 *   cfl_conv_wfinal_many        finishes n such layers at once: dV = s dW - (s / n^2)(dW . V) V + reg V, dg = (dW . V) / n, db -- ONE
 *                               slab-sum launch and ONE finalisation launch for all of them (up to 32 layers per launch pair; the few
 *                               filters of >= 2^20 elements keep their coalesced three-launch form).  caches[i] = the layer's cache
 *                               buffer with CFL_CONV_CACHE_SCALE valid (its header holds scale and n^2); dg[i] / db[i] nullable.
 * Bit-identical to the per-layer path (the same per-element code in the same order).                                              */
size_t cfl_conv_wgrad_slab_bytes(const CflConv *conv);
int cfl_conv2d_wn_wgrad_slabs(const CflConv *conv, const float *x, const float *y, const float *dy, int32_t dy_subpixel,
                              float *slab, size_t slab_bytes, cfl_stream_t stream);
int cfl_conv_wfinal_many(int32_t n, const CflConv *convs, float *const *slabs, const float *const *V, const float *const *g,
                         void *const *caches, float reg_const, float *const *dV, float *const *dg, float *const *db,
                         cfl_stream_t stream);

/* The forward with two store epilogues of the MrCGAN stacks folded in (cache nullable as above):
 *   residual  dev [B,OH,OW,Co] or NULL: y = act(conv + b + residual) -- the join of a residual block
 *             (cfl/models/blocks.py:150-170: lrelu(conv_b(...) + h)) as the epilogue of its second convolution
 *   subpixel  != 0: y is stored 2x sub-pixel shuffled, [B, 2 OH, 2 OW, Co/4] (conv2d_subpixel, cfl/layers.py:212-250, is an
 *             index remap of the convolution's output; the activation commutes with it).  Co % 4 == 0.            */
int cfl_conv2d_wn_fwd_fused(const CflConv *conv, const float *x, const float *V, const float *g, const float *b,
                            const float *residual, int32_t subpixel, float *y, void *workspace, size_t workspace_bytes,
                            void *cache, size_t cache_bytes, int32_t *cache_flags, cfl_stream_t stream);

/* Weight-normalised TRANSPOSED convolution (conv2d_transpose_weight_norm, cfl/layers.py:253-361):
 * x [B,H,W,Ci], V [KH,KW,Co,Ci] (norm over kh,kw,ci per OUTPUT channel), y [B,H*stride,W*stride,Co],
 * 'SAME'.  `conv` describes the layer in its own terms (H,W,Ci = input; Co = output).  Same
 * nullable-pointer rules as cfl_conv2d_wn_bwd.                                              */
size_t cfl_conv_transpose_workspace_bytes(const CflConv *conv);
int cfl_conv2d_transpose_wn_fwd(const CflConv *conv, const float *x, const float *V, const float *g,
                                const float *b, float *y, void *workspace, size_t workspace_bytes,
                                cfl_stream_t stream);
int cfl_conv2d_transpose_wn_bwd(const CflConv *conv, const float *x, const float *V, const float *g,
                                const float *y, const float *dy, float reg_const, float *dx, float *dV,
                                float *dg, float *db, void *workspace, size_t workspace_bytes,
                                cfl_stream_t stream);

/* ---- glue of the MrCGAN stacks (cfl/models/blocks.py:25-438, cfl/models/cfl.py:730-806, 951-1063).
 * A fully connected weight-norm layer (cfl/layers.py:28-97) is cfl_conv2d_wn_* with H = W = KH =
 * KW = stride = 1, Ci = inputs, Co = outputs.                                                */
#define CFL_EW_NONE 0
#define CFL_EW_LRELU 1   /* cfl/ops.py:10-12 */
#define CFL_EW_RELU 2
#define CFL_EW_TANH 3
#define CFL_EW_SIGMOID 4
/* y = act(x);  dx = dy * act'(.) evaluated from the POST-activation y (also used to apply the
 * fixed lrelu masks of the gradient-penalty double backward);  y = act(a + b) (residual join);
 * y += alpha * x.                                                                            */
int cfl_ew_act_fwd(const float *x, float *y, int64_t n, int act, cfl_stream_t stream);
int cfl_ew_act_bwd(const float *y, const float *dy, float *dx, int64_t n, int act, cfl_stream_t stream);
int cfl_ew_add_act(const float *a, const float *b, float *y, int64_t n, int act, cfl_stream_t stream);
/* acc += dy * act'(y): backward of a residual join towards its skip input, accumulated onto the gradient that arrived
 * through the block's convolutions (cfl/models/blocks.py:150-170 differentiated)                              */
int cfl_ew_act_bwd_add(const float *y, const float *dy, float *acc, int64_t n, int act, cfl_stream_t stream);
int cfl_ew_axpy(float alpha, const float *x, float *y, int64_t n, cfl_stream_t stream);
/* y = clip(x * mul + add, lo, hi): data / ae / latent normaliser (cfl/ops.py:66-143, 302-349) applied
 * to a batch that does not enter the fused pair kernels (images for the conv stacks, GAN inputs). */
int cfl_ew_affine_clip(const float *x, float *y, int64_t n, const CflNorm *norm, cfl_stream_t stream);
/* per-channel mean / norm of NHWC images (cfl/ops.py:84-106): y[i] = clip(x[i] * mul[i % C] + add[i % C]);
 * mul / add are HOST arrays of C <= 4 floats, lo / hi / has_* are taken from `clip`.                    */
int cfl_ew_affine_clip_channels(const float *x, float *y, int64_t n, int C, const float *mul, const float *add,
                                const CflNorm *clip, cfl_stream_t stream);
/* Input transformers of cfl/ops.py:38-63, 262-299 on NHWC batches x [B,H,W,C] -> y [B,h,w,C]:
 * mode 0 = crop / zero-pad window (offsets [B,2] = (row, col) of tf.random_crop, NULL = the central window of
 * resize_image_with_crop_or_pad); mode 1 = tf.image.resize_images bilinear (align_corners = False).
 * flip [B] (nullable) = per-sample left-right flip applied to the result (tf.image.random_flip_left_right). */
int cfl_image_transform(const float *x, int64_t B, int H, int W, int C, float *y, int h, int w,
                        const int32_t *offsets, const int32_t *flip, int mode, cfl_stream_t stream);
/* conv2d_subpixel scale 2 (cfl/layers.py:212-250): x [B,H,W,C] -> y [B,2H,2W,C/4],
 * y[b,2h+i,2w+j,c] = act(x[b,h,w,(2i+j)*C/4+c]); bwd scatters dy * act'(y) back (y may be NULL
 * when act == CFL_EW_NONE).                                                                  */
int cfl_subpixel2x_fwd(const float *x, float *y, int64_t B, int H, int W, int C, int act, cfl_stream_t stream);
int cfl_subpixel2x_bwd(const float *y, const float *dy, float *dx, int64_t B, int H, int W, int C, int act,
                       cfl_stream_t stream);
/* out[r] = [a[r, :na], b[r, :nb]]  (tf.concat(zs, -1), cfl/models/blocks.py:65)               */
int cfl_concat_cols(const float *a, int na, const float *b, int nb, int64_t rows, float *out, cfl_stream_t stream);
/* dst[r, dst_off + c] = src[r, src_off + c], c < ncols: column slices / concats of row-major matrices.   */
int cfl_copy_cols(const float *src, int src_ld, int src_off, float *dst, int dst_ld, int dst_off, int ncols,
                  int64_t rows, cfl_stream_t stream);
/* cgan conditioning of the discriminator (cfl/models/blocks.py:182-195, 382-395): out[b,h,w,:] =
 * concat(hmap[b,h,w,:C1], t[b,:C2]) (t == NULL: zeros); backward: dh = d[..., :C1] (nullable),
 * dt[b,:] = sum_{h,w} d[b,h,w,C1:] (nullable).  HW = pixels per sample.                                  */
int cfl_tile_concat_channels(const float *hmap, int C1, const float *t, int C2, int64_t samples, int HW, float *out,
                             cfl_stream_t stream);
int cfl_tile_concat_channels_bwd(const float *d, int C1, int C2, int64_t samples, int HW, float *dh, float *dt,
                                 cfl_stream_t stream);
/* one_prototype_activations: out[r, :] = P[r, c[r], :], P [B,K,L] (cfl/models/cfl.py:535-546)   */
int cfl_gather_prototype(const float *P, const int32_t *c, int64_t B, int K, int L, float *out, cfl_stream_t stream);
/* *loss = weight * mean(sigmoid_cross_entropy_with_logits(logits, label)); *frac_pos = mean(logits > 0);
 * dlogits (nullable) = or += weight/n * (sigmoid - label).   cfl/models/cfl.py:955-1037         */
int cfl_bce_logits(const float *logits, int64_t n, float label, float weight, float *loss, float *frac_pos,
                   float *dlogits, int accumulate, cfl_stream_t stream);
/* latent losses on d_r = sum_j (a[r,j]-b[r,j])^2 (cfl/models/cfl.py:1001-1063): mode 0 weight*mean(d);
 * mode 1 weight*mean(max(0, sqrt(d+1e-7) - margin)^2); mode 2 weight*mean(max(0, margin - sqrt(d+1e-7))^2).
 * da (nullable) = or += d loss / d a.                                                           */
int cfl_rowdist_loss(const float *a, const float *b, int64_t B, int L, int mode, float margin, float weight,
                     float *loss, float *da, int accumulate, cfl_stream_t stream);
/* X_hat = X + lambda_dra * sqrt(population variance of all of X) * eps[row] (cfl/models/cfl.py:742-745) */
size_t cfl_perturb_workspace_bytes(void);
int cfl_perturb(const float *x, const float *eps, int64_t B, int64_t N, float lambda_dra, float *out,
                void *workspace, size_t workspace_bytes, cfl_stream_t stream);
/* gradient penalty (cfl/models/cfl.py:986-991) from u = d D(X_hat) / d X_hat [B,N]:
 * *loss = lambda_gp * mean((||u_r|| - 1)^2);  v (nullable) = d loss / d u;  rowloss: dev scratch [B]. */
int cfl_grad_penalty(const float *u, int64_t B, int64_t N, float lambda_gp, float *loss, float *v,
                     float *rowloss, cfl_stream_t stream);

/* ROC AUC (ties get half credit, = sklearn.metrics.roc_auc_score) and sign accuracy of a split's scores:
 * the `roc_auc_score` / accuracy of dist_eval, cfl/utils.py:227-274, cfl/bin/evaluate_total.py:100-118.
 *   scores_pos [n_pos], scores_neg [n_neg] : dev fp32;  out : dev double[2] = {auc, accuracy}
 *   accuracy = (#(pos > 0) + #(neg <= 0)) / (n_pos + n_neg)                                            */
size_t cfl_auc_workspace_bytes(int64_t n_pos, int64_t n_neg);
int cfl_auc(const float *scores_pos, int64_t n_pos, const float *scores_neg, int64_t n_neg, double *out,
            void *workspace, size_t workspace_bytes, cfl_stream_t stream);

/* Optional per-kernel timing (bench.py's roofline object).  While enabled,
 * every kernel the library launches is bracketed by two HIP events recorded on
 * the caller's stream.  cfl_profile_read() synchronises those events, adds the
 * elapsed milliseconds / launch counts per kernel kind into the two arrays of
 * CFL_K_COUNT entries and clears the record list.  Not for production steps.   */
enum {
    CFL_K_COLNORM = 0, CFL_K_PROJ, CFL_K_MID, CFL_K_GRAD, CFL_K_FINALIZE,
    CFL_K_ADAM, CFL_K_GATHER, CFL_K_DP /* the kernels of the one-shot exchange (ABI 6) */, CFL_K_COUNT = 8
};
int cfl_profile_enable(int on);
int cfl_profile_read(double *ms_sum, int64_t *launches);

/* One-shot gradient exchange of the data-parallel step (new functionality; the reference is single-device).  An opt-in
 * alternative to all-reducing [gradient | scalars] with RCCL, shaped for point-to-point xGMI links: reduce-scatter by
 * direct stores into the owners' slot arrays, TF-Adam on the owned 1/world of theta / m / v (the Adam slots are SHARDED:
 * rank r keeps slice r of m and v current), all-gather of the updated theta slices by direct stores into the peers' stage
 * buffers.  Deterministic (rank-ordered sums, one writer per parameter), bit-identical parameters on every rank.  Host
 * side: cfl/dp_exchange.py; protocol and memory model: csrc/cfl_dp.hip.
 *
 * Exchange memory.  Slots, stage buffers and flag words are polled by running kernels while peers store into them, which
 * HIP supports for FINE-GRAINED device allocations only; the caller allocates them here and shares them with the other
 * ranks' processes through the 64-byte hipIpc handle.  The caller owns every allocation / mapping (free / close them).
 *   cfl_dp_alloc       *ptr = zero-filled device memory on the current device (fine_grained != 0:
 *                      hipExtMallocWithFlags(hipDeviceMallocFinegrained), else hipMalloc)
 *   cfl_dp_ipc_export  handle64: 64 bytes to send to the peers;  cfl_dp_ipc_open maps a peer's allocation here
 *
 * Let n = floats of the exchanged buffer (a multiple of 4), slice = floats per rank (a multiple of 4, slice * world >= n;
 * rank r owns [r * slice, min((r + 1) * slice, n)) ), generation = a number unique to the step (never 0).
 *   cfl_dp_rs_push    src dev [n] (this rank's [gradient | scalars]); peer_rows[s] = row `rank` (slice floats) of rank
 *                     s's slot array of the current parity, peer_flags[s] = this rank's A flag word there (HOST arrays
 *                     [world] of device pointers; entry `rank` = the local ones); ticket: dev uint32, zero before the
 *                     first call (the kernels leave it zero)
 *   cfl_dp_rs_adam    gslots dev [world][slice] (local, current parity), flags dev [world] (local A flags); waits for
 *                     them (at most timeout_s seconds of wall-clock time; <= 0: 30 s), sums the rows in rank order,
 *                     writes its slice of the sum to sum_out [n], applies TF-Adam with sum / world to its slice of the
 *                     first n_adam floats (theta, m, v), stores the updated theta slice (past n_adam: the sums) into
 *                     peer_stage[r] [n] of every peer r != rank and raises its B flag peer_flags[r] there.  *lost (dev
 *                     int32, zeroed by the caller) is set when the wait gives up; the update is then NaN (loud)
 *   cfl_dp_rs_gather  stage dev [n] (local, current parity), flags dev [world] (local B flags): waits for every peer's
 *                     flag, copies the peers' slices into theta (floats < n_adam) and sum_out (the rest)               */
int cfl_dp_alloc(void **ptr, size_t bytes, int32_t fine_grained);
int cfl_dp_free(void *ptr);
int cfl_dp_ipc_export(void *ptr, void *handle64);
int cfl_dp_ipc_open(const void *handle64, void **ptr);
int cfl_dp_ipc_close(void *ptr);
int cfl_dp_rs_push(const float *src, int64_t n, int64_t slice, float *const *peer_rows, uint32_t *const *peer_flags,
                   int32_t world, uint32_t generation, uint32_t *ticket, cfl_stream_t stream);
int cfl_dp_rs_adam(float *theta, float *m, float *v, const float *gslots, const uint32_t *flags, int32_t world,
                   int32_t rank, int64_t n, int64_t n_adam, int64_t slice, float *sum_out, float *const *peer_stage,
                   uint32_t *const *peer_flags, float lr_t, float beta1, float beta2, float eps, uint32_t generation,
                   int32_t *lost, double timeout_s, uint32_t *ticket, cfl_stream_t stream);
int cfl_dp_rs_gather(float *theta, float *sum_out, const float *stage, const uint32_t *flags, int32_t world, int32_t rank,
                     int64_t n, int64_t n_adam, int64_t slice, uint32_t generation, int32_t *lost, double timeout_s,
                     cfl_stream_t stream);
/* ... that also writes the kept bf16 planes of EVERY weight of theta (the peers' slices as it copies them, this rank's own
 * slice -- updated by cfl_dp_rs_adam on the same stream -- as it passes over it) and sets planes->valid (ABI 5);
 * n_adam must be cfl_layout(shape).total.  planes may be NULL: identical to cfl_dp_rs_gather.                          */
int cfl_dp_rs_gather_planes(const CflShape *shape, float *theta, float *sum_out, const float *stage, const uint32_t *flags,
                            int32_t world, int32_t rank, int64_t n, int64_t n_adam, int64_t slice, uint32_t generation,
                            int32_t *lost, double timeout_s, CflThetaPlanes *planes, cfl_stream_t stream);

/* ---- ABI 6: the WHOLE data-parallel step behind one call ------------------------------------------------------------
 * Round 6.  The reference is single-device; SURVEY 8(e) / BASELINE north_star ask for row shards + one
 * gradient exchange per step.  Until ABI 5 the host drove that step as 3-5 separate calls per iteration; from ABI 6 one call
 * runs K iterations of
 *
 *   proj -> mid -> grad (fused tail emits [gradient | scalars])  ->  exchange  ->  TF-Adam (+ planes)
 *
 * with either exchange:
 *   CflDpExchange  the one-shot exchange above with its reduce-scatter FUSED into the weight-gradient launch: the tile
 *                  finishers store their finished gradient entries straight into the owner rank's slot array (peer memory)
 *                  and the launch's last workgroup raises this rank's arrival flags -- no cfl_dp_rs_push launch, no flat
 *                  gradient round trip: proj, mid, grad(+push), cfl_dp_rs_adam, cfl_dp_rs_gather_planes = 5 launches;
 *   CflAllReduce   a caller-supplied all-reduce (signature of ncclAllReduce: the caller passes RCCL's entry point and its
 *                  communicator, the library does not link RCCL) enqueued on the launch stream between the weight-gradient
 *                  launch and cfl_adam_tf_planes: 3 launches + the collective + 1;
 *   neither        no exchange at all (a one-rank group; "step without collective" measurements).
 */
#define CFL_DP_MAX_WORLD 16
typedef struct {
    int32_t world, rank;
    int64_t n, n_adam, slice;                      /* floats: exchanged buffer, parameters (cfl_layout.total), per-rank slice */
    /* everything below is per step parity p = step & 1 (double buffering, csrc/cfl_dp.hip) */
    float *slots[2];                               /* LOCAL slot array [world][slice] */
    float *stage[2];                               /* LOCAL stage buffer [n] */
    uint32_t *flags_a[2], *flags_b[2];             /* LOCAL flag words [world] */
    float *peer_rows[2][CFL_DP_MAX_WORLD];         /* row `rank` of rank r's slot array, mapped here ([rank] = the local one) */
    float *peer_stage[2][CFL_DP_MAX_WORLD];        /* rank r's stage buffer, mapped here */
    uint32_t *peer_flag_a[2][CFL_DP_MAX_WORLD];    /* word `rank` of rank r's A / B flag arrays, mapped here */
    uint32_t *peer_flag_b[2][CFL_DP_MAX_WORLD];
    /* DEVICE copy of {peer_rows[p], peer_flag_a[p]} for the fused push: void *[2 parities][2 kinds][CFL_DP_MAX_WORLD],
     * kind 0 = rows, kind 1 = A flags (512 bytes; the caller fills it once).  NULL: the push stays a launch of its own */
    void *dev_tables;
    uint32_t *tickets;                             /* device, 4 zeroed words ([0] push / fused push, [1] Adam) */
    int32_t *lost;                                 /* set to 1 by a kernel whose bounded wait gave up (pinned host or device) */
    double timeout_s;
    uint64_t step;                                 /* exchanges so far (parity / generation); ADVANCED BY THE LIBRARY */
} CflDpExchange;

typedef int (*cfl_allreduce_fn)(const void *sendbuf, void *recvbuf, size_t count, int dtype, int op, void *comm, void *stream);
typedef struct {
    cfl_allreduce_fn fn;                           /* ncclAllReduce of the librccl.so the process already uses */
    void *comm;                                    /* ncclComm_t */
    int32_t dtype, op;                             /* ncclFloat32 (7), ncclSum (0) */
    int32_t world;                                 /* ranks (the gradient / scalar sums are divided by it) */
} CflAllReduce;

/* The exchange + update of ONE step on `gradbuf` = this rank's [gradient | scalars | pad] (n floats): with `pushed` != 0 the
 * weight-gradient launch has already stored it into the owners' slots (fused push); otherwise cfl_dp_rs_push runs first.
 * Then cfl_dp_rs_adam and cfl_dp_rs_gather_planes.  scalars_copy (nullable, device-accessible, 16 floats): a second copy of
 * the global scalar SUMS (divide by world).  Advances ex->step.                                                        */
int cfl_dp_exchange_step(const CflShape *shape, CflDpExchange *ex, int32_t pushed, float *theta, float *m, float *v,
                         float *gradbuf, float lr_t, float beta1, float beta2, float eps, CflThetaPlanes *planes,
                         float *scalars_copy, cfl_stream_t stream);
/* workspace of a training call of `rows` rows per group that carries a validation batch of `val_rows` pairs per group
 * (cfl_pair_train_val_steps_idx_planes / cfl_pair_dp_steps_idx_planes: a data-parallel rank trains its shard but scores the
 * whole validation batch, val_rows > rows); >= cfl_workspace_bytes(shape, rows, 2).  0 on error.                         */
size_t cfl_workspace_bytes_val(const CflShape *shape, int64_t rows, int64_t val_rows);
/* can a training call of this shape / row count push from inside its weight-gradient launch (fused tail plans)? */
int cfl_dp_push_fusable(const CflShape *shape, int64_t rows);

/* One data-parallel step on dense row blocks / on rows of a resident table (cfl_pair_step_fwd_bwd[_idx]_planes + exchange +
 * update).  gradbuf: [cfl_layout.total floats of gradient | 16 scalars | pad to n]; theta / m / v as cfl_pair_train_step.
 * Exactly one of ex / ar may be non-NULL (both NULL: no exchange, plain cfl_adam_tf_planes).                           */
int cfl_pair_dp_step_planes(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss, const float *const x4[4],
                            int64_t B, float *theta, float *m, float *v, float *gradbuf, float lr_t, float beta1,
                            float beta2, float eps, CflThetaPlanes *planes, CflDpExchange *ex, const CflAllReduce *ar,
                            void *workspace, size_t workspace_bytes, cfl_stream_t stream);
int cfl_pair_dp_step_idx_planes(const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss, const float *table,
                                int64_t table_rows, const int32_t *const idx4[4], int64_t idx_stride, int64_t B,
                                float *theta, float *m, float *v, float *gradbuf, float lr_t, float beta1, float beta2,
                                float eps, CflThetaPlanes *planes, CflDpExchange *ex, const CflAllReduce *ar,
                                void *workspace, size_t workspace_bytes, cfl_stream_t stream);
/* K data-parallel iterations over windows of the device pair lists: cfl_pair_train_val_steps_idx_planes with the exchange
 * inside every iteration (this rank trains rows [shard_lo, shard_lo + rows) of every window).  val_mask == NULL: no
 * validation fetch (the val_* arguments are ignored); otherwise the masked iterations score the next validation batch as
 * extra rows of their own launches -- EVERY rank scores the whole batch (theta is replicated: identical scores) -- and
 * ring_slots[k] receives [global scalar SUMS (x 1/world on the host) | scores].                                         */
int cfl_pair_dp_steps_idx_planes(
    const CflShape *shape, const CflNorm *norm, const CflLossCfg *loss, const float *table, int64_t table_rows,
    const int32_t *pos_pairs, int64_t n_pos, const int32_t *neg_pairs, int64_t n_neg, int64_t pos_head, int64_t neg_head,
    int64_t batch_rows, int64_t shard_lo, int64_t rows, const uint8_t *switched, int64_t nsteps,
    const float *val_table, int64_t val_table_rows, const int32_t *val_pos_pairs, int64_t n_val_pos,
    const int32_t *val_neg_pairs, int64_t n_val_neg, int64_t val_pos_head, int64_t val_neg_head, int64_t val_batch_rows,
    const uint8_t *val_switched, const uint8_t *val_mask, float *const *ring_slots,
    float *theta, float *m, float *v, float *gradbuf, float lr, float beta1, float beta2, float eps,
    float *beta1_power, float *beta2_power, CflThetaPlanes *planes, CflDpExchange *ex, const CflAllReduce *ar,
    void *workspace, size_t workspace_bytes, cfl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CFL_HIP_H */
