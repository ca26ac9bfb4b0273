"""ctypes binding of libcfl_hip.so (include/cfl_hip.h) for torch-ROCm tensors.

PyTorch is only plumbing here (device memory + streams); all arithmetic of the
pair-distance hot path runs in the hand-written gfx950 kernels behind the C ABI.
There is NO CPU / eager fallback: if the shared library is missing, importing
this module's ``lib()`` raises, and every op raises on non-CUDA tensors.
"""
import ctypes as C
import os

# Kernel arguments in DEVICE memory: every launch of the step starts with scalar loads from its argument block, and with
# the block in host memory each of them is a round trip over PCIe -- measured 46.8 vs 35.9 us per step (+3.6 us per
# launch, tools/kernarg_ab.sh).  ROCm 7 defaults to device memory on this GPU; this only pins the default against an
# inherited environment, and has no effect once the HIP runtime is initialised.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

import numpy as np  # noqa: E402
import torch  # noqa: E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CFL_HIP_LIB') or os.path.join(
    os.path.dirname(_HERE), 'lib', 'libcfl_hip.so')

DIST_TYPES = {'pcd': 0, 'monomer': 1, 'siamese': 2}
ACT_TYPES = {None: 0, 'linear': 0, 'sigmoid': 1, 'tanh': 2, 'relu': 3}

SCALAR_NAMES = ('total', 'reg', 'thres', 'loss_pos', 'loss_neg', 'cd', 'accuracy',
                'mean_d_pos', 'mean_d_neg', 'mean_o_pos', 'mean_o_neg', 'threshold',
                'dist_adapt_pos', 'dist_adapt_neg')
S_COUNT = 16
S_ERROR = 15      # sticky error word of the scalars array (CFL_S_ERROR, include/cfl_hip.h)


class CflShape(C.Structure):
    _fields_ = [('D', C.c_int32), ('L', C.c_int32), ('K', C.c_int32),
                ('dist_type', C.c_int32), ('weight_norm', C.c_int32),
                ('has_bias', C.c_int32), ('act_type', C.c_int32),
                ('directed', C.c_int32)]


class CflHead(C.Structure):
    _fields_ = [('w', C.c_int64), ('b', C.c_int64), ('g', C.c_int64),
                ('n', C.c_int32), ('npad', C.c_int32)]


class _CflEnc(C.Structure):
    _fields_ = [('outputs', CflHead), ('proto', CflHead), ('mono', CflHead)]


class CflLayout(C.Structure):
    _fields_ = [('enc', _CflEnc * 2), ('thr', C.c_int64), ('total', C.c_int64)]


class CflNorm(C.Structure):
    _fields_ = [('mul', C.c_float), ('add', C.c_float), ('lo', C.c_float),
                ('hi', C.c_float), ('has_lo', C.c_int32), ('has_hi', C.c_int32), ('valid_cols', C.c_int32)]


class CflLossCfg(C.Structure):
    _fields_ = [('use_threshold', C.c_int32), ('pos_weight', C.c_float),
                ('caffe_margin', C.c_float), ('lambda_m', C.c_float),
                ('reg_const', C.c_float)]


class CflThetaPlanes(C.Structure):
    _fields_ = [('buf', C.c_void_p), ('valid', C.c_int32)]


DP_MAX_WORLD = 16
GRADBUF_PAD = 64      # gradbuf = [cfl_layout.total floats of gradient | 16 scalars | 48 pad] (include/cfl_hip.h, ABI 6)


class CflDpExchange(C.Structure):
    """one rank's view of the one-shot exchange memory (include/cfl_hip.h, ABI 6); filled by cfl.dp_exchange.OneShotExchange"""
    _fields_ = [('world', C.c_int32), ('rank', C.c_int32), ('n', C.c_int64), ('n_adam', C.c_int64), ('slice', C.c_int64),
                ('slots', C.c_void_p * 2), ('stage', C.c_void_p * 2), ('flags_a', C.c_void_p * 2), ('flags_b', C.c_void_p * 2),
                ('peer_rows', (C.c_void_p * DP_MAX_WORLD) * 2), ('peer_stage', (C.c_void_p * DP_MAX_WORLD) * 2),
                ('peer_flag_a', (C.c_void_p * DP_MAX_WORLD) * 2), ('peer_flag_b', (C.c_void_p * DP_MAX_WORLD) * 2),
                ('dev_tables', C.c_void_p), ('tickets', C.c_void_p), ('lost', C.c_void_p), ('timeout_s', C.c_double),
                ('step', C.c_uint64)]


class CflAllReduce(C.Structure):
    """a caller-supplied all-reduce with ncclAllReduce's signature (include/cfl_hip.h, ABI 6); cfl.rccl.Communicator.c_struct()"""
    _fields_ = [('fn', C.c_void_p), ('comm', C.c_void_p), ('dtype', C.c_int32), ('op', C.c_int32), ('world', C.c_int32)]


class CflPlanInfo(C.Structure):
    _fields_ = ([(n, C.c_char * 40) for n in ('proj', 'mid', 'grad', 'tail')] +
                [(n, C.c_int32) for n in ('launches', 'per_call_plane_split', 'S', 'P', 'rows_padded', 'column_jobs',
                                          'proj_tile_rows', 'proj_workgroups', 'grad_tile_d', 'grad_workgroups',
                                          'grad_waves', 'fused_tail', 'reads_planes', 'xcd_aligned')])


class CflConv(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('B', 'H', 'W', 'Ci', 'Co', 'KH', 'KW', 'stride', 'act')]


CONV_ACTS = {None: 0, 'linear': 0, 'lrelu': 1, 'relu': 2}

EXPORTS = ('cfl_version', 'cfl_last_error', 'cfl_layout', 'cfl_workspace_bytes',
           'cfl_pair_scores', 'cfl_pair_step_fwd_bwd', 'cfl_pair_train_step', 'cfl_adam_tf',
           'cfl_gather_rows', 'cfl_profile_enable', 'cfl_profile_read', 'cfl_pair_input_grad',
           'cfl_conv_workspace_bytes', 'cfl_conv_uses_direct_kernel', 'cfl_conv2d_wn_fwd', 'cfl_conv2d_wn_bwd',
           'cfl_pair_scores_idx', 'cfl_pair_step_fwd_bwd_idx', 'cfl_pair_train_step_idx', 'cfl_reload_env',
           'cfl_pair_train_steps_idx', 'cfl_pair_scores_idx4', 'cfl_mt19937_reshuffle', 'cfl_dp_alloc', 'cfl_dp_free', 'cfl_dp_ipc_export', 'cfl_dp_ipc_open', 'cfl_dp_ipc_close',
           'cfl_dp_rs_push', 'cfl_dp_rs_adam', 'cfl_dp_rs_gather',
           'cfl_scalars_status', 'cfl_theta_planes_bytes', 'cfl_pair_train_step_planes', 'cfl_pair_train_step_idx_planes',
           'cfl_pair_train_steps_idx_planes', 'cfl_pair_step_fwd_bwd_planes', 'cfl_pair_step_fwd_bwd_idx_planes',
           'cfl_adam_tf_planes', 'cfl_dp_rs_gather_planes', 'cfl_plan_describe', 'cfl_crc32c',
           'cfl_train_val_fusable', 'cfl_pair_train_val_steps_idx_planes',
           'cfl_dp_exchange_step', 'cfl_dp_push_fusable', 'cfl_pair_dp_step_planes', 'cfl_pair_dp_step_idx_planes',
           'cfl_pair_dp_steps_idx_planes', 'cfl_workspace_bytes_val')

KERNEL_NAMES = ('colnorm', 'proj', 'mid', 'grad', 'finalize', 'adam', 'gather', 'dp_exchange')
K_COUNT = 8

_lib = None


class CflHipError(RuntimeError):
    pass


def lib():
    """Load libcfl_hip.so (built by ``__graft_entry__.build()``); fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CflHipError(
            'HIP extension missing: {} (run `python -c "import __graft_entry__ as g; '
            'g.build()"`); there is no CPU fallback for the cfl hot path'.format(LIB_PATH))
    L = C.CDLL(LIB_PATH)
    L.cfl_version.restype = C.c_int
    L.cfl_last_error.restype = C.c_char_p
    L.cfl_layout.argtypes = [C.POINTER(CflShape), C.POINTER(CflLayout)]
    L.cfl_layout.restype = C.c_int
    L.cfl_workspace_bytes.argtypes = [C.POINTER(CflShape), C.c_int64, C.c_int32]
    L.cfl_workspace_bytes.restype = C.c_size_t
    L.cfl_pair_scores.argtypes = [
        C.POINTER(CflShape), C.POINTER(CflNorm), C.c_void_p, C.c_void_p, C.c_int64,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.cfl_pair_scores.restype = C.c_int
    L.cfl_pair_step_fwd_bwd.argtypes = [
        C.POINTER(CflShape), C.POINTER(CflNorm), C.POINTER(CflLossCfg),
        C.POINTER(C.c_void_p), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_size_t, C.c_void_p]
    L.cfl_pair_step_fwd_bwd.restype = C.c_int
    a = list(L.cfl_pair_step_fwd_bwd.argtypes)
    L.cfl_pair_step_fwd_bwd_planes.argtypes = a[:8] + [C.POINTER(CflThetaPlanes)] + a[8:]
    L.cfl_pair_step_fwd_bwd_planes.restype = C.c_int
    L.cfl_pair_train_step.argtypes = [
        C.POINTER(CflShape), C.POINTER(CflNorm), C.POINTER(CflLossCfg),
        C.POINTER(C.c_void_p), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_size_t,
        C.c_void_p]
    L.cfl_pair_train_step.restype = C.c_int
    a = list(L.cfl_pair_train_step.argtypes)
    L.cfl_pair_train_step_planes.argtypes = a[:14] + [C.POINTER(CflThetaPlanes)] + a[14:]
    L.cfl_pair_train_step_planes.restype = C.c_int
    L.cfl_theta_planes_bytes.argtypes = [C.POINTER(CflShape)]
    L.cfl_theta_planes_bytes.restype = C.c_size_t
    L.cfl_plan_describe.argtypes = [C.POINTER(CflShape), C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                    C.POINTER(CflPlanInfo)]
    L.cfl_plan_describe.restype = C.c_int
    L.cfl_scalars_status.argtypes = [C.c_void_p]
    L.cfl_scalars_status.restype = C.c_int
    L.cfl_pair_scores_idx.argtypes = [
        C.POINTER(CflShape), C.POINTER(CflNorm), C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.c_int64,
        C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.cfl_pair_scores_idx.restype = C.c_int
    L.cfl_pair_scores_idx4.argtypes = L.cfl_pair_scores_idx.argtypes
    L.cfl_pair_scores_idx4.restype = C.c_int
    L.cfl_pair_step_fwd_bwd_idx.argtypes = [
        C.POINTER(CflShape), C.POINTER(CflNorm), C.POINTER(CflLossCfg), C.c_void_p, C.c_int64,
        C.POINTER(C.c_void_p), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
        C.c_void_p]
    L.cfl_pair_step_fwd_bwd_idx.restype = C.c_int
    a = list(L.cfl_pair_step_fwd_bwd_idx.argtypes)
    L.cfl_pair_step_fwd_bwd_idx_planes.argtypes = a[:11] + [C.POINTER(CflThetaPlanes)] + a[11:]
    L.cfl_pair_step_fwd_bwd_idx_planes.restype = C.c_int
    L.cfl_pair_train_step_idx.argtypes = [
        C.POINTER(CflShape), C.POINTER(CflNorm), C.POINTER(CflLossCfg), C.c_void_p, C.c_int64,
        C.POINTER(C.c_void_p), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]
    L.cfl_pair_train_step_idx.restype = C.c_int
    a = list(L.cfl_pair_train_step_idx.argtypes)
    L.cfl_pair_train_step_idx_planes.argtypes = a[:17] + [C.POINTER(CflThetaPlanes)] + a[17:]
    L.cfl_pair_train_step_idx_planes.restype = C.c_int
    L.cfl_pair_train_steps_idx.argtypes = [
        C.POINTER(CflShape), C.POINTER(CflNorm), C.POINTER(CflLossCfg), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
        C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_char_p, C.c_int64, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
        C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_size_t, C.c_void_p]
    L.cfl_pair_train_steps_idx.restype = C.c_int
    a = list(L.cfl_pair_train_steps_idx.argtypes)
    L.cfl_pair_train_steps_idx_planes.argtypes = a[:27] + [C.POINTER(CflThetaPlanes)] + a[27:]
    L.cfl_pair_train_steps_idx_planes.restype = C.c_int
    L.cfl_train_val_fusable.argtypes = [C.POINTER(CflShape), C.c_int64, C.c_int64]
    L.cfl_train_val_fusable.restype = C.c_int
    L.cfl_pair_train_val_steps_idx_planes.argtypes = [
        C.POINTER(CflShape), C.POINTER(CflNorm), C.POINTER(CflLossCfg), C.c_void_p, C.c_int64,
        C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_char_p, C.c_int64,
        C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
        C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p),
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
        C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(CflThetaPlanes), C.c_void_p, C.c_size_t, C.c_void_p]
    L.cfl_pair_train_val_steps_idx_planes.restype = C.c_int
    L.cfl_mt19937_reshuffle.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int64, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
    L.cfl_mt19937_reshuffle.restype = C.c_int
    L.cfl_reload_env.restype = C.c_int
    L.cfl_adam_tf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_int64, C.c_float, C.c_float, C.c_float,
                              C.c_float, C.c_float, C.c_void_p]
    L.cfl_adam_tf.restype = C.c_int
    L.cfl_adam_tf_planes.argtypes = [C.POINTER(CflShape), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.POINTER(CflThetaPlanes), C.c_void_p]
    L.cfl_adam_tf_planes.restype = C.c_int
    L.cfl_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                  C.c_void_p, C.c_void_p]
    L.cfl_gather_rows.restype = C.c_int
    L.cfl_pair_input_grad.argtypes = [C.POINTER(CflShape), C.POINTER(CflNorm), C.c_int64, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cfl_pair_input_grad.restype = C.c_int
    L.cfl_conv_workspace_bytes.argtypes = [C.POINTER(CflConv)]
    L.cfl_conv_workspace_bytes.restype = C.c_size_t
    L.cfl_conv_uses_direct_kernel.argtypes = [C.POINTER(CflConv), C.c_int]
    L.cfl_conv2d_wn_fwd.argtypes = [C.POINTER(CflConv)] + [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p]
    L.cfl_conv2d_wn_fwd.restype = C.c_int
    L.cfl_conv2d_wn_bwd.argtypes = ([C.POINTER(CflConv)] + [C.c_void_p] * 5 + [C.c_float] +
                                    [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p])
    L.cfl_conv2d_wn_bwd.restype = C.c_int
    L.cfl_dp_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_int32]
    L.cfl_dp_free.argtypes = [C.c_void_p]
    L.cfl_dp_ipc_export.argtypes = [C.c_void_p, C.c_void_p]
    L.cfl_dp_ipc_open.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.cfl_dp_ipc_close.argtypes = [C.c_void_p]
    L.cfl_dp_rs_push.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32,
                                 C.c_uint32, C.c_void_p, C.c_void_p]
    L.cfl_dp_rs_adam.argtypes = ([C.c_void_p] * 5 + [C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                                     C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)] + [C.c_float] * 4 +
                                 [C.c_uint32, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p])
    L.cfl_dp_rs_gather.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_uint32,
                                                      C.c_void_p, C.c_double, C.c_void_p]
    a = list(L.cfl_dp_rs_gather.argtypes)
    L.cfl_dp_rs_gather_planes.argtypes = [C.POINTER(CflShape)] + a[:12] + [C.POINTER(CflThetaPlanes)] + a[12:]
    for f in (L.cfl_dp_alloc, L.cfl_dp_free, L.cfl_dp_ipc_export, L.cfl_dp_ipc_open, L.cfl_dp_ipc_close, L.cfl_dp_rs_push,
              L.cfl_dp_rs_adam, L.cfl_dp_rs_gather, L.cfl_dp_rs_gather_planes):
        f.restype = C.c_int
    # ABI 6: the whole data-parallel step behind one call
    DPX, ARP, TP = C.POINTER(CflDpExchange), C.POINTER(CflAllReduce), C.POINTER(CflThetaPlanes)
    L.cfl_dp_exchange_step.argtypes = [C.POINTER(CflShape), DPX, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_float, C.c_float, C.c_float, C.c_float, TP, C.c_void_p, C.c_void_p]
    L.cfl_dp_push_fusable.argtypes = [C.POINTER(CflShape), C.c_int64]
    L.cfl_workspace_bytes_val.argtypes = [C.POINTER(CflShape), C.c_int64, C.c_int64]
    L.cfl_workspace_bytes_val.restype = C.c_size_t
    L.cfl_pair_dp_step_planes.argtypes = [
        C.POINTER(CflShape), C.POINTER(CflNorm), C.POINTER(CflLossCfg), C.POINTER(C.c_void_p), C.c_int64, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, TP, DPX, ARP, C.c_void_p, C.c_size_t, C.c_void_p]
    L.cfl_pair_dp_step_idx_planes.argtypes = [
        C.POINTER(CflShape), C.POINTER(CflNorm), C.POINTER(CflLossCfg), C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.c_int64,
        C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, TP, DPX, ARP,
        C.c_void_p, C.c_size_t, C.c_void_p]
    L.cfl_pair_dp_steps_idx_planes.argtypes = [
        C.POINTER(CflShape), C.POINTER(CflNorm), C.POINTER(CflLossCfg), C.c_void_p, C.c_int64,
        C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_char_p, C.c_int64,
        C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
        C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p),
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
        C.POINTER(C.c_float), C.POINTER(C.c_float), TP, DPX, ARP, C.c_void_p, C.c_size_t, C.c_void_p]
    for f in (L.cfl_dp_exchange_step, L.cfl_dp_push_fusable, L.cfl_pair_dp_step_planes, L.cfl_pair_dp_step_idx_planes,
              L.cfl_pair_dp_steps_idx_planes):
        f.restype = C.c_int
    L.cfl_profile_enable.argtypes = [C.c_int]
    L.cfl_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    if L.cfl_version() != 6:
        raise CflHipError('libcfl_hip.so ABI version mismatch')
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise CflHipError('libcfl_hip error {}: {}'.format(
            rc, lib().cfl_last_error().decode()))


def check_scalars(host_scalars):
    """Raise CflHipError if a read-back scalars array carries the sticky error word (a training kernel gave up on an
    in-launch hand-off; the parameters are NaN-poisoned from that step on).  cfl_scalars_status of the C ABI."""
    a = np.ascontiguousarray(np.asarray(host_scalars, dtype=np.float32)[:S_COUNT])
    _check(lib().cfl_scalars_status(a.ctypes.data))


class ThetaPlanes(object):
    """Caller-kept bf16 planes of theta (CflThetaPlanes): a device buffer beside theta, m, v that the fused training
    step keeps current, so that the bf16x3 projection of the next step needs no per-call split of the weights."""

    def __init__(self, shape, device):
        n = lib().cfl_theta_planes_bytes(C.byref(shape))
        if n == 0:
            raise CflHipError('cfl_theta_planes_bytes: ' + lib().cfl_last_error().decode())
        self.buf = torch.empty(n // 2, dtype=torch.int16, device=device)
        self.c = CflThetaPlanes(self.buf.data_ptr(), 0)

    def invalidate(self):
        """theta was changed by something other than the fused training step"""
        self.c.valid = 0

    @property
    def valid(self):
        return bool(self.c.valid)


def _planes(p):
    return C.byref(p.c) if p is not None else None


def make_shape(D, L, K, dist_type='pcd', weight_norm=False, has_bias=True,
               act_type=None, directed=False):
    return CflShape(int(D), int(L), int(K), DIST_TYPES[dist_type],
                    int(bool(weight_norm)), int(bool(has_bias)),
                    ACT_TYPES[act_type], int(bool(directed)))


def make_norm(mul=1.0, add=0.0, lo=None, hi=None, valid_cols=0):
    """valid_cols: true feature width when the rows are zero-padded to a multiple of 64 (0 = no padding)."""
    return CflNorm(float(mul), float(add), float(lo if lo is not None else 0.0),
                   float(hi if hi is not None else 0.0), int(lo is not None),
                   int(hi is not None), int(valid_cols))


def make_loss(use_threshold=True, pos_weight=None, caffe_margin=None,
              lambda_m=0.0, reg_const=0.0):
    return CflLossCfg(int(bool(use_threshold)), float(pos_weight or 0.0),
                      float(caffe_margin or 0.0), float(lambda_m or 0.0),
                      float(reg_const or 0.0))


def layout(shape):
    lay = CflLayout()
    _check(lib().cfl_layout(C.byref(shape), C.byref(lay)))
    return lay


def plan_describe(shape, rows, groups=2, train=True, planes_kept=True):
    """The kernels and launch geometry of a call of this shape, from the library's own planner (cfl_plan_describe):
    a dict of kernel names (proj / mid / grad / tail) and integers (launches, S, P, workgroups, ...).  Host-only."""
    info = CflPlanInfo()
    _check(lib().cfl_plan_describe(C.byref(shape), int(rows), int(groups), int(bool(train)), int(bool(planes_kept)),
                                   C.byref(info)))
    out = {}
    for name, _ in CflPlanInfo._fields_:
        v = getattr(info, name)
        out[name] = v.decode() if isinstance(v, bytes) else int(v)
    return out


def workspace_bytes(shape, rows, groups):
    n = lib().cfl_workspace_bytes(C.byref(shape), int(rows), int(groups))
    if n == 0:
        raise CflHipError('cfl_workspace_bytes: ' + lib().cfl_last_error().decode())
    return n


def workspace_bytes_val(shape, rows, val_rows):
    """workspace of a training call of `rows` rows per group that carries `val_rows` validation pairs per group"""
    n = lib().cfl_workspace_bytes_val(C.byref(shape), int(rows), int(val_rows))
    if n == 0:
        raise CflHipError('cfl_workspace_bytes_val: ' + lib().cfl_last_error().decode())
    return n


def _dev(t, dtype=torch.float32):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise CflHipError('cfl hot path needs CUDA/HIP tensors (no CPU fallback)')
    if t.dtype != dtype or not t.is_contiguous():
        raise CflHipError('expected contiguous {} tensor'.format(dtype))
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """the HIP stream torch would launch on now, as an integer handle (torch.cuda.current_stream().cuda_stream costs ~8 us of
    Python per call -- 400 calls per MrCGAN step; the raw getter a fraction of a microsecond)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def pair_scores(shape, norm, xs, xt, theta, workspace, scores=None, dists=None):
    n = xs.shape[0]
    if scores is None:
        scores = torch.empty(n, dtype=torch.float32, device=xs.device)
    _check(lib().cfl_pair_scores(
        C.byref(shape), C.byref(norm), _dev(xs), _dev(xt), n, _dev(theta),
        _dev(scores), _dev(dists) if dists is not None else None,
        workspace.data_ptr(), workspace.numel() * workspace.element_size(), _stream()))
    return scores


def pair_step_fwd_bwd(shape, norm, loss, x4, theta, grad, scalars, workspace, planes=None):
    """planes: the caller's kept plane buffer (read; split from theta first when stale) -- the data-parallel step"""
    B = x4[0].shape[0]
    arr = (C.c_void_p * 4)(*[_dev(x) for x in x4])
    _check(lib().cfl_pair_step_fwd_bwd_planes(
        C.byref(shape), C.byref(norm), C.byref(loss), arr, B, _dev(theta), _dev(grad),
        _dev(scalars), _planes(planes), workspace.data_ptr(),
        workspace.numel() * workspace.element_size(), _stream()))


def pair_train_step(shape, norm, loss, x4, theta, m, v, grad, scalars, workspace, lr_t,
                    beta1, beta2, eps=1e-8, planes=None):
    B = x4[0].shape[0]
    arr = (C.c_void_p * 4)(*[_dev(x) for x in x4])
    _check(lib().cfl_pair_train_step_planes(
        C.byref(shape), C.byref(norm), C.byref(loss), arr, B, _dev(theta), _dev(m), _dev(v),
        _dev(grad), _dev(scalars), float(lr_t), float(beta1), float(beta2), float(eps), _planes(planes),
        workspace.data_ptr(), workspace.numel() * workspace.element_size(), _stream()))


class IndexStreams(object):
    """The index side of an indexed call: `ptrs` = device addresses of the 2 (scoring) or 4 (training) int32 index
    streams, `stride` in elements (2 walks one column of an [n, 2] pair array in place), `n` rows per stream.
    `keep` holds the tensors the addresses point into."""

    def __init__(self, ptrs, stride, n, keep=None):
        self.arr = (C.c_void_p * len(ptrs))(*[int(p) for p in ptrs])
        self.stride, self.n, self.keep = int(stride), int(n), keep

    def pair(self, k):
        """the 2 streams (src, dst) of pair group k of a 4-stream training batch (0 = positive, 1 = negative)"""
        return IndexStreams([self.arr[2 * k], self.arr[2 * k + 1]], self.stride, self.n, keep=self.keep)

    @classmethod
    def from_tensors(cls, tensors):
        """dense int32 device vectors, one per stream"""
        for t in tensors:
            _dev(t, torch.int32)
        return cls([t.data_ptr() for t in tensors], 1, tensors[0].shape[0], keep=list(tensors))


def _table(table):
    _dev(table)
    return table.data_ptr(), table.shape[0]


def pair_scores_idx(shape, norm, table, streams, theta, workspace, scores=None, dists=None):
    """pair_scores with rows src = table[idx0], dst = table[idx1] (cfl_pair_scores_idx)."""
    if scores is None:
        scores = torch.empty(streams.n, dtype=torch.float32, device=table.device)
    tp, rows = _table(table)
    _check(lib().cfl_pair_scores_idx(
        C.byref(shape), C.byref(norm), tp, rows, streams.arr, streams.stride, streams.n, _dev(theta),
        _dev(scores), _dev(dists) if dists is not None else None, workspace.data_ptr(),
        workspace.numel() * workspace.element_size(), _stream()))
    return scores


def pair_scores_idx4(shape, norm, table, streams, theta, workspace, scores=None):
    """Scores of the (stream 0, stream 1) pairs followed by the (stream 2, stream 3) pairs in one call
    (cfl_pair_scores_idx4): [2 n]."""
    if scores is None:
        scores = torch.empty(2 * streams.n, dtype=torch.float32, device=table.device)
    tp, rows = _table(table)
    _check(lib().cfl_pair_scores_idx4(
        C.byref(shape), C.byref(norm), tp, rows, streams.arr, streams.stride, streams.n, _dev(theta),
        _dev(scores), None, workspace.data_ptr(), workspace.numel() * workspace.element_size(), _stream()))
    return scores


def pair_step_fwd_bwd_idx(shape, norm, loss, table, streams, theta, grad, scalars, workspace, planes=None):
    tp, rows = _table(table)
    _check(lib().cfl_pair_step_fwd_bwd_idx_planes(
        C.byref(shape), C.byref(norm), C.byref(loss), tp, rows, streams.arr, streams.stride, streams.n,
        _dev(theta), _dev(grad), _dev(scalars), _planes(planes), workspace.data_ptr(),
        workspace.numel() * workspace.element_size(), _stream()))


def pair_train_step_idx(shape, norm, loss, table, streams, theta, m, v, grad, scalars, workspace, lr_t, beta1,
                        beta2, eps=1e-8, planes=None):
    tp, rows = _table(table)
    _check(lib().cfl_pair_train_step_idx_planes(
        C.byref(shape), C.byref(norm), C.byref(loss), tp, rows, streams.arr, streams.stride, streams.n,
        _dev(theta), _dev(m), _dev(v), _dev(grad), _dev(scalars), float(lr_t), float(beta1), float(beta2),
        float(eps), _planes(planes), workspace.data_ptr(), workspace.numel() * workspace.element_size(), _stream()))


def pair_train_steps_idx(shape, norm, loss, table, pos_pairs, neg_pairs, pos_head, neg_head, batch_rows, shard_lo,
                         rows, switched, nsteps, theta, m, v, grad, scalars, workspace, lr, beta1, beta2, eps,
                         beta1_power, beta2_power, planes=None):
    """nsteps consecutive training steps over windows of the device pair lists (cfl_pair_train_steps_idx).
    Returns the advanced (beta1_power, beta2_power) as python floats holding float32 values."""
    tp, trows = _table(table)
    b1p, b2p = C.c_float(beta1_power), C.c_float(beta2_power)
    sw = bytes(bytearray(int(bool(x)) for x in switched)) if switched is not None else None
    _check(lib().cfl_pair_train_steps_idx_planes(
        C.byref(shape), C.byref(norm), C.byref(loss), tp, trows, _dev(pos_pairs, torch.int32), int(pos_pairs.shape[0]),
        _dev(neg_pairs, torch.int32), int(neg_pairs.shape[0]), int(pos_head), int(neg_head), int(batch_rows),
        int(shard_lo), int(rows), sw,
        int(nsteps), _dev(theta), _dev(m), _dev(v), _dev(grad), _dev(scalars), float(lr), float(beta1),
        float(beta2), float(eps), C.byref(b1p), C.byref(b2p), _planes(planes), workspace.data_ptr(),
        workspace.numel() * workspace.element_size(), _stream()))
    return b1p.value, b2p.value


def train_val_fusable(shape, rows, val_rows):
    """can a training step of `rows` rows per group carry a validation batch of `val_rows` pairs per group as extra
    scoring rows of its own launches (cfl_pair_train_val_steps_idx_planes)?"""
    return lib().cfl_train_val_fusable(C.byref(shape), int(rows), int(val_rows)) == 1


def _flags(xs):
    return bytes(bytearray(int(bool(x)) for x in xs)) if xs is not None else None


def pair_train_val_steps_idx(shape, norm, loss, win, vwin, val_mask, slot_ptrs, theta, m, v, grad, scalars, workspace, lr,
                             beta1, beta2, eps, beta1_power, beta2_power, planes=None):
    """win.nsteps training steps over windows of the device pair lists; the steps with val_mask[i] set also score the next
    validation batch of `vwin` inside their own launches and leave [scalars | scores] in the next of `slot_ptrs`
    (device-accessible addresses, e.g. rows of a pinned host tensor).  Returns the advanced (beta1_power, beta2_power)."""
    tp, trows = _table(win.table)
    vp, vrows = _table(vwin.table)
    b1p, b2p = C.c_float(beta1_power), C.c_float(beta2_power)
    slots = (C.c_void_p * max(1, len(slot_ptrs)))(*[int(p) for p in slot_ptrs])
    _check(lib().cfl_pair_train_val_steps_idx_planes(
        C.byref(shape), C.byref(norm), C.byref(loss), tp, trows, _dev(win.pos_pairs, torch.int32), int(win.pos_pairs.shape[0]),
        _dev(win.neg_pairs, torch.int32), int(win.neg_pairs.shape[0]), int(win.pos_head), int(win.neg_head),
        int(win.batch_rows), int(win.shard_lo), int(win.rows), _flags(win.switched), int(win.nsteps),
        vp, vrows, _dev(vwin.pos_pairs, torch.int32), int(vwin.pos_pairs.shape[0]), _dev(vwin.neg_pairs, torch.int32),
        int(vwin.neg_pairs.shape[0]), int(vwin.pos_head), int(vwin.neg_head), int(vwin.batch_rows), _flags(vwin.switched),
        _flags(val_mask), slots, _dev(theta), _dev(m), _dev(v), _dev(grad), _dev(scalars), float(lr), float(beta1),
        float(beta2), float(eps), C.byref(b1p), C.byref(b2p), _planes(planes), workspace.data_ptr(),
        workspace.numel() * workspace.element_size(), _stream()))
    return b1p.value, b2p.value


# ---- ABI 6: one data-parallel step / K windowed steps behind ONE library call ------------------------------------------
def _exchange_args(ex, ar):
    """(CflDpExchange* | None, CflAllReduce* | None) for the C ABI; both None: no exchange"""
    return (C.byref(ex) if ex is not None else None), (C.byref(ar) if ar is not None else None)


def pair_dp_step(shape, norm, loss, batch, theta, m, v, gradbuf, workspace, lr_t, beta1, beta2, eps=1e-8, planes=None, ex=None,
                 ar=None):
    """forward / backward of this rank's rows, the gradient exchange (ex: one-shot exchange with the push fused into the
    weight-gradient launch; ar: a caller-supplied all-reduce enqueued on the launch stream) and TF-Adam + planes: one call.
    batch = 4 dense device tensors or (table, IndexStreams)."""
    if gradbuf.numel() != theta.numel() + GRADBUF_PAD:
        raise CflHipError('gradbuf must hold the parameter count + %d floats' % GRADBUF_PAD)
    exp, arp = _exchange_args(ex, ar)
    wsb = workspace.numel() * workspace.element_size()
    if isinstance(batch[1], IndexStreams):
        tp, rows = _table(batch[0])
        st = batch[1]
        _check(lib().cfl_pair_dp_step_idx_planes(
            C.byref(shape), C.byref(norm), C.byref(loss), tp, rows, st.arr, st.stride, st.n, _dev(theta), _dev(m), _dev(v),
            _dev(gradbuf), float(lr_t), float(beta1), float(beta2), float(eps), _planes(planes), exp, arp,
            workspace.data_ptr(), wsb, _stream()))
    else:
        arr = (C.c_void_p * 4)(*[_dev(x) for x in batch])
        _check(lib().cfl_pair_dp_step_planes(
            C.byref(shape), C.byref(norm), C.byref(loss), arr, batch[0].shape[0], _dev(theta), _dev(m), _dev(v), _dev(gradbuf),
            float(lr_t), float(beta1), float(beta2), float(eps), _planes(planes), exp, arp, workspace.data_ptr(), wsb,
            _stream()))


def pair_dp_steps_idx(shape, norm, loss, win, theta, m, v, gradbuf, workspace, lr, beta1, beta2, eps, beta1_power, beta2_power,
                      planes=None, ex=None, ar=None, vwin=None, val_mask=None, slot_ptrs=None):
    """win.nsteps data-parallel iterations over windows of the device pair lists in ONE call (cfl_pair_dp_steps_idx_planes); with
    vwin / val_mask / slot_ptrs the masked iterations carry the validation fetch.  Returns the advanced Adam powers."""
    if gradbuf.numel() != theta.numel() + GRADBUF_PAD:
        raise CflHipError('gradbuf must hold the parameter count + %d floats' % GRADBUF_PAD)
    tp, trows = _table(win.table)
    b1p, b2p = C.c_float(beta1_power), C.c_float(beta2_power)
    exp, arp = _exchange_args(ex, ar)
    if vwin is not None:
        vp, vrows = _table(vwin.table)
        slots = (C.c_void_p * max(1, len(slot_ptrs)))(*[int(p) for p in slot_ptrs])
        val = (vp, vrows, _dev(vwin.pos_pairs, torch.int32), int(vwin.pos_pairs.shape[0]), _dev(vwin.neg_pairs, torch.int32),
               int(vwin.neg_pairs.shape[0]), int(vwin.pos_head), int(vwin.neg_head), int(vwin.batch_rows), _flags(vwin.switched),
               _flags(val_mask), slots)
    else:
        val = (None, 0, None, 0, None, 0, 0, 0, 0, None, None, None)
    _check(lib().cfl_pair_dp_steps_idx_planes(
        C.byref(shape), C.byref(norm), C.byref(loss), tp, trows, _dev(win.pos_pairs, torch.int32), int(win.pos_pairs.shape[0]),
        _dev(win.neg_pairs, torch.int32), int(win.neg_pairs.shape[0]), int(win.pos_head), int(win.neg_head),
        int(win.batch_rows), int(win.shard_lo), int(win.rows), _flags(win.switched), int(win.nsteps), *val,
        _dev(theta), _dev(m), _dev(v), _dev(gradbuf), float(lr), float(beta1), float(beta2), float(eps), C.byref(b1p),
        C.byref(b2p), _planes(planes), exp, arp, workspace.data_ptr(), workspace.numel() * workspace.element_size(), _stream()))
    return b1p.value, b2p.value


def dp_push_fusable(shape, rows):
    """does a training call of this shape push its gradient from inside the weight-gradient launch (fused-tail plans)?"""
    return lib().cfl_dp_push_fusable(C.byref(shape), int(rows)) == 1


def mt19937_reshuffle(state, rows, want32=False):
    """(rows[perm], new_state[, int32 copy]) with perm = RandomState.permutation(len(rows)) drawn from the legacy
    generator `state` (the tuple of RandomState.get_state()); host-only, runs without the interpreter lock."""
    key = np.array(state[1], dtype=np.uint32)
    pos = C.c_int32(int(state[2]))
    in_dtype = np.asarray(rows).dtype
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    flat = rows.reshape(rows.shape[0], -1)
    out = np.empty_like(flat)
    out32 = np.empty(flat.shape, dtype=np.int32) if want32 else None
    _check(lib().cfl_mt19937_reshuffle(key.ctypes.data, C.byref(pos), flat.shape[0], flat.ctypes.data, flat.shape[1],
                                       out.ctypes.data, None, out32.ctypes.data if want32 else None))
    new_state = (state[0], key, int(pos.value), state[3], state[4])
    res = out.reshape(rows.shape)
    if in_dtype != np.int64:            # pairs[perm] keeps the dtype of `pairs`
        res = res.astype(in_dtype)
    if want32:
        return res, new_state, out32.reshape(rows.shape)
    return res, new_state


def reload_env():
    """Re-read CFL_EXACT_FP32 / CFL_DEBUG_* (the launch plans are cached per process otherwise)."""
    return lib().cfl_reload_env()


def adam_tf_planes(shape, theta, m, v, grad, lr_t, beta1, beta2, eps=1e-8, grad_scale=1.0, planes=None):
    """TF-Adam over the whole theta of `shape` that also writes the kept bf16 planes of the updated weights
    (cfl_adam_tf_planes: the update of a data-parallel step)."""
    if theta.numel() != layout(shape).total or grad.numel() < theta.numel():
        raise CflHipError('adam_tf_planes: theta / grad do not have the parameter count of the shape')
    _check(lib().cfl_adam_tf_planes(C.byref(shape), _dev(theta), _dev(m), _dev(v), _dev(grad), float(lr_t),
                                    float(beta1), float(beta2), float(eps), float(grad_scale), _planes(planes),
                                    _stream()))


def adam_tf(theta, m, v, grad, lr_t, beta1, beta2, eps=1e-8, grad_scale=1.0):
    _check(lib().cfl_adam_tf(_dev(theta), _dev(m), _dev(v), _dev(grad), theta.numel(),
                             float(lr_t), float(beta1), float(beta2), float(eps),
                             float(grad_scale), _stream()))


def pair_input_grad(shape, norm, B, theta, workspace, dx_src=None, dx_dst=None):
    """dL/d(input rows) of the last cfl_pair_step_fwd_bwd on this workspace: two [2B, D]
    tensors (source side, destination side; positive rows first)."""
    D = shape.D
    if dx_src is None:
        dx_src = torch.empty(2 * B, D, dtype=torch.float32, device=theta.device)
    if dx_dst is None:
        dx_dst = torch.empty(2 * B, D, dtype=torch.float32, device=theta.device)
    _check(lib().cfl_pair_input_grad(C.byref(shape), C.byref(norm), B, _dev(theta), workspace.data_ptr(),
                                     workspace.numel() * workspace.element_size(), _dev(dx_src),
                                     _dev(dx_dst), _stream()))
    return dx_src, dx_dst


def make_conv(B, H, W, Ci, Co, KH, KW, stride, act=None):
    return CflConv(B, H, W, Ci, Co, KH, KW, stride, CONV_ACTS[act])


def conv_out_hw(conv):
    return -(-conv.H // conv.stride), -(-conv.W // conv.stride)


def conv_uses_direct_kernel(conv, product):
    """product 'fwd' / 'dx' / 'dw': does this shape run on the 3x3 halo-tile kernels (csrc/conv_halo*.h)?  'stem': is it the
    3-channel 4x4 stride-2 image stem with kernels of its own (csrc/conv_stem.h)?"""
    return lib().cfl_conv_uses_direct_kernel(C.byref(conv), {'fwd': 0, 'dx': 1, 'dw': 2, 'stem': 3}[product]) == 1


def conv_workspace(conv, device):
    n = lib().cfl_conv_workspace_bytes(C.byref(conv))
    if n == 0:
        raise CflHipError('cfl_conv_workspace_bytes: ' + lib().cfl_last_error().decode())
    return torch.empty((n + 3) // 4, dtype=torch.float32, device=device)


def conv2d_wn_fwd(conv, x, V, g, b, ws, y=None):
    """x [B,H,W,Ci] NHWC, V [KH,KW,Ci,Co] HWIO -> y [B,OH,OW,Co]."""
    oh, ow = conv_out_hw(conv)
    if y is None:
        y = torch.empty(conv.B, oh, ow, conv.Co, dtype=torch.float32, device=x.device)
    _check(lib().cfl_conv2d_wn_fwd(C.byref(conv), _dev(x), _dev(V), _dev(g) if g is not None else None,
                                   _dev(b) if b is not None else None, _dev(y), ws.data_ptr(),
                                   ws.numel() * 4, _stream()))
    return y


def conv2d_wn_bwd(conv, x, V, g, y, dy, ws, reg_const=0.0, need_dx=True, need_db=True):
    dx = torch.empty_like(x) if need_dx else None
    dV = torch.empty_like(V)
    dg = torch.empty_like(g) if g is not None else None
    db = torch.empty(conv.Co, dtype=torch.float32, device=x.device) if need_db else None
    _check(lib().cfl_conv2d_wn_bwd(C.byref(conv), _dev(x), _dev(V), _dev(g) if g is not None else None,
                                   _dev(y), _dev(dy), float(reg_const),
                                   _dev(dx) if dx is not None else None, _dev(dV),
                                   _dev(dg) if dg is not None else None,
                                   _dev(db) if db is not None else None, ws.data_ptr(), ws.numel() * 4,
                                   _stream()))
    return dx, dV, dg, db


def profile_enable(on):
    lib().cfl_profile_enable(int(bool(on)))


def profile_read():
    """{kernel: (total_ms, launches)} since the last read (synchronises)."""
    ms = (C.c_double * K_COUNT)()
    cnt = (C.c_int64 * K_COUNT)()
    _check(lib().cfl_profile_read(ms, cnt))
    return {KERNEL_NAMES[i]: (ms[i], cnt[i]) for i in range(len(KERNEL_NAMES)) if cnt[i]}


def gather_rows(table, idx, out=None):
    n, D = idx.shape[0], table.shape[1]
    if out is None:
        out = torch.empty(n, D, dtype=torch.float32, device=table.device)
    _check(lib().cfl_gather_rows(_dev(table), _dev(idx, torch.int64), n, D, _dev(out),
                                 _stream()))
    return out


# ---------------------------------------------------------------------------
# theta <-> named variables in the reference (TensorFlow) layout [D, N]
# ---------------------------------------------------------------------------
def wt_to_frag(wt):
    """Wt[npad][D] -> fragment-major Wf[nt][g][q][c16][e] (see csrc/cfl_hip.hip):
    column c = 16nt + c16, d = 16g + 4q + e; one contiguous 1 KiB block per (nt, g)."""
    npad, D = wt.shape
    return wt.reshape(npad // 16, 16, D // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous()


def frag_to_wt(flat, npad, D):
    return flat.reshape(npad // 16, D // 16, 4, 16, 4).permute(0, 3, 1, 2, 4).reshape(npad, D)


def head_views(lay, shape, e):
    """Yield (scope-relative variable name, kind, CflHead) for encoder e."""
    enc = lay.enc[e]
    for name, h in (('outputs', enc.outputs), ('proto', enc.proto), ('mono', enc.mono)):
        if h.w >= 0:
            yield name, h


def pack_theta(shape, params, params_dst=None, thr=1e-6, device='cpu'):
    """Build the flat device array from {'outputs/W': [D,L], 'outputs/b': ..,
    'outputs/g': .., 'proto/W': [D,K*L], .., 'mono/W': [L,K], 'mono/g'} dicts."""
    lay = layout(shape)
    theta = torch.zeros(lay.total, dtype=torch.float32)
    D = shape.D
    for e, p in enumerate((params, params_dst)):
        if p is None:
            continue
        for name, h in head_views(lay, shape, e):
            W = torch.as_tensor(np.asarray(p[name + '/W'], dtype=np.float32))
            if name == 'mono':
                blk = theta[h.w:h.w + shape.L * h.npad].view(shape.L, h.npad)
                blk[:, :h.n] = W
            else:
                wt = torch.zeros(h.npad, D, dtype=torch.float32)
                wt[:h.n] = W.t()
                theta[h.w:h.w + h.npad * D] = wt_to_frag(wt).reshape(-1)
            if h.b >= 0:
                theta[h.b:h.b + h.n] = torch.as_tensor(np.asarray(p[name + '/b'], dtype=np.float32))
            if h.g >= 0:
                theta[h.g:h.g + h.n] = torch.as_tensor(np.asarray(p[name + '/g'], dtype=np.float32))
    theta[lay.thr] = float(thr)
    return theta.to(device)


def unpack_theta(shape, theta):
    """Inverse of pack_theta: (params, params_dst|None, thr) as numpy arrays."""
    lay = layout(shape)
    th = theta.detach().cpu()
    D = shape.D
    out = []
    for e in range(2 if shape.directed else 1):
        p = {}
        for name, h in head_views(lay, shape, e):
            if name == 'mono':
                p['mono/W'] = th[h.w:h.w + shape.L * h.npad].view(shape.L, h.npad)[:, :h.n].numpy().copy()
            else:
                wt = frag_to_wt(th[h.w:h.w + h.npad * D], h.npad, D)
                p[name + '/W'] = wt[:h.n].t().contiguous().numpy().copy()
            if h.b >= 0:
                p[name + '/b'] = th[h.b:h.b + h.n].numpy().copy()
            if h.g >= 0:
                p[name + '/g'] = th[h.g:h.g + h.n].numpy().copy()
        out.append(p)
    return out[0], (out[1] if shape.directed else None), float(th[lay.thr])
