"""PairEngine: device state + step driver of the pair-distance hot path.

Owns the flat parameter array ``theta`` (layout: include/cfl_hip.h), its Adam
slots, the flat gradient, the scalar read-back buffer and the scratch workspace,
and drives one training step as

    single GPU      cfl_pair_train_step[_idx]_planes             (3 launches, TF-Adam fused into the last one)
    data parallel   cfl_pair_dp_step[s]_idx_planes (ABI 6): ONE library call per step / per K windowed steps runs
                    proj -> mid -> grad -> exchange -> update, where the exchange is
                      * RCCL's ncclAllReduce of [gradient | scalars], called BY the library on the launch stream between its
                        weight-gradient launch and cfl_adam_tf_planes (default; cfl/rccl.py hands over the entry point), or
                      * the one-shot exchange with the push fused into the weight-gradient launch (CFL_DP_EXCHANGE=oneshot:
                        cfl/dp_exchange.py, csrc/cfl_dp.hip): 5 launches per step, no collective library in the loop.
                    Without a raw RCCL communicator (gloo test mode, torch fallback): forward / backward, then
                    torch.distributed.all_reduce and cfl_adam_tf_planes from Python (`_exchange_and_update`).

which replaces the per-iteration ``sess.run([summary, [s_optim], ...])`` of
cfl/bin/train_dist.py:81-82 and cfl/models/cfl.py:1399-1414.  Data parallelism
(one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI) is new
functionality: the reference is single-device.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import hipabi as H


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def dp_active():
    """Does a training step take the data-parallel form (forward / backward -> exchange -> Adam)?  With more than one rank,
    and -- CFL_FORCE_DP=1 -- with a process group of ONE rank too: the one-GPU way to run the exact code path of a
    multi-GPU job, RCCL communicator, stream-ordered all-reduce of the gradient buffer and all (tests, bench.py --force-dp)."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('CFL_FORCE_DP') == '1'


def init_from_env():
    """Called by the CLI entry points before any model is built: caps the host thread pool (quiet_host_threads) and,
    under torchrun (WORLD_SIZE > 1 in the environment), binds this process to its GPU and joins the RCCL process
    group.  One process per GPU."""
    import os
    quiet_host_threads()
    forced = os.environ.get('CFL_FORCE_DP') == '1'
    if (int(os.environ.get('WORLD_SIZE', '1')) > 1 or forced) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if forced:      # a one-rank group outside torchrun
            os.environ.setdefault('MASTER_PORT', '29517')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('CFL_DIST_BACKEND', 'nccl')
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local if backend == 'nccl' else local % max(torch.cuda.device_count(), 1))
        if backend == 'nccl':
            dist.init_process_group(backend, device_id=torch.device('cuda', torch.cuda.current_device()))
        else:
            dist.init_process_group(backend)
    return world_size()


def prepare_run_dirs(paths, reset):
    """--reset and directory creation with several ranks: rank 0 clears and creates, the others wait."""
    import os
    import shutil
    if rank() == 0:
        for path in paths:
            if reset and os.path.exists(path):
                shutil.rmtree(path)
            os.makedirs(path, exist_ok=True)
    if world_size() > 1:
        dist.barrier()


def reduce_gradients(flat_grad):
    """The ONE exchange of a data-parallel step: sum the flat fp32
    buffer [gradient | loss / accuracy scalars] (same layout on every rank) over all ranks;
    returns the factor that turns the sums into the global-batch means.  Backend "nccl" is RCCL
    over xGMI on MI355X; the CPU tests run the same code over gloo."""
    n = world_size()
    if dp_active():
        comm = hot_communicator()
        if comm is not None and flat_grad.is_cuda:
            comm.all_reduce_sum_(flat_grad)      # ncclAllReduce on the launch stream itself (cfl/rccl.py)
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / n


def hot_communicator():
    """The RCCL communicator of the per-step exchange, or None (gloo test mode; CFL_DP_ALLREDUCE=torch: the collective
    goes through torch.distributed, i.e. ProcessGroupNCCL's own stream + two events per step).  COLLECTIVE on first use:
    PairEngine.__init__ calls it on every rank."""
    import os
    if not dp_active() or dist.get_backend() != 'nccl' or os.environ.get('CFL_DP_ALLREDUCE', 'direct') == 'torch':
        return None
    from . import rccl
    return rccl.default_communicator()


def finalize():
    """end of a run: destroy the hot communicator, then the process group (every rank)"""
    from . import rccl
    rccl.shutdown()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def quiet_host_threads(n=4):
    """The training loop's host side is one launching thread plus the reshuffle worker; torch's intra-op OpenMP pool
    defaults to one thread per core and its idle threads SPIN after every parallel region (a host-side tensor copy
    of a few hundred KB is enough).  On a many-core box inside a CPU-quota'd container that spinning alone exhausts
    the quota and the kernel then freezes every thread of the process -- including the one launching kernels -- for
    the rest of each 100 ms period (measured: 60 ms stalls, GPU idle).  The command-line entry points cap the pool
    unless the user chose a size (OMP_NUM_THREADS)."""
    import os
    if os.environ.get('OMP_NUM_THREADS'):
        return
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)


def shard_rows(n_rows, rank_=None, world=None):
    """Rows [lo, hi) of a global batch that belong to this rank: contiguous, equal
    slices, so the union over ranks is the single-GPU batch bit-exactly."""
    world = world_size() if world is None else world
    rank_ = rank() if rank_ is None else rank_
    if n_rows % world:
        raise ValueError('global batch %d is not divisible by world size %d' % (n_rows, world))
    per = n_rows // world
    return rank_ * per, (rank_ + 1) * per


class PairEngine(object):
    def __init__(self, D, L, K, dist_type='pcd', weight_norm=False, has_bias=True,
                 act_type=None, directed=False, norm=None, loss=None, lr=1e-3,
                 beta1=0.9, beta2=0.999, eps=1e-8, device='cuda', params=None,
                 params_dst=None, thr=1e-6, batch_size=None):
        H.lib()
        self.shape = H.make_shape(D, L, K, dist_type, weight_norm, has_bias, act_type, directed)
        self.layout = H.layout(self.shape)
        self.norm = norm if norm is not None else H.make_norm()
        self.loss = loss if loss is not None else H.make_loss()
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise H.CflHipError('PairEngine runs on a HIP device only (no CPU fallback)')
        self.theta = H.pack_theta(self.shape, params, params_dst, thr, self.device)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        # ONE flat buffer [gradient (layout of theta) | the step's loss / accuracy scalars | pad]: the kernels
        # write both parts, and the data-parallel exchange is a single all-reduce of it
        n = self.theta.numel()
        self.gradbuf = torch.zeros(n + H.GRADBUF_PAD, dtype=torch.float32, device=self.device)
        self.grad = self.gradbuf[:n]
        self.scalars = self.gradbuf[n:n + H.S_COUNT]
        self._scalar_scale = 1.0
        # TF keeps beta1_power / beta2_power as float32 variables
        self.beta1_power = np.float32(beta1)
        self.beta2_power = np.float32(beta2)
        self.global_step = 0
        self._ws = {}
        # bf16 planes of theta, kept current by the fused training step (single GPU): the bf16x3 projection of large
        # batches then needs no per-call split of the weights (include/cfl_hip.h: CflThetaPlanes)
        self.planes = H.ThetaPlanes(self.shape, self.device)
        if batch_size:
            self._workspace(batch_size, 2)
        # data-parallel exchange: RCCL all-reduce (default) or the library's one-shot push + summing Adam
        # (CFL_DP_EXCHANGE=oneshot: cfl/dp_exchange.py, csrc/cfl_dp.hip)
        self._oneshot = None
        import os
        self._comm = hot_communicator()    # (collective: created here, on every rank, not inside the first step)
        if dp_active() and os.environ.get('CFL_DP_EXCHANGE', 'allreduce') == 'oneshot':
            from .dp_exchange import OneShotExchange
            self._oneshot = OneShotExchange(self)   # (a one-rank group too: CFL_FORCE_DP, the one-GPU form of the same code)

    # -- plumbing ----------------------------------------------------------
    def _workspace(self, rows, groups, val_rows=0):
        """val_rows: the call also carries a validation batch of that many pairs per group (larger than `rows` under data
        parallelism: a rank trains its shard and scores the whole batch)"""
        key = (int(rows), int(groups)) if val_rows <= rows else (int(rows), int(groups), int(val_rows))
        ws = self._ws.get(key)
        if ws is None:
            n = H.workspace_bytes(self.shape, rows, groups) if val_rows <= rows else H.workspace_bytes_val(self.shape, rows, val_rows)
            ws = torch.empty(n // 4, dtype=torch.float32, device=self.device)
            # a few sizes per group count stay cached: under data parallelism the training step (rows = batch / world) and
            # the validation fetch (rows = batch) alternate, and evicting each other meant a reallocation per read-back
            same = [k for k in self._ws if k[1] == groups and len(k) == len(key)]
            for k in same[:max(0, len(same) - 2)]:
                del self._ws[k]
            self._ws[key] = ws
        return ws

    def lr_t(self):
        one = np.float32(1)
        return float(np.float32(self.lr) * np.sqrt(one - self.beta2_power) /
                     (one - self.beta1_power))

    @property
    def world_size(self):
        return world_size()

    # -- the hot path ------------------------------------------------------
    def fwd_bwd(self, batch):
        """batch = (pos_src, pos_dst, neg_src, neg_dst) device tensors [B, D], or (table, IndexStreams)."""
        if isinstance(batch[1], H.IndexStreams):
            ws = self._workspace(batch[1].n, 2)
            H.pair_step_fwd_bwd_idx(self.shape, self.norm, self.loss, batch[0], batch[1], self.theta, self.grad,
                                    self.scalars, ws)
        else:
            ws = self._workspace(batch[0].shape[0], 2)
            H.pair_step_fwd_bwd(self.shape, self.norm, self.loss, batch, self.theta,
                                self.grad, self.scalars, ws)
        self._scalar_scale = 1.0

    def apply_adam(self, grad_scale=1.0):
        H.adam_tf_planes(self.shape, self.theta, self.m, self.v, self.grad, self.lr_t(), self.beta1,
                         self.beta2, self.eps, grad_scale, planes=self.planes)
        self._advance()

    def _advance(self):
        self.beta1_power = np.float32(self.beta1_power * np.float32(self.beta1))
        self.beta2_power = np.float32(self.beta2_power * np.float32(self.beta2))
        self.global_step += 1

    def step(self, batch):
        """One training step on this rank's shard of the row batch: `batch` is either the 4 dense device
        tensors (pos_src, pos_dst, neg_src, neg_dst) or a (table, IndexStreams) pair -- rows picked from a
        resident feature table by index (cfl.input_data.ResidentFeatures.next_indexed)."""
        indexed = isinstance(batch[1], H.IndexStreams)
        rows = batch[1].n if indexed else batch[0].shape[0]
        ws = self._workspace(rows, 2)
        if not dp_active():
            # single GPU: Adam fused into the last kernel of the step
            if indexed:
                H.pair_train_step_idx(self.shape, self.norm, self.loss, batch[0], batch[1], self.theta, self.m,
                                      self.v, self.grad, self.scalars, ws, self.lr_t(), self.beta1, self.beta2,
                                      self.eps, planes=self.planes)
            else:
                H.pair_train_step(self.shape, self.norm, self.loss, batch, self.theta, self.m,
                                  self.v, self.grad, self.scalars, ws, self.lr_t(), self.beta1,
                                  self.beta2, self.eps, planes=self.planes)
            self._scalar_scale = 1.0
            self._advance()
            return
        native = self.dp_native()
        if native is not None:
            # the whole step behind one library call: forward / backward, the exchange, the update
            H.pair_dp_step(self.shape, self.norm, self.loss, batch, self.theta, self.m, self.v, self.gradbuf, ws, self.lr_t(),
                           self.beta1, self.beta2, self.eps, planes=self.planes, **native)
            self._after_native_steps(1, native)
            return
        if indexed:
            H.pair_step_fwd_bwd_idx(self.shape, self.norm, self.loss, batch[0], batch[1], self.theta, self.grad,
                                    self.scalars, ws, planes=self.planes)
        else:
            H.pair_step_fwd_bwd(self.shape, self.norm, self.loss, batch, self.theta, self.grad, self.scalars, ws,
                                planes=self.planes)
        self._exchange_and_update()

    def dp_native(self):
        """How the library itself runs the exchange of a data-parallel step (ABI 6), as keyword arguments of
        hipabi.pair_dp_step / pair_dp_steps_idx: the one-shot exchange, or RCCL's all-reduce through the raw communicator;
        None when the collective has to go through torch.distributed (gloo test mode, CFL_DP_ALLREDUCE=torch, RCCL not
        loadable through ctypes) -- the step is then driven from Python (`_exchange_and_update`).  CFL_DP_NATIVE=0: never."""
        import os
        if not dp_active() or os.environ.get('CFL_DP_NATIVE', '1') in ('0', ''):
            return None
        if os.environ.get('CFL_DP_NO_COLLECTIVE', '0') == '1':
            # measurements only (bench.py `without_collective`): the launches of a data-parallel step and its stand-alone Adam
            # with NO exchange -- every rank trains on its own rows
            return dict()
        if self._oneshot is not None:
            return dict(ex=self._oneshot.c)
        if os.environ.get('CFL_DP_ALLREDUCE', 'direct') == 'torch':
            return None
        comm = hot_communicator()
        return dict(ar=comm.c_struct()) if comm is not None else None

    def _native_scale(self, native):
        """factor that turns the scalars a library-driven step leaves behind into global-batch means"""
        return 1.0 / world_size() if native else 1.0

    def _after_native_steps(self, n, native):
        self._scalar_scale = self._native_scale(native)
        if self._oneshot is not None:
            self._oneshot.after_library_steps()
        for _ in range(n):
            self._advance()

    def _exchange_and_update(self):
        """the ONE exchange of a data-parallel step + the replicated Adam apply"""
        if self._oneshot is not None:
            self._scalar_scale = self._oneshot.exchange_and_adam(self, self.lr_t())   # (its gather re-writes the planes)
            self._advance()
            return
        # sum of [flat fp32 gradient | scalars] over xGMI
        scale = reduce_gradients(self.gradbuf)
        self._scalar_scale = scale
        self.apply_adam(scale)

    def step_windows(self, win):
        """`win.nsteps` consecutive single-GPU training steps over windows of the device pair lists
        (cfl.input_data.ResidentFeatures.next_windows): one library call, the launches of all steps are enqueued
        back to back."""
        ws = self._workspace(win.rows, 2)
        native = self.dp_native() if dp_active() else None
        if native is not None:
            # data parallel: the same windows, every iteration forward / backward on this rank's rows, the exchange, the update --
            # ONE library call for the iterations between two read-backs (cfl_pair_dp_steps_idx_planes)
            b1p, b2p = H.pair_dp_steps_idx(self.shape, self.norm, self.loss, win, self.theta, self.m, self.v, self.gradbuf, ws,
                                           np.float32(self.lr), self.beta1, self.beta2, self.eps, self.beta1_power,
                                           self.beta2_power, planes=self.planes, **native)
            self.beta1_power, self.beta2_power = np.float32(b1p), np.float32(b2p)
            self.global_step += win.nsteps
            self._scalar_scale = self._native_scale(native)
            if self._oneshot is not None:
                self._oneshot.after_library_steps()
            return
        if dp_active():
            # ... through torch.distributed (gloo test mode, torch fallback): the exchange inside a Python loop
            itemsize = 4
            if (win.nsteps <= 0 or win.pos_head < 0 or win.neg_head < 0
                    or win.pos_head + win.nsteps * win.batch_rows > win.pos_pairs.shape[0]
                    or win.neg_head + win.nsteps * win.batch_rows > win.neg_pairs.shape[0]):
                raise H.CflHipError('step window runs past the pair lists (as cfl_pair_train_steps_idx rejects it)')
            pp, npair = win.pos_pairs.data_ptr(), win.neg_pairs.data_ptr()
            for i in range(win.nsteps):
                po = pp + 2 * itemsize * (win.pos_head + i * win.batch_rows + win.shard_lo)
                no = npair + 2 * itemsize * (win.neg_head + i * win.batch_rows + win.shard_lo)
                c0 = 1 if (win.switched is not None and win.switched[i]) else 0
                streams = H.IndexStreams([po + 4 * c0, po + 4 * (1 - c0), no + 4 * c0, no + 4 * (1 - c0)], 2, win.rows,
                                         keep=[win.pos_pairs, win.neg_pairs])
                H.pair_step_fwd_bwd_idx(self.shape, self.norm, self.loss, win.table, streams, self.theta, self.grad,
                                        self.scalars, ws, planes=self.planes)
                self._exchange_and_update()
            return
        b1p, b2p = H.pair_train_steps_idx(
            self.shape, self.norm, self.loss, win.table, win.pos_pairs, win.neg_pairs, win.pos_head, win.neg_head,
            win.batch_rows, win.shard_lo, win.rows, win.switched, win.nsteps, self.theta, self.m, self.v, self.grad,
            self.scalars, ws, np.float32(self.lr), self.beta1, self.beta2, self.eps, self.beta1_power,
            self.beta2_power, planes=self.planes)
        self.beta1_power, self.beta2_power = np.float32(b1p), np.float32(b2p)
        self.global_step += win.nsteps
        self._scalar_scale = 1.0

    def step_windows_val(self, win, vwin, val_mask, slot_ptrs):
        """step_windows with the validation fetch inside the steps (single GPU; cfl_pair_train_val_steps_idx_planes): the
        steps with val_mask[i] set also score the next validation batch of `vwin` as extra rows of their own projection /
        row-math launches and write [scalars | scores of the positive, then the negative validation pairs] into the next
        of `slot_ptrs` (rows of a pinned host buffer: no copy command, no second launch pair per read-back)."""
        ws = self._workspace(win.rows, 2, vwin.batch_rows)
        if dp_active():
            native = self.dp_native()
            if native is None:
                raise H.CflHipError('step_windows_val under data parallelism needs the library-driven exchange '
                                    '(PairEngine.dp_native); score separately')
            # every rank scores the WHOLE validation batch (theta is replicated: identical scores on every rank); the ring slot
            # receives the global scalar sums from the exchange's last kernel
            b1p, b2p = H.pair_dp_steps_idx(self.shape, self.norm, self.loss, win, self.theta, self.m, self.v, self.gradbuf, ws,
                                           np.float32(self.lr), self.beta1, self.beta2, self.eps, self.beta1_power,
                                           self.beta2_power, planes=self.planes, vwin=vwin, val_mask=val_mask,
                                           slot_ptrs=slot_ptrs, **native)
            self.beta1_power, self.beta2_power = np.float32(b1p), np.float32(b2p)
            self.global_step += win.nsteps
            self._scalar_scale = self._native_scale(native)
            if self._oneshot is not None:
                self._oneshot.after_library_steps()
            return
        b1p, b2p = H.pair_train_val_steps_idx(
            self.shape, self.norm, self.loss, win, vwin, val_mask, slot_ptrs, self.theta, self.m, self.v, self.grad,
            self.scalars, ws, np.float32(self.lr), self.beta1, self.beta2, self.eps, self.beta1_power, self.beta2_power,
            planes=self.planes)
        self.beta1_power, self.beta2_power = np.float32(b1p), np.float32(b2p)
        self.global_step += win.nsteps
        self._scalar_scale = 1.0

    def read_scalars(self):
        """Host copy of the last step's scalars (synchronises the stream).  Under data parallelism they are the
        global-batch values: every scalar is a mean over this rank's rows, the shards are equal-sized, and the
        sums travelled in the gradient all-reduce."""
        raw = self.scalars.cpu().numpy()
        self.check_health(raw)
        vals = raw * np.float32(self._scalar_scale)
        return dict(zip(H.SCALAR_NAMES, (float(x) for x in vals)))

    def check_health(self, host_scalars=None):
        """Raise CflHipError when a training kernel reported a lost in-launch hand-off (the sticky error word of the
        scalars array, which also travels in the data-parallel gradient sum) or the one-shot exchange timed out."""
        if host_scalars is None:
            host_scalars = self.scalars.cpu().numpy()
        H.check_scalars(host_scalars)
        if self._oneshot is not None:
            self._oneshot.check()

    def scores(self, xs, xt):
        """max(thr,1e-6) - dist(src, dst), the value the reference fetches as
        ``val_s_pos_predicts.outputs`` (cfl/utils.py:245).  (xs, xt) dense device rows, or
        (table, IndexStreams of 2 streams)."""
        if isinstance(xt, H.IndexStreams):
            ws = self._workspace(xt.n, 1)
            return H.pair_scores_idx(self.shape, self.norm, xs, xt, self.theta, ws)
        ws = self._workspace(xs.shape[0], 1)
        return H.pair_scores(self.shape, self.norm, xs, xt, self.theta, ws)

    def scores_pos_neg(self, table, streams, out=None):
        """Scores of an indexed labeled batch, positive pairs then negative pairs, in one library call
        (the validation fetch of the training loop): [2 n] (written into `out` when given)."""
        ws = self._workspace(streams.n, 2)
        return H.pair_scores_idx4(self.shape, self.norm, table, streams, self.theta, ws, scores=out)

    # -- checkpoint payload --------------------------------------------------
    def sync_state(self):
        """COLLECTIVE under the one-shot exchange (a no-op otherwise): its Adam slots are sharded over the ranks, so every
        rank calls this before the chief reads m / v for a checkpoint (state_dict / PairModel.checkpoint_state)."""
        if self._oneshot is not None:
            self._oneshot.sync_optimizer_state(self)

    def require_complete_slots(self):
        """A checkpoint must never be written from sharded Adam slots: raise unless sync_state() ran since the last step."""
        if self._oneshot is not None and self._oneshot.slots_dirty:
            raise H.CflHipError('the Adam slots are sharded over the ranks (CFL_DP_EXCHANGE=oneshot) and a step was taken '
                                'since the last PairEngine.sync_state(): every rank must call sync_state() before the '
                                'chief reads m / v for a checkpoint')

    def state_dict(self):
        self.require_complete_slots()
        return dict(theta=self.theta.cpu(), m=self.m.cpu(), v=self.v.cpu(),
                    beta1_power=float(self.beta1_power), beta2_power=float(self.beta2_power),
                    global_step=self.global_step)

    def set_theta(self, theta):
        """overwrite the parameters (checkpoint load, assignment): the kept bf16 planes are stale afterwards"""
        self.theta.copy_(theta)
        self.planes.invalidate()

    def load_state_dict(self, sd):
        self.set_theta(sd['theta'])
        self.m.copy_(sd['m'])
        self.v.copy_(sd['v'])
        self.beta1_power = np.float32(sd['beta1_power'])
        self.beta2_power = np.float32(sd['beta2_power'])
        self.global_step = int(sd['global_step'])

    def named_variables(self):
        p, pd, thr = H.unpack_theta(self.shape, self.theta)
        return p, pd, thr
