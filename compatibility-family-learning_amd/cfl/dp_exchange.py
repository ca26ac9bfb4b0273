"""One-shot gradient exchange of the data-parallel pair step (CFL_DP_EXCHANGE=oneshot; new functionality -- the
reference is single-device).

Default data-parallel step: proj -> mid -> grad -> RCCL all-reduce of [gradient | scalars] -> Adam (cfl/engine.py).
With the one-shot exchange the collective is three kernels of the library instead (csrc/cfl_dp.hip), shaped for
point-to-point xGMI links -- every byte crosses exactly one link, every link carries 1/N of the buffer twice:

    cfl_dp_rs_push    slice s of this rank's [gradient | scalars] -> rank s's slot array (peer memory mapped here through
                      hipIpc), then this rank's A flag in every peer = the step's generation
    cfl_dp_rs_adam    waits for the N local A flags, sums the N rows of ITS slice in rank order, applies TF-Adam to its
                      slice of theta / m / v, pushes the updated theta slice (and the summed scalars) into every peer's
                      stage buffer, raises its B flag there
    cfl_dp_rs_gather  waits for the peers' B flags, copies their slices from the local stage buffer into theta

Round 6 (ABI 6): the PUSH IS FUSED into the weight-gradient launch -- its tile finishers store their finished entries of
[gradient | scalars] straight into the owner's slot array and its last workgroup raises the A flags (csrc/pair_grad.h:
GradFuse::dp_*) -- so a data-parallel step is proj, mid, grad(+push), cfl_dp_rs_adam, cfl_dp_rs_gather_planes: five launches,
all of them behind ONE library call per K iterations (cfl_pair_dp_steps_idx_planes; PairEngine.step / step_windows).  This
module owns the exchange memory and describes it to the library as a CflDpExchange (addresses of both parities, the device
tables of the fused push, the step counter the library advances).  CFL_DP_PUSH_SEPARATE=1 keeps cfl_dp_rs_push as a launch of
its own (A/B runs).

The Adam slots are SHARDED: rank r keeps slice r of m and v current (Adam work / N); `sync_optimizer_state()` -- a
collective every rank calls before the chief writes a checkpoint -- gathers them.  After a step every rank holds
bit-identical parameters (one writer per slice), the global scalar sums, and ITS slice of the gradient sum.

Exchange memory is ONE fine-grained device allocation per rank (cfl_dp_alloc: hipExtMallocWithFlags(
hipDeviceMallocFinegrained) -- the only kind of device memory for which HIP promises that a peer's stores become visible to
a running kernel), shared once, at construction, as a raw 64-byte hipIpc handle over the existing process group.
HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the environment (dmabuf IPC; cfl.engine.init_from_env sets it).
Waits are bounded by wall-clock time (CFL_DP_TIMEOUT_S, default 60 s): a late rank is waited for, a dead one makes the
update NaN and `check()` -- called wherever the host reads the scalars back -- raises.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import hipabi as H


class _Raw(object):
    """a raw device address as a __cuda_array_interface__ object, so that torch can alias it"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}


def slice_floats(n, world):
    """floats per rank of an n-float exchange buffer (n a multiple of 4): whole float4s, slice * world >= n"""
    return (-(-(n // 4) // world)) * 4


def owned_range(n_adam, slice_, rank):
    """[lo, hi) of the first n_adam floats (the parameters) that rank `rank` owns: it sums their gradients, applies Adam to
    them and keeps their Adam slots; the ranges of all ranks tile [0, n_adam) exactly once"""
    lo = min(rank * slice_, n_adam)
    return lo, min(lo + slice_, n_adam)


class OneShotExchange(object):
    def __init__(self, engine):
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        if self.world > H.DP_MAX_WORLD:
            raise H.CflHipError('one-shot exchange supports up to %d ranks' % H.DP_MAX_WORLD)
        dev = engine.device
        L = H.lib()
        self.n = int(engine.gradbuf.numel())
        self.n_adam = int(engine.theta.numel())
        # everything the kernels will require, checked BEFORE any collective (a failure on a subset of the ranks in the
        # middle of the handle exchange would hang the others)
        if self.n % 4 or self.n_adam % 4:
            raise H.CflHipError('gradient buffer / parameter length must be a multiple of 4')
        for name in ('theta', 'm', 'v', 'gradbuf'):
            if getattr(engine, name).data_ptr() % 16:
                raise H.CflHipError('engine.%s must be 16-byte aligned' % name)
        self.slice = slice_floats(self.n, self.world)
        self.timeout_s = float(os.environ.get('CFL_DP_TIMEOUT_S', '60'))
        # layout of the allocation (floats): slots [2][world][slice] | stage [2][n] | A flags [2][64] | B flags [2][64]
        self.o_slots = 0
        self.o_stage = self.o_slots + 2 * self.world * self.slice
        self.o_flags_a = self.o_stage + 2 * self.n
        self.o_flags_b = self.o_flags_a + 2 * 64
        self.total = self.o_flags_b + 2 * 64
        torch.cuda.set_device(dev if dev.index is not None else torch.cuda.current_device())
        fine = 0 if os.environ.get('CFL_DP_COARSE') == '1' else 1     # (escape hatch for experiments on one GPU only)
        # Allocation / export / mapping can fail on a SUBSET of the ranks (out of memory, IPC mode): every phase ends with an
        # exchange of (ok, message) so that all ranks raise together instead of some hanging in the next collective
        self.base, handle, err = None, C.create_string_buffer(64), None
        try:
            base = C.c_void_p()
            H._check(L.cfl_dp_alloc(C.byref(base), 4 * self.total, fine))
            self.base = base.value
            H._check(L.cfl_dp_ipc_export(self.base, handle))
        except H.CflHipError as e:
            err = str(e)
        # the kernels' `lost` word lives in PINNED HOST memory (they write it only when a wait gives up): check() reads it
        # without touching the stream
        self.lost = torch.zeros(1, dtype=torch.int32).pin_memory()
        torch.cuda.synchronize(dev)
        handles = [None] * self.world
        dist.all_gather_object(handles, (err, handle.raw))
        self._raise_together([h[0] for h in handles], 'allocating / exporting the exchange memory')
        handles = [h[1] for h in handles]
        self._peer_base = []
        err = None
        for r in range(self.world):
            if r == self.rank:
                self._peer_base.append(self.base)
                continue
            p = C.c_void_p()
            try:
                H._check(L.cfl_dp_ipc_open(handles[r], C.byref(p)))
            except H.CflHipError as e:
                err = err or ('peer %d: %s' % (r, e))
                p = C.c_void_p(0)
            self._peer_base.append(p.value)
        errs = [None] * self.world
        dist.all_gather_object(errs, err)
        self._raise_together(errs, 'mapping the peers\' exchange memory')
        self.local = torch.as_tensor(_Raw(self.base, self.total, '<f4'), device=dev)   # alias (diagnostics, tests)
        # the library's view: every address of both parities, the device tables of the fused push, the step counter
        c = self.c = H.CflDpExchange()
        c.world, c.rank, c.n, c.n_adam, c.slice = self.world, self.rank, self.n, self.n_adam, self.slice
        tables = []
        for par in range(2):
            c.slots[par] = self._slot_row(self.base, par, 0)
            c.stage[par] = self._stage(self.base, par)
            c.flags_a[par] = self._flag(self.base, self.o_flags_a, par, 0)
            c.flags_b[par] = self._flag(self.base, self.o_flags_b, par, 0)
            for r, b in enumerate(self._peer_base):
                c.peer_rows[par][r] = self._slot_row(b, par, self.rank)
                c.peer_stage[par][r] = self._stage(b, par)
                c.peer_flag_a[par][r] = self._flag(b, self.o_flags_a, par, self.rank)
                c.peer_flag_b[par][r] = self._flag(b, self.o_flags_b, par, self.rank)
            pad = [0] * (H.DP_MAX_WORLD - self.world)
            tables += [c.peer_rows[par][r] for r in range(self.world)] + pad
            tables += [c.peer_flag_a[par][r] for r in range(self.world)] + pad
        self._tables = torch.tensor(tables, dtype=torch.int64).to(dev)     # [2 parities][rows, A flags][DP_MAX_WORLD]
        self._tickets = torch.zeros(4, dtype=torch.int32, device=dev)
        c.dev_tables = self._tables.data_ptr()
        c.tickets = self._tickets.data_ptr()
        c.lost = self.lost.data_ptr()
        c.timeout_s = self.timeout_s
        c.step = 0
        torch.cuda.synchronize(dev)
        dist.barrier()                   # nobody pushes before everyone has mapped everyone
        self.m_v_sharded = self.world > 1
        # True between a step and the next sync_optimizer_state(): this rank's m / v are current for its own slice only,
        # and a checkpoint written now would carry stale slots for (world - 1) / world of the parameters
        self.slots_dirty = False

    def _raise_together(self, errs, what):
        bad = [(r, e) for r, e in enumerate(errs) if e]
        if bad:
            raise H.CflHipError('one-shot exchange: %s failed on rank(s) %s: %s' % (what, [r for r, _ in bad], bad[0][1]))

    # -- addresses -------------------------------------------------------------------------------------------------
    def _slot_row(self, base, par, row):
        return base + 4 * (self.o_slots + (par * self.world + row) * self.slice)

    def _stage(self, base, par):
        return base + 4 * (self.o_stage + par * self.n)

    def _flag(self, base, off, par, r):
        return base + 4 * (off + par * 64 + r)

    @property
    def step(self):
        """exchanges so far (the library advances the counter inside its calls)"""
        return int(self.c.step)

    def after_library_steps(self):
        """bookkeeping behind a library call that ran exchanges (PairEngine): the Adam slots are sharded again"""
        self.slots_dirty = self.m_v_sharded

    def exchange_and_adam(self, engine, lr_t):
        """engine.gradbuf (this rank's [gradient | scalars], already computed into the flat buffer) -> every rank's theta updated
        with the mean gradient; this rank's slice of m / v updated; engine.gradbuf = [this rank's slice of the gradient sum |
        the global scalar sums].  The separate-push form (cfl_dp_rs_push first): what PairEngine.fwd_bwd + this gives when the
        step is not taken through the one-call entry points.  Returns the factor that turns the scalar sums into means."""
        H._check(H.lib().cfl_dp_exchange_step(C.byref(engine.shape), C.byref(self.c), 0, engine.theta.data_ptr(),
                                              engine.m.data_ptr(), engine.v.data_ptr(), engine.gradbuf.data_ptr(), float(lr_t),
                                              float(engine.beta1), float(engine.beta2), float(engine.eps),
                                              H._planes(engine.planes), None, H._stream()))
        self.after_library_steps()
        return 1.0 / self.world

    def owned(self):
        """[lo, hi) of the parameters whose Adam slots this rank keeps current"""
        return owned_range(self.n_adam, self.slice, self.rank)

    def sync_optimizer_state(self, engine):
        """COLLECTIVE: make m and v complete on every rank (each rank owns a slice).  Called by every rank before the
        chief writes a checkpoint.  A no-op when no step was taken since the last call (the flag is the same on every
        rank: steps are collective)."""
        if not self.slots_dirty:
            return
        for t in (engine.m, engine.v):
            pad = torch.zeros(self.world * self.slice, dtype=torch.float32, device=t.device)
            lo, hi = self.owned()
            mine = torch.zeros(self.slice, dtype=torch.float32, device=t.device)
            mine[:hi - lo] = t[lo:hi]
            if dist.get_backend() == 'gloo':
                parts = [torch.empty(self.slice, dtype=torch.float32) for _ in range(self.world)]
                dist.all_gather(parts, mine.cpu())
                pad.copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(pad, mine)
            t.copy_(pad[:t.numel()])
        self.slots_dirty = False

    def check(self):
        """host side of the bounded waits: raise when a peer never arrived.  No synchronisation: `lost` is pinned host memory
        the kernels write only when a wait gives up; the parameters are NaN from that step on, so every rank's loss turns
        non-finite in the same iteration and the training loops stop there whichever rank sees this word first."""
        if int(self.lost[0]) != 0:
            raise H.CflHipError('one-shot gradient exchange: a peer did not arrive within {:.0f} s at or before step {} '
                                '(the parameters are NaN from that step on; CFL_DP_TIMEOUT_S sets the bound)'
                                .format(self.timeout_s, self.step))

    def close(self):
        L = H.lib()
        if getattr(self, 'base', None):
            torch.cuda.synchronize()
            for r, b in enumerate(self._peer_base):
                if r != self.rank:
                    L.cfl_dp_ipc_close(b)
            L.cfl_dp_free(self.base)
            self.base = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
