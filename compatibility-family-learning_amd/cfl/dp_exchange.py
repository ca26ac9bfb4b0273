"""One-shot gradient exchange of the data-parallel pair step (CFL_DP_EXCHANGE=oneshot; new functionality -- the
reference is single-device).

Default data-parallel step: proj -> mid -> grad -> RCCL all-reduce of [gradient | scalars] -> Adam (cfl/engine.py).
With the one-shot exchange the collective is three kernels of the library instead (csrc/cfl_dp.hip):

    cfl_dp_push   this rank's buffer -> its slot in EVERY peer's exchange buffer (peer memory mapped here through
                  hipIpc), then this rank's flag word in every peer's flag array = the step's generation
    cfl_dp_wait   one wave waits until all `world` local flags carry the generation
    cfl_dp_adam   sums the `world` local slots in rank order, leaves the sum in the engine's buffer (what the
                  all-reduce would have left: gradient sums | scalar sums) and applies TF-Adam with sum / world

No ring, no reduction tree: one hop per peer over the point-to-point xGMI links, the same summation order on every
rank (bit-identical parameters on all ranks by construction).  Slots are double-buffered by step parity (see the
header of csrc/cfl_dp.hip for why the flags then suffice).

The peers' buffers are exchanged ONCE, at construction, as torch IPC handles over the existing process group
(all_gather_object): `torch.multiprocessing.reductions.reduce_tensor` on the owner, the rebuild function on the peers.
HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the environment (dmabuf IPC; cfl.engine.init_from_env sets it).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import hipabi as H


class OneShotExchange(object):
    def __init__(self, engine):
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        if self.world > 16:
            raise H.CflHipError('one-shot exchange supports up to 16 ranks')
        dev = engine.device
        self.n = int(engine.gradbuf.numel())
        if self.n % 4:
            raise H.CflHipError('gradient buffer length must be a multiple of 4')
        # [parity][rank][n] slots and [parity][64] flag words (one 256-byte line per parity), written by the peers
        self.slots = torch.zeros(2, self.world, self.n, dtype=torch.float32, device=dev)
        self.flags = torch.zeros(2, 64, dtype=torch.int32, device=dev)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lost = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        from torch.multiprocessing.reductions import reduce_tensor
        mine = (reduce_tensor(self.slots), reduce_tensor(self.flags))
        handles = [None] * self.world
        dist.all_gather_object(handles, mine)
        self._peer_tensors = []          # keep the mappings alive
        self._slot_base, self._flag_base = [], []
        for r in range(self.world):
            if r == self.rank:
                ps, pf = self.slots, self.flags
            else:
                (fs, as_), (ff, af) = handles[r]
                ps, pf = fs(*as_), ff(*af)
                # touching the mapping once from this device also enables peer access to the owner's memory
                torch.empty(1, dtype=torch.float32, device=dev).copy_(ps.view(-1)[:1])
            self._peer_tensors.append((ps, pf))
            self._slot_base.append(ps.data_ptr())
            self._flag_base.append(pf.data_ptr())
        torch.cuda.synchronize(dev)
        dist.barrier()                   # nobody pushes before everyone has mapped everyone
        self.step = 0

    def exchange_and_adam(self, engine, lr_t):
        """engine.gradbuf (this rank's [gradient | scalars]) -> sum over ranks in engine.gradbuf; theta, m, v updated
        with the mean gradient."""
        par, gen = self.step & 1, (self.step + 1) & 0xffffffff
        if gen == 0:
            gen = 1
        slot_ptrs = (C.c_void_p * self.world)(*[b + 4 * ((par * self.world + self.rank) * self.n)
                                                 for b in self._slot_base])
        flag_ptrs = (C.c_void_p * self.world)(*[b + 4 * (par * 64 + self.rank) for b in self._flag_base])
        L = H.lib()
        st = H._stream()
        H._check(L.cfl_dp_push(engine.gradbuf.data_ptr(), self.n, slot_ptrs, flag_ptrs, self.world, gen,
                               self.ticket.data_ptr(), st))
        H._check(L.cfl_dp_wait(self.flags[par].data_ptr(), self.world, gen, self.lost.data_ptr(), st))
        H._check(L.cfl_dp_adam(engine.theta.data_ptr(), engine.m.data_ptr(), engine.v.data_ptr(),
                               self.slots[par].data_ptr(), self.world, self.n, int(engine.theta.numel()),
                               engine.gradbuf.data_ptr(), float(lr_t), float(engine.beta1), float(engine.beta2),
                               float(engine.eps), self.lost.data_ptr(), st))
        self.step += 1
        return 1.0 / self.world
