"""RCCL called directly, on the stream the step's kernels are launched on (new functionality: the reference is
single-device; 8(b) "RCCL communicator owned by the Python side").

`torch.distributed.all_reduce` with backend "nccl" runs the collective on ProcessGroupNCCL's OWN stream and fences it
against the caller's stream with two events.  For a 33 us step whose next launch needs the reduced gradient that is two
cross-stream dependencies per step: measured on one MI355X with a one-rank group, 59.3 us per step against 35.5 us for
the same launches without the collective (profiles/r05_a_bench.json) -- the fixed cost of the exchange was 2/3 of a step
before a byte crossed a link.  Here the communicator is created once from the existing process group (the unique id
travels over it) and `ncclAllReduce` is enqueued on the caller's stream like any other kernel of the step:

    proj -> mid -> grad -> ncclAllReduce([gradient | scalars]) -> adam (+ planes)        one stream, no events

The library is the librccl.so that torch itself loaded (one RCCL per process).  backend "nccl" IS RCCL on ROCm.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_FLOAT32 = 7
NCCL_SUM = 0


class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('CFL_RCCL_LIB') or os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
    L = C.CDLL(path)
    L.ncclGetErrorString.restype = C.c_char_p
    L.ncclGetErrorString.argtypes = [C.c_int]
    L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclCommDestroy.argtypes = [C.c_void_p]
    for f in (L.ncclGetUniqueId, L.ncclCommInitRank, L.ncclAllReduce, L.ncclCommDestroy):
        f.restype = C.c_int
    _lib = L
    return L


class RcclError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise RcclError('%s: %s' % (what, lib().ncclGetErrorString(rc).decode()))


class Communicator(object):
    """One RCCL communicator over the ranks of the default torch process group (COLLECTIVE constructor: every rank
    calls it; the current device must be this rank's GPU)."""

    def __init__(self):
        if not dist.is_initialized():
            raise RcclError('a torch.distributed process group must exist (the unique id travels over it)')
        L = lib()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        uid = _UniqueId()
        if self.rank == 0:
            _check(L.ncclGetUniqueId(C.byref(uid)), 'ncclGetUniqueId')
        # (the raw 128 bytes: ctypes truncates a c_char array at the first NUL when it is read as bytes)
        box = [C.string_at(C.addressof(uid), 128) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        C.memmove(C.addressof(uid), box[0], 128)
        comm = C.c_void_p()
        _check(L.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank), 'ncclCommInitRank')
        self.comm = comm

    def all_reduce_sum_(self, tensor, stream=None):
        """in-place fp32 sum over the ranks, enqueued on `stream` (default: torch's current stream)"""
        if tensor.dtype != torch.float32 or not tensor.is_cuda or not tensor.is_contiguous():
            raise RcclError('all_reduce_sum_ expects a contiguous fp32 device tensor')
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        p = tensor.data_ptr()
        _check(lib().ncclAllReduce(p, p, tensor.numel(), NCCL_FLOAT32, NCCL_SUM, self.comm, st), 'ncclAllReduce')

    def close(self):
        if getattr(self, 'comm', None):
            torch.cuda.synchronize()
            lib().ncclCommDestroy(self.comm)
            self.comm = None



_default = None


_unavailable = False


def default_communicator():
    """The process-wide communicator of the hot exchange, created on first use (COLLECTIVE), or None when it cannot be
    created on EVERY rank -- librccl not loadable through ctypes, ncclCommInitRank failing somewhere: the ranks agree on the
    outcome over the torch process group (a MIN all-reduce of a success flag), so that either all of them enqueue
    ncclAllReduce on their launch streams or all of them fall back to torch.distributed.all_reduce; a split would hang."""
    global _default, _unavailable
    if _default is None and not _unavailable:
        comm, err = None, None
        try:
            comm = Communicator()
        except Exception as e:          # noqa: BLE001 (anything: OSError from ctypes, RcclError, AttributeError of a missing symbol)
            err = e
        ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32, device='cuda')
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            _default = comm
        else:
            _unavailable = True
            if comm is not None:
                comm.close()
            import logging
            logging.getLogger(__name__).warning('direct RCCL all-reduce unavailable (%s on this rank): the gradient exchange goes '
                                                'through torch.distributed.all_reduce', err or 'failed on another rank')
    return _default


def shutdown():
    global _default, _unavailable
    if _default is not None:
        _default.close()
        _default = None
    _unavailable = False
