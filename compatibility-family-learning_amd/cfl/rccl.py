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
    """One RCCL communicator over the ranks of the default torch process group.  Built by `default_communicator()` in
    PHASES with an agreement step after each: a constructor that is one collective from end to end cannot
    fail symmetrically -- a rank whose librccl does not load would sit in the agreement all-reduce while the others sit in
    the broadcast of the unique id."""

    def __init__(self, comm, rank, world):
        self.comm, self.rank, self.world = comm, rank, world
        self._c = None

    def all_reduce_sum_(self, tensor, stream=None):
        """in-place fp32 sum over the ranks, enqueued on `stream` (default: torch's current stream)"""
        if tensor.dtype != torch.float32 or not tensor.is_cuda or not tensor.is_contiguous():
            raise RcclError('all_reduce_sum_ expects a contiguous fp32 device tensor')
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        p = tensor.data_ptr()
        _check(lib().ncclAllReduce(p, p, tensor.numel(), NCCL_FLOAT32, NCCL_SUM, self.comm, st), 'ncclAllReduce')

    def c_struct(self):
        """CflAllReduce for libcfl_hip's data-parallel entry points (include/cfl_hip.h, ABI 6): RCCL's own ncclAllReduce and
        this communicator -- the library calls it between its weight-gradient launch and the update, on the launch stream"""
        if self._c is None:
            from .hipabi import CflAllReduce
            fn = C.cast(lib().ncclAllReduce, C.c_void_p).value
            self._c = CflAllReduce(fn, self.comm.value if isinstance(self.comm, C.c_void_p) else int(self.comm),
                                   NCCL_FLOAT32, NCCL_SUM, self.world)
        return self._c

    def close(self):
        if getattr(self, 'comm', None):
            torch.cuda.synchronize()
            lib().ncclCommDestroy(self.comm)
            self.comm = None


_default = None
_unavailable = False


def _all_ok(ok):
    """agreement over the torch process group: did the phase succeed on EVERY rank?  (MIN all-reduce of a flag; with one rank
    nothing to agree on)"""
    if dist.get_world_size() == 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def _warn_fallback(phase, err):
    import logging
    logging.getLogger(__name__).warning('direct RCCL all-reduce unavailable (%s: %s): the gradient exchange goes through '
                                        'torch.distributed.all_reduce', phase, err or 'failed on another rank')


def default_communicator():
    """The process-wide communicator of the hot exchange, created on first use (COLLECTIVE: every rank calls it), or None when it
    cannot be created on EVERY rank; the ranks then ALL fall back to torch.distributed.all_reduce (a split would hang).  Three
    phases, each followed by an agreement over the torch process group, so that a failure in one of them is seen by every
    rank BEFORE anybody enters the next collective:
      1. load librccl through ctypes and resolve the symbols                      -> MIN all-reduce
      2. rank 0 draws the unique id; (ok, id) is broadcast                        -> every rank reads `ok` from the broadcast
      3. ncclCommInitRank (collective inside RCCL: every rank enters it)          -> MIN all-reduce
    CFL_RCCL_FAIL_PHASE=<n>[:<rank>] makes phase n fail (on that rank; tests)."""
    global _default, _unavailable
    if _default is not None or _unavailable:
        return _default
    if not dist.is_initialized():
        raise RcclError('a torch.distributed process group must exist (the unique id travels over it)')
    rank, world = dist.get_rank(), dist.get_world_size()
    fail = os.environ.get('CFL_RCCL_FAIL_PHASE', '')
    fail_phase, _, fail_rank = fail.partition(':')

    def injected(phase):
        return fail_phase == str(phase) and (fail_rank == '' or int(fail_rank) == rank)

    # phase 1: the library
    L, err = None, None
    try:
        if injected(1):
            raise OSError('injected failure (CFL_RCCL_FAIL_PHASE)')
        L = lib()
    except Exception as e:          # noqa: BLE001 (OSError from ctypes, AttributeError of a missing symbol)
        err = e
    if not _all_ok(L is not None):
        _unavailable = True
        _warn_fallback('loading librccl', err)
        return None
    # phase 2: the unique id (rank 0 draws it; its success travels WITH the id)
    uid, err = _UniqueId(), None
    box = [None]
    if rank == 0:
        try:
            if injected(2):
                raise RcclError('injected failure (CFL_RCCL_FAIL_PHASE)')
            _check(L.ncclGetUniqueId(C.byref(uid)), 'ncclGetUniqueId')
            # (the raw 128 bytes: ctypes truncates a c_char array at the first NUL when it is read as bytes)
            box = [C.string_at(C.addressof(uid), 128)]
        except Exception as e:      # noqa: BLE001
            err = e
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        _unavailable = True
        _warn_fallback('ncclGetUniqueId on rank 0', err)
        return None
    C.memmove(C.addressof(uid), box[0], 128)
    # phase 3: the communicator (every rank enters ncclCommInitRank: the failures that remain are RCCL's own, symmetric or not)
    comm, err = C.c_void_p(), None
    try:
        if injected(3):
            raise RcclError('injected failure (CFL_RCCL_FAIL_PHASE)')
        _check(L.ncclCommInitRank(C.byref(comm), world, uid, rank), 'ncclCommInitRank')
    except Exception as e:          # noqa: BLE001
        err, comm = e, None
    if not _all_ok(comm is not None):
        _unavailable = True
        if comm is not None:
            Communicator(comm, rank, world).close()
        _warn_fallback('ncclCommInitRank', err)
        return None
    _default = Communicator(comm, rank, world)
    return _default


def shutdown():
    global _default, _unavailable
    if _default is not None:
        _default.close()
        _default = None
    _unavailable = False
