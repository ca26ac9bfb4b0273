"""Feasibility probe (GPU box, two processes on the one GPU): fine-grained device memory allocated through the library,
exported with hipIpc, opened by a second process, wrapped as torch tensors on both sides; the peer writes, the owner
reads.  Also times the headline step with theta in fine-grained memory against ordinary memory."""
import ctypes as C
import os
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
    sys.path.insert(0, p)


class Raw(object):
    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {'shape': (nfloats,), 'typestr': '<f4', 'data': (int(ptr), False), 'version': 2}


def lib():
    L = C.CDLL(os.path.join(ROOT, 'compatibility-family-learning_amd', 'lib', 'libcfl_hip.so'))
    L.cfl_last_error.restype = C.c_char_p
    L.cfl_dp_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_int32]
    L.cfl_dp_ipc_export.argtypes = [C.c_void_p, C.c_void_p]
    L.cfl_dp_ipc_open.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    return L


def peer(q_in, q_out):
    torch.cuda.set_device(0)
    torch.zeros(1, device='cuda')
    L = lib()
    fine, handle = q_in.get()
    p = C.c_void_p()
    rc = L.cfl_dp_ipc_open(handle, C.byref(p))
    if rc:
        q_out.put(('open failed', L.cfl_last_error().decode()))
        return
    t = torch.as_tensor(Raw(p.value, 1024), device='cuda')
    t.fill_(3.5)
    torch.cuda.synchronize()
    q_out.put(('ok', float(t[7].item())))
    q_in.get()


def main():
    mp.set_start_method('spawn')
    torch.cuda.set_device(0)
    torch.zeros(1, device='cuda')
    L = lib()
    for fine in (1, 0):
        p = C.c_void_p()
        rc = L.cfl_dp_alloc(C.byref(p), 4096, fine)
        print('alloc fine=%d rc=%d %s' % (fine, rc, L.cfl_last_error().decode() if rc else hex(p.value)))
        if rc:
            continue
        h = C.create_string_buffer(64)
        rc = L.cfl_dp_ipc_export(p, h)
        print('  export rc=%d %s' % (rc, L.cfl_last_error().decode() if rc else 'ok'))
        if rc:
            continue
        q_in, q_out = mp.Queue(), mp.Queue()
        pr = mp.Process(target=peer, args=(q_in, q_out))
        pr.start()
        q_in.put((fine, h.raw))
        print('  peer:', q_out.get(timeout=120))
        mine = torch.as_tensor(Raw(p.value, 1024), device='cuda')
        torch.cuda.synchronize()
        print('  owner reads', float(mine[7].item()), float(mine.sum().item()))
        q_in.put('bye')
        pr.join()
    # step time with theta in fine-grained memory
    import numpy as np
    from cfl import hipabi as H
    from cfl.engine import PairEngine
    from oracle import cfl_oracle as O
    D, L_, K, B = 4096, 20, 3, 512
    cfg = O.EncoderCfg(D=D, L=L_, K=K)
    params = O.init_encoder_params(cfg, np.random.RandomState(0), np.float32)
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    pool = [tuple(torch.randn(B, D, generator=g, device='cuda').abs_() * 13 for _ in range(4)) for _ in range(12)]
    for fine in (0, 1, 0, 1):
        eng = PairEngine(D, L_, K, norm=H.make_norm(1 / 58.4), params=params, batch_size=B)
        if fine:
            n = eng.theta.numel()
            p = C.c_void_p()
            assert L.cfl_dp_alloc(C.byref(p), 4 * n, 1) == 0
            t = torch.as_tensor(Raw(p.value, n), device='cuda')
            t.copy_(eng.theta)
            eng.theta = t
        for i in range(100):
            eng.step(pool[i % 12])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(500):
            eng.step(pool[i % 12])
        torch.cuda.synchronize()
        print('theta fine-grained=%d: %.2f us per step' % (fine, (time.perf_counter() - t0) / 500 * 1e6))


if __name__ == '__main__':
    main()
