"""where do the ranks' parameters differ in the failing 4-rank case (cfl pcd D=4096 L=36 K=4)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')):
    sys.path.insert(0, _p)
import numpy as np


def ranges(mask):
    idx = np.flatnonzero(mask)
    if idx.size == 0:
        return []
    cuts = np.flatnonzero(np.diff(idx) > 1)
    starts = np.r_[idx[0], idx[cuts + 1]]
    ends = np.r_[idx[cuts], idx[-1]]
    return list(zip(starts.tolist(), ends.tolist()))[:12], int(idx.size)


def worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', CFL_DP_MAX_BLOCKS='64')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cfl import engine, hipabi as H
    from cfl.engine import PairEngine
    from oracle import cfl_oracle as O
    D, L, K, B = 4096, int(os.environ.get('L', 36)), int(os.environ.get('K', 4)), 64 * world
    rng = np.random.RandomState(3)
    cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type='pcd', style='cfl')
    p0 = O.init_encoder_params(cfg, rng, np.float32)
    if os.environ.get('PERTURB', '1') == '1':
        for k in p0:
            p0[k] = (p0[k] + 0.05 * rng.randn(*p0[k].shape).astype(np.float32) * (0.1 if k.endswith('/W') else 1.0)).astype(np.float32)

    def mk(exchange):
        os.environ['CFL_DP_EXCHANGE'] = exchange
        return PairEngine(D, L, K, 'pcd', weight_norm=True, has_bias=True, norm=H.make_norm(1.0 / 31.9098), loss=H.make_loss(),
                          lr=1e-3, device='cuda', params=p0, thr=0.5, batch_size=B)
    a, b = mk('allreduce'), mk('oneshot')
    lay = H.layout(b.shape)
    lo, hi = engine.shard_rows(B)
    for step in range(int(os.environ.get('STEPS', 5))):
        full = [torch.from_numpy((np.abs(rng.randn(B, D)) * 8.0).astype(np.float32)).cuda() for _ in range(4)]
        shard = [x[lo:hi].contiguous() for x in full]
        a.step(shard); b.step(shard)
        torch.cuda.synchronize()
        th = b.theta.detach().cpu()
        ths = [torch.empty_like(th) for _ in range(world)] if rank == 0 else None
        dist.gather(th, ths, dst=0)
        ga = a.theta.detach().cpu()
        if rank == 0:
            print('   NaN count: all-reduce engine', int(torch.isnan(a.theta).sum()), 'one-shot engine', int(torch.isnan(b.theta).sum()),
                  'one-shot gradbuf (own slice region only is meaningful)', int(torch.isnan(b.gradbuf).sum()),
                  'scalars', b.gradbuf[lay.total:lay.total + 16].cpu().numpy().round(4).tolist(), flush=True)
            print('step', step, 'n', b._oneshot.n, 'n_adam', b._oneshot.n_adam, 'slice', b._oneshot.slice, 'total', lay.total, flush=True)
            for r in range(world):
                d = (ths[r] != ths[0]).numpy() | np.isnan(ths[r].numpy())
                print('  rank', r, 'theta differs from rank 0 at', ranges(d), ' vs all-reduce engine (rank 0) max', float((ths[r] - ga).abs().max()), flush=True)
    b.sync_state()
    if rank == 0:
        print('m NaN at', ranges(np.isnan(b.m.cpu().numpy())), 'v NaN at', ranges(np.isnan(b.v.cpu().numpy())))
        print('heads:', [(n, getattr(lay.enc[0], n).w, getattr(lay.enc[0], n).npad) for n in ('outputs', 'proto')])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=worker, args=(r, world, 29500 + os.getpid() % 2000, None)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(600) for p in procs]
