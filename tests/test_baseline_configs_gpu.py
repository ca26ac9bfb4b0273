"""BASELINE.json workloads at their own shapes (-m gpu).

* config 1 (headline): 4096-d features, PCD K=3, L=20, batch 512 -- 100 training steps beside the oracle in float64
  AND in float32 on identical batches: three deviation series per step (HIP vs fp64, HIP vs fp32 CPU, fp32 CPU vs fp64)
  recorded and capped at the values in CAPS (observed series: profiles/r04_trajectory_*.json), then 2 x 100 000
  held-out pairs scored by the HIP scoring path and by the oracle on its own trained weights: AUC within 1e-4
  (SURVEY 8(d) "AUC check").
* config 4 shape (2048-d, K=5, L=20, batch 1024, weight-norm heads) and config 3 (1024-d siamese L=256, hinge margin
  100, pos_weight 0.0625) plus the dyadic script's pcd run: 30 steps the same way.
* config 5 (MrCGAN 64x64x3, latent 64, K=2, z=20, srgan, lambda_gp 0.5, m_prj 0.2, m_enc 0.05;
  experiments/dyadic/run_gen.sh:25-53): one full post-epoch step -- every loss part and the D / G gradients --
  against oracle/gan_oracle.py at B=20, and the loss parts at the reference batch size B=100.
"""
import numpy as np
import pytest
import torch

from oracle import cfl_oracle as O

pytestmark = pytest.mark.gpu

NV = 58.388599          # experiments/monomer/run.sh:18


def _planted(gen, B, D, teacher, back, s, noise):
    """post-ReLU-like features; positive targets planted by a hidden linear teacher, negative targets planted from
    unrelated sources (same marginals: only the pairing separates the classes)"""
    r = lambda: torch.randn(B, D, generator=gen, device='cuda')
    ps = r().abs_() * s
    pd = ((ps @ teacher) @ back + noise * s * r()).abs_()
    ns = r().abs_() * s
    other = r().abs_() * s
    nd = ((other @ teacher) @ back + noise * s * r()).abs_()
    return [t.contiguous() for t in (ps, pd, ns, nd)]


def _trajectory(name, style, D, K, L, B, steps, lkw, n_eval, nv, caps, min_auc=None, dist_type='pcd'):
    """HIP beside the NumPy oracle in float64 AND in float32 (the precision of the reference's TensorFlow CPU path) on
    identical batches.  Three series are measured per step, recorded (CFL_RECORD_DIR=<dir> writes
    <dir>/trajectory_<name>.json; the committed copies are profiles/r04_trajectory_*.json) and asserted:
        hip_vs_fp64    |loss_HIP - loss_fp64| / max(1, |loss_fp64|)
        hip_vs_fp32    |loss_HIP - loss_fp32cpu| / max(1, ...)        (north_star's reference IS an fp32 CPU path)
        fp32_vs_fp64   |loss_fp32cpu - loss_fp64| / max(1, ...)       (what fp32 arithmetic itself costs here)
    `caps` = (cap on hip_vs_fp64, cap on hip_vs_fp32): twice the worst value observed when the bars were set, floored
    at 1e-6 (the observed series are in the committed JSON); all are below north_star's 1e-5.  Step 0 (before any update: pure forward precision) is held
    to 1e-6.  AUC on the held-out pairs within 1e-4 of the float64 oracle's."""
    import json
    import os
    from cfl import hipabi as H
    from cfl.engine import PairEngine
    rng = np.random.RandomState(0)
    cfg = O.EncoderCfg(D=D, L=L, K=K, dist_type=dist_type, style=style)
    p = O.init_encoder_params(cfg, rng, np.float32)
    lcfg = O.LossCfg(**lkw)
    thr0 = 1e-6
    tr = O.OracleTrainer(cfg, lcfg, lr=1e-3, dtype=np.float64, params={k: v.astype(np.float64) for k, v in p.items()})
    tr32 = O.OracleTrainer(cfg, lcfg, lr=1e-3, dtype=np.float32, params={k: v.copy() for k, v in p.items()})
    eng = PairEngine(D, L, K, dist_type, weight_norm=cfg.weight_norm, has_bias=cfg.has_bias, norm=H.make_norm(1.0 / nv),
                     loss=H.make_loss(**lkw), lr=1e-3, device='cuda', params=p, thr=thr0, batch_size=B)
    gen = torch.Generator(device='cuda')
    gen.manual_seed(633)
    teacher = torch.randn(D, 64, generator=gen, device='cuda') / D ** 0.5
    back = torch.randn(64, D, generator=gen, device='cuda') / 8.0
    s = nv / 4.5
    hip64, hip32, cpu32 = [], [], []
    for it in range(steps):
        b = _planted(gen, B, D, teacher, back, s, 0.3)
        eng.step(b)
        host = [x.cpu().numpy() for x in b]
        sc = tr.step(tuple(x.astype(np.float64) / nv for x in host))
        sc32 = tr32.step(tuple(x / np.float32(nv) for x in host))
        got = eng.read_scalars()['total']
        den = max(1.0, abs(sc['total']))
        hip64.append(abs(got - sc['total']) / den)
        hip32.append(abs(got - float(sc32['total'])) / den)
        cpu32.append(abs(float(sc32['total']) - sc['total']) / den)
    record = {'name': name, 'shape': dict(style=style, dist_type=dist_type, D=D, K=K, L=L, B=B, steps=steps, loss=lkw, nv=nv),
              'hip_vs_fp64': hip64, 'hip_vs_fp32cpu': hip32, 'fp32cpu_vs_fp64': cpu32,
              'max': dict(hip_vs_fp64=max(hip64), hip_vs_fp32cpu=max(hip32), fp32cpu_vs_fp64=max(cpu32)),
              'caps_asserted': dict(hip_vs_fp64=caps[0], hip_vs_fp32cpu=caps[1], step0=1e-6)}
    show = lambda e: ['%.1e' % x for x in e[:12]] + ['max %.1e' % max(e)]
    # held-out pairs, scored in chunks: HIP scores with the HIP-trained weights, the oracles with their own
    sp, sn, rp, rn, qp, qn = [], [], [], [], [], []
    chunk = 8192
    for _ in range((n_eval + chunk - 1) // chunk):
        ps, pd, ns, nd = _planted(gen, chunk, D, teacher, back, s, 0.3)
        sp.append(eng.scores(ps, pd).cpu().numpy())
        sn.append(eng.scores(ns, nd).cpu().numpy())
        f = lambda t: t.cpu().numpy().astype(np.float64) / nv
        g = lambda t: t.cpu().numpy() / np.float32(nv)
        rp.append(tr.scores(f(ps), f(pd)))
        rn.append(tr.scores(f(ns), f(nd)))
        qp.append(tr32.scores(g(ps), g(pd)))
        qn.append(tr32.scores(g(ns), g(nd)))
    sp, sn, rp, rn, qp, qn = (np.concatenate(a)[:n_eval] for a in (sp, sn, rp, rn, qp, qn))
    ev_h, ev_o = O.dist_eval(sp.astype(np.float64), sn.astype(np.float64)), O.dist_eval(rp, rn)
    ev_q = O.dist_eval(qp.astype(np.float64), qn.astype(np.float64))
    scale = max(1.0, float(np.abs(rp).max()))
    cpu32_dev = max(np.abs(qp - rp).max(), np.abs(qn - rn).max())
    hip_dev = max(np.abs(sp - rp).max(), np.abs(sn - rn).max())
    record['eval'] = dict(pairs=int(n_eval), auc_hip=ev_h['auc'], auc_fp64=ev_o['auc'], auc_fp32cpu=ev_q['auc'],
                          accuracy_hip=ev_h['accuracy'], accuracy_fp64=ev_o['accuracy'],
                          max_score_dev_hip=float(hip_dev), max_score_dev_fp32cpu=float(cpu32_dev), score_scale=scale)
    out_dir = os.environ.get('CFL_RECORD_DIR')
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'trajectory_%s.json' % name), 'w') as fh:
            json.dump(record, fh, indent=1)
    assert hip64[0] <= 1e-6, show(hip64)
    assert max(hip64) <= caps[0], (show(hip64), show(cpu32))
    assert max(hip32) <= caps[1], (show(hip32), show(cpu32))
    if min_auc is not None:
        assert min_auc < ev_o['auc'] < 0.9999, ev_o       # a non-trivial ranking problem
    assert abs(ev_h['auc'] - ev_o['auc']) <= 1e-4, (ev_h, ev_o)
    assert abs(ev_h['accuracy'] - ev_o['accuracy']) <= 1e-3, (ev_h, ev_o)
    assert hip_dev <= max(1e-4 * scale, 4.0 * cpu32_dev), (hip_dev, cpu32_dev, scale)
    return record


# (cap on |HIP - fp64|, cap on |HIP - fp32 CPU|) per workload: 2x the worst value observed on MI355X when the bars
# were set -- the observed series are committed as profiles/r03_trajectory_<name>.json
CAPS = {   # round 4 (projection on the bf16 matrix cores from kept planes): observed series in profiles/r04_trajectory_*.json
    'headline': (2.5e-6, 5e-6),               # observed 1.2e-6 / 2.5e-6 (fp32 CPU vs fp64: 1.3e-6: HIP is the closer one to fp64)
    'config4': (1e-6, 1e-6),                  # observed 8.9e-8 / 3.6e-7
    'config3_pcd': (1e-6, 1e-6),              # observed 1.1e-7 / 1.5e-7
    'config3_siamese_hinge': (1e-6, 1e-6),    # observed 7.2e-8 / 2.1e-7
}


def test_headline_config_100_steps_and_auc_on_100k_pairs():
    """BASELINE config 1: Monomer-style 4096-d, `Dist` model, K=3, L=20, B=512."""
    r = _trajectory('headline', 'dist', 4096, 3, 20, 512, 100, dict(), 100000, NV, CAPS['headline'], min_auc=0.52)
    print('headline: worst rel loss diff vs fp64 %.2e, vs fp32 cpu %.2e (fp32 cpu vs fp64 %.2e); AUC hip %.6f fp64 %.6f' % (
        r['max']['hip_vs_fp64'], r['max']['hip_vs_fp32cpu'], r['max']['fp32cpu_vs_fp64'], r['eval']['auc_hip'],
        r['eval']['auc_fp64']))


def test_config4_shape_trajectory_and_auc():
    """BASELINE config 4 shape: 2048-d latents, PCD K=5, L=20, B=1024, weight-normalised CFL heads with the
    polyvore flags (--pos-weight .25 --use-threshold)."""
    _trajectory('config4', 'cfl', 2048, 5, 20, 1024, 30, dict(use_threshold=True, pos_weight=0.25), 32768, 1.0, CAPS['config4'])


def test_dyadic_pcd_shape_trajectory_and_auc():
    """The pcd run of the dyadic script (experiments/dyadic/run.sh: 1024-d, --num-components 3 --latent-size 64
    --pos-weight 0.0625 --use-threshold) -- NOT BASELINE config 3, which is the siamese / hinge model below."""
    _trajectory('config3_pcd', 'cfl', 1024, 3, 64, 512, 30, dict(use_threshold=True, pos_weight=0.0625), 32768, 31.9098,
                CAPS['config3_pcd'])


def test_config3_siamese_hinge_trajectory_and_auc():
    """BASELINE config 3, the model itself (experiments/dyadic/run.sh:40-50): 1024-d GoogLeNet latents, siamese
    distance, latent_size 256, contrastive hinge --caffe-margin 100, --pos-weight 0.0625, no threshold in the encoder
    loss (the threshold trains under its own Adam), weight-normalised heads, B=512: 30 steps + AUC on 32768 pairs."""
    _trajectory('config3_siamese_hinge', 'cfl', 1024, 1, 256, 512, 30,
                dict(use_threshold=False, caffe_margin=100.0, pos_weight=0.0625), 32768, 31.9098,
                CAPS['config3_siamese_hinge'], dist_type='siamese')


def _close(name, got, want, rtol):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = max(float(np.abs(want).max()), 1e-12)
    err = float(np.abs(got - want).max())
    assert err <= rtol * scale, '%s: max err %.3e vs scale %.3e' % (name, err, scale)


@pytest.mark.parametrize('B', [20, 100])
def test_config5_mrcgan_post_epoch_step_64x64(B):
    """BASELINE config 5 at its own shape, B = 20 and the reference batch B = 100: every loss part and every D / G gradient
    tensor of one post-epoch step against oracle/gan_oracle.py in float64, with the SAME oracle evaluated in float32 on the CPU
    as the yardstick (tests/parity_series.py): HIP must be no further from float64 than twice the fp32 CPU evaluation is --
    loss parts: relative difference; gradients: per tensor, the fraction of entries within 5e-4 of the tensor's scale and the
    worst entry (a near-zero pre-activation on the other side of an lrelu / relu kink changes single entries visibly, in
    the fp32 CPU evaluation exactly as here: tests/test_activation_masks_gpu.py holds the masks equal and gets 2e-5)."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from parity_series import ParitySeries
    from cfl.models import mrcgan as M
    from oracle import gan_oracle as GO
    shape, Ld, zd = (64, 64, 3), 64, 20
    cfgkw = dict(m_enc=0.05, m_prj=0.2, lambda_gp=0.5)
    o = GO.GanOracle('srgan', shape, 'tanh', zd, Ld, seed=1, **cfgkw)
    o32 = GO.GanOracle('srgan', shape, 'tanh', zd, Ld, seed=1, dtype=torch.float32, **cfgkw)
    ph = M.GanPhase('srgan', shape, 'tanh', zd, Ld, B, torch.device('cuda'), np.random.RandomState(0),
                    lambda_dra=0.5, **cfgkw)
    for net, ref, pre in ((ph.gen, o.gp, 'Generator/'), (ph.disc, o.dp, 'Discriminator/')):
        assert set(net.pool.order) == set(pre + k for k in ref)
        net.pool.load({pre + k: v.numpy() for k, v in ref.items()})
    rng = np.random.RandomState(11)
    N = int(np.prod(shape))
    batch = [np.tanh(rng.randn(B, N)), 0.3 * rng.randn(B, Ld), 0.3 * rng.randn(B, Ld), 0.3 * rng.randn(B, Ld),
             0.3 * rng.randn(B, Ld), rng.randn(B, zd), rng.rand(B, 1)]
    dev = lambda a: torch.tensor(np.asarray(a, np.float32), device='cuda')
    ph.step(*[dev(b) for b in batch], apply=False)
    s = ph.read_scalars()
    tb = [torch.tensor(b) for b in batch]
    d_total, g_total, parts, d_grads, g_grads = o.losses_and_grads(*tb)
    d32, g32, parts32, d_grads32, g_grads32 = o32.losses_and_grads(*[torch.tensor(np.asarray(b, np.float32)) for b in batch])
    ref = dict(d_total_loss=float(d_total), g_total_loss=float(g_total))
    ref.update({k: float(v) for k, v in parts.items() if k in s})
    ref32 = dict(d_total_loss=float(d32), g_total_loss=float(g32))
    ref32.update({k: float(v) for k, v in parts32.items() if k in s})
    assert set(ref) >= {'d_loss_real', 'd_loss_fake', 'd_grad_loss', 'd_loss_d', 'g_loss', 'g_loss_d', 'g_loss_d_neg'}
    ser = ParitySeries('config5_loss_parts_b%d' % B, floor=1e-5, meta=dict(shape=shape, B=B))
    for k, r in ref.items():
        ser.add(0, k, s[k], r, ref32[k])
    ser.check()
    gd = ph.disc.pool.named(ph.disc.pool.grad)
    gg = ph.gen.pool.named(ph.gen.pool.grad)
    bad, table, nets = [], [], {}
    for pre, got, want, twin in (('Discriminator/', gd, d_grads, d_grads32), ('Generator/', gg, g_grads, g_grads32)):
        gscale = max(float(t.abs().max()) for t in want.values())
        e_hip = e_cpu = n_ref = 0.0
        for k, t in want.items():
            w = t.numpy()
            scale = max(float(np.abs(w).max()), 1e-3 * gscale)
            dh = got[pre + k].astype(np.float64) - w
            dc = twin[k].numpy().astype(np.float64) - w
            d, d32_ = np.abs(dh) / scale, np.abs(dc) / scale
            e_hip, e_cpu, n_ref = e_hip + float((dh * dh).sum()), e_cpu + float((dc * dc).sum()), n_ref + float((w * w).sum())
            row = dict(tensor=pre + k, hip_share_off=float((d > 5e-4).mean()), hip_worst=float(d.max()),
                       fp32cpu_share_off=float((d32_ > 5e-4).mean()), fp32cpu_worst=float(d32_.max()),
                       hip_rel_l2=float(np.sqrt((dh * dh).sum() / max((w * w).sum(), 1e-300))),
                       fp32cpu_rel_l2=float(np.sqrt((dc * dc).sum() / max((w * w).sum(), 1e-300))))
            table.append(row)
            # per tensor: the share of entries off by more than 5e-4 of the tensor's scale is no larger than twice the fp32
            # CPU evaluation's (floor: 0.5 % of the entries).  (WHICH entries a flipped kink moves, and by how much, differs
            # between any two fp32 evaluations: the worst entry is recorded, the L2 bar below is what is asserted)
            if row['hip_share_off'] > max(0.005, 2.0 / w.size, 2.0 * row['fp32cpu_share_off']):     # (floor: 0.5 % or two entries)
                bad.append(row)
        # per network: the whole gradient vector is no further (L2) from the float64 gradient than twice the fp32 CPU
        # evaluation's (floor 1e-5 relative)
        nets[pre] = dict(hip_rel_l2=float(np.sqrt(e_hip / n_ref)), fp32cpu_rel_l2=float(np.sqrt(e_cpu / n_ref)))
        if nets[pre]['hip_rel_l2'] > max(1e-5, 2.0 * nets[pre]['fp32cpu_rel_l2']):
            bad.append((pre, nets[pre]))
    out_dir = os.environ.get('CFL_RECORD_DIR')
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'series_config5_gradients_b%d.json' % B), 'w') as fh:
            json.dump({'bar': 'per network: relative L2 error of the gradient vector <= max(1e-5, 2 x fp32 CPU); per tensor: share of '
                              'entries off by > 5e-4 of the tensor scale <= max(0.5 %, 2 entries, 2 x fp32 CPU share)', 'B': B,
                       'networks': nets, 'tensors': table}, fh, indent=1)
    assert not bad, bad


def test_exact_fp32_headline_trajectory_keeps_the_round3_bar():
    """CFL_EXACT_FP32=1 (k-ordered fp32 MFMA in both contractions; read when a plan is first built, hence a fresh
    process): the headline trajectory at the bars the exact path held before the bf16x3 projection became the default
    (|HIP - fp64| <= 4e-6, |HIP - fp32 CPU| <= 2e-6 over 100 steps; round 3 observed 1.9e-6 / 6.3e-7) -- so that the exact
    path cannot regress unnoticed behind the default arithmetic's looser caps."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys; sys.path[:0] = [%r, %r, %r]\n'
            'import test_baseline_configs_gpu as T\n'
            'r = T._trajectory("headline_exact_fp32", "dist", 4096, 3, 20, 512, 100, dict(), 16384, T.NV, (4e-6, 2e-6), min_auc=0.52)\n'
            'print("EXACT", r["max"]["hip_vs_fp64"], r["max"]["hip_vs_fp32cpu"])\n'
            % (root, os.path.join(root, 'compatibility-family-learning_amd'), os.path.join(root, 'tests')))
    env = dict(os.environ, CFL_EXACT_FP32='1')
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert 'EXACT' in r.stdout
