"""B sweep of the headline step (run on the GPU box): separates the fixed per-step overhead (launch / dispatch
latencies of the 3 launches, latency chains) from the asymptotic per-row cost.

    python tools/b_sweep.py > gpurun_out/b_sweep.md

Each line is one `bench.py --batch-size B` run (median of >= 50 repeats); the least-squares line
t(B) = t0 + B / rate through the points gives the fixed overhead t0 and the asymptotic rows/s."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for B in (256, 512, 1024, 2048, 4096, 8192):
    batch_mib = 4 * B * 4096 * 4 >> 20
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--batch-size', str(B), '--steps', '20',
                        '--warmup', '5', '--pool-mib', str(max(384, 3 * batch_mib)), '--no-cpu-baseline',
                        '--no-cli-loop', '--no-other-configs', '--no-live-traffic', '--no-dp-form'], capture_output=True, text=True, cwd=ROOT)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    if not line:
        print('B=%d failed: %s' % (B, r.stderr[-500:]), file=sys.stderr)
        continue
    d = json.loads(line[-1])
    k = d['roofline']['kernels']
    rows.append((B, 1e3 * d['ms_per_step'], d['value'], d['roofline']['step']['hbm_frac'],
                 {n: k[n]['avg_us'] for n in k}))
print('| B (rows/step) | us/step | M rows/s | step HBM frac (16*D*B / t / 8 TB/s) | event intervals (us): proj / mid / grad |')
print('|---|---|---|---|---|')
for B, us, v, hf, k in rows:
    print('| %d | %.2f | %.2f | %.3f | %s |' % (B, us, v / 1e6, hf, ' / '.join('%.1f' % k.get(n, 0) for n in ('proj', 'mid', 'grad'))))
if len(rows) >= 3:
    Bs = np.array([r[0] for r in rows], float)
    ts = np.array([r[1] for r in rows], float)
    A = np.stack([np.ones_like(Bs), Bs], 1)
    (t0, slope), *_ = np.linalg.lstsq(A, ts, rcond=None)
    print()
    print('fit t(B) = t0 + B / rate: t0 = %.1f us fixed per step, asymptotic rate = %.1f M rows/s = %.2f TB/s of input '
          '(%.1f %% of 8 TB/s)' % (t0, 1.0 / slope, 65536.0 / slope / 1e6, 65536.0 / slope / 1e6 / 8.0 * 100))
