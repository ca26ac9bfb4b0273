"""Three deviation series per compared quantity (round 5):

    hip_vs_fp64      |x_HIP    - x_fp64| / max(1, |x_fp64|)       what the test is about
    fp32cpu_vs_fp64  |x_fp32cpu - x_fp64| / max(1, |x_fp64|)      what fp32 arithmetic itself costs on this graph: the SAME oracle
                                                                code evaluated in float32 on the CPU (north_star's reference IS
                                                                an fp32 CPU path)
    hip_vs_fp32cpu   |x_HIP - x_fp32cpu| / max(1, |x_fp64|)

The bar is  hip_vs_fp64 <= max(floor, factor * (largest fp32cpu_vs_fp64 over ALL steps of the run)), checked at the end:
"the HIP trajectory never leaves float64 by more than `factor` times what an fp32 evaluation of the same graph leaves it by
over the same steps" -- instead of a fixed tolerance chosen to pass.  (Over all steps, not step by step: once a kink has
flipped a free-running trajectory leaves float64 about tenfold per step, and WHICH step that happens in differs between any
two fp32 evaluations -- the CPU twin's own result depends on the host's BLAS threading -- so a per-step ratio would compare
the luck of two draws.)
After step 0 the conv / GAN trajectories are chaotic in fp32 (a near-zero pre-activation lands on the other side of an
lrelu / relu kink and changes a slope from 0.2 to 1: tests/test_activation_masks_gpu.py), for the CPU evaluation exactly
as for HIP, which is why the fp32 twin is the right yardstick.  CFL_RECORD_DIR=<dir> writes <dir>/series_<name>.json;
the committed copies are profiles/r05_series_*.json.
"""
import json
import os


class ParitySeries(object):
    def __init__(self, name, floor, factor=2.0, meta=None):
        self.name, self.floor, self.factor = name, float(floor), float(factor)
        self.rows = []           # (step, key, hip_vs_fp64, fp32cpu_vs_fp64, hip_vs_fp32cpu)
        self.meta = dict(meta or {})
        self._worst32 = 0.0
        self.failures = []

    def add(self, step, key, hip, f64, f32, floor=None):
        den = max(1.0, abs(float(f64)))
        h64 = abs(float(hip) - float(f64)) / den
        c32 = abs(float(f32) - float(f64)) / den
        h32 = abs(float(hip) - float(f32)) / den
        self.rows.append((int(step), key, h64, c32, h32))
        self._worst32 = max(self._worst32, c32)
        return h64, c32

    def record(self):
        out_dir = os.environ.get('CFL_RECORD_DIR')
        if not out_dir:
            return
        os.makedirs(out_dir, exist_ok=True)
        steps = sorted(set(r[0] for r in self.rows))
        per_step = {s: dict(hip_vs_fp64=max(r[2] for r in self.rows if r[0] == s),
                            fp32cpu_vs_fp64=max(r[3] for r in self.rows if r[0] == s),
                            hip_vs_fp32cpu=max(r[4] for r in self.rows if r[0] == s)) for s in steps}
        with open(os.path.join(out_dir, 'series_%s.json' % self.name), 'w') as fh:
            json.dump({'name': self.name, 'meta': self.meta, 'floor': self.floor, 'factor': self.factor,
                       'bar': 'hip_vs_fp64 <= max(floor, factor * max over all steps of fp32cpu_vs_fp64)',
                       'worst': dict(hip_vs_fp64=max(r[2] for r in self.rows), fp32cpu_vs_fp64=self._worst32,
                                     hip_vs_fp32cpu=max(r[4] for r in self.rows)),
                       'per_step_max': per_step,
                       'rows': [dict(step=r[0], key=r[1], hip_vs_fp64=r[2], fp32cpu_vs_fp64=r[3], hip_vs_fp32cpu=r[4])
                                for r in self.rows]}, fh, indent=1)

    def check(self):
        self.record()
        bar = max(self.floor, self.factor * self._worst32)
        self.failures = [(r[0], r[1], r[2], bar, r[3]) for r in self.rows if r[2] > bar]
        assert not self.failures, ('HIP further from float64 than %.1f x the fp32 CPU evaluation (step, key, hip_vs_fp64, '
                                   'bar, fp32cpu_vs_fp64): %r' % (self.factor, self.failures[:6]))
