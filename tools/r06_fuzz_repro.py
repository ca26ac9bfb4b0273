"""the one failing case of tools/r06_fuzz.sh (cfl monomer D=256 L=1 K=8 B=257 directed pos_weight 0.5): bf16x3 vs exact fp32 kernels,
and neighbouring shapes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
os.environ['CFL_FUZZ_GRAD_CAP'] = '1e-9'      # make every case "fail" so that the assertion prints its worst tensor
import tests.test_hip_parity as T
from cfl import hipabi
T.H = hipabi
for exact in ('0', '1'):
    os.environ['CFL_EXACT_FP32'] = exact
    for (L, K, B, nv, directed) in ((1, 8, 257, 1.0, True), (1, 8, 257, 1.0, False), (2, 8, 257, 1.0, True), (1, 4, 257, 1.0, True),
                                    (1, 8, 64, 1.0, True), (1, 8, 257, 16.0, True)):
        try:
            T.test_step_fwd_bwd('cfl', 'monomer', 256, L, K, None, B, nv, {'pos_weight': 0.5}, directed)
            print('exact', exact, (L, K, B, nv, directed), 'passes at 1e-9?!')
        except AssertionError as e:
            print('exact', exact, (L, K, B, nv, directed), repr(e)[:200], flush=True)
