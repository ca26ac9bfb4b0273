"""repro of the 4-rank fuzz failure (cfl pcd D=4096 L=36 K=4, one row per rank): the same per-rank shape on a one-rank group"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'compatibility-family-learning_amd')]
import tests.test_dp_step_gpu as T
cases = [('cfl', 'pcd', 4096, 36, 4, b, {'reg_const': 1e-4}, False, 5) for b in (1, 2, 4, 64)] + \
        [('cfl', 'pcd', 4096, 36, 4, 1, {}, False, 5), ('dist', 'pcd', 4096, 36, 4, 1, {}, False, 5), ('cfl', 'pcd', 4096, 20, 3, 1, {}, False, 5)]
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
for c in cases:
    code = T._FAMILIES % dict(root=ROOT, port=str(38700 + os.getpid() % 1000), cases=[c])
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')]
    print(c[:6], lines[-1] if lines else ('rc %d: ' % r.returncode) + r.stderr.strip().splitlines()[-1][:300], flush=True)
