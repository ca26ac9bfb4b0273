"""Build libcfl_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Every source is compiled to its own object (in parallel, rebuilt only when it or a header it may include changed)
and the objects are linked into lib/libcfl_hip.so."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ('cfl_hip', 'cfl_conv', 'cfl_gan', 'cfl_eval', 'cfl_dp')
SRCS = [os.path.join(HERE, 'csrc', n + '.hip') for n in NAMES]
SRC = SRCS[0]
HEADERS = [os.path.join(HERE, 'csrc', h) for h in ('gemm_gather.h', 'conv_halo.h', 'conv_halo_wgrad.h', 'conv_stem.h', 'theta_planes.h',
                                                    'pair_proj.h', 'pair_grad.h', 'pair_mid.h', 'pair_finalize.h')] + \
          [os.path.join(os.path.dirname(HERE), 'include', 'cfl_hip.h')]
OBJ_DIR = os.path.join(HERE, 'build')
OUT = os.path.join(HERE, 'lib', 'libcfl_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC']


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    extra = [f for f in os.environ.get('CFL_HIPCC_FLAGS', '').split() if f]
    # the objects of a build with extra flags (-DCFL_STAMPS, ablations) must never be mistaken for production objects by a
    # later plain build: the flags of the last build are recorded beside the objects, a mismatch makes every object stale
    stamp = os.path.join(OBJ_DIR, 'flags.txt')
    flags_now = ' '.join(FLAGS + extra)
    try:
        with open(stamp) as f:
            same_flags = f.read() == flags_now
    except OSError:
        same_flags = False
    if not same_flags:
        force = True
    jobs = []
    for src, name in zip(SRCS, NAMES):
        obj = os.path.join(OBJ_DIR, name + '.o')
        if force or _stale(obj, [src] + HEADERS + [os.path.abspath(__file__)]):
            cmd = [HIPCC] + FLAGS + extra + ['-c', src, '-o', obj]
            if verbose:
                cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
            jobs.append(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 5)) as ex:
            for rc, cmd in zip(ex.map(subprocess.call, jobs), jobs):
                if rc:
                    raise subprocess.CalledProcessError(rc, cmd)
    with open(stamp, 'w') as f:
        f.write(flags_now)
    objs = [os.path.join(OBJ_DIR, n + '.o') for n in NAMES]
    if jobs or _stale(OUT, objs):
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', OUT])
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
