"""Build libcfl_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, 'csrc', f) for f in ('cfl_hip.hip', 'cfl_conv.hip', 'cfl_gan.hip', 'cfl_eval.hip', 'cfl_dp.hip')]
SRC = SRCS[0]
OUT = os.path.join(HERE, 'lib', 'libcfl_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def build(force=False, verbose=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    deps = SRCS + [os.path.join(HERE, 'csrc', 'gemm_gather.h'), os.path.join(HERE, 'csrc', 'conv_halo.h'),
                   os.path.join(os.path.dirname(HERE), 'include', 'cfl_hip.h')]
    if (not force and os.path.exists(OUT)
            and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)):
        return OUT
    cmd = [HIPCC, '-O3', '--offload-arch=gfx950', '-std=c++17', '-shared', '-fPIC',
           ] + SRCS + ['-o', OUT]
    if verbose:
        cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
