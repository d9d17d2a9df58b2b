"""Builds libposeprobe_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m poseprobe_amd.build_ext [--force]
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libposeprobe_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-shared', '-Wno-pass-failed',
         '-Wno-unused-result', '-Wno-unused-value']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(HERE, '..', 'include', 'poseprobe_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not is_stale():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + os.environ.get('PP_EXTRA_HIPCC_FLAGS', '').split() + sources() + ['-o', OUT]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(OUT)
