"""Builds libposeprobe_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m poseprobe_amd.build_ext [--force]

One object per .hip source (csrc/_obj/*.o, compiled in parallel, rebuilt when the source or ANY header / generated
include of csrc/ or the public header is newer), then one link step.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
OUT = os.path.join(HERE, 'libposeprobe_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-Wno-pass-failed', '-Wno-unused-result',
         '-Wno-unused-value']
# per-source flags.  The files with split-precision MFMA kernels: the SLP vectoriser turns adjacent scalar fp32 arithmetic (epilogues, operand
# conversion) into v_pk_*_f32, which on gfx950 does not overlap with MFMAs (tools/mfma_valu_probe.hip: two v_pk_fma_f32 between
# MFMAs double the loop time, four plain VALU are free; MI355X_MICROARCH.md lists the same).  Measured: k_gemm_tn_split 123 -> 110 us, scene step 3.22 -> 3.10 ms.  NOT for
# pp_mlp_fused.hip: its fp32-instruction kernels are matrix-pipe bound and their serial epilogues are shorter packed
# (k_warp_fused_bwd 256 -> 282 us without SLP)
SOURCE_FLAGS = {f: ['-fno-slp-vectorize'] for f in ('pp_mlp_split.hip', 'pp_nerf.hip')}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def headers():
    """Everything a translation unit may include: csrc/*.h, generated csrc/*.inc, the public C header."""
    return (glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inc'))
            + [os.path.join(HERE, '..', 'include', 'poseprobe_hip.h')])


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')


def stale_sources():
    newest_header = max(os.path.getmtime(h) for h in headers())
    out = []
    for s in sources():
        o = _obj(s)
        if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_header):
            out.append(s)
    return out


def is_stale():
    if not os.path.exists(OUT) or stale_sources():
        return True
    return any(os.path.getmtime(_obj(s)) > os.path.getmtime(OUT) for s in sources())


def build(force=False, verbose=True):
    if not force and not is_stale():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    extra = os.environ.get('PP_EXTRA_HIPCC_FLAGS', '').split()
    os.makedirs(OBJ, exist_ok=True)
    todo = sources() if force else stale_sources()

    def compile_one(src):
        cmd = [hipcc] + FLAGS + SOURCE_FLAGS.get(os.path.basename(src), []) + extra + ['-c', src, '-o', _obj(src)]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + [_obj(s) for s in sources()] + ['-o', OUT]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(OUT)
