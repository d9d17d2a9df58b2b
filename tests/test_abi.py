"""CPU-only checks of the drop-in boundary: the shared library loads, exports every symbol include/poseprobe_hip.h
declares (no compute calls - there is no GPU here), argument validation works, and the product never imports the
oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from poseprobe_amd import _lib, build_ext
    build_ext.build()
    protos = _lib.parse_header()
    assert len(protos) >= 27
    L = ctypes.CDLL(_lib.SO_PATH)
    for name in protos:
        assert hasattr(L, name), f'{name} declared in the header but not exported'
    assert hasattr(L, 'pp_last_error')
    L2 = _lib.lib()
    assert L2.pp_abi_version() == _lib.header_abi_version() == 3
    assert not hasattr(L, 'pp_set_option') and not hasattr(L, 'pp_get_option')     # no process-wide switches left in the ABI


def test_pp_scene_struct_matches_header_layout():
    from poseprobe_amd._lib import pp_scene
    # 6 floats + 3 ints + 5 floats + int + float + 4 ints = 20 x 4 bytes, no padding
    assert ctypes.sizeof(pp_scene) == 20 * 4
    hdr = open(os.path.join(ROOT, 'include', 'poseprobe_hip.h')).read()
    body = hdr[hdr.index('typedef struct {'):hdr.index('} pp_scene;')]
    fields = re.findall(r'(?:float|int32_t)\s+(\w+)', body)
    assert fields == [f[0] for f in pp_scene._fields_]


def test_a_stale_library_is_refused(tmp_path, monkeypatch):
    """The binding compares pp_abi_version() with the header's PP_ABI_VERSION before anything is called (a caller
    built against another ABI would pass its stream where a pointer is read)."""
    from poseprobe_amd import _lib
    hdr = tmp_path / 'poseprobe_hip.h'
    hdr.write_text(open(_lib.HEADER).read().replace('#define PP_ABI_VERSION 3', '#define PP_ABI_VERSION 4'))
    monkeypatch.setattr(_lib, 'HEADER', str(hdr))
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'header_abi_version', lambda path=None: 4)
    with pytest.raises(_lib.PoseProbeError, match='ABI'):
        _lib.lib()


def test_options_live_in_caller_owned_contexts():
    """No process-wide option state: two contexts hold different values side by side, the compiled-in defaults are immutable
    (a NULL context cannot be written), unknown names and out-of-range values are refused."""
    from poseprobe_amd import _lib
    a, b = _lib.Context(mlp_split=0, nerf_split=0), _lib.Context()
    assert a.get('mlp_split') == 0 and b.get('mlp_split') == 31 == _lib.library_default('mlp_split')
    assert a.get('nerf_split') == 0 and b.get('nerf_split') == 1
    b.set('mlp_wgs', 48)
    assert a.get('mlp_wgs') == 0 and b.get('mlp_wgs') == 48 and _lib.library_default('mlp_wgs') == 0
    L = _lib.lib()
    assert L.pp_context_set_option(None, b'mlp_split', 0) == -1 and b'null context' in L.pp_last_error()
    with pytest.raises(_lib.PoseProbeError):
        a.set('no_such_option', 1)
    with pytest.raises(_lib.PoseProbeError):
        a.set('mlp_split', 64)
    assert set(a.options()) == set(_lib.OPTION_NAMES)


def test_null_arguments_are_rejected_with_a_message():
    from poseprobe_amd import _lib
    L = _lib.lib()
    rc = L.pp_pose_bwd(None, None, 3, None, None)
    assert rc == -1
    assert b'pp_pose_bwd' in L.pp_last_error()
    with pytest.raises(_lib.PoseProbeError):
        _lib.call('pp_pose_bwd', None, None, 3, None, None)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it."""
    pkg = os.path.join(ROOT, 'poseprobe_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in txt.replace('the oracle', '').replace('oracle)', '') or f == 'synthetic.py', \
                    f'{f} mentions the oracle'
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    uses = [l for l in bench.splitlines() if 'from oracle' in l or 'import oracle' in l]
    assert len(uses) == 3 and 'voxurf_oracle' in uses[0] and 'scene_nerf' in uses[1] and 'voxurf_oracle' in uses[2]
    # inside the CPU legs only - cpu_baseline(), its scene-branch helper cpu_baseline_scene() and the PSNR-parity checker
    # cpu_baseline_psnr() (where the oracle is the reference trajectory the HIP engine is compared with) - nothing that is
    # measured as `value` or shipped
    marks = [bench.index('def cpu_baseline('), bench.index('def cpu_baseline_scene('), bench.index('def cpu_baseline_psnr('),
             bench.index('def dual_branch_leg(')]
    pos, at = [], 0
    for u in uses:
        at = bench.index(u, at)
        pos.append(at)
        at += 1
    assert marks[0] < pos[0] < marks[1] < pos[1] < marks[2] < pos[2] < marks[3]


def test_ops_refuse_cpu_tensors():
    import torch
    from poseprobe_amd import ops
    a = torch.zeros(4)
    with pytest.raises(RuntimeError, match='CUDA'):
        ops.alpha2weight_fwd(a, a, 1, a, a, a, a)
