"""bench.py --gpus N without RANK in the environment starts its own N ranks (CPU test of the launch logic with a stub
child: no GPU, no torch.distributed.run)."""
import io
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(tmp_path, rc):
    p = tmp_path / 'stub_launcher.py'
    p.write_text('import json, os, sys\n'
                 'print(json.dumps({"argv": sys.argv[1:], "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))\n'
                 f'sys.exit({rc})\n')
    return [sys.executable, str(p)]


def test_launch_relays_rank0_output_and_return_code(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    for rc in (0, 7):
        buf = io.StringIO()
        got = bench.launch_ranks(4, ['--gpus', '4', '--steps', '3'], device_count=lambda: 8, launcher=_stub(tmp_path, rc), out=buf)
        assert got == rc
        rec = json.loads(buf.getvalue())
        assert rec['argv'][0] == os.path.join(ROOT, 'bench.py') and rec['argv'][1:] == ['--gpus', '4', '--steps', '3']
        assert rec['ipc'] == os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def test_launch_refuses_more_ranks_than_gpus(tmp_path, capsys):
    sys.path.insert(0, ROOT)
    import bench
    buf = io.StringIO()
    assert bench.launch_ranks(8, ['--gpus', '8'], device_count=lambda: 1, launcher=_stub(tmp_path, 0), out=buf) == 3
    assert buf.getvalue() == ''
    assert 'only 1 GPU' in capsys.readouterr().err


def test_default_launcher_is_torch_distributed_run(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class FakeProc:
        stdout = []

        def wait(self):
            return 0

    def fake_popen(cmd, **kw):
        seen['cmd'] = cmd
        return FakeProc()

    monkeypatch.setattr(bench.subprocess, 'Popen', fake_popen)
    assert bench.launch_ranks(2, ['--gpus', '2'], device_count=lambda: 2) == 0
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=2' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-3].endswith('bench.py') and cmd[-2:] == ['--gpus', '2']


def test_plain_invocation_with_more_gpus_than_present_exits_nonzero():
    """No GPU in the build container: `python bench.py --gpus 2` must fail fast and clearly, never with a traceback."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 3 and 'GPU(s) are visible' in r.stderr and 'Traceback' not in r.stderr
