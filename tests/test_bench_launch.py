"""bench.py --gpus N starts its own ranks: the parent spawns the launcher before it has
imported torch, passes the ranks' JSON line on and leaves with their exit code."""
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_parent_spawns_the_launcher_before_touching_the_gpu(monkeypatch, capsys):
    bench = _load_bench()
    seen = {}

    def fake_run(cmd, stdout=None, text=None):
        seen["cmd"] = cmd
        seen["torch_loaded_by_bench"] = "torch" in sys.modules and getattr(sys.modules["torch"], "_bench_marker", False)
        return types.SimpleNamespace(returncode=0, stdout='NCCL banner\n{"metric": "x", "value": 1.0, "n_gpus": 4}\n')

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0])["n_gpus"] == 4          # ONE line on stdout, the ranks' result


def test_parent_passes_a_failure_on(monkeypatch):
    bench = _load_bench()
    monkeypatch.setattr(subprocess, "run", lambda cmd, stdout=None, text=None: types.SimpleNamespace(returncode=3, stdout=""))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3


@pytest.mark.gpu
def test_bench_gpus_2_runs_by_itself_on_one_gpu():
    """Two ranks on the one GPU of the test box (gloo moves the tensors: RCCL refuses two ranks on one device): the
    launch, the strong-scaled frame, the re-cut bands, the JSON line."""
    env = dict(os.environ, LENTIL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--width", "960", "--height", "540", "--samples", "64", "--f-hi", "0.001", "--no-config5"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["scaling"] == "strong"
    assert d["value"] > 0 and d["config"]["visits_per_gpu"] > 0
    assert "960x540 frame tiled over 2 GPU(s)" in d["config"]["workload"]
    assert isinstance(d["ranks"], list) and len(d["ranks"]) == 2
