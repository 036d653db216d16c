"""bench.py's launch contract (no GPU needed): `--gpus N` must never silently measure another job."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)   # top level: standard library imports and constants only
    return mod


def test_world_size_mismatch_is_an_error():
    """Under a launcher WORLD_SIZE must equal --gpus: a warning would leave a one-GPU number labelled as an N-GPU run."""
    for world, gpus in (("4", "2"), ("2", "1")):
        env = dict(os.environ, WORLD_SIZE=world, RANK="0", LOCAL_RANK="0")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus], env=env, capture_output=True, text=True,
                             timeout=120)
        assert out.returncode != 0
        assert "WORLD_SIZE" in out.stderr and not out.stdout.strip()


def test_self_launch_starts_fresh_ranks_before_torch(monkeypatch):
    """Without WORLD_SIZE, --gpus N > 1 starts N ranks under torch.distributed.run (the driver's command line) as child
    processes, passes the original arguments on and returns their exit code; torch is not imported by the launching process."""
    bench = _bench_module()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--log2n", "20"])
    had_torch = "torch" in sys.modules
    assert bench.launch_ranks(4) == 7
    assert had_torch or "torch" not in sys.modules
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--log2n", "20"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")


def test_main_dispatches_to_the_launcher(monkeypatch):
    bench = _bench_module()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setattr(bench, "launch_ranks", lambda n: 5 if n == 2 else 0)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 5
    else:
        raise AssertionError("main() must exit with the children's code")
