"""bench.py's N > 1 entry (no GPU needed): started directly with --gpus N it must launch its N ranks itself -- as children, through
torch.distributed.run on 127.0.0.1, before anything in the parent touches a GPU -- and leave with their exit code; under
torch.distributed.run it must refuse a --gpus that disagrees with WORLD_SIZE.  The child launch is replaced by a stub."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.syspath_prepend(ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    return importlib.import_module("bench")


def test_self_launch_builds_the_torchrun_command(bench, monkeypatch):
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = list(cmd), dict(env)
        return Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    rc = bench.self_launch(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert rc == 7                                                    # the ranks' exit code is the parent's
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"       # (the container's hostname may not resolve)
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 <= port < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]      # every flag travels to the ranks
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"      # dmabuf IPC, or RCCL cannot share device memory between ranks


def test_main_launches_ranks_only_when_the_gpus_exist(bench, monkeypatch):
    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda n, argv: calls.append((n, list(argv))) or 0)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "only 1 visible" in str(e.value) and not calls            # asked for more GPUs than there are: no launch, a message
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and calls == [(2, ["--gpus", "2", "--steps", "2"])]


def test_main_refuses_a_world_size_that_disagrees_with_gpus(bench, monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "4")                             # (under torch.distributed.run)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setattr(bench, "self_launch", lambda n, argv: pytest.fail("a rank must not launch ranks"))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "does not match WORLD_SIZE 4" in str(e.value)
