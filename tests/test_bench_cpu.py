"""bench.py's launch logic (no GPU needed): `--gpus N` must end up with N ranks or refuse to run."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_single_gpu_needs_no_spawn():
    assert bench.resolve_world(1, {}) == (1, False)
    assert bench.resolve_world(1, {"WORLD_SIZE": "1"}) == (1, False)


def test_multi_gpu_without_launcher_spawns_ranks():
    assert bench.resolve_world(8, {}) == (8, True)
    assert bench.resolve_world(2, {"PATH": "/usr/bin"}) == (2, True)


def test_multi_gpu_under_launcher_runs_in_place():
    assert bench.resolve_world(4, {"WORLD_SIZE": "4", "RANK": "3"}) == (4, False)


@pytest.mark.parametrize("gpus,ws", [(8, "1"), (1, "2"), (4, "8")])
def test_world_size_mismatch_refuses(gpus, ws):
    with pytest.raises(SystemExit) as e:
        bench.resolve_world(gpus, {"WORLD_SIZE": ws})
    assert "refusing" in str(e.value)


def test_spawn_command_is_the_drivers_launch_line():
    seen = {}

    def fake_run(cmd, env):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    rc = bench.spawn_ranks(4, ["--gpus", "4", "--steps", "2"], environ={"PATH": "/usr/bin"}, run=fake_run)
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 < port < 65536
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"       # dmabuf IPC: RCCL across processes needs it


def test_config3_is_one_million_in_total():
    assert bench.N_TOTAL_SHARDED == 1_000_000 and bench.N_PER_GPU == 100_000


def test_julia_probe_reports_unavailable_or_a_version():
    """the bench line says by itself whether the reference's runtime is on the box (cpu_baseline.reference_julia)."""
    import shutil

    got = bench.probe_julia()
    if shutil.which("julia") is None:
        assert got == "unavailable"
    else:
        assert "julia" in got.lower()
