"""bench.py's own launcher on the CPU: `--gpus 2` without a torchrun environment must start two ranks (gloo),
reduce their counters and print one JSON line from rank 0; a world size that differs from --gpus is an error.
The `stub` workload does no GPU work, so this runs in the CPU-only container."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_gpus_2_spawns_two_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0",
                        "--workload", "stub"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 2
    assert out["value"] > 0


def test_world_size_mismatch_is_an_error():
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2
    assert "differs from --gpus" in p.stderr


def test_torchrun_environment_is_respected():
    """Under torch.distributed.run (the driver's launch form) bench.py must NOT spawn again: rank and world come
    from the environment."""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29653", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "stub"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0])["ranks_seen"] == 2


def test_single_rank_stub_line():
    """--gpus 1 (the driver's default call) with no launcher environment: no spawn, one JSON line."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--workload", "stub"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1
