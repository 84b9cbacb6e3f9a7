"""The RCCL code path executed on the one GPU of the test box: torch.distributed's "nccl" backend (= RCCL on ROCm) with a
world of ONE rank, in a child process that creates the process group before anything else touches the GPU
(AWARE_FORCE_COLLECTIVES=1 makes aware_amd.parallel run its collectives for a single rank too).  Covers communicator set-up,
the float64 metric all-reduces (SUM / MAX), the flat 1 681 960-float gradient bucket, the barrier and bench.py under
torch.distributed.run -- the launch form of the driver's multi-GPU runs.  Multi-rank semantics are covered on
gloo (tests/test_distributed_cpu.py, world size 2)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import torch
from aware_amd import parallel
rank, world, local = parallel.init_distributed()
import torch.distributed as dist
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", torch.cuda.current_device())
sums, maxes = parallel.reduce_metrics({"bit_errors": 3, "bits": 5120, "seconds": 768.5}, {"wall": 0.4125}, device=dev)
assert sums == {"bit_errors": 3.0, "bits": 5120.0, "seconds": 768.5} and maxes == {"wall": 0.4125}, (sums, maxes)
shapes = [(512, 128), (1024, 512), (1024, 1024), (40, 1024), (512,), (1024,), (1024,), (40,)]
g = torch.Generator(device="cuda").manual_seed(1)
grads = [torch.randn(s, device=dev, generator=g) for s in shapes]
assert sum(t.numel() for t in grads) == 1681960
keep = [t.clone() for t in grads]
parallel.all_reduce_gradients(grads, average=True)
torch.cuda.synchronize()
assert all(torch.equal(a, b) for a, b in zip(grads, keep))          # one rank: the average is the input, bit for bit
parallel.barrier()
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.get_device_name(0))
""" % ROOT


def _env(port):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AWARE_FORCE_COLLECTIVES="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_rccl_collectives_with_one_rank():
    p = subprocess.run([sys.executable, "-c", CHILD], env=_env(29671), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def test_bench_under_torchrun_with_rccl():
    """bench.py as the driver launches it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), here with N = 1 and
    the collectives forced on: RCCL barrier on both sides of the timed region, metric all-reduce on device tensors, one JSON
    line from rank 0."""
    env = {k: v for k, v in _env(0).items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", "29672", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1",
                        "--warmup", "0", "--clips-per-gpu", "32", "--no-cpu-baseline", "--no-profile"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["ber_percent_clean"] == 0.0


def test_default_bench_line_carries_the_contract_objects():
    """`python bench.py` (here with 32 clips per GPU and one step): ONE JSON line with the contract's keys, a `roofline`
    object for the dominant kernel timed live (the f16 two-term conv kernel from 32 clips on), a `cpu_baseline` object from the
    oracle on a bounded sample, and the `exact_pipe` leg."""
    env = {k: v for k, v in _env(0).items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                         "AWARE_FORCE_COLLECTIVES")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--clips-per-gpu", "32"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["unit"] == "waveform-seconds/sec" and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["steps"] == 1 and out["n_gpus"] == 1 and out["value"] > 0 and out["ber_percent_clean"] == 0.0
    assert abs(out["value"] - 32 * 3.0 / (out["ms_per_step"] * 1e-3)) < 0.01 * out["value"]
    r = out["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and "gemm_clip_h2_kernel" in r["kernel"]
    assert 0 < r["achieved"] < r["peak"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and abs(r["peak"] - 2500 / 3) < 0.1
    assert r["traffic"] is None or r["traffic"] > 0                 # null unless a stored counter profile matches batch and sources
    assert r["dsp_hbm"]["bound"] == "hbm" and 0 < r["dsp_hbm"]["frac"] < 1
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and 0 < c["value"] < out["value"] and "clips" in c["sample"]
    e = out["exact_pipe"]
    assert e["conv_pipe"] == "bf16x3" and 0 < e["value"] <= 1.2 * out["value"] and e["ber_percent"] == 0.0


def test_training_workload_with_rccl_all_reduce():
    """`bench.py --workload train` under torch.distributed.run with one rank and the collectives forced on: the gradient bucket's
    all-reduce and the metric reductions run on RCCL inside the timed region."""
    env = {k: v for k, v in _env(0).items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", "29673", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "train",
                        "--steps", "2", "--warmup", "1", "--clips-per-gpu", "16"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and 0 < out["loss"] < 10
