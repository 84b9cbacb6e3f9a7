"""world_size-2 gloo test of the N>1 path: shard by clip, no data-path collective, counters
SUM-reduced and wall time MAX-reduced (aware_amd/parallel.py)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from aware_amd import parallel
    r, w, _ = parallel.init_distributed()
    assert (r, w) == (rank, world)
    costs = [188, 63, 626, 300, 100, 450, 75, 210]
    mine = parallel.shard_by_cost(costs, world)[rank]
    sums, maxes = parallel.reduce_metrics({"bit_errors": rank + 1, "bits": 20 * len(mine), "seconds": sum(costs[i] for i in mine)},
                                          {"wall": 1.0 + rank}, device="cpu")
    parallel.barrier()
    q.put((rank, mine, sums, maxes))
    torch.distributed.destroy_process_group()


def test_gloo_world2_shard_and_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    all_idx = sorted(res[0][1] + res[1][1])
    assert all_idx == list(range(8))
    for _, _, sums, maxes in res:
        assert sums["bit_errors"] == 3.0 and sums["bits"] == 160.0 and sums["seconds"] == 2012.0
        assert maxes["wall"] == 2.0


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from aware_amd import parallel
    parallel.init_distributed(cpu_only=True)
    # the detector's parameter shapes (model card): gradients differ per rank, the average must come back on both
    shapes = [(512, 128), (1024, 512), (1024, 1024), (40, 1024), (512,), (1024,), (1024,), (40,)]
    g = torch.Generator().manual_seed(100 + rank)
    grads = [torch.randn(s, generator=g) for s in shapes]
    mine = [t.clone() for t in grads]
    parallel.all_reduce_gradients(grads, average=True)
    q.put((rank, [float(t.double().sum()) for t in mine], [float(t.double().sum()) for t in grads], sum(t.numel() for t in grads)))
    torch.distributed.destroy_process_group()


def test_gloo_world2_gradient_all_reduce():
    """EXTENSION (detector training): the flat-bucket gradient all-reduce of aware_amd/parallel.py on two gloo ranks --
    every rank ends with the average of the two ranks' gradients, tensor by tensor."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, m0, a0, n0), (_, m1, a1, n1) = res
    assert n0 == n1 == 1681960                                     # SURVEY: 1 681 960 parameters
    for x0, x1, y0, y1 in zip(m0, m1, a0, a1):
        assert abs(y0 - y1) < 1e-6 * max(1.0, abs(y0))             # both ranks hold the same tensors
        assert abs(y0 - 0.5 * (x0 + x1)) < 1e-3 * max(1.0, abs(x0) + abs(x1))
