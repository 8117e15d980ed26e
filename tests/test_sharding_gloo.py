"""N > 1 host path on CPU: two gloo ranks shard a batch by clip, reduce the wall time with MAX and
gather the clips back in order (the only collectives the sampling path has)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mst_amd  # noqa: F401
from mst_amd import sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, ws, port, gb, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        lo, hi = sharding.shard_range(gb, rank, ws)
        local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1, 1).expand(-1, 3, 1, 4).contiguous()
        sharding.barrier()
        t = sharding.max_over_ranks(1.0 + rank)
        full = sharding.gather_clips(local, gb)
        rep = sharding.group_report(1.0 + rank, 10.0)          # every rank's own rate, world size and backend from the process group
        q.put((rank, lo, hi, t, full[:, 0, 0, 0].tolist(), sharding.rank_seed(7, rank), sharding.world(), rep))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_timing_and_gather():
    ws, gb = 2, 7            # ragged on purpose: shards of 4 and 3 clips
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, gb, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(ws))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, full0, s0, w0, g0), (r1, lo1, hi1, t1, full1, s1, w1, g1) = res
    for g, r in ((g0, 0), (g1, 1)):
        assert g["initialized"] and g["world_size"] == 2 and g["backend"] == "gloo" and g["rank"] == r
        assert g["per_rank_units_per_s"] == [10.0, 5.0]
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7)          # disjoint, contiguous, covers the batch
    assert t0 == t1 == 2.0                                # max over ranks
    assert full0 == full1 == [float(i) for i in range(gb)]
    assert s0 != s1 and w0 == (0, 2) and w1 == (1, 2)


def test_single_process_defaults():
    assert sharding.world() == (0, 1)
    assert sharding.shard_range(64, 0, 1) == (0, 64)
    assert [sharding.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sharding.max_over_ranks(1.5) == 1.5
    assert sharding.group_report(2.0, 8.0) == {"initialized": False, "world_size": 1, "backend": None, "per_rank_units_per_s": [4.0]}
    x = torch.zeros(2, 3, 1, 4)
    assert sharding.gather_clips(x, 2) is x
