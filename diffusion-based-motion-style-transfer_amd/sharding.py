"""Multi-GPU host logic for the sampling path: clips are independent (no cross-sample op anywhere
in model/mdm_forstyledataset.py:602-625 or diffusion/gaussian_diffusion.py:311-585), so a batch is
partitioned over ranks by clip with NO data-path collective; ranks only meet at barriers, at the
max-over-ranks timing reduction and at an optional final gather of the finished clips.
One process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_range(global_batch, rank, world_size):
    """Contiguous [start, stop) of the clips rank `rank` denoises: sizes differ by at most one."""
    base, extra = divmod(global_batch, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def rank_seed(seed, rank, pass_index=0):
    """Distinct Philox keys per (rank, pass) so ranks never draw the same noise stream."""
    return (int(seed) * 1000003 + rank * 7919 + pass_index) & 0x7FFFFFFFFFFFFFFF


def barrier(device=None):
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(seconds, device="cpu"):
    """Whole-job wall time = the slowest rank's."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def group_report(own_seconds, units, device="cpu"):
    """What the PROCESS GROUP says about a timed region (bench.py puts it into its JSON line, so that a multi-GPU record proves the
    collective backend saw N ranks -- `n_gpus` alone is read from the launcher's WORLD_SIZE): world size and backend from
    torch.distributed, every rank's own rate (units / its own wall time) through an all_gather."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"initialized": False, "world_size": 1, "backend": None, "per_rank_units_per_s": [round(units / own_seconds, 3)]}
    ws = dist.get_world_size()
    t = torch.tensor([own_seconds], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(ws)]
    dist.all_gather(out, t)
    return {"initialized": True, "world_size": ws, "backend": str(dist.get_backend()), "rank": dist.get_rank(),
            "per_rank_units_per_s": [round(units / float(o.item()), 3) for o in out]}


def gather_clips(local, global_batch):
    """All ranks' finished clips in global order ([global_batch, F, 1, T]); ragged shards are padded
    to the largest shard for the collective and trimmed afterwards."""
    rank, ws = world()
    if ws == 1:
        return local
    sizes = [shard_range(global_batch, r, ws)[1] - shard_range(global_batch, r, ws)[0] for r in range(ws)]
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(out, buf)
    return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)
