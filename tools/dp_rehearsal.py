"""Two-rank rehearsal of the data-parallel paths on ONE card (both ranks use cuda:0, gloo carries the collectives; the pool
offers no multi-GPU box to the builder).  What it checks is equality, not speed:

  fine-tune   each rank differentiates its 64-clip shard through the NATIVE training node with the event-gated per-layer bucket
              reducer (finetune_dp.LayerBucketReducer: all-reduce of bucket l launched behind layer l's gradient event while the
              backward pass is still enqueuing); rank 0 then differentiates the full 128-clip batch alone and compares all 96
              reduced tensors with it; every bucket must have been launched from inside the backward pass.
  sampling    a ragged global batch (5 clips -> 3 + 2) denoised shard-wise with per-rank Philox keys (sharding.rank_seed),
              gathered with sharding.gather_clips: the gathered tensor must hold each rank's clips in global order, the masked
              rows bit-equal to the content clips, and no two ranks may have drawn the same noise.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/dp_rehearsal.py
Prints one JSON line on rank 0 (committed as profiles/r03_dp_rehearsal_2ranks_one_gpu.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import torch.distributed as dist

import mst_amd  # noqa: F401
from mst_amd import sharding, synthetic as syn
from mst_amd.finetune_dp import LayerBucketReducer


def rel(a, b):
    return float((a - b).norm() / b.norm())


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    out = {"world": world, "backend": "gloo, both ranks on cuda:0"}

    # ------------------------------------------------------------------ fine-tune: reduced buckets == full-batch gradients
    from test_gpu_train import _style_model
    B = 64                                                # clips per rank (BASELINE configs[3])
    model = _style_model(dropout=0.0).eval()              # identical seeded weights on every rank; no dropout: a pure function of the batch
    Fe, T = 181, 76
    g = torch.Generator(device="cpu").manual_seed(1234)
    x_all = torch.randn(world * B, Fe, 1, T, generator=g).to(dev)
    tgt_all = torch.randn(world * B, Fe, 1, T, generator=g).to(dev)
    t_all = torch.randint(0, 1000, (world * B,), generator=g).to(dev)
    emb_all = torch.randn(world * B, 512, generator=g).to(dev)
    lo, hi = rank * B, (rank + 1) * B

    def loss_of(sl):
        return ((model(x_all[sl], t_all[sl], y={"text_embed": emb_all[sl]}) - tgt_all[sl]) ** 2).mean()

    red = LayerBucketReducer(model)
    red.zero_grad()
    loss_of(slice(lo, hi)).backward()
    red.finish()
    torch.cuda.synchronize(dev)
    reduced = {n: p.grad.clone() for n, p in model.named_parameters() if p.requires_grad}
    out["finetune"] = {"clips_per_rank": B, "buckets": len(red.buckets), "launch_order": red.launch_order,
                       "launched_in": red.launched_in, "bucket_bytes": red.bucket_bytes()}
    assert red.launched_in == ["backward"] * 8, red.launched_in
    assert red.launch_order == list(range(7, -1, -1)), red.launch_order
    dist.barrier()
    if rank == 0:
        red.enabled = False                               # the reference: the whole 2 x 64-clip batch in one process, no exchange
        red.zero_grad()
        loss_of(slice(0, world * B)).backward()
        red.finish()
        torch.cuda.synchronize(dev)
        errs = {n: rel(reduced[n], p.grad) for n, p in model.named_parameters() if p.requires_grad}
        worst = max(errs, key=errs.get)
        out["finetune"].update(tensors_compared=len(errs), worst_tensor=worst, worst_rel_l2=errs[worst])
        # same products; other summation orders (row blocks, split-K partition, mean of two means) and per-call gradient scales
        assert len(errs) == 96 and errs[worst] < 2e-5, (worst, errs[worst])
    dist.barrier()

    # ------------------------------------------------------------------ sampling: ragged shards, per-rank Philox keys, gather
    from mst_amd.engine import DenoiserEngine, Schedule, SAMPLER_DDPM
    from mst_amd.diffusion.gaussian_diffusion import schedule_tables
    GB, F2, T2, steps, seed = 5, 181, 76, 10, 77
    s0, s1 = sharding.shard_range(GB, rank, world)
    n = s1 - s0
    w = syn.denoiser_state(seed, F2)
    eng = DenoiserEngine(F2, T2, max(n, 1), device=dev)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
    tab, tmap = schedule_tables("cosine", 1000, "100")
    sch = Schedule(tab, tmap, dev)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    txt = to(syn.normal(seed, "txt", (GB, 512)))[s0:s1].contiguous()
    x = to(syn.normal(seed, "xT", (GB, F2, 1, T2)))[s0:s1].contiguous()
    motion_all = to(syn.normal(seed, "motion", (GB, F2, 1, T2)))
    motion = motion_all[s0:s1].contiguous()
    mask = to(syn.root_horizontal_mask(n, F2, T2))
    eng.set_text(txt)
    key = sharding.rank_seed(seed, rank, 0)
    eng.sample_loop(sch, x, steps - 1, 0, SAMPLER_DDPM, mask=mask, motion=motion, mask_noise=True, seed=key)
    torch.cuda.synchronize(dev)
    full = sharding.gather_clips(x.cpu(), GB)            # (gloo moves CUDA tensors only in all_reduce / broadcast; RCCL gathers device tensors)
    keys = [None] * world
    dist.all_gather_object(keys, key)
    noise0 = eng.philox_normal(n, T2, key, 0)[:1].cpu()
    firsts = [torch.empty_like(noise0) for _ in range(world)]
    dist.all_gather(firsts, noise0)
    assert full.shape[0] == GB and torch.equal(full[s0:s1], x.cpu())
    assert torch.equal(full[:, :3], motion_all[:, :3].cpu())                     # inpainted rows of EVERY rank's clips, in global order
    assert len(set(keys)) == world and not torch.equal(firsts[0], firsts[1])
    out["sampling"] = {"global_batch": GB, "shards": [list(sharding.shard_range(GB, r, world)) for r in range(world)],
                       "philox_keys": keys, "gathered_shape": list(full.shape), "masked_rows_bit_exact": True}
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
