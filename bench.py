"""Headline benchmark: denoised motion clips/sec of a full p_sample_loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: a complete 1000-step DDPM `p_sample_loop`
(BASELINE.json configs[1]: batch 64 synthetic (263,1,196) clips, 8-layer/512-dim denoiser,
root_horizontal inpainting, cosine schedule, FIXED_SMALL variance, x0-prediction).  Every rank runs
its own batch on its own GPU (sampling shards by clip, no data-path collective: weak scaling);
`value` = clips all ranks denoised / max-over-ranks wall time.  Inputs (weights, x_T, text embedding,
mask, content clip) are resident in HBM before the timed region; per-step noise is generated in
the fused step kernel (Philox), as the reference draws randn_like on the device inside its loop.

The JSON line also carries
  roofline      the dominant kernel family by device time: algorithmic FLOPs per launch / its
                average launch duration measured with HIP events on the launch stream inside the
                timed region (every 16th denoise step is instrumented), against the dense
                f16/bf16 MFMA peak.
  cpu_baseline  the CPU oracle (a port of the reference's fp32 path; oracle/) timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0     # dense f16/bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
D, FF, H = 512, 1024, 4


def flops_per_launch(family, rows, T, F, kin_pad):
    """Algorithmic FLOPs (2*M*N*K) of one launch of a kernel family; rows = clips through the
    transformer, S = T + 1 tokens each."""
    S = T + 1
    M = rows * S
    return {
        "qkv_gemm": 2.0 * M * 3 * D * D,
        "attention": 2.0 * 2 * rows * H * S * S * (D // H),
        "outproj_ln_gemm": 2.0 * M * D * D,
        "ffn1_gelu_gemm": 2.0 * M * FF * D,
        "ffn2_ln_gemm": 2.0 * M * D * FF,
        "embed_in": 2.0 * rows * T * D * F,
        "embed_out_step": 2.0 * rows * T * F * D,
        "cond_token": 0.0,
        "qkv_attention_fused": 2.0 * M * 3 * D * D + 2.0 * 2 * rows * H * S * S * (D // H),
    }[family]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed p_sample_loop passes")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--denoise-steps", type=int, default=1000)
    ap.add_argument("--cfg", action="store_true", help="configs[2]: classifier-free guidance (doubled batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--shared-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (use with --backend gloo)")
    ap.add_argument("--cpu-sample-steps", type=int, default=24)
    args = ap.parse_args()

    import numpy as np
    import torch
    import mst_amd  # noqa: F401
    from mst_amd import sharding, synthetic as syn
    from mst_amd.engine import DenoiserEngine, Schedule, SAMPLER_DDPM
    from mst_amd.diffusion.gaussian_diffusion import schedule_tables

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dist = None
    if args.shared_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    F, T, B, NS = 263, 196, args.batch, args.denoise_steps
    rows = 2 * B if args.cfg else B
    seed = 20261003
    eng = DenoiserEngine(F, T, rows, device=dev)
    w = syn.denoiser_state(seed, F)
    pe = syn.positional_table(5000, 512)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(pe))
    tab, tmap = schedule_tables("cosine", 1000, "" if NS == 1000 else str(NS))
    sch = Schedule(tab, tmap, dev)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # per-rank inputs (rank offsets the stream so ranks denoise different clips)
    txt = to(syn.normal(seed + rank, "bench/txt", (B, 512)))
    x_T = to(syn.normal(seed + rank, "bench/xT", (B, F, 1, T)))
    motion = to(syn.normal(seed + rank, "bench/motion", (B, F, 1, T)))
    mask = to(syn.root_horizontal_mask(B, F, T))
    scale = to(np.full((B,), 2.5, np.float32)) if args.cfg else None
    eng.set_text(txt, cfg=args.cfg)

    def one_pass(k):
        x = x_T.clone()
        eng.sample_loop(sch, x, NS - 1, 0, SAMPLER_DDPM, cfg=args.cfg, scale=scale, mask=mask, motion=motion,
                        mask_noise=True, seed=sharding.rank_seed(seed, rank, k))
        return x

    def barrier():
        sharding.barrier(dev)

    for k in range(args.warmup):
        one_pass(k)
    eng.profile(True, 16)
    barrier()
    t0 = time.perf_counter()
    last = None
    for k in range(args.steps):
        last = one_pass(args.warmup + k)
    barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    assert torch.isfinite(last).all()
    assert torch.equal(last[:, :3], motion[:, :3]), "inpainted rows must equal the content clip exactly"
    dt = sharding.max_over_ranks(dt, dev if args.backend == "nccl" else "cpu")

    if rank == 0:
        clips = world * B * args.steps
        value = clips / dt
        slices = eng.loop_slices(B, args.cfg)        # un-instrumented steps run this many clip slices concurrently;
        # the event-timed (every 16th) steps run as one full-batch slice, so per-launch work below is the full batch
        fam_ms = {k: (ms / n if n else 0.0) for k, (ms, n) in prof.items()}
        fam_tot = {k: ms for k, (ms, n) in prof.items()}
        dom = max(fam_tot, key=fam_tot.get)
        fl = flops_per_launch(dom, rows, T, F, 320)
        achieved = fl / (fam_ms[dom] * 1e-3) / 1e12 if fam_ms[dom] > 0 else 0.0
        # whole path: 7.353 GFLOP per clip per denoise step (14.706 with CFG), SURVEY.md section 8d
        flops_per_clip = 7.353e9 * NS * (2 if args.cfg else 1)
        traffic, traffic_src = pmc_traffic(dom, rows)
        line = {
            "metric": f"denoised motion clips/sec ({NS}-step DDPM, Bx263x196)",
            "value": round(value, 4), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 MFMA operands, fp32 accumulate/stream", "data": "synthetic",
            "config": {"workload": ("configs[2]" if args.cfg else "configs[1]") +
                       f": batch {B}/GPU x (263,1,196), {NS}-step DDPM p_sample_loop, 8-layer/512-dim denoiser, "
                       "root_horizontal inpainting" + (", classifier-free guidance scale 2.5 (doubled batch)" if args.cfg else ""),
                       "global_batch": world * B, "denoise_steps": NS, "parallelism": f"clip-sharded x{world}, no collective"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_us": round(1e3 * fam_ms[dom], 2), "launches_timed": prof[dom][1],
                         "clips_per_timed_launch": rows, "concurrent_slices_elsewhere": slices,
                         "whole_path_tflops": round(value * flops_per_clip * 1e-12, 2),
                         "kernel_avg_us": {k: round(1e3 * v, 2) for k, v in fam_ms.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, pe, tab, tmap, B, F, T, NS, args.cpu_sample_steps, seed)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def pmc_traffic(family, rows):
    """HBM bytes per launch of the dominant kernel from the PMC passes (tools/pmc_traffic.sh: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE runs with full-batch launches, units and gfx950 correction per the microarch guide), committed as
    profiles/r01_final_pmc_traffic.json.  Counters cannot be read from inside this process, so the figure is null unless that
    file was collected for the same per-launch work (64 clips, no CFG)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_final_pmc_traffic.json")
    key = "ln_gemm" if family in ("outproj_ln_gemm", "ffn2_ln_gemm") else family
    try:
        k = json.load(open(path))["kernels"][key]
    except (OSError, KeyError, ValueError):
        return None, None
    if rows != 64:
        return None, None
    return k["hbm_bytes"], "profiles/r01_final_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 64-clip launches)"


def cpu_baseline(w, pe, tab, tmap, B, F, T, NS, sample_steps, seed):
    """The oracle (CPU port of the reference's fp32 path) on the host cores, on a bounded sample:
    `sample_steps` denoise steps of the same batch-B loop, extrapolated linearly to NS steps (steps
    cost the same).  Reported next to the GPU number, never a target."""
    import numpy as np
    import torch
    from mst_amd import synthetic as syn
    from oracle import denoiser, diffusion
    torch.set_num_threads(min(16, os.cpu_count() or 1))      # the box's CPU share for one GPU
    cores = torch.get_num_threads()
    shape = (B, F, 1, T)
    txt = syn.normal(seed, "bench/txt", (B, 512))
    motion = syn.normal(seed, "bench/motion", shape)
    mask = syn.root_horizontal_mask(B, F, T)
    g = torch.Generator().manual_seed(0)
    noise_fn = lambda k: torch.randn(shape, generator=g)
    with torch.no_grad():
        t0 = time.perf_counter()
        diffusion.sample_loop(lambda xx, tt: denoiser.forward(w, pe, xx, tt, txt), tab, tmap, shape, noise_fn, "ddpm", True,
                              mask, motion, init_image=motion, skip_timesteps=NS - sample_steps)
        dt = time.perf_counter() - t0
    per_step = dt / sample_steps
    return {"value": round(B / (per_step * NS), 5), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{sample_steps} of {NS} denoise steps at batch {B} ({dt:.1f} s), extrapolated linearly",
            "s_per_denoise_step": round(per_step, 4)}


if __name__ == "__main__":
    main()
