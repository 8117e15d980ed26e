"""Headline benchmark: denoised motion clips/sec of a full p_sample_loop on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: spawns the N ranks itself, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

--mode sample (default).  One "step" = one pass of the hot path over one batch: a complete 1000-step DDPM
`p_sample_loop` (BASELINE.json configs[1]: batch 64 synthetic (263,1,196) clips, 8-layer/512-dim denoiser,
root_horizontal inpainting, cosine schedule, FIXED_SMALL variance, x0-prediction; --cfg = configs[2]).  Every rank
runs its own batch on its own GPU (sampling shards by clip, no data-path collective: weak scaling); `value` = clips all
ranks denoised / max-over-ranks wall time.  Inputs (weights, x_T, text embedding, mask, content clip) are resident in
HBM before the timed region; per-step noise is generated in the fused step kernel (Philox), as the reference draws
randn_like on the device inside its loop.

The JSON line also carries
  roofline      the dominant kernel by device time PER ROCPROF SYMBOL (out-proj and FFN2 launches are one symbol, the
                LayerNorm-epilogue GEMM): algorithmic FLOPs per launch / average launch duration measured with HIP
                events on the launch stream inside the timed region (every 50th denoise step is instrumented), against
                the dense f16/bf16 MFMA peak; `families` lists every kernel family with its MFMA fraction and its
                achieved algorithmic HBM GB/s (the bandwidth-bound ones: embed_in, embed_out_step, the LN GEMMs).
  boundary      the same workload called through the drop-in surface (`diffusion.p_sample_loop(model, shape,
                model_kwargs=...)`, model = StyleDiffusion) in both noise modes, one pass each, next to the
                engine-level `value`.
  cpu_baseline  the CPU oracle (a port of the reference's fp32 path; oracle/) timed on this box's host cores on a
                bounded sample of the same workload (rank 0, N = 1 only).

--mode finetune.  BASELINE.json configs[3]: one step = one data-parallel fine-tune iteration
(`few_shot_style_finetune_losses`, DDIM-20 / skip 700, 64 text-to-motion clips per rank + the single-clip style branch,
backward, per-layer bucketed gradient all-reduce over RCCL overlapped with the backward pass, AdamW);
value = text-to-motion clips all ranks trained on / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0     # dense f16/bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBPS = 8000.0        # HBM3E spec peak (same table); ~6300 GB/s is what a streaming copy achieves
HBM_ACHIEVABLE_GBPS = 6300.0
D, FF, H = 512, 1024, 4

# kernel families (engine profiling names) -> rocprof symbol they are launches of
SYMBOL = {
    "outproj_ln_gemm": "k_gemm_dma<64,512,2,2,2,1,RowsDirect,DEpiResidLN,64>",
    "ffn2_ln_gemm": "k_gemm_dma<64,512,2,2,2,1,RowsDirect,DEpiResidLN,64>",
    "qkv_attention_fused": "k_qkv_attention2<13>",
    "ffn1_gelu_gemm": "k_gemm_dma<128,256,2,2,3,1,RowsDirect,DEpiBiasF16<true>,32>",
    "layer_tail_fused": "k_layer_tail",
    # (as rocprofv3 prints them at 263 features: 3 blocks of 16 output features per wave, ancestral step, no CFG; kpad 288 = 9 k-steps)
    "embed_out_step": "k_embed_out<3,1,1>",
    "embed_in": "k_embed_in<9> (writes the conditioning tokens too; + k_frames_f16 on the first step of a loop, later steps get their f16 frame rows from the previous step's epilogue)",
    "cond_token": "k_cond_token",
    "qkv_gemm": "k_gemm_dma<...,DEpiBiasF16<false>>",
    "attention": "k_attention<7>",
}


def family_work(family, rows, clips, T, F):
    """(algorithmic FLOPs, algorithmic HBM bytes) of ONE launch of a kernel family.  rows = clips through the
    transformer (2 x clips under CFG), S = T + 1 tokens each.  Bytes = compulsory traffic:
    every input read once, every output written once (f16 operands, the stream as an f16 hi + lo pair = 4 B/element,
    fp32 clip tensors); weights included; intermediates that stay on chip are not counted."""
    S = T + 1
    M = rows * S
    stream = M * D * 4.0                                 # hi + lo
    op = M * D * 2.0                                     # one f16 operand copy of a [M,512] tensor
    clip = clips * F * T * 4.0                           # one fp32 [B,F,1,T] tensor
    att_fl = 2.0 * 2 * rows * H * S * S * (D // H)
    table = {
        "qkv_gemm": (2.0 * M * 3 * D * D, op + 3 * op + 3 * D * D * 2),
        "attention": (att_fl, 3 * op + op),
        "qkv_attention_fused": (2.0 * M * 3 * D * D + att_fl, op + op + 3 * D * D * 2),
        "outproj_ln_gemm": (2.0 * M * D * D, op + stream + stream + D * D * 2),
        "ffn1_gelu_gemm": (2.0 * M * FF * D, op + 2 * op + FF * D * 2),
        "ffn2_ln_gemm": (2.0 * M * D * FF, 2 * op + stream + stream + FF * D * 2),
        "layer_tail_fused": (2.0 * M * D * D + 4.0 * M * FF * D, op + stream + stream + (D * D + 2 * FF * D) * 2),
        "embed_in": (2.0 * clips * T * D * F, clip + rows * T * D * 4.0 + D * F * 2),
        "embed_out_step": (2.0 * rows * T * F * D, rows * T * D * 4.0 + 3 * clip + clip + F * D * 2),      # the last stream is read as hi + lo
        "cond_token": (0.0, rows * D * 4.0 * 2),
    }
    return table[family]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed passes (sample: p_sample_loop passes; finetune: iterations)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=("sample", "finetune"), default="sample")
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--denoise-steps", type=int, default=1000)
    ap.add_argument("--cfg", action="store_true", help="configs[2]: classifier-free guidance (doubled batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boundary", action="store_true", help="skip the via-boundary legs (one extra pass per noise mode)")
    ap.add_argument("--via-boundary", choices=("philox", "torch"), default=None,
                    help="time the drop-in call diffusion.p_sample_loop(model, shape, model_kwargs=...) AS the headline value")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--shared-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (use with --backend gloo)")
    ap.add_argument("--cpu-sample-steps", type=int, default=24)
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` run plainly: start the N ranks with torch.distributed.run as a child process.  Nothing in
    this process has touched the GPU (torch is not even imported yet), so there is no exec-after-GPU-init hazard."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def init_dist(args):
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    return world, rank, dev, dist


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.mode == "finetune":
        return finetune_main(args)
    return sample_main(args)


# ---------------------------------------------------------------------------------------------- sampling
def build_boundary(seed, F, w, dev):
    """The drop-in objects of the reference's scripts: StyleDiffusion + InpaintingGaussianDiffusion from the factories
    (utils/model_util.py), loaded with the bench's seeded weights."""
    import types
    import torch
    from mst_amd.diffusion.inpainting_gaussian_diffusion import InpaintingGaussianDiffusion
    from mst_amd.model.mdm_forstyledataset import StyleDiffusion
    from mst_amd.utils import model_util
    a = types.SimpleNamespace(dataset="humanml", latent_dim=512, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                              emb_trans_dec=False, diffusion_steps=1000, noise_schedule="cosine", sigma_small=True,
                              lambda_vel=0.0, lambda_rcxyz=0.0, lambda_fc=0.0)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model = StyleDiffusion(**model_util.get_transfer_args(a))
        diffusion = model_util.create_gaussian_diffusion(a, InpaintingGaussianDiffusion, "")
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert not unexpected
    return model.to(dev).eval(), diffusion


def sample_main(args):
    import numpy as np
    import torch
    import mst_amd  # noqa: F401
    from mst_amd import sharding, synthetic as syn
    from mst_amd.engine import DenoiserEngine, Schedule, SAMPLER_DDPM
    from mst_amd.diffusion.gaussian_diffusion import schedule_tables

    world, rank, dev, dist = init_dist(args)
    F, T, B, NS = 263, 196, args.batch, args.denoise_steps
    rows = 2 * B if args.cfg else B
    seed = 20261003
    w = syn.denoiser_state(seed, F)
    pe = syn.positional_table(5000, 512)
    tab, tmap = schedule_tables("cosine", 1000, "" if NS == 1000 else str(NS))
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # per-rank inputs (rank offsets the stream so ranks denoise different clips)
    txt = to(syn.normal(seed + rank, "bench/txt", (B, 512)))
    x_T = to(syn.normal(seed + rank, "bench/xT", (B, F, 1, T)))
    motion = to(syn.normal(seed + rank, "bench/motion", (B, F, 1, T)))
    mask = to(syn.root_horizontal_mask(B, F, T))
    scale = to(np.full((B,), 2.5, np.float32)) if args.cfg else None

    model = diffusion = None
    if args.via_boundary or not args.no_boundary:
        model, diffusion = build_boundary(seed, F, w, dev)
        if args.cfg:
            from mst_amd.model.cfg_sampler import ClassifierFreeSampleModel
            model = ClassifierFreeSampleModel(model)

    def boundary_pass(noise_source):
        """sample/demo_style_transfer.py:240-256's call, batch B: everything the script passes, nothing engine-specific."""
        diffusion.noise_source = noise_source
        y = {"text_embed": txt, "mask": torch.ones(B, 1, 1, T, device=dev), "inpainting_mask": mask, "inpainted_motion": motion}
        if args.cfg:
            y["scale"] = scale
        with torch.no_grad():
            return diffusion.p_sample_loop(model, (B, F, 1, T), noise=x_T, clip_denoised=False, model_kwargs={"y": y},
                                           skip_timesteps=0, init_image=None, progress=False, dump_steps=None,
                                           const_noise=False)

    if args.via_boundary:
        eng = None
        one_pass = lambda k: boundary_pass(args.via_boundary)
    else:
        eng = DenoiserEngine(F, T, rows, device=dev)
        eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(pe))
        sch = Schedule(tab, tmap, dev)
        eng.set_text(txt, cfg=args.cfg)

        def one_pass(k):
            x = x_T.clone()
            eng.sample_loop(sch, x, NS - 1, 0, SAMPLER_DDPM, cfg=args.cfg, scale=scale, mask=mask, motion=motion,
                            mask_noise=True, seed=sharding.rank_seed(seed, rank, k))
            return x

    barrier = lambda: sharding.barrier(dev)
    for k in range(args.warmup):
        one_pass(k)
    if eng is not None:
        eng.profile(True, 50)
    barrier()
    t0 = time.perf_counter()
    last = None
    for k in range(args.steps):
        last = one_pass(args.warmup + k)
    barrier()
    dt = time.perf_counter() - t0
    prof, ev_us = {}, 0.0
    if eng is not None:
        prof = eng.profile_read()
        ev_us = max(0.0, eng.profile_event_overhead_us())
        eng.profile(False)
    assert torch.isfinite(last).all()
    assert torch.equal(last[:, :3], motion[:, :3]), "inpainted rows must equal the content clip exactly"
    group = sharding.group_report(dt, B * args.steps, dev if args.backend == "nccl" else "cpu")      # before the max: every rank's OWN time
    dt = sharding.max_over_ranks(dt, dev if args.backend == "nccl" else "cpu")

    boundary = None
    if rank == 0 and world == 1 and not args.via_boundary and not args.no_boundary:
        boundary = {"call": "diffusion.p_sample_loop(model, (B,263,1,196), noise=x_T, clip_denoised=False, model_kwargs={'y': ...})"}
        with torch.no_grad():                                # one model call first: the module builds its engine and uploads its weights there (one-time set-up,
            model(x_T, torch.zeros(B, dtype=torch.long, device=dev),      # not part of a loop's rate)
                  y={"text_embed": txt, "mask": torch.ones(B, 1, 1, T, device=dev), **({"scale": scale} if args.cfg else {})})
        for mode in ("philox", "torch"):
            boundary_pass(mode) if NS <= 100 else None       # short loops: warm the allocator; 1000-step passes are their own warm-up
            torch.cuda.synchronize(dev)
            tb = time.perf_counter()
            out = boundary_pass(mode)
            torch.cuda.synchronize(dev)
            boundary[f"{mode}_noise_clips_per_s"] = round(B / (time.perf_counter() - tb), 3)
            assert torch.equal(out[:, :3], motion[:, :3])
        boundary["note"] = ("philox = in-kernel counter-based noise (what `value` uses); torch = th.randn_like per step in the "
                            "reference's draw order, stacked in bounded chunks and handed to the fused loop")

    if rank == 0:
        clips = world * B * args.steps
        value = clips / dt
        flops_per_clip = 7.353e9 * NS * (2 if args.cfg else 1)
        line = {
            "metric": f"denoised motion clips/sec ({NS}-step DDPM, Bx263x196)",
            "value": round(value, 4), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 MFMA operands, fp32 accumulate/stream", "data": "synthetic",
            "config": {"workload": workload_label(B, args.cfg) +
                       f": batch {B}/GPU x (263,1,196), {NS}-step DDPM p_sample_loop, 8-layer/512-dim denoiser, "
                       "root_horizontal inpainting" + (", classifier-free guidance scale 2.5 (doubled batch)" if args.cfg else "") +
                       (f", called through diffusion.p_sample_loop ({args.via_boundary} noise)" if args.via_boundary else ""),
                       "global_batch": world * B, "denoise_steps": NS, "parallelism": f"clip-sharded x{world}, no collective"},
        }
        line["distributed"] = group
        if eng is not None:
            line["roofline"] = roofline(prof, rows, B, T, F, value, flops_per_clip, eng.loop_slices(B, args.cfg), ev_us)
        else:
            line["roofline"] = {"bound": "mfma", "achieved": round(value * flops_per_clip * 1e-12, 2), "peak": MFMA_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": round(value * flops_per_clip * 1e-12 / MFMA_PEAK_TFLOPS, 4),
                                "traffic": None, "kernel": "whole path (no per-kernel events through the boundary)"}
        if boundary is not None:
            line["boundary"] = boundary
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, pe, tab, tmap, B, F, T, NS, args.cpu_sample_steps, seed)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def workload_label(B, cfg):
    """Which BASELINE.json config a (batch per GPU, CFG) pair is; anything else is named as what it is, not as a config."""
    if cfg:
        return "configs[2]" if B == 64 else f"configs[2]'s path at batch {B} (not a BASELINE size)"
    if B == 64:
        return "configs[1]"
    if B == 128:
        return "configs[4]'s per-GPU share (1024 clips over 8 GPUs = 128 per GPU; the trainable stack as the denoiser)"
    return f"configs[1]'s path at batch {B} (not a BASELINE size)"


def roofline(prof, rows, clips, T, F, value, flops_per_clip, slices, ev_us=0.0):
    """prof: family -> (total ms, launches) of the event-timed launches (instrumented steps run as ONE full-batch slice,
    so per-launch work is the full batch).  The dominant kernel is chosen by total time per rocprof SYMBOL.
    ev_us: what an EMPTY event pair reports (calibrated by the engine); it is subtracted once per timed launch, which brings
    the event-bracketed durations onto rocprofv3's kernel durations (profiles/r02_kernel_stats_*: agreement within a few %)."""
    prof = {k: (max(ms - n * ev_us * 1e-3, 1e-9), n) for k, (ms, n) in prof.items()}
    fam = {}
    for k, (ms, n) in prof.items():
        if n == 0:
            continue
        fl, by = family_work(k, rows, clips, T, F)
        us = 1e3 * ms / n
        fam[k] = {"symbol": SYMBOL.get(k, k), "launches_timed": n, "avg_launch_us": round(us, 2),
                  "gflop_per_launch": round(fl * 1e-9, 2), "tflops": round(fl / us * 1e-6, 1),
                  "mfma_frac": round(fl / us * 1e-6 / MFMA_PEAK_TFLOPS, 4),
                  "algorithmic_mb_per_launch": round(by * 1e-6, 1), "hbm_gbps": round(by / us * 1e-3, 0),
                  "hbm_frac_of_peak": round(by / us * 1e-3 / HBM_PEAK_GBPS, 3),
                  "share_of_timed_device_time": 0.0}
    tot_ms = sum(prof[k][0] for k in fam) or 1.0
    by_symbol = {}
    for k in fam:
        fam[k]["share_of_timed_device_time"] = round(prof[k][0] / tot_ms, 3)
        by_symbol.setdefault(fam[k]["symbol"], []).append(k)
    dom_sym = max(by_symbol, key=lambda s: sum(prof[k][0] for k in by_symbol[s]))
    members = by_symbol[dom_sym]
    d_ms = sum(prof[k][0] for k in members)
    d_n = sum(prof[k][1] for k in members)
    d_fl = sum(family_work(k, rows, clips, T, F)[0] * prof[k][1] for k in members)
    d_by = sum(family_work(k, rows, clips, T, F)[1] * prof[k][1] for k in members)
    achieved = d_fl / (d_ms * 1e-3) * 1e-12
    traffic, traffic_src = pmc_traffic(members, rows)
    return {"bound": "mfma", "kernel": dom_sym, "families_of_that_symbol": members,
            "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
            "avg_launch_us": round(1e3 * d_ms / d_n, 2), "launches_timed": d_n,
            "share_of_timed_device_time": round(d_ms / tot_ms, 3),
            "algorithmic_gflop_per_launch": round(d_fl / d_n * 1e-9, 2),
            "hbm_gbps": round(d_by / (d_ms * 1e-3) * 1e-9, 0),
            "traffic": traffic, "traffic_source": traffic_src,
            "clips_per_timed_launch": rows, "concurrent_slices_elsewhere": slices,
            "whole_path_tflops": round(value * flops_per_clip * 1e-12, 2),
            "whole_path_frac": round(value * flops_per_clip * 1e-12 / MFMA_PEAK_TFLOPS, 4),
            "hbm_peak_gbps": HBM_PEAK_GBPS, "hbm_achievable_gbps": HBM_ACHIEVABLE_GBPS,
            "event_pair_overhead_us_subtracted": round(ev_us, 2),
            "families": fam}


def pmc_traffic(families, rows):
    """HBM bytes per launch of the dominant kernel from the PMC passes (tools/pmc_traffic.sh: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE runs with full-batch launches, units and gfx950 correction per the microarch guide), committed under
    profiles/.  Counters cannot be read from inside this process, so the figure is null unless that file was collected for
    the same per-launch work (64 clips, no CFG) and the same kernel."""
    prof_dir = os.path.join(ROOT, "profiles")
    keys = {"outproj_ln_gemm": "ln_gemm", "ffn2_ln_gemm": "ln_gemm"}
    if rows != 64:
        return None, None
    # Only a collection made with THIS build of the kernels counts: tools/pmc_traffic.py records the library's source hash, and a
    # file from another build (any csrc/ change since) reads as "no traffic figure", never as a stale number.
    from mst_amd import _native
    have = _native.built_hash()
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json"):
        try:
            doc = json.load(open(os.path.join(prof_dir, name)))
            kernels = doc["kernels"]
        except (OSError, KeyError, ValueError):
            continue
        if not have or doc.get("source_hash") != have:
            return None, f"profiles/{name} was collected with library build {doc.get('source_hash')}, this run is {have}: not quoted"
        for f in families:
            k = kernels.get(keys.get(f, f))
            if k:
                return k["hbm_bytes"], f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 64-clip launches, same library build {have})"
    return None, None


def cpu_baseline(w, pe, tab, tmap, B, F, T, NS, sample_steps, seed):
    """The oracle (CPU port of the reference's fp32 path) on the host cores, on a bounded sample:
    `sample_steps` denoise steps of the same batch-B loop, extrapolated linearly to NS steps (steps
    cost the same).  Reported next to the GPU number, never a target."""
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
            "s_per_denoise_step": round(per_step, 4),
            # one bounded sample on whatever the host is doing: rounds 3-5 measured 0.099 .. 0.129 clips/s for this same sample on the
            # pool's boxes (profiles/r0*_bench_*.json) -- quote the baseline as that range, the GPU / CPU ratio as ~800 .. 1 100
            "observed_range_clips_per_s": [0.099, 0.129]}


# ---------------------------------------------------------------------------------------------- fine-tuning
def finetune_flops(B):
    """Algorithmic FLOPs of ONE fine-tune iteration on one rank: every pass of a clip through the trainable
    denoiser is 7.353 GFLOP forward and twice that backward (dgrad + wgrad); the objective makes B + 6 such passes (the B-clip
    text-to-motion call and the 6 chained single-clip steps); the frozen motion encoder sees the B clips once, forward (7.3 GFLOP)
    and backward to its input only (dgrad, as much again)."""
    return 3 * 7.353e9 * (B + 6) + 2 * 7.3e9 * B


def wgrad_flops(B):
    """Algorithmic FLOPs of all k_wgrad_tr launches of ONE iteration: dW = dY^T X over the token rows of every trainable pass (the B-clip
    call and the six chained single-clip steps, differentiated in one pass), four weight matrices per layer, eight layers."""
    rows = (B + 6) * 197
    return 8 * 2.0 * rows * (1536 * 512 + 512 * 512 + 2 * 1024 * 512)


def finetune_roofline(B, world, s_per_iter, wgrad=None):
    """Whole-iteration MFMA roofline + the training path's dominant kernel (k_wgrad_tr by rocprofv3 share, profiles/r0*_finetune_kernel_
    stats_streams1.csv), timed IN THIS RUN with HIP events on the stream it is launched on (`wgrad`: total ms, launches, event-pair
    overhead of a few extra iterations behind the timed region); the committed rocprofv3 summary of an earlier run stays beside it."""
    fl = finetune_flops(B)
    achieved = world * fl / s_per_iter * 1e-12
    out = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS * world, "unit": "TFLOP/s",
           "frac": round(achieved / (MFMA_PEAK_TFLOPS * world), 4), "traffic": None,
           "algorithmic_gflop_per_iteration_per_gpu": round(fl * 1e-9, 1),
           "kernel": None, "note": "whole iteration (objective + backward + AdamW) over the dense MFMA peak"}
    # The other roofline: the activation tape.  Algorithmic HBM bytes per layer in units of rows x 512 bytes (every input read once, every
    # output written once): forward 50 at batch size (QKV 8, attention 8, the fused tail 34: att 2 + the stream's hi / lo
    # 4 + x1 read back 4 in, z1 / x1 / pre / hid / z2 / the stream out 24) and 56 on the small path (out-proj + LayerNorm 14, FFN1 10, FFN2
    # + LayerNorm 16 instead of the tail), dgrad chain 76, weight / bias gradients 34 (frozen stacks: none).  One iteration = the B-clip
    # call (all three), the frozen motion encoder on B clips (forward + dgrad) and the six chained single-clip calls (all three).
    S, unit = 197, 512
    tape = 8 * unit * (B * S * (50 + 76 + 34) + B * S * (50 + 76) + 6 * S * (56 + 76 + 34))
    out["tape_traffic"] = {"algorithmic_gb_per_iteration_per_gpu": round(tape * 1e-9, 2),
                           "achieved_tb_per_s": round(world * tape / s_per_iter * 1e-12, 3), "hbm_peak_tb_per_s": 8.0 * world,
                           "frac": round(tape / s_per_iter * 1e-12 / 8.0, 4),
                           "note": "the training launches sit on NEITHER roofline: each moves 52-150 MB of tape at 1.5-3.3 TB/s (docs/LAB_NOTES.md R5.7, "
                                   "R6.2); what bounds the iteration is the dependent chain on the caller's stream (R6.3)"}
    if wgrad and wgrad.get("launches"):
        n, iters = wgrad["launches"], wgrad["iterations"]
        avg_us = 1e3 * wgrad["total_ms"] / n - wgrad["event_pair_overhead_us"]
        tflops = wgrad_flops(B) * iters / (avg_us * 1e-6 * n) * 1e-12
        out["kernel"] = "k_wgrad_tr"
        out["dominant_kernel"] = {
            "kernel": "k_wgrad_tr (weight gradients dW = dY^T X, both operands read transposed from row-major activations)",
            "measured": f"HIP events on the launch stream around every launch of {iters} extra iterations behind the timed region",
            "launches_timed": n, "launches_per_iteration": n // iters, "avg_launch_us": round(avg_us, 2),
            "event_pair_overhead_us_subtracted": round(wgrad["event_pair_overhead_us"], 2),
            "algorithmic_gflop_per_iteration": round(wgrad_flops(B) * 1e-9, 1), "achieved_tflops": round(tflops, 1),
            "frac_of_mfma_peak": round(tflops / MFMA_PEAK_TFLOPS, 4)}
        # HBM bytes of the batch-size launches of that kernel by PMC (tools/r6_train_pmc.sh: FETCH_SIZE x 2 + WRITE_SIZE, separate passes
        # over one stack forward + backward at 64 clips on one stream) -- quoted only from a file measured on THIS library build
        try:
            from mst_amd import _native
            pm = json.load(open(os.path.join(ROOT, "profiles", "r06_train_pmc_traffic.json")))
            if pm.get("source_hash") == _native.built_hash() and B == pm.get("clips"):
                k = pm["kernels"]["k_wgrad_tr"]
                out["traffic"] = int(k["hbm_bytes"])
                out["traffic_source"] = ("profiles/r06_train_pmc_traffic.json: k_wgrad_tr at 64 clips, mean of the four weight-gradient shapes of a layer "
                                         f"({k['launches']} launches), same library build {pm['source_hash']}; every training kernel is in that file")
        except (OSError, KeyError, ValueError):
            pass
    # A committed rocprofv3 summary of the same command from an earlier run (possibly another build), with the file's hash: the share of
    # device time that names the dominant kernel comes from there.
    for name in ("r06_finetune_kernel_stats_streams1.csv", "r05_finetune_kernel_stats_streams1.csv", "r04_finetune_kernel_stats_streams1.csv"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            import csv
            import hashlib
            raw = open(path, "rb").read()
            rows = list(csv.DictReader(raw.decode().splitlines()))
            rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
            tot = sum(float(r["TotalDurationNs"]) for r in rows)
            top = rows[0]
            out["reference_profile"] = {
                "file": f"profiles/{name}", "sha256_16": hashlib.sha256(raw).hexdigest()[:16],
                "what": "rocprofv3 --kernel-trace --stats of tools/finetune_bench.py, an EARLIER run (not this one)",
                "kernel": top["Name"][:120], "kernel_share_of_device_time": round(float(top["TotalDurationNs"]) / tot, 3),
                "kernel_avg_launch_us": round(float(top["AverageNs"]) * 1e-3, 2)}
            break
        except (OSError, KeyError, ValueError, IndexError):
            continue
    return out


def finetune_main(args):
    """BASELINE.json configs[3]: data-parallel few-shot fine-tune iterations (train/finetune_style_diffusion.py's objective,
    diffusion/gaussian_diffusion.py:1317-1399), 64 text-to-motion clips per rank, gradients of the 96 trainable tensors
    mean-reduced over ranks in 8 per-layer buckets (finetune_dp.LayerBucketReducer) while the backward pass is still running."""
    import types
    import numpy as np
    import torch
    import mst_amd  # noqa: F401
    from mst_amd import sharding, synthetic as syn
    from mst_amd.diffusion.resample import host_to_device_async
    from mst_amd.finetune_dp import LayerBucketReducer
    from mst_amd.model.mdm_forstyledataset import StyleDiffusion
    from mst_amd.optim import FusedAdamW
    from mst_amd.utils import model_util

    world, rank, dev, dist = init_dist(args)
    B, F, T, seed = args.batch, 263, 196, 20261003
    a = types.SimpleNamespace(dataset="humanml", latent_dim=512, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                              emb_trans_dec=False, diffusion_steps=1000, noise_schedule="cosine", sigma_small=True,
                              lambda_vel=0.0, lambda_rcxyz=0.0, lambda_fc=0.0)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model, d_ddim, _ = model_util.creat_serval_diffusion(a, StyleDiffusion, "ddim20")
    sd = {k: torch.from_numpy(np.ascontiguousarray(syn.tensor_for(seed, k, tuple(v.shape)))) for k, v in model.state_dict().items()
          if not k.endswith(".pe") and "clip_model" not in k}
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).train()
    to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    t2m = to(syn.normal(seed + rank, "ft/t2m", (B, F, 1, T)))                   # the rank's shard of the global batch
    content = to(syn.normal(seed, "ft/content", (1, F, 1, T)))                  # the style example is replicated
    style = to(syn.normal(seed, "ft/style", (1, F, 1, T)))
    emb = to(syn.normal(seed, "ft/text", (1, 512)))                             # post-CLIP embedding (CLIP is outside the engine)
    y1 = {"y": {"text": ["a"], "text_embed": emb, "mask": torch.ones(1, 1, 1, T, device=dev),
                "inpainting_mask": to(syn.root_horizontal_mask(1, F, T)), "inpainted_motion": content}}
    yB = {"y": {"text": ["a"] * B, "text_embed": emb.expand(B, -1).contiguous(), "mask": torch.ones(B, 1, 1, T, device=dev),
                "inpainting_mask": to(syn.root_horizontal_mask(B, F, T)), "inpainted_motion": t2m}}
    gen = torch.Generator(device="cpu").manual_seed(seed + rank)
    opt = FusedAdamW(model.parameters_wo_enc(), lr=1e-5, weight_decay=0.0)
    red = LayerBucketReducer(model)

    # MST_FT_OVERLAP_BACKWARD=1: few_shot_style_finetune_losses(overlap_backward=True) -- measured in round 6, no gain (LAB_NOTES R6.10); off
    OVERLAP = os.environ.get("MST_FT_OVERLAP_BACKWARD", "0") == "1"

    def iteration(reduce=True):
        # range((1000 - 700) / 1000 * 20), training_loop.py:247; onto the device as UniformSampler.sample does it (pinned staging, no
        # blocking copy: the host is not tied to the GPU once per iteration -- diffusion/resample.py host_to_device_async)
        tt = host_to_device_async(torch.randint(0, 6, (B,), generator=gen), dev)
        red.zero_grad()
        red.enabled = reduce
        terms = d_ddim.few_shot_style_finetune_losses(model, t2m, tt, content, style, skip_steps=700, model_kwargs=y1,
                                                      model_t2m_kwargs=yB, semantic_guidance=1, use_ddim=1, Ls=10,
                                                      overlap_backward=OVERLAP)
        terms["loss"].backward()
        red.finish()
        opt.step()
        return terms["loss"]

    def timed(n, reduce=True):
        sharding.barrier(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            loss = iteration(reduce)
        timed.host_s = time.perf_counter() - t0              # the host has enqueued everything; how far ahead of the GPU it is says who bounds the loop
        sharding.barrier(dev)
        return time.perf_counter() - t0, float(loss)

    for _ in range(max(1, args.warmup)):
        iteration()
    dt, loss = timed(args.steps)
    host_s = timed.host_s
    group = sharding.group_report(dt, B * args.steps, dev if args.backend == "nccl" else "cpu")
    dt = sharding.max_over_ranks(dt, dev if args.backend == "nccl" else "cpu")
    comm = None
    if world > 1:
        # how much of the exchange is hidden: the same iterations with the collective skipped (buckets still filled), and the
        # eight bucket all-reduces alone, back to back
        dt_nocomm, _ = timed(args.steps, reduce=False)
        dt_nocomm = sharding.max_over_ranks(dt_nocomm, dev if args.backend == "nccl" else "cpu")
        sharding.barrier(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            red.allreduce_only()
        sharding.barrier(dev)
        dt_ar = sharding.max_over_ranks(time.perf_counter() - t0, dev if args.backend == "nccl" else "cpu")
        exposed = max(0.0, dt - dt_nocomm)
        comm = {"bytes_per_iteration": int(sum(red.bucket_bytes())), "buckets": len(red.bucket_bytes()),
                "allreduce_alone_ms": round(1e3 * dt_ar / args.steps, 3),
                "iteration_without_collective_ms": round(1e3 * dt_nocomm / args.steps, 3),
                "exposed_ms": round(1e3 * exposed / args.steps, 3),
                "fraction_hidden": round(1.0 - min(1.0, exposed / dt_ar), 3) if dt_ar > 0 else None,
                "launched_during_backward": red.launched_in.count("backward"), "launched_after_backward": red.launched_in.count("flush")}
    wgrad = None
    if rank == 0 and dev.type == "cuda":
        # the dominant kernel, timed in this run: a few MORE iterations with every k_wgrad_tr launch bracketed by HIP events
        try:
            engines = [ent["eng"] for ent in model.__dict__.get("_mst_engines", {}).values()]      # the module's engine + the chain's own instance
            for eng in engines:
                eng.profile(True, 1)
            for _ in range(3):
                iteration(reduce=False)
            torch.cuda.synchronize(dev)
            tot, n, ov = 0.0, 0, 0.0
            for eng in engines:
                ms, k = eng.profile_read().get("wgrad_tr", (0.0, 0))
                tot, n, ov = tot + ms, n + k, max(ov, eng.profile_event_overhead_us())
                eng.profile(False)
            if n:
                wgrad = {"total_ms": tot, "launches": n, "iterations": 3, "event_pair_overhead_us": ov}
        except Exception as exc:                                  # never lose the bench line to the instrumentation
            wgrad = {"error": repr(exc)[:200]}
    if rank == 0:
        value = world * B * args.steps / dt
        line = {"metric": "fine-tune text-to-motion clips/sec (few_shot_style_finetune_losses, ddim20 / skip 700, Bx263x196)",
                "value": round(value, 3), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
                "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f16 MFMA operands, fp32 accumulate / gradients / optimizer", "data": "synthetic",
                "config": {"workload": f"configs[3]: data-parallel fine-tune, {B} clips/GPU x (263,1,196), one 64-clip objective call + "
                           "6 chained single-clip steps + frozen motion encoder + backward + AdamW per iteration",
                           "global_batch": world * B, "parallelism": f"dp{world}, 8 per-layer gradient buckets, all-reduce overlapped with backward"},
                "iterations_per_s": round(args.steps / dt, 3), "final_loss": round(loss, 5), "allreduce": comm, "distributed": group,
                "host_enqueue_ms_per_step": round(1e3 * host_s / args.steps, 3)}     # rank 0's Python + launch time per iteration, no synchronisation inside
        line["roofline"] = finetune_roofline(B, world, dt / args.steps, wgrad)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
