"""Is the sampling loop host-bound?  Time until mst_sample_loop RETURNS (all launches enqueued) vs until the GPU is done."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mst_amd
from mst_amd import synthetic as syn
from mst_amd.engine import DenoiserEngine, Schedule, SAMPLER_DDPM
from mst_amd.diffusion.gaussian_diffusion import schedule_tables
dev = torch.device("cuda:0")
F, T = 263, 196
import os
for B in [int(b) for b in os.environ.get("HB_BATCHES", "64,16,1").split(",")]:
    eng = DenoiserEngine(F, T, B, device=dev)
    w = syn.denoiser_state(1, F)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
    tab, tmap = schedule_tables("cosine", 1000, "")
    sch = Schedule(tab, tmap, dev)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    eng.set_text(to(syn.normal(1, "t", (B, 512))))
    x0 = to(syn.normal(1, "x", (B, F, 1, T))); motion = to(syn.normal(1, "m", (B, F, 1, T))); mask = to(syn.root_horizontal_mask(B, F, T))
    n = 300
    for rep in range(2):
        x = x0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.sample_loop(sch, x, n - 1, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=rep)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B} slices={eng.loop_slices(B)}: enqueue {1e6*(t1-t0)/n:.0f} us/step, total {1e6*(t2-t0)/n:.0f} us/step")
