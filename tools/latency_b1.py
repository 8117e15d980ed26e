"""configs[0] on the GPU: single (181,1,76) clip, 100 respaced DDPM steps -- wall time per loop."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mst_amd
from mst_amd import synthetic as syn
from mst_amd.engine import DenoiserEngine, Schedule, SAMPLER_DDPM
from mst_amd.diffusion.gaussian_diffusion import schedule_tables
dev = torch.device("cuda:0")
for (F, T, B, resp, n) in ((181, 76, 1, "100", 100), (263, 196, 1, "", 1000), (181, 76, 1, "ddim20", 6)):
    eng = DenoiserEngine(F, T, max(B, 2), device=dev)
    w = syn.denoiser_state(1, F)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
    tab, tmap = schedule_tables("cosine", 1000, resp)
    sch = Schedule(tab, tmap, dev)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    eng.set_text(to(syn.normal(1, "t", (B, 512))))
    x0 = to(syn.normal(1, "x", (B, F, 1, T))); motion = to(syn.normal(1, "m", (B, F, 1, T))); mask = to(syn.root_horizontal_mask(B, F, T))
    for rep in range(3):
        x = x0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.sample_loop(sch, x, n - 1, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=rep)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"F={F} T={T} B={B} steps={n}: {dt*1e3:.2f} ms per loop, {dt/n*1e6:.1f} us per denoise step, {B/dt:.2f} clips/s")
