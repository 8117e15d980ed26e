"""Event-timed embed kernels: plain forward (output projection with a bias-and-store epilogue) against a sampling step (the diffusion
update in the epilogue), 64 clips -- how much of the embed-out launch is its epilogue."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mst_amd
from mst_amd import synthetic as syn
from mst_amd.engine import DenoiserEngine, Schedule, SAMPLER_DDPM
from mst_amd.diffusion.gaussian_diffusion import schedule_tables
dev = torch.device("cuda:0")
F, T, B = 263, 196, int(os.environ.get("EB", "64"))
eng = DenoiserEngine(F, T, B, device=dev)
w = syn.denoiser_state(1, F)
eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
eng.set_text(to(syn.normal(1, "t", (B, 512))))
x = to(syn.normal(1, "x", (B, F, 1, T))); t = torch.full((B,), 500, dtype=torch.int64, device=dev)
motion = to(syn.normal(1, "m", (B, F, 1, T))); mask = to(syn.root_horizontal_mask(B, F, T))
tab, tmap = schedule_tables("cosine", 1000, "")
sch = Schedule(tab, tmap, dev)
def show(tag):
    torch.cuda.synchronize()
    p = eng.profile_read(); ev = eng.profile_event_overhead_us()
    print(tag, {k: round(1e3 * ms / n - ev, 2) for k, (ms, n) in p.items() if n})
for _ in range(3): eng.forward(x, t)
eng.profile(True, 1)
for _ in range(20): eng.forward(x, t)
show("forward (MODE 0)      ")
eng.profile(True, 1)
eng.sample_loop(sch, x.clone(), 39, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=1)
show("loop, philox          ")
nz = torch.randn((40,) + tuple(x.shape), device=dev)
eng.profile(True, 1)
eng.sample_loop(sch, x.clone(), 39, 0, SAMPLER_DDPM, mask=mask, motion=motion, noise=nz)
show("loop, noise buffer    ")
eng.profile(True, 1)
eng.sample_loop(sch, x.clone(), 39, 0, SAMPLER_DDPM, seed=1)
show("loop, philox, no mask ")
