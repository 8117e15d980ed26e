"""Single-clip step (the demo's / configs[0]'s path): event-timed launches per kernel family + the loop's wall time per step."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mst_amd
from mst_amd import synthetic as syn
from mst_amd.engine import DenoiserEngine, Schedule, SAMPLER_DDPM
from mst_amd.diffusion.gaussian_diffusion import schedule_tables
dev = torch.device("cuda:0")
F, T, B = int(os.environ.get("BF", "181")), int(os.environ.get("BT", "76")), 1
eng = DenoiserEngine(F, T, 2, device=dev)
w = syn.denoiser_state(1, F)
eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
tab, tmap = schedule_tables("cosine", 1000, "100")
sch = Schedule(tab, tmap, dev)
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
eng.set_text(to(syn.normal(1, "t", (B, 512))))
x0 = to(syn.normal(1, "x", (B, F, 1, T))); motion = to(syn.normal(1, "m", (B, F, 1, T))); mask = to(syn.root_horizontal_mask(B, F, T))
for rep in range(3):
    x = x0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.sample_loop(sch, x, 99, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=rep)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"F={F} T={T}: {dt / 100 * 1e6:.1f} us per denoise step (wall)")
eng.profile(True, 1)
eng.sample_loop(sch, x0.clone(), 99, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=1)
torch.cuda.synchronize()
p = eng.profile_read(); ev = eng.profile_event_overhead_us()
tot = 0
for k, (ms, n) in p.items():
    if n:
        us = 1e3 * ms / n - ev
        print(f"  {k:22s} {n / 100:5.1f} scopes/step  {us:7.2f} us each  {us * n / 100:7.1f} us/step")
        tot += us * n / 100
print(f"  sum of event-timed scopes {tot:.1f} us/step")
