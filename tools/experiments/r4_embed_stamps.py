"""Phase stamps of k_embed_out (diagnostic library build with -DEMB_PROBE, selected through MST_ENGINE_LIB): the median workgroup of the
last launch of a 40-step loop.   EB=<clips> MST_ENGINE_LIB=$PWD/ab_libs/lib_embprobe.so python tools/experiments/r4_embed_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mst_amd  # noqa: E402,F401
from mst_amd import _native as N, synthetic as syn  # noqa: E402
from mst_amd.diffusion.gaussian_diffusion import schedule_tables  # noqa: E402
from mst_amd.engine import SAMPLER_DDPM, DenoiserEngine, Schedule  # noqa: E402

dev = torch.device("cuda:0")
F, T, B = 263, 196, int(os.environ.get("EB", "21"))
eng = DenoiserEngine(F, T, B, device=dev)
w = syn.denoiser_state(1, F)
eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
eng.set_text(to(syn.normal(1, "t", (B, 512))))
x = to(syn.normal(1, "x", (B, F, 1, T)))
motion = to(syn.normal(1, "m", (B, F, 1, T)))
mask = to(syn.root_horizontal_mask(B, F, T))
tab, tmap = schedule_tables("cosine", 1000, "")
sch = Schedule(tab, tmap, dev)
eng.sample_loop(sch, x.clone(), 39, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=1)
torch.cuda.synchronize()
buf = np.zeros((512, 8, 8), np.uint64)
lib = N.lib()
lib.mst_probe_read.argtypes = [C.c_void_p]
lib.mst_probe_read.restype = C.c_int
assert lib.mst_probe_read(buf.ctypes.data) == 0
tiles = min(512, (B * T + 63) // 64)
st = buf[:tiles].astype(np.float64) * 0.01      # us (100 MHz clock)
t0 = st[:, :, 0].min(axis=1, keepdims=True)
names = ["start", "rows issued (+ noise)", "rows landed, barrier", "GEMM done", "tile scattered", "update applied", "frame rows written"]
last = st.max(axis=1) - t0                      # last wave at each mark
first = st.min(axis=1) - t0
med, medf = np.median(last, axis=0), np.median(first, axis=0)
print(f"k_embed_out, {B} clips = {tiles} workgroups (median workgroup, us since its first wave started):")
for i in range(1, 7):
    print(f"  {names[i]:28s} last wave {med[i]:6.2f} (+{med[i] - med[i - 1]:5.2f})   first wave {medf[i]:6.2f}")
    if i == 3:
        print(f"  {'epilogue loads issued':28s} last wave {med[7]:6.2f} (+{med[7] - med[3]:5.2f})   first wave {medf[7]:6.2f}")
print("  workgroup starts spread over", round(float(t0.max() - t0.min()), 2), "us; launch =", round(float(st[:, :, 6].max() - t0.min()), 2), "us")
