import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mst_amd
from mst_amd import synthetic as syn
from mst_amd.engine import DenoiserEngine
dev = torch.device("cuda:0")
F, T, B = 263, 196, 2
S = T + 1
np.set_printoptions(linewidth=250)
def run(ntb, zero):
    os.environ["MST_TAIL_NTB"] = ntb; os.environ["MST_SMALL_M"] = "0"
    eng = DenoiserEngine(F, T, B, device=dev)
    w = dict(syn.denoiser_state(1, F))
    for k in list(w):
        if k.startswith("seqTransEncoder.layers.0.") and any(z in k for z in zero):
            w[k] = np.zeros_like(w[k])
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    eng.set_text(to(syn.normal(1, "t", (B, 512))))
    x = to(syn.normal(1, "x", (B, F, 1, T))); t = torch.full((B,), 500, dtype=torch.int64, device=dev)
    eng.debug_stop_after(0, 5)
    eng.forward(x, t)
    return eng.debug_buffer("hs", B * S, 512).float().cpu().numpy()
for lib in (None, "ab_libs/lib_schedb.so"):
    if lib: os.environ["MST_ENGINE_LIB"] = os.path.abspath(lib)
    import subprocess
    code = ("import os,sys,numpy as np;sys.path.insert(0,'.');exec(open('tools/experiments/r4_ntb_cmp.py').read().split('for lib in')[0]);"
            "d=np.abs(run('3',())-run('4',()));print('%s', np.round(d.max(1)[:192].reshape(-1,16).max(1),3))" % (lib or "in-tree"))
    print(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout.strip().splitlines()[-1])
