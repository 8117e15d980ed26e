"""Where does the resident-group trunk differ from the two-kernel path?  att / hx buffers after a forward call, 1 and 2 layers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mst_amd
from mst_amd import synthetic as syn
from mst_amd.engine import DenoiserEngine
SEED, F, T, B = 20261003, 263, 196, 12
S = T + 1
dev = torch.device("cuda:0")
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for NL in (1, 2, 8):
    eng = DenoiserEngine(F, T, 64, num_layers=NL, device=dev)
    w = syn.denoiser_state(SEED, F)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
    x = cu(syn.normal(SEED, "dbg/x", (B, F, 1, T)))
    t = torch.full((B,), 300, device=dev, dtype=torch.long)
    eng.set_text(cu(syn.normal(SEED, "dbg/txt", (B, 512))))
    res = {}
    for on in (0, 1):
        eng.set_trunk_groups(bool(on))
        out = eng.forward(x, t).clone()
        torch.cuda.synchronize()
        res[on] = (out, eng.debug_buffer("att", B * S, 512).float().cpu().numpy().reshape(B, S, 4, 128),
                   eng.debug_buffer("hx", B * S, 512).float().cpu().numpy().reshape(B, S, 512))
    eng.set_trunk_groups(False)
    try:
        eng.trunk_check(); ok = "no give-up"
    except Exception as e:
        ok = str(e)
    print(f"== {NL} layer(s): {ok}; out max diff {float((res[0][0] - res[1][0]).abs().max()):.3e}")
    da = np.abs(res[0][1] - res[1][1]); dh = np.abs(res[0][2] - res[1][2])
    print("   att: max diff", da.max(), "| per head", da.max(axis=(0, 1, 3)), "| clips with a difference", np.nonzero(da.max(axis=(1, 2, 3)) > 0)[0].tolist())
    rows = np.nonzero(da.max(axis=(0, 2, 3)) > 0)[0]
    print("   att rows with a difference:", rows[:20].tolist(), "...", len(rows))
    print("   hx : max diff", dh.max(), "| clips", np.nonzero(dh.max(axis=(1, 2)) > 0)[0].tolist())
    rows = np.nonzero(dh.max(axis=(0, 2)) > 1e-6)[0]
    print("   hx rows with a difference > 1e-6:", rows[:30].tolist(), "...", len(rows), "| by tile:", [int((dh[:, a:b].max() > 1e-6)) for a, b in ((0, 64), (64, 112), (112, 160), (160, 197))])
    cols = np.nonzero(dh.max(axis=(0, 1)) > 1e-6)[0]
    print("   hx cols with a difference > 1e-6:", len(cols), cols[:16].tolist())
