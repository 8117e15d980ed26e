"""Round 6 debugging aid: the sum-of-halves gradient check of tests/test_gpu_train_fullsize.py in a loop (an intermittent 4e-5 deviation
was seen once behind the hid / x1 changes of the fused training tail): which tensors deviate, in which call."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import test_gpu_train_fullsize as T
from conftest import rel_l2
from mst_amd.engine import LAYER_TENSORS
eng, w = T.big_engine()
h, r = T.stream(T.B)
p, seed = float(os.environ.get("P", 0.0)), 3
ref = None
for it in range(int(os.environ.get("N", 12))):
    _, d_full, g_full = T.engine_grads(eng, w, h, r, p, seed)
    _, d_a, g_a = T.engine_grads(eng, w, h[:32].contiguous(), r[:32].contiguous(), p, seed)
    torch.cuda.synchronize()
    if ref is None:
        ref = ([g.clone() for g in g_full], [g.clone() for g in g_a], d_full.clone(), d_a.clone())
        continue
    bad = [f"full L{i // 12}.{LAYER_TENSORS[i % 12]} {rel_l2(a.cpu().numpy(), b.cpu().numpy()):.1e}" for i, (a, b) in enumerate(zip(g_full, ref[0])) if not torch.equal(a, b)]
    bad += [f"half L{i // 12}.{LAYER_TENSORS[i % 12]} {rel_l2(a.cpu().numpy(), b.cpu().numpy()):.1e}" for i, (a, b) in enumerate(zip(g_a, ref[1])) if not torch.equal(a, b)]
    if not torch.equal(d_full, ref[2]): bad.append("d_in full")
    if not torch.equal(d_a, ref[3]): bad.append("d_in half")
    print("iteration", it, "differs from the first:", bad[:12] if bad else "nothing")
