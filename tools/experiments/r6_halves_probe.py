import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import torch
import test_gpu_train_fullsize as T
from conftest import rel_l2
from mst_amd.engine import LAYER_TENSORS
eng, w = T.big_engine()
h, r = T.stream(T.B)
p, seed = 0.0, 0
_, d_full, g_full = T.engine_grads(eng, w, h, r, p, seed)
_, d_a, g_a = T.engine_grads(eng, w, h[:32].contiguous(), r[:32].contiguous(), p, seed)
_, d_b, g_b = T.engine_grads(eng, w, h[32:].contiguous(), r[32:].contiguous(), p, seed)
errs = [(rel_l2((a + b).cpu().numpy(), f.cpu().numpy()), f"L{i // 12}.{LAYER_TENSORS[i % 12]}") for i, (f, a, b) in enumerate(zip(g_full, g_a, g_b))]
errs.sort(reverse=True)
print(os.environ.get("MST_TRAIN_FUSE_TAIL", "1"), "worst five:", [(f"{e:.2e}", n) for e, n in errs[:5]], "median", f"{errs[len(errs)//2][0]:.2e}")
