import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, numpy as np
import test_gpu_train as T
from conftest import rel_l2
for tag, rows, p in [("xia", 3, 0.0), ("hml", 2, 0.1)]:
    eng, w = T.engine_for(tag)
    S, h, r = T.stream_input(tag, rows)
    seed = 99
    params = T.layer_params(w, True)
    href = h.clone().requires_grad_(True)
    masks = T.engine_masks(eng, seed, p, rows, S) if p > 0 else None
    ref = T.torch_stack(href, params, masks)
    (ref * r).sum().backward()
    out, tape = eng.train_forward(h, p, seed)
    grads = [torch.zeros_like(q) for q in params]
    d_in = eng.train_backward(tape, r, p, seed, grads)
    print(tag, p, "out", rel_l2(out.cpu().numpy(), ref.detach().cpu().numpy()), "d_in", rel_l2(d_in.cpu().numpy(), href.grad.cpu().numpy()))
    e = [rel_l2(g.cpu().numpy(), q.grad.cpu().numpy()) for g, q in zip(grads, params)]
    for i in range(12):
        print("  %-28s max over layers %.2e  min %.2e" % (T.LAYER_TENSORS[i], max(e[i::12]), min(e[i::12])))
