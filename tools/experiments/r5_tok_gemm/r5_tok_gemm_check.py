"""k_tok_gemm (csrc/mst_tokgemm.h) against the slab-ring GEMMs it replaces in the training path at batch size: two engines in one
process (MST_TOK_GEMM=0 / 1 at creation), the same weights and inputs; forward output, dL/dh and all 96 parameter gradients must be
BIT-identical (same MFMA, same k order, same epilogue code), then both are timed (forward, backward; events on the launch stream).
    python tools/experiments/r5_tok_gemm/r5_tok_gemm_check.py [clips]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import mst_amd  # noqa: F401,E402
from mst_amd import synthetic as syn  # noqa: E402
from mst_amd.engine import DenoiserEngine, LAYER_TENSORS  # noqa: E402
from test_gpu_train import layer_params  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
FE, T, D = 263, 196, 512
S = T + 1
dev = torch.device("cuda:0")
w = syn.denoiser_state(1234, FE, layer_prefix="seqTransEncoder.layers.")


def make(tok, **env):
    os.environ["MST_TOK_GEMM"] = str(tok)
    for k in ("MST_TOK_BT", "MST_TOK_MASK"):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    eng = DenoiserEngine(FE, T, B, device=dev)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, layer_prefix="seqTransEncoder.layers.",
                        pe=torch.from_numpy(syn.positional_table(5000, 512)))
    return eng


g = torch.Generator(device="cpu").manual_seed(7)
h = torch.randn(B, S, D, generator=g).to(dev)
r = torch.randn(B, S, D, generator=g).to(dev)


def run(eng, p, seed):
    out, tape = eng.train_forward(h, p, seed)
    grads = [torch.zeros_like(q) for q in layer_params(w, False)]
    d_in = eng.train_backward(tape, r, p, seed, grads)
    return out, d_in, grads


def timed(eng, p, seed, n=10):
    grads = [torch.zeros_like(q) for q in layer_params(w, False)]
    for _ in range(2):
        out, tape = eng.train_forward(h, p, seed)
        eng.train_backward(tape, r, p, seed, grads)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(n):
        ev[0].record()
        out, tape = eng.train_forward(h, p, seed)
        ev[1].record()
        eng.train_backward(tape, r, p, seed, grads)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1])
        tb += ev[1].elapsed_time(ev[2])
    return tf / n, tb / n


ring = make(0)
variants = {"tok bt64": make(1), "tok bt32": make(1, MST_TOK_BT=32)}
bad = 0
names = ["out", "d_in"] + [f"L{i // 12}.{LAYER_TENSORS[i % 12]}" for i in range(96)]
for p in (0.0, 0.1):
    a = run(ring, p, 99)
    for vn, eng in variants.items():
        b = run(eng, p, 99)
        same = [torch.equal(a[0], b[0]), torch.equal(a[1], b[1])] + [torch.equal(x, y) for x, y in zip(a[2], b[2])]
        diff = [i for i, s in enumerate(same) if not s]
        print(f"{vn}, dropout {p}: {len(same) - len(diff)} of {len(same)} tensors bit-identical to the ring path", [names[i] for i in diff][:8], flush=True)
        if diff:
            i = diff[0]
            x = (a[0], a[1], *a[2])[i].float()
            y = (b[0], b[1], *b[2])[i].float()
            print("  first difference:", names[i], "rel", float((x - y).norm() / x.norm()), flush=True)
        bad += len(diff)
for name, eng in [("ring", ring)] + list(variants.items()):
    tf, tb = timed(eng, 0.1, 5)
    print(f"{name}: forward {tf:.3f} ms, backward {tb:.3f} ms ({B} clips, 8 layers)", flush=True)
sys.exit(1 if bad else 0)
