"""Forward + backward of the trainable encoder stack at the fine-tune batch (64 clips x 197 tokens, dropout 0.1):
the native path (mst_train_forward / mst_train_backward) beside torch autograd over nn.TransformerEncoder
(fp32 = what the reference runs, and bf16 autocast) on the same GPU.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import torch.nn as nn

import mst_amd  # noqa: F401
from mst_amd import synthetic as syn
from mst_amd.engine import DenoiserEngine, LAYER_TENSORS

B = int(os.environ.get("TB_BATCH", 64))
T, F, D, L = 196, 263, 512, 8
S = T + 1
P = 0.1
dev = torch.device("cuda:0")
ITERS = int(os.environ.get("TB_ITERS", 5))


def timed(fn, iters=ITERS, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


w = syn.denoiser_state(20261003, F, layer_prefix="seqTransEncoder.layers.")
eng = DenoiserEngine(F, T, B, device=dev)
eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, layer_prefix="seqTransEncoder.layers.",
                    pe=torch.from_numpy(syn.positional_table(5000, 512)))
h = torch.randn(B, S, D, device=dev)
r = torch.randn(B, S, D, device=dev)
grads = [torch.zeros(torch.from_numpy(w[f"seqTransEncoder.layers.{i}.{k}"]).shape, device=dev) for i in range(L) for k in LAYER_TENSORS]
tape = eng.train_tape(B, S)
res = {"batch": B, "tokens": B * S, "dropout": P}
res["native_fwd_ms"] = timed(lambda: eng.train_forward(h, P, 1, tape))
res["native_bwd_ms"] = timed(lambda: eng.train_backward(tape, r, P, 1, None if os.environ.get("TB_FROZEN") else grads))   # TB_FROZEN=1: input gradient only (a frozen stack)
flop_fwd = 8 * 905.76e6 * B
res["native_fwd_tflops"] = flop_fwd / res["native_fwd_ms"] / 1e9
res["native_fwdbwd_tflops"] = 3 * flop_fwd / (res["native_fwd_ms"] + res["native_bwd_ms"]) / 1e9

if os.environ.get("TB_NATIVE_ONLY"):
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}))
    sys.exit(0)
layer = nn.TransformerEncoderLayer(d_model=D, nhead=4, dim_feedforward=1024, dropout=P, activation="gelu")
enc = nn.TransformerEncoder(layer, num_layers=L).to(dev).train()
sd = {k[len("seqTransEncoder."):]: torch.from_numpy(v) for k, v in w.items() if k.startswith("seqTransEncoder.")}
enc.load_state_dict(sd)
seq = h.permute(1, 0, 2).contiguous().requires_grad_(True)
rs = r.permute(1, 0, 2).contiguous()


def torch_step(autocast):
    enc.zero_grad(set_to_none=True)
    seq.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        out = enc(seq)
    (out.float() * rs).sum().backward()


res["torch_fp32_fwdbwd_ms"] = timed(lambda: torch_step(False), iters=3, warm=1)
res["torch_bf16_fwdbwd_ms"] = timed(lambda: torch_step(True), iters=3, warm=1)
res["native_fwdbwd_ms"] = res["native_fwd_ms"] + res["native_bwd_ms"]
res["speedup_vs_torch_fp32"] = res["torch_fp32_fwdbwd_ms"] / res["native_fwdbwd_ms"]
res["speedup_vs_torch_bf16"] = res["torch_bf16_fwdbwd_ms"] / res["native_fwdbwd_ms"]
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}))
