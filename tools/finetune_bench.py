"""One fine-tune iteration of the reference's objective (few_shot_style_finetune_losses, gd.py:1317-1399:
DDIM-20 sub-schedule, skip 700 -> 6 chained in-graph denoising steps of the 64-clip text-to-motion batch + the
single-clip style branch + the frozen motion encoder) followed by backward, model.train() (dropout 0.1):
native training node vs the same module evaluated with torch ops (fp32, and bf16 autocast) on the same GPU."""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]       # tests/torch_reference.py: the fp32 torch-op evaluation compared against
import numpy as np
import torch

import mst_amd  # noqa: F401
import mst_amd.synthetic as syn
from mst_amd.model.mdm_forstyledataset import StyleDiffusion
from mst_amd.utils import model_util

dev = torch.device("cuda:0")
B = int(os.environ.get("FB_BATCH", 64))
F, T = 263, 196
args = types.SimpleNamespace(dataset="humanml", latent_dim=512, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                             emb_trans_dec=False, diffusion_steps=1000, noise_schedule="cosine", sigma_small=True,
                             lambda_vel=0.0, lambda_rcxyz=0.0, lambda_fc=0.0)
model, d_ddim, _ = model_util.creat_serval_diffusion(args, StyleDiffusion, "ddim20")
sd = {k: torch.from_numpy(np.ascontiguousarray(syn.tensor_for(7, k, tuple(v.shape)))) for k, v in model.state_dict().items()
      if not k.endswith(".pe") and "clip_model" not in k}
model.load_state_dict(sd, strict=False)
model.motion_enc.mdm_model.set_text_encoder(lambda texts: torch.stack([torch.from_numpy(syn.normal(7, "text/" + t, (512,))) for t in texts]))
model = model.to(dev).train()
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
t2m = to(syn.normal(7, "t2m", (B, F, 1, T)))
content = to(syn.normal(7, "content", (1, F, 1, T)))
style = to(syn.normal(7, "style", (1, F, 1, T)))
mask1 = to(syn.root_horizontal_mask(1, F, T))
maskB = to(syn.root_horizontal_mask(B, F, T))
emb1 = to(syn.normal(7, "text/a person walks", (1, 512)))          # post-CLIP embeddings (CLIP is outside the engine and absent here)
y1 = {"y": {"text": ["a person walks"], "text_embed": emb1, "mask": torch.ones(1, 1, 1, T, device=dev), "inpainting_mask": mask1, "inpainted_motion": content}}
yB = {"y": {"text": ["a person walks"] * B, "text_embed": emb1.expand(B, -1).contiguous(), "mask": torch.ones(B, 1, 1, T, device=dev), "inpainting_mask": maskB, "inpainted_motion": t2m}}
tt = torch.randint(0, 20, (B,), device=dev)
from mst_amd.optim import FusedAdamW
opts = {"native": FusedAdamW(model.parameters_wo_enc(), lr=1e-5), "torch": torch.optim.AdamW(model.parameters_wo_enc(), lr=1e-5)}


backend = "native"


def iteration(autocast=False):
    opt = opts[backend]
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        terms = d_ddim.few_shot_style_finetune_losses(model, t2m, tt, content, style, skip_steps=700, model_kwargs=y1,
                                                      model_t2m_kwargs=yB, semantic_guidance=1, use_ddim=1, Ls=10)
    terms["loss"].backward()
    opt.step()
    return float(terms["loss"])


def timed(fn, iters, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


res = {"batch": B, "shape": [F, 1, T], "objective": "few_shot_style_finetune_losses ddim20 skip700 + AdamW step"}
res["native_ms"] = timed(iteration, int(os.environ.get("FB_ITERS", 3)))
res["native_loss"] = iteration()
if not os.environ.get("FB_NATIVE_ONLY"):
    from torch_reference import use_torch_ops
    use_torch_ops(model)
    backend = "torch"
    res["torch_fp32_ms"] = timed(iteration, 2)
    res["torch_fp32_loss"] = iteration()
    res["torch_bf16_ms"] = timed(lambda: iteration(True), 2)
    res["speedup_vs_torch_fp32"] = res["torch_fp32_ms"] / res["native_ms"]
    res["speedup_vs_torch_bf16"] = res["torch_bf16_ms"] / res["native_ms"]
res["clips_per_s_native"] = B / res["native_ms"] * 1e3
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}))
