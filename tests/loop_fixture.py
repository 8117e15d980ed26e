"""Inputs of the training-loop golden (tests/golden/make_golden_train.py) rebuilt from the seeds, and the recorded
draw sequence the reference consumed."""
import contextlib
import types

import numpy as np
import torch

import mst_amd  # noqa: F401
import mst_amd.synthetic as syn
from conftest import SEED

STEPS_PER_EPOCH, NUM_STEPS = 4, 11
F, T, B = 181, 76, 2
PROMPTS = ["a person walks proudly", "an old man jumps"]
ARGS = dict(dataset="stylexia_posrot", batch_size=B, lr=1e-4, log_interval=1, save_interval=1000, resume_checkpoint="",
            weight_decay=0.01, lr_anneal_steps=40, style_finetune=1, semantic_guidance=1, skip_steps=700, num_steps=NUM_STEPS,
            overwrite=True, use_ddim=1, diffusion_steps=1000, Ls=10.0)


def batches():
    data = []
    for i in range(STEPS_PER_EPOCH):
        motion = torch.from_numpy(syn.normal(SEED, f"loop/t2m/{i}", (B, F, 1, T)))
        fm = torch.ones(B, 1, 1, T)
        fm[1, ..., T - 5 - i:] = 0
        mask = torch.from_numpy(syn.root_horizontal_mask(B, F, T))
        data.append((motion, {"y": {"text": PROMPTS, "mask": fm, "inpainting_mask": mask, "inpainted_motion": motion}}))
    content = torch.from_numpy(syn.normal(SEED, "loop/content", (1, F, 1, T)))
    style = torch.from_numpy(syn.normal(SEED, "loop/style", (1, F, 1, T)))
    cond_style = {"y": {"text": PROMPTS[:1], "mask": torch.ones(1, 1, 1, T),
                        "inpainting_mask": torch.from_numpy(syn.root_horizontal_mask(1, F, T)), "inpainted_motion": style}}
    return data, ((content, cond_style),)


def diffusion_args():
    return types.SimpleNamespace(dataset="stylexia_posrot", latent_dim=512, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                                 emb_trans_dec=False, diffusion_steps=1000, noise_schedule="cosine", sigma_small=True,
                                 lambda_vel=0.0, lambda_rcxyz=0.0, lambda_fc=0.0)


def build_model(device):
    """StyleDiffusion with the golden's seeded weights, eval mode (dropout / cond mask off), + the ddim20 diffusion."""
    from mst_amd.model.mdm_forstyledataset import StyleDiffusion
    from mst_amd.utils import model_util
    model, d_ddim, _ = model_util.creat_serval_diffusion(diffusion_args(), StyleDiffusion, "ddim20")
    sd = {k: torch.from_numpy(np.ascontiguousarray(syn.tensor_for(SEED, k, tuple(v.shape))))
          for k, v in model.state_dict().items() if not k.endswith(".pe") and "clip_model" not in k}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    model.motion_enc.mdm_model.set_text_encoder(
        lambda texts: torch.stack([torch.from_numpy(syn.normal(SEED, "text/" + t, (512,))) for t in texts]))
    return model.to(device).eval(), d_ddim


@contextlib.contextmanager
def recorded_noise(tag):
    state = {"k": 0}
    orig = (torch.randn, torch.randn_like, torch.rand_like)

    def draw(shape, device):
        a = syn.normal(SEED, f"{tag}/noise/{state['k']}", tuple(shape))
        state["k"] += 1
        return torch.from_numpy(a).to(device)

    def draw_u(x, **kw):
        a = syn.uniform(SEED, f"{tag}/uniform/{state['k']}", tuple(x.shape), 0.0, 1.0)
        state["k"] += 1
        return torch.from_numpy(a).to(x.device)

    torch.randn = lambda *s, device=None, **kw: draw(s[0] if isinstance(s[0], (tuple, list)) else s, device)
    torch.randn_like = lambda x, **kw: draw(x.shape, x.device)
    torch.rand_like = draw_u
    try:
        yield
    finally:
        torch.randn, torch.randn_like, torch.rand_like = orig


def run_loop(device, save_dir, backend):
    """Drive our TrainInpaintingLoop exactly as the golden drove the reference's; returns (loop, record)."""
    from mst_amd.diffusion import logger
    from mst_amd.train.training_loop import TrainInpaintingLoop
    model, diffusion = build_model(device)
    if backend == "torch":                                   # the tests' own fp32 torch-op evaluation (tests/torch_reference.py)
        from torch_reference import use_torch_ops
        use_torch_ops(model)
    logger.configure(dir=save_dir)
    args = types.SimpleNamespace(save_dir=save_dir, **ARGS)
    data, style_data = batches()
    platform = types.SimpleNamespace(report_scalar=lambda **k: None, close=lambda: None)
    loop = TrainInpaintingLoop(args, platform, model, data, diffusion=diffusion, style_data=style_data)
    rec = {"loss": [], "rot_mse": [], "text_cosine": [], "t": [], "lr": [], "grad_norm": [], "param_norm": []}
    orig = diffusion.few_shot_style_finetune_losses

    def losses(*a, **k):
        terms = orig(*a, **k)
        rec["loss"].append(float(terms["loss"].detach()))
        rec["rot_mse"].append(terms["rot_mse"].detach().cpu().numpy().copy())
        rec["text_cosine"].append(float(terms["text_cosine"].detach()))
        rec["t"].append(a[2].cpu().numpy().copy())
        return terms

    diffusion.few_shot_style_finetune_losses = losses
    orig_opt = loop.mp_trainer.optimize

    def optimize(opt):
        rec["lr"].append(opt.param_groups[0]["lr"])
        if not hasattr(opt, "last_sq_norms"):
            g, p = loop.mp_trainer._compute_norms()
            rec["grad_norm"].append(float(g))
            rec["param_norm"].append(float(p))
            return orig_opt(opt)
        r = orig_opt(opt)
        rec["grad_norm"].append(loop.mp_trainer.last_norms[0])
        rec["param_norm"].append(loop.mp_trainer.last_norms[1])
        return r

    loop.mp_trainer.optimize = optimize
    np.random.seed(SEED % (2 ** 31))
    with recorded_noise("loop"):
        loop.run_loop()
    return loop, {k: np.asarray(v) for k, v in rec.items()}
