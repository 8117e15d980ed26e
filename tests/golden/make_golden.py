"""Generate the golden vectors that pin `oracle/` to the reference.

Runs ONLY in the authoring container, where the reference is mounted read-only at /root/reference
(it never travels to the GPU box; the tests read the committed .npz files, not the reference).
The reference has no tests or fixtures of its own, so these vectors are the
reference ITSELF executed here (torch 2.10 CPU, fp32) on seeded inputs:

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

In-process shims (nothing under /root/reference is modified, nothing is installed):
  * np.float / np.int aliases (used at import time by the reference's quaternion/resample modules)
  * a stub `clip` module whose tokenizer maps each prompt to a seeded [512] vector and whose
    text encoder is the identity (CLIP itself is third-party and absent; the engine's input is
    the post-CLIP embedding)
  * a stub `model.smpl` (smplx + SMPL body files are absent; never called on this path)
  * torch.load of the two pretrained checkpoints returns {} and the strict-key asserts are
    bypassed; every parameter is then overwritten with `mst_amd.synthetic.tensor_for(seed, key)`
  * torch.randn / torch.randn_like are replaced, while a loop runs, by a recorded deterministic
    sequence (`synthetic.normal(seed, "noise/<k>")`) so the oracle can replay identical noise.
Only inputs' seeds and the reference's outputs are stored (weights and inputs are regenerated
from the seeds by the tests).
"""
import contextlib
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import mst_amd  # noqa: E402
from mst_amd import synthetic as syn  # noqa: E402

REF = "/root/reference"
SEED = 20261003


# ------------------------------------------------------------------------------------------ shims
def install_shims():
    np.float = float
    np.int = int
    sys.path.insert(0, REF)
    clip = types.ModuleType("clip")
    clip.model = types.ModuleType("clip.model")

    class FakeClip(nn.Module):
        def __init__(self):
            super().__init__()
            self.p = nn.Parameter(torch.zeros(1))

        def encode_text(self, tok):
            return tok

    clip.load = lambda *a, **k: (FakeClip(), None)
    clip.tokenize = lambda texts, **k: torch.stack([text_embedding(t) for t in texts])
    clip.model.convert_weights = lambda m: None
    sys.modules["clip"] = clip
    sys.modules["clip.model"] = clip.model
    smpl = types.ModuleType("model.smpl")

    class SMPL(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    smpl.SMPL = SMPL
    smpl.JOINTSTYPE_ROOT = {}
    sys.modules["model.smpl"] = smpl


def text_embedding(prompt):
    return torch.from_numpy(syn.normal(SEED, "text/" + prompt, (512,)))


@contextlib.contextmanager
def recorded_noise(tag):
    """Replace torch.randn / randn_like by the k-th tensor of a seeded sequence."""
    state = {"k": 0}
    orig = (torch.randn, torch.randn_like, torch.rand_like)

    def draw(shape):
        a = syn.normal(SEED, f"{tag}/noise/{state['k']}", tuple(shape))
        state["k"] += 1
        return torch.from_numpy(a)

    torch.randn = lambda *shape, **kw: draw(shape[0] if isinstance(shape[0], (tuple, list)) else shape)
    torch.randn_like = lambda x, **kw: draw(x.shape)

    def draw_uniform(x, **kw):          # th.rand_like (gaussian_diffusion.py:1332)
        a = syn.uniform(SEED, f"{tag}/uniform/{state['k']}", tuple(x.shape), 0.0, 1.0)
        state["k"] += 1
        return torch.from_numpy(a)

    torch.rand_like = draw_uniform
    try:
        yield state
    finally:
        torch.randn, torch.randn_like, torch.rand_like = orig


# -------------------------------------------------------------------------------------- reference
def build_reference_model(mdm, njoints):
    orig_load = torch.load
    torch.load = lambda *a, **k: {}
    mdm.MotionEncoder.load_model_wo_clip = lambda self, model, sd: None
    mdm.StyleDiffusion.load_model = lambda self, model, sd: None
    try:
        model = mdm.StyleDiffusion(
            modeltype="", njoints=njoints, nfeats=1, num_actions=1, translation=True,
            pose_rep="rot6d", glob=True, glob_rot=True, latent_dim=512, ff_size=1024,
            num_layers=8, num_heads=4, dropout=0.1, activation="gelu", data_rep="hml_vec",
            cond_mode="text", cond_mask_prob=0.1, action_emb="tensor", arch="trans_enc",
            emb_trans_dec=False, clip_version="ViT-B/32", dataset="stylexia_posrot",
            mdm_path="x", semantic_discriminator_path="y")
    finally:
        torch.load = orig_load
    sd = {}
    for k, v in model.state_dict().items():
        if k.endswith(".pe") or "clip_model" in k:
            continue
        sd[k] = torch.from_numpy(np.ascontiguousarray(syn.tensor_for(SEED, k, tuple(v.shape))))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    assert all(k.endswith(".pe") or "clip_model" in k for k in missing), missing
    return model.eval()


def args_for(respacing_steps=1000, schedule="cosine"):
    a = types.SimpleNamespace()
    a.diffusion_steps = respacing_steps
    a.noise_schedule = schedule
    a.sigma_small = True
    a.lambda_vel = a.lambda_rcxyz = a.lambda_fc = 0.0
    return a


def main():
    install_shims()
    gd = importlib.import_module("diffusion.gaussian_diffusion")
    rs = importlib.import_module("diffusion.respace")
    igd = importlib.import_module("diffusion.inpainting_gaussian_diffusion")
    mdm = importlib.import_module("model.mdm_forstyledataset")
    cfg = importlib.import_module("model.cfg_sampler")
    # utils.model_util imports utils.parser_util -> argparse only; fine
    mu = importlib.import_module("utils.model_util")

    # ---------------------------------------------------------------- 1. schedule tables
    sched = {}
    names = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
             "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
             "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
             "sqrt_recipm1_alphas_cumprod", "posterior_variance",
             "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")
    for schedule in ("cosine", "linear"):
        for resp in ("", "ddim20", "100", "10,20,30"):
            d = mu.create_gaussian_diffusion(args_for(1000, schedule), igd.InpaintingGaussianDiffusion,
                                             timestep_respacing=resp)
            key = f"{schedule}|{resp}"
            for n in names:
                sched[f"{key}|{n}"] = getattr(d, n)
            sched[f"{key}|timestep_map"] = np.array(d.timestep_map, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "schedules.npz"), **sched)

    # ---------------------------------------------------------------- 2. inpainting masks
    masks = {}
    for modname, F in (("stylexia_posrot_utils", 181), ("bandai_posrot_utils", 190),
                       ("humanml_utils", 263)):
        m = importlib.import_module("data_loaders." + modname)
        for name in ("root", "root_horizontal", "y_rotation", "upper_body", "lower_body",
                     "root_horizontal,lower_body"):
            full = m.get_inpainting_mask(name, (2, F, 1, 5))
            assert full.dtype == np.float64 and full.shape == (2, F, 1, 5)
            row = full[0, :, 0, 0]
            assert (full == row.reshape(1, F, 1, 1)).all()
            masks[f"{modname}|{name}"] = row.astype(np.uint8)
        pre = m.get_inpainting_mask("prefix", (1, F, 1, 30), prefix_length=20)
        masks[f"{modname}|prefix20_T30"] = pre[0, 0, 0].astype(np.uint8)
        assert (pre == pre[0, 0, 0].reshape(1, 1, 1, 30)).all()
    np.savez_compressed(os.path.join(HERE, "masks.npz"), **masks)

    # ---------------------------------------------------------------- 3./4./5. model + diffusion
    out = {}

    def keep(tag, a):
        """HML-shape step outputs: store clip 1 only (clip 0 = the t=0/1 edge is covered by xia)."""
        return a[1:] if tag == "hml" else a

    for tag, F, T in (("xia", 181, 76), ("hml", 263, 196)):
        model = build_reference_model(mdm, F)
        # positional table must equal the product-side generator bit for bit
        pe_ref = model.motion_enc.mdm_model.sequence_pos_encoder.pe[:, 0].numpy()
        assert np.array_equal(pe_ref, syn.positional_table(5000, 512))
        B = 2
        x = torch.from_numpy(syn.normal(SEED, f"{tag}/x", (B, F, 1, T)))
        t = torch.tensor([3, 957])
        prompts = ["a person walks proudly", "an old man jumps"]
        y = {"text": prompts, "mask": torch.ones(B, 1, 1, T)}
        with torch.no_grad():
            out[f"{tag}|fwd_cond"] = model(x, t, y=y).numpy()
            out[f"{tag}|fwd_uncond"] = keep(tag, model(x, t, y={**y, "uncond": True}).numpy())
            out[f"{tag}|prior_fwd"] = keep(tag, model.motion_enc.mdm_model(x, t, y=y).numpy())
            y_cfg = {**y, "scale": torch.tensor([2.5, 1.5])}
            out[f"{tag}|cfg"] = cfg.ClassifierFreeSampleModel(model)(x, t, y_cfg).numpy()
            lens = [T, T - 17]
            fm = torch.zeros(B, 1, 1, T)
            for i, n in enumerate(lens):
                fm[i, ..., :n] = 1
            mu_vec, txt = model.motion_enc(x, y={"mask": fm, "text": prompts})
            out[f"{tag}|motion_enc_mu"] = mu_vec.numpy()

        # ---- single steps through the reference's diffusion objects (model in the loop)
        d_full = mu.create_gaussian_diffusion(args_for(), igd.InpaintingGaussianDiffusion, "")
        d_ddim = mu.create_gaussian_diffusion(args_for(), igd.InpaintingGaussianDiffusion, "ddim20")
        d_100 = mu.create_gaussian_diffusion(args_for(), igd.InpaintingGaussianDiffusion, "100")
        d_base = mu.create_gaussian_diffusion(args_for(), rs.SpacedDiffusion, "")
        mask = torch.from_numpy(syn.root_horizontal_mask(B, F, T))
        motion = torch.from_numpy(syn.normal(SEED, f"{tag}/motion", (B, F, 1, T)))
        yk = {"y": {**y, "inpainting_mask": mask, "inpainted_motion": motion}}
        with torch.no_grad():
            with recorded_noise(f"{tag}/q"):
                out[f"{tag}|q_sample"] = d_full.q_sample(motion, torch.tensor([10, 700]),
                                                         model_kwargs=yk).numpy()
            for name, dd, tt in (("full", d_full, [0, 500]), ("ddim", d_ddim, [0, 19]),
                                 ("r100", d_100, [1, 99])):
                tt = torch.tensor(tt)
                with recorded_noise(f"{tag}/ps_{name}"):
                    r = dd.p_sample(model, x, tt, clip_denoised=False, model_kwargs=yk)
                out[f"{tag}|p_sample_{name}|sample"] = keep(tag, r["sample"].numpy())
                out[f"{tag}|p_sample_{name}|pred_xstart"] = keep(tag, r["pred_xstart"].numpy())
                with recorded_noise(f"{tag}/dd_{name}"):
                    r = dd.ddim_sample(model, x, tt, clip_denoised=False, model_kwargs=yk)
                out[f"{tag}|ddim_sample_{name}|sample"] = keep(tag, r["sample"].numpy())
                with recorded_noise(f"{tag}/dd5_{name}"):
                    r = dd.ddim_sample(model, x, tt, clip_denoised=False, model_kwargs=yk, eta=0.5)
                out[f"{tag}|ddim_sample_eta_{name}|sample"] = keep(tag, r["sample"].numpy())
            # base-class (non-inpainting) step: noise is NOT masked, blend still applies
            with recorded_noise(f"{tag}/ps_base"):
                r = d_base.p_sample(model, x, torch.tensor([7, 400]), clip_denoised=False,
                                    model_kwargs=yk)
            out[f"{tag}|p_sample_base|sample"] = keep(tag, r["sample"].numpy())

            # ---- loops
            if tag == "xia":
                shp = (1, F, 1, T)
                y1 = {"y": {"text": prompts[:1], "mask": torch.ones(1, 1, 1, T),
                            "inpainting_mask": mask[:1], "inpainted_motion": motion[:1]}}
                # config 1: 100 respaced DDPM steps, single clip (BASELINE.json configs[0])
                with recorded_noise("xia/loop100"):
                    s = d_100.p_sample_loop(model, shp, clip_denoised=False, model_kwargs=y1)
                out["xia|loop100|sample"] = s.numpy()
                # the demo setting: ddim20, skip 14, init_image, dump_all_xstart
                with recorded_noise("xia/demo"):
                    dump = d_ddim.ddim_sample_loop(model, shp, clip_denoised=False, model_kwargs=y1,
                                                   skip_timesteps=14, init_image=motion[:1],
                                                   dump_all_xstart=True)
                out["xia|demo|xstart"] = torch.cat(dump).numpy()
                # neutralisation pre-pass shape: frozen prior as denoiser, stop_timesteps
                y_n = {"y": {"text": prompts[:1], "mask": torch.ones(1, 1, 1, T),
                             "inpainting_mask": torch.zeros(shp), "inpainted_motion": motion[:1]}}
                with recorded_noise("xia/neutral"):
                    dump = d_full.p_sample_loop(model.motion_enc.mdm_model, shp, clip_denoised=False,
                                                model_kwargs=y_n, skip_timesteps=0,
                                                init_image=motion[:1], stop_timesteps=990,
                                                dump_all_xstart=True)
                out["xia|neutral|xstart_last"] = dump[-1].numpy()
                out["xia|neutral|n"] = np.array(len(dump))
                # CFG-wrapped model inside a DDPM loop (BASELINE config 3 at toy length)
                y_c = {"y": {**y1["y"], "scale": torch.tensor([2.5])}}
                with recorded_noise("xia/cfgloop"):
                    s = d_full.p_sample_loop(cfg.ClassifierFreeSampleModel(model), shp,
                                             clip_denoised=False, model_kwargs=y_c,
                                             skip_timesteps=990, init_image=motion[:1])
                out["xia|cfgloop|sample"] = s.numpy()
                # ---- fine-tune objective (few_shot_style_finetune_losses) in eval mode: dropout and the
                # Bernoulli cond mask are off, every draw is recorded; loss terms + one gradient norm
                t2m = torch.from_numpy(syn.normal(SEED, "xia/t2m", (B, F, 1, T)))
                fm = torch.ones(B, 1, 1, T)
                fm[1, ..., T - 9:] = 0
                y_t2m = {"y": {"text": prompts, "mask": fm, "inpainting_mask": mask.double(),
                               "inpainted_motion": t2m}}
                style = torch.from_numpy(syn.normal(SEED, "xia/style", shp))
                for use_ddim, dd in ((1, d_ddim), (0, d_full)):
                    model.zero_grad()
                    for p_ in model.parameters_wo_enc():
                        p_.requires_grad_(True)
                    with torch.enable_grad(), recorded_noise(f"xia/ft{use_ddim}"):
                        terms = dd.few_shot_style_finetune_losses(
                            model, t2m, torch.tensor([2, 4]), motion[:1], style, skip_steps=700 if use_ddim else 995,
                            model_kwargs=y1, model_t2m_kwargs=y_t2m, semantic_guidance=1, use_ddim=use_ddim, Ls=10)
                    with torch.enable_grad():
                        terms["loss"].backward()
                    out[f"xia|ft{use_ddim}|rot_mse"] = terms["rot_mse"].detach().numpy()
                    out[f"xia|ft{use_ddim}|text_cosine"] = terms["text_cosine"].detach().numpy()
                    out[f"xia|ft{use_ddim}|loss"] = terms["loss"].detach().numpy()
                    gsd = dict(model.named_parameters())
                    out[f"xia|ft{use_ddim}|grad_l0_inproj"] = gsd["seqTransEncoder.layers.0.self_attn.in_proj_weight"].grad[:8, :8].numpy().copy()
                    out[f"xia|ft{use_ddim}|grad_l7_lin2_norm"] = np.array(gsd["seqTransEncoder.layers.7.linear2.weight"].grad.norm().item())
                model.zero_grad()
            else:
                shp = (B, F, 1, T)
                with recorded_noise("hml/tail8"):
                    s = d_full.p_sample_loop(model, shp, clip_denoised=False, model_kwargs=yk,
                                             skip_timesteps=992, init_image=motion)
                out["hml|tail8|sample"] = s.numpy()
    np.savez_compressed(os.path.join(HERE, "denoise.npz"), **out)
    for f in ("schedules.npz", "masks.npz", "denoise.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
