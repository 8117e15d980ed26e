"""Golden vectors that pin the TRAINING path to the reference, not to the tests' own torch restatement.

The reference's `few_shot_style_finetune_losses` (diffusion/gaussian_diffusion.py:1317-1399) is RUN here (authoring container only:
the reference is mounted at /root/reference and never travels) on seeded inputs with recorded noise, model in eval mode (dropout and
the Bernoulli text mask off; every random draw recorded), `terms["loss"].backward()`, and for each case the fixture keeps

  * the loss terms,
  * EVERY one of the 96 trainable gradients (`parameters_wo_enc()`): its norm (f64) and a fixed 64-dimensional random projection
    (`fold_project`: the flat gradient folded into rows of 4 096 with seeded row weights, then a seeded 64 x 4 096 matrix -- every
    element carries a non-zero random weight); every 1-D gradient (biases, LayerNorm gains) additionally IN FULL,
  * d loss / d x_start of the text-to-motion batch (through the frozen motion encoder, the eight trainable layers' input gradients
    and the pose embedding): in full at the Xia shape, norm + projection at the HumanML shape.

Cases: `xia|ft1` / `xia|ft0` -- the inputs and noise tags of tests/golden/make_golden.py's loss-term goldens (DDIM-20 skip 700: six chained
steps; DDPM skip 995: five), `hml|ft1` -- two (263, 1, 196) clips through the text-to-motion branch and a 196-frame content / style pair.

    python tests/golden/make_golden_grads.py          # rewrites tests/golden/ft_grads.npz  (about a minute on 8 cores)
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import SEED, syn  # noqa: E402

PROJ_DIM, FOLD = 64, 4096
PROMPTS = ["a person walks proudly", "an old man jumps"]


def fold_project(name, g):
    """[n] float -> (norm, [64] projection).  Linear in g; the weights depend on the tensor's NAME and size only."""
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    rows = (g.size + FOLD - 1) // FOLD
    buf = np.zeros(rows * FOLD, dtype=np.float64)
    buf[:g.size] = g
    w = syn.normal(SEED, f"gradproj/rows/{name}", (rows,)).astype(np.float64)
    a = syn.normal(SEED, "gradproj/matrix", (PROJ_DIM, FOLD)).astype(np.float64)
    return float(np.linalg.norm(g)), (a @ (w @ buf.reshape(rows, FOLD))).astype(np.float64)


def run_case(out, tag, model, dd, F, T, use_ddim, skip):
    B = 2
    shp = (1, F, 1, T)
    base = tag.split("|")[0]
    mask = torch.from_numpy(syn.root_horizontal_mask(B, F, T))
    motion = torch.from_numpy(syn.normal(SEED, f"{base}/motion", (B, F, 1, T)))
    y1 = {"y": {"text": PROMPTS[:1], "mask": torch.ones(1, 1, 1, T), "inpainting_mask": mask[:1], "inpainted_motion": motion[:1]}}
    t2m = torch.from_numpy(syn.normal(SEED, f"{base}/t2m", (B, F, 1, T))).requires_grad_(True)
    fm = torch.ones(B, 1, 1, T)
    fm[1, ..., T - 9:] = 0
    y_t2m = {"y": {"text": PROMPTS, "mask": fm, "inpainting_mask": mask.double(), "inpainted_motion": t2m.detach()}}
    style = torch.from_numpy(syn.normal(SEED, f"{base}/style", shp))
    model.zero_grad()
    for p_ in model.parameters_wo_enc():
        p_.requires_grad_(True)
    with torch.enable_grad(), mg.recorded_noise(f"{base}/ft{use_ddim}"):
        terms = dd.few_shot_style_finetune_losses(model, t2m, torch.tensor([2, 4]), motion[:1], style, skip_steps=skip, model_kwargs=y1,
                                                  model_t2m_kwargs=y_t2m, semantic_guidance=1, use_ddim=use_ddim, Ls=10)
    with torch.enable_grad():
        terms["loss"].backward()
    for k in ("rot_mse", "text_cosine", "loss"):
        out[f"{tag}|{k}"] = terms[k].detach().numpy()
    names = []
    trainable = {id(p) for p in model.parameters_wo_enc()}
    for name, p in model.named_parameters():
        if id(p) not in trainable:
            continue
        assert p.grad is not None, name
        g = p.grad.numpy()
        n, pr = fold_project(name, g)
        names.append(name)
        out[f"{tag}|norm|{name}"] = np.array(n)
        out[f"{tag}|proj|{name}"] = pr
        if g.ndim == 1:
            out[f"{tag}|full|{name}"] = g.copy()
    assert len(names) == 96, len(names)
    out[f"{tag}|names"] = np.array(names)
    gx = t2m.grad.numpy()
    n, pr = fold_project("x_start", gx)
    out[f"{tag}|dx_norm"] = np.array(n)
    out[f"{tag}|dx_proj"] = pr
    if base == "xia":
        out[f"{tag}|dx_full"] = gx.copy()
    print(tag, "loss", float(terms["loss"]), "steps", terms["rot_mse"].shape[0], "|dL/dx_start|", n,
          "largest / smallest gradient norm", max(float(out[f"{tag}|norm|{k}"]) for k in names), min(float(out[f"{tag}|norm|{k}"]) for k in names))
    model.zero_grad()


def main():
    mg.install_shims()
    torch.set_num_threads(8)
    igd = importlib.import_module("diffusion.inpainting_gaussian_diffusion")
    mdm = importlib.import_module("model.mdm_forstyledataset")
    mu = importlib.import_module("utils.model_util")
    out = {}
    for base, F, T in (("xia", 181, 76), ("hml", 263, 196)):
        model = mg.build_reference_model(mdm, F)
        d_full = mu.create_gaussian_diffusion(mg.args_for(), igd.InpaintingGaussianDiffusion, "")
        d_ddim = mu.create_gaussian_diffusion(mg.args_for(), igd.InpaintingGaussianDiffusion, "ddim20")
        run_case(out, f"{base}|ft1", model, d_ddim, F, T, 1, 700)
        if base == "xia":
            run_case(out, f"{base}|ft0", model, d_full, F, T, 0, 995)
    # the loss terms of the two Xia cases must be those of denoise.npz (same inputs, same recorded draws)
    old = np.load(os.path.join(HERE, "denoise.npz"))
    for u in (0, 1):
        for k in ("rot_mse", "text_cosine", "loss"):
            assert np.array_equal(out[f"xia|ft{u}|{k}"], old[f"xia|ft{u}|{k}"]), (u, k)
    np.savez_compressed(os.path.join(HERE, "ft_grads.npz"), **out)
    print("ft_grads.npz", os.path.getsize(os.path.join(HERE, "ft_grads.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
