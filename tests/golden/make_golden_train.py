"""Golden vectors for SURVEY section 8f-1: the reference's own `TrainInpaintingLoop`
(train/training_loop.py:42-348) driven for 12 seeded steps on the CPU, dropout off (model.eval()),
every random draw recorded (torch noise through make_golden.recorded_noise, timesteps through
np.random.seed).  Captured per step: loss terms, grad/param norms as the trainer logs them, the sampled
timesteps, the learning rate; at the end: the checkpoint files' key lists and parameter slices.

Run in the authoring container only (imports /root/reference):
    python tests/golden/make_golden_train.py      ->  tests/golden/train_loop.npz

Shims beyond make_golden.install_shims (ordinary ModuleNotFoundError work-arounds, nothing under
/root/reference is modified): `blobfile` (join/dirname/exists/BlobFile over the local filesystem), `imageio`
and `utils.process_smpl_from_hybrik` (plotting / SMPL conversion helpers imported at module top, unused here)."""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import SEED, syn  # noqa: E402

STEPS_PER_EPOCH, NUM_STEPS = 4, 11          # run_loop performs (NUM_STEPS // 4 + 1) * 4 = 12 steps
F, T, B = 181, 76, 2
PROMPTS = ["a person walks proudly", "an old man jumps"]
ARGS = dict(dataset="stylexia_posrot", batch_size=B, lr=1e-4, log_interval=1, save_interval=1000, resume_checkpoint="",
            weight_decay=0.01, lr_anneal_steps=40, style_finetune=1, semantic_guidance=1, skip_steps=700, num_steps=NUM_STEPS,
            overwrite=True, use_ddim=1, diffusion_steps=1000, Ls=10.0)


def batches():
    """The (motion, cond) batches of one epoch and the style example, all seeded."""
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


def main():
    mg.install_shims()
    bf = types.ModuleType("blobfile")
    bf.join, bf.dirname, bf.exists = os.path.join, os.path.dirname, os.path.exists
    bf.BlobFile = lambda p, m="rb": open(p, m)
    sys.modules["blobfile"] = bf
    sys.modules["imageio"] = types.ModuleType("imageio")
    h = types.ModuleType("utils.process_smpl_from_hybrik")
    h.amass_to_pose = h.pos2hmlrep = None
    sys.modules["utils.process_smpl_from_hybrik"] = h
    import importlib
    mdm = importlib.import_module("model.mdm_forstyledataset")
    tl = importlib.import_module("train.training_loop")
    logger = importlib.import_module("diffusion.logger")
    model_util = importlib.import_module("utils.model_util")
    igd = importlib.import_module("diffusion.inpainting_gaussian_diffusion")

    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = mg.build_reference_model(mdm, F)               # eval(): dropout and the Bernoulli cond mask are off
    dargs = mg.args_for()
    diffusion = model_util.create_gaussian_diffusion(dargs, igd.InpaintingGaussianDiffusion, "ddim20")
    tmp = tempfile.mkdtemp(prefix="mst_loop_")
    logger.configure(dir=tmp, format_strs=[])
    args = types.SimpleNamespace(save_dir=tmp, **ARGS)
    data, style_data = batches()
    platform = types.SimpleNamespace(report_scalar=lambda **k: None, close=lambda: None)
    loop = tl.TrainInpaintingLoop(args, platform, model, data, diffusion=diffusion, style_data=style_data)

    rec = {"loss": [], "rot_mse": [], "text_cosine": [], "grad_norm": [], "param_norm": [], "t": [], "lr": []}
    orig_losses = diffusion.few_shot_style_finetune_losses

    def losses(*a, **k):
        terms = orig_losses(*a, **k)
        rec["loss"].append(float(terms["loss"]))
        rec["rot_mse"].append(terms["rot_mse"].detach().numpy().copy())
        rec["text_cosine"].append(float(terms["text_cosine"]))
        rec["t"].append(a[2].numpy().copy())
        return terms

    diffusion.few_shot_style_finetune_losses = losses
    orig_norms = loop.mp_trainer._compute_norms

    def norms(*a, **k):
        g, p = orig_norms(*a, **k)
        rec["grad_norm"].append(g)
        rec["param_norm"].append(p)
        rec["lr"].append(loop.opt.param_groups[0]["lr"])
        return g, p

    loop.mp_trainer._compute_norms = norms
    np.random.seed(SEED % (2 ** 31))
    with mg.recorded_noise("loop"):
        loop.run_loop()
    assert len(rec["loss"]) == (NUM_STEPS // STEPS_PER_EPOCH + 1) * STEPS_PER_EPOCH, len(rec["loss"])

    out = {k: np.asarray(v) for k, v in rec.items()}
    out["final_lr"] = np.array(loop.opt.param_groups[0]["lr"])
    out["final_step"] = np.array(loop.step)
    files = sorted(os.listdir(tmp))
    out["files"] = np.array("\n".join(f for f in files if f.endswith(".pt")))
    ck = torch.load(os.path.join(tmp, [f for f in files if f.startswith("model")][-1]))
    out["ckpt_keys"] = np.array("\n".join(ck.keys()))
    for k in ("seqTransEncoder.layers.0.self_attn.in_proj_weight", "seqTransEncoder.layers.7.linear2.weight",
              "seqTransEncoder.layers.3.norm1.weight", "seqTransEncoder.layers.5.linear1.bias"):
        ck[k] = ck[k].detach()
        out["param|" + k] = ck[k].reshape(-1)[:64].numpy().copy()
        out["delta|" + k] = (ck[k] - torch.from_numpy(np.ascontiguousarray(syn.tensor_for(SEED, k, tuple(ck[k].shape))))).reshape(-1)[:64].numpy().copy()
    opt = torch.load(os.path.join(tmp, [f for f in files if f.startswith("opt")][-1]))
    out["opt_state_count"] = np.array(len(opt["state"]))
    out["opt_group_keys"] = np.array("\n".join(sorted(opt["param_groups"][0].keys())))
    first = opt["state"][sorted(opt["state"].keys())[0]]
    out["opt_state_keys"] = np.array("\n".join(sorted(first.keys())))
    out["opt_step"] = np.array(float(first["step"]))
    np.savez_compressed(os.path.join(HERE, "train_loop.npz"), **out)
    print("losses", np.round(out["loss"], 5))
    print("grad_norm", np.round(out["grad_norm"], 4), "param_norm", out["param_norm"][:2])
    print("files", files, "ckpt keys", len(ck), "opt states", len(opt["state"]))
    print("train_loop.npz", os.path.getsize(os.path.join(HERE, "train_loop.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
