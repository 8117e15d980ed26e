"""SURVEY section 8f-1, host side (no GPU): our `TrainInpaintingLoop` control flow against what the REFERENCE's loop
(train/training_loop.py:42-348) did in its 12-step golden run (tests/golden/train_loop.npz): step / epoch / save
arithmetic, the timestep draws (same np.random call pattern), the learning-rate anneal, checkpoint file names, the
96-key model file and the torch-AdamW-layout optimizer file.  The objective is replaced by a stand-in quadratic (the
diffusion kernels are GPU-only and raise on CPU tensors); loss parity is tests/test_gpu_training_loop.py."""
import os
import types

import numpy as np
import torch

from conftest import SEED
import loop_fixture as lf


class StandInDiffusion:
    num_timesteps = 20

    def few_shot_style_finetune_losses(self, model, x_start, t, content, style, **kw):
        self.ts.append(t.numpy().copy())
        reg = sum((p ** 2).sum() for p in model.parameters_wo_enc()) * 1e-6
        return {"loss": reg, "rot_mse": reg.detach().expand(6), "text_cosine": reg.detach()}


def test_loop_control_flow_matches_reference_run(golden, tmp_path, monkeypatch):
    from mst_amd.diffusion import logger
    from mst_amd.model.mdm_forstyledataset import StyleDiffusion
    from mst_amd.train.training_loop import TrainInpaintingLoop
    from mst_amd.utils import model_util
    g = golden["train_loop"]
    import mst_amd.train.training_loop as tl
    model, _, _ = model_util.creat_serval_diffusion(lf.diffusion_args(), StyleDiffusion, "ddim20")
    monkeypatch.setattr(tl, "FusedAdamW", torch.optim.AdamW)       # no GPU here: the product's optimizer kernel raises on CPU tensors
    diffusion = StandInDiffusion()
    diffusion.ts = []
    logger.configure(dir=str(tmp_path))
    args = types.SimpleNamespace(save_dir=str(tmp_path), **lf.ARGS)
    data, style_data = lf.batches()
    lrs = []
    platform = types.SimpleNamespace(report_scalar=lambda **k: None, close=lambda: None)
    loop = TrainInpaintingLoop(args, platform, model, data, diffusion=diffusion, style_data=style_data)
    assert type(loop.opt) is torch.optim.AdamW                      # the stand-in installed above
    step = loop.opt.step
    loop.opt.step = lambda *a, **k: (lrs.append(loop.opt.param_groups[0]["lr"]), step(*a, **k))[1]
    np.random.seed(SEED % (2 ** 31))
    loop.run_loop()
    assert loop.step == int(g["final_step"]) == 12
    assert np.array_equal(np.asarray(diffusion.ts), g["t"])          # same np.random draws -> same indices
    assert np.allclose(lrs, g["lr"], rtol=1e-12) and abs(loop.opt.param_groups[0]["lr"] - float(g["final_lr"])) < 1e-15
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".pt"))
    assert "\n".join(files) == str(g["files"])
    ck = torch.load(os.path.join(tmp_path, "model000000012.pt"))
    assert "\n".join(ck.keys()) == str(g["ckpt_keys"]) and len(ck) == 96
    opt = torch.load(os.path.join(tmp_path, "opt000000012.pt"))
    assert len(opt["state"]) == int(g["opt_state_count"]) == 96
    first = opt["state"][sorted(opt["state"].keys())[0]]
    assert "\n".join(sorted(first.keys())) == str(g["opt_state_keys"]) and float(first["step"]) == float(g["opt_step"])
    assert "\n".join(sorted(opt["param_groups"][0].keys())) == str(g["opt_group_keys"])
    kv = logger.get_current().name2val
    assert kv["step"] == 11 and kv["samples"] == 12 * lf.B and "grad_norm" in kv and "param_norm" in kv


def test_resume_step_parsing_and_lookup(tmp_path):
    from mst_amd.train.training_loop import find_resume_checkpoint, parse_resume_step_from_filename
    assert parse_resume_step_from_filename("/x/y/model000001200.pt") == 1200
    assert parse_resume_step_from_filename("/x/y/weights.pt") == 0
    for s in (0, 300, 1200):
        open(tmp_path / f"model{s:09d}.pt", "wb").close()
        open(tmp_path / f"opt{s:09d}.pt", "wb").close()
    assert find_resume_checkpoint(str(tmp_path), "model").endswith("model000001200.pt")
    assert find_resume_checkpoint(str(tmp_path), "opt").endswith("opt000001200.pt")
