"""INTEGRATION.md Option A on a CPU box: a stand-in reference checkout (packages named like the reference's, the shadowed
modules poisoned so that importing the checkout's copy fails the test) + `PYTHONPATH=shim`, then the import lines of
sample/demo_style_transfer.py:10-20, train/finetune_style_diffusion.py:10-19 and utils/model_util.py:1-5 in a fresh
interpreter: the denoise-path names must resolve to this repository's classes, every other module to the checkout."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

POISON = "raise ImportError('the checkout copy of a shadowed module was imported')\n"
CHECKOUT = {
    "utils/__init__.py": "", "utils/parser_util.py": "def get_cond_mode(args):\n    return 'text'\n", "utils/fixseed.py": "def fixseed(s):\n    return s\n",
    "utils/dist_util.py": "def dev():\n    return 'cpu'\n", "utils/model_util.py": POISON,
    "diffusion/__init__.py": "", "diffusion/nn.py": "def mean_flat(x):\n    return x\n", "diffusion/gaussian_diffusion.py": POISON,
    "diffusion/respace.py": POISON, "diffusion/inpainting_gaussian_diffusion.py": POISON,
    "model/__init__.py": "", "model/smpl.py": "SMPL = object\n", "model/mdm_forstyledataset.py": POISON, "model/cfg_sampler.py": POISON,
    "train/__init__.py": "", "train/train_platforms.py": "class NoPlatform:\n    pass\n", "train/training_loop.py": POISON,
}
SCRIPT = textwrap.dedent("""
    from utils.fixseed import fixseed
    from utils.parser_util import get_cond_mode
    from utils import dist_util
    from utils.model_util import load_model_wo_controlmdm, creat_serval_diffusion, load_model_wo_moenc, creat_ddpm_ddim_diffusion
    from diffusion.inpainting_gaussian_diffusion import InpaintingGaussianDiffusion
    from diffusion import gaussian_diffusion as gd
    from diffusion.respace import SpacedDiffusion, space_timesteps
    from diffusion.nn import mean_flat
    from model.mdm_forstyledataset import StyleDiffusion, MDM, MotionEncoder
    from model.cfg_sampler import ClassifierFreeSampleModel
    from model.smpl import SMPL
    from train.training_loop import TrainInpaintingLoop
    from train.train_platforms import NoPlatform
    import utils.parser_util, diffusion.nn
    mine = (creat_serval_diffusion, InpaintingGaussianDiffusion, gd.GaussianDiffusion, SpacedDiffusion, StyleDiffusion, MDM,
            ClassifierFreeSampleModel, TrainInpaintingLoop, load_model_wo_moenc)
    assert all(o.__module__.startswith("mst_amd.") for o in mine), [o.__module__ for o in mine]
    assert utils.parser_util.__file__.startswith(CHECKOUT) and diffusion.nn.__file__.startswith(CHECKOUT)
    assert dist_util.dev() == 'cpu' and fixseed(3) == 3 and mean_flat(1) == 1
    import types
    a = types.SimpleNamespace(dataset="stylexia_posrot", latent_dim=512, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                              emb_trans_dec=False, diffusion_steps=1000, noise_schedule="cosine", sigma_small=True,
                              lambda_vel=0.0, lambda_rcxyz=0.0, lambda_fc=0.0)
    model, d_ddim, d_plain = creat_serval_diffusion(a, StyleDiffusion, "ddim20")          # demo_style_transfer.py:57-62
    assert isinstance(d_ddim, InpaintingGaussianDiffusion) and d_ddim.num_timesteps == 20 and len(model.state_dict()) > 96
    print("SHIM_OK")
""")


def test_reference_import_lines_resolve_through_the_shim(tmp_path):
    for rel, body in CHECKOUT.items():
        f = tmp_path / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(body)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "shim"))
    code = f"CHECKOUT = {str(tmp_path)!r}\n" + SCRIPT
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SHIM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---------------------------------------------------------------------------------------- the REAL checkout, where it exists
REFERENCE = "/root/reference"
REAL_SCRIPT = textwrap.dedent("""
    import sys, types
    import numpy as np
    np.float, np.int = float, int                      # numpy-2 aliases the reference's quaternion / resample modules use at import time
    for name in ("blobfile", "smplx"):                 # absent third-party packages some host-glue modules import (CLIP stays absent:
        sys.modules.setdefault(name, types.ModuleType(name))      # the engine-backed model then takes post-CLIP embeddings from y['text_embed'])
    # the import lines of sample/demo_style_transfer.py:10-17 and train/finetune_style_diffusion.py:10-19 that concern the denoise
    # path and its host glue (data loaders, BVH export and visualisation need assets / packages this container does not have)
    from utils.fixseed import fixseed
    from utils.parser_util import eval_inpainting_style_args, finetune_inpainting_style_args
    from utils import dist_util
    from utils.model_util import load_model_wo_controlmdm, creat_serval_diffusion, load_model_wo_moenc, creat_ddpm_ddim_diffusion
    from diffusion.inpainting_gaussian_diffusion import InpaintingGaussianDiffusion
    from train.training_loop import TrainInpaintingLoop
    from train.train_platforms import NoPlatform
    from data_loaders.tensors import collate
    from data_loaders.humanml.scripts.motion_process import recover_from_ric
    import utils.parser_util, utils.fixseed, utils.dist_util, train.train_platforms, data_loaders.tensors, diffusion.nn
    for mod in (utils.parser_util, utils.fixseed, utils.dist_util, train.train_platforms, data_loaders.tensors, diffusion.nn):
        assert mod.__file__.startswith(REFERENCE + "/"), mod.__file__          # the checkout's own files
    mine = (creat_serval_diffusion, creat_ddpm_ddim_diffusion, load_model_wo_moenc, InpaintingGaussianDiffusion, TrainInpaintingLoop)
    assert all(o.__module__.startswith("mst_amd.") for o in mine), [o.__module__ for o in mine]
    import diffusion.gaussian_diffusion, diffusion.respace, model.mdm_forstyledataset, model.cfg_sampler
    for mod in (diffusion.gaussian_diffusion, diffusion.respace, model.mdm_forstyledataset, model.cfg_sampler):
        assert "diffusion-based-motion-style-transfer_amd" in mod.__file__, mod.__file__
    # the factory call of demo_style_transfer.py:57-62 with the reference's OWN argument parser defaults
    sys.argv = ["demo", "--model_path", "/nonexistent/model.pt"]
    from utils.parser_util import get_cond_mode
    a = types.SimpleNamespace(dataset="stylexia_posrot", latent_dim=512, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                              emb_trans_dec=False, diffusion_steps=1000, noise_schedule="cosine", sigma_small=True,
                              lambda_vel=0.0, lambda_rcxyz=0.0, lambda_fc=0.0, unconstrained=False)
    from model.mdm_forstyledataset import StyleDiffusion
    model, d_ddim, d_plain = creat_serval_diffusion(a, StyleDiffusion, "ddim20")
    assert get_cond_mode(a) == "text" and d_ddim.num_timesteps == 20 and d_plain.num_timesteps == 1000
    batch = collate([{"inp": __import__("torch").zeros(181, 1, 76), "lengths": 76, "text": "a"}])   # the reference's collate feeds our model
    assert batch[1]["y"]["mask"].shape[-1] == 76
    print("REAL_SHIM_OK")
""")


def test_shim_against_the_real_reference_checkout():
    """The same adoption path run inside the actual reference tree (read-only; present in the authoring container only)."""
    import pytest
    if not os.path.isdir(os.path.join(REFERENCE, "diffusion")):
        pytest.skip("no reference checkout on this box")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "shim"), PYTHONDONTWRITEBYTECODE="1")
    code = f"REFERENCE = {REFERENCE!r}\n" + REAL_SCRIPT
    r = subprocess.run([sys.executable, "-c", code], cwd=REFERENCE, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "REAL_SHIM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
