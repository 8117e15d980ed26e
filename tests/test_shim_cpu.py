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
