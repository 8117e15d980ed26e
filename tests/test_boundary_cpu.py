"""Host-side logic of the drop-in boundary that needs no GPU: factories, state-dict layout,
schedule tables, respacing, argument errors."""
import types

import numpy as np
import pytest
import torch

import mst_amd  # noqa: F401
import mst_amd.synthetic
from mst_amd.diffusion import gaussian_diffusion as gd
from mst_amd.diffusion.inpainting_gaussian_diffusion import InpaintingGaussianDiffusion
from mst_amd.diffusion.respace import SpacedDiffusion, space_timesteps
from mst_amd.model.mdm_forstyledataset import StyleDiffusion
from mst_amd.utils import model_util


def make_args(dataset="stylexia_posrot"):
    return types.SimpleNamespace(dataset=dataset, latent_dim=512, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                                 emb_trans_dec=False, diffusion_steps=1000, noise_schedule="cosine", sigma_small=True,
                                 lambda_vel=0.0, lambda_rcxyz=0.0, lambda_fc=0.0)


@pytest.mark.parametrize("sched", ["cosine", "linear"])
@pytest.mark.parametrize("resp", ["", "ddim20", "100", "10,20,30"])
def test_product_tables_equal_reference(golden, sched, resp):
    a = make_args()
    a.noise_schedule = sched
    d = model_util.create_gaussian_diffusion(a, InpaintingGaussianDiffusion, resp)
    g = golden["schedules"]
    assert np.array_equal(np.array(d.timestep_map), g[f"{sched}|{resp}|timestep_map"])
    for name in d.TABLES:
        assert np.array_equal(getattr(d, name), g[f"{sched}|{resp}|{name}"]), name
    assert d.num_timesteps == len(d.timestep_map)


def test_factories_return_reference_types():
    m, d1, d2 = model_util.creat_serval_diffusion(make_args(), StyleDiffusion, "ddim20")
    assert isinstance(d1, InpaintingGaussianDiffusion) and d1.num_timesteps == 20
    assert type(d2) is SpacedDiffusion and d2.num_timesteps == 1000
    m, d1, d2 = model_util.creat_ddpm_ddim_diffusion(make_args(), StyleDiffusion, "ddim20")
    assert isinstance(d2, InpaintingGaussianDiffusion) and d2.num_timesteps == 1000
    assert m.njoints == 181 and m.nfeats == 1 and m.cond_mode == "text" and m.cond_mask_prob == 0.1
    assert model_util.get_transfer_args(make_args("humanml"))["njoints"] == 263
    assert model_util.get_transfer_args(make_args("bandai-1_posrot"))["njoints"] == 190


def test_checkpoint_layout_is_the_references_96_tensors():
    """train/training_loop.py:312-348 saves state_dict minus motion_enc.* / clip_model.*: exactly the
    96 tensors seqTransEncoder.layers.{0-7}.*; load_model_wo_moenc must accept that file."""
    m, _, _ = model_util.creat_serval_diffusion(make_args(), StyleDiffusion, "ddim20")
    saved = {k: v for k, v in m.state_dict().items() if not k.startswith("motion_enc.") and "clip_model." not in k}
    assert len(saved) == 96
    assert sorted(saved) == sorted(k for i in range(8) for k in mst_amd.synthetic.layer_keys(i))
    assert saved["seqTransEncoder.layers.0.self_attn.in_proj_weight"].shape == (1536, 512)
    assert saved["seqTransEncoder.layers.7.linear2.weight"].shape == (512, 1024)
    model_util.load_model_wo_moenc(m, saved)
    with pytest.raises(AssertionError):
        model_util.load_model_wo_moenc(m, {**saved, "bogus.weight": torch.zeros(1)})
    prior_keys = [k for k in m.state_dict() if k.startswith("motion_enc.mdm_model.")]
    assert "motion_enc.mdm_model.sequence_pos_encoder.pe" in prior_keys
    assert m.state_dict()["motion_enc.mdm_model.sequence_pos_encoder.pe"].shape == (5000, 1, 512)
    assert [p.requires_grad for p in m.motion_enc.parameters()].count(True) == 0
    assert len(m.parameters_wo_enc()) == 96


def test_positional_table_matches_generator():
    m, _, _ = model_util.creat_serval_diffusion(make_args(), StyleDiffusion, "")
    pe = m.motion_enc.mdm_model.sequence_pos_encoder.pe[:, 0].numpy()
    assert np.array_equal(pe, mst_amd.synthetic.positional_table(5000, 512))


def test_cpu_tensors_are_rejected_not_computed_on_the_host():
    m, d1, _ = model_util.creat_serval_diffusion(make_args(), StyleDiffusion, "ddim20")
    x = torch.zeros(1, 181, 1, 76)
    with pytest.raises(RuntimeError, match="GPU"):
        d1.q_sample(x, torch.tensor([3]), model_kwargs={"y": {"inpainting_mask": torch.zeros_like(x)}})
    with pytest.raises(RuntimeError, match="GPU"):
        with torch.no_grad():
            m.eval()(x, torch.tensor([3]), y={"text_embed": torch.zeros(1, 512)})


def test_reference_error_behaviour():
    with pytest.raises(NotImplementedError):
        gd.get_named_beta_schedule("sqrt", 10)
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim600")
    _, d1, _ = model_util.creat_serval_diffusion(make_args(), StyleDiffusion, "ddim20")
    with pytest.raises(NotImplementedError):
        d1.ddim_sample_loop(None, (1, 181, 1, 76), const_noise=True)
    with pytest.raises(AttributeError):
        d1.training_losses(None)
    with pytest.raises(ValueError):
        gd.GaussianDiffusion(betas=np.array([0.1]), model_mean_type=gd.ModelMeanType.START_X,
                             model_var_type=gd.ModelVarType.FIXED_SMALL, loss_type=gd.LossType.MSE, lambda_pose=2.0)
