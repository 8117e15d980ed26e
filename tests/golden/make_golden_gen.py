"""Golden vectors for SURVEY section 8 rows f-4 and f-2, produced by RUNNING THE REFERENCE here (authoring container only):

  gen.npz      the reference's `CompMDMGeneratedDataset` (data_loaders/humanml/motion_loaders/comp_v6_model_dataset.py:146-240)
               driven with a stub text-to-motion dataloader, the seeded StyleDiffusion wrapped in the reference's
               ClassifierFreeSampleModel, a 100-step respaced SpacedDiffusion and recorded noise: generated clips, multimodality
               repeats, bookkeeping.
  (same file)  the neutralisation pre-pass exactly as train/finetune_style_diffusion.py:195-212 issues it: the frozen prior as
               the denoiser, `stop_timesteps = 900`, `dump_all_xstart=True` -> 100 x0-hat tensors.  Stored: 4 of them in full,
               and of all 100 the norm and a fixed 64-dimensional random projection (5.5 MB of tensors would not be a small fixture).

    python tests/golden/make_golden_gen.py        # rewrites tests/golden/gen.npz

Shims as in make_golden.py (imported from it); additionally `data_loaders.humanml.networks.*` (the T2M evaluator networks the
module imports at the top but the class never touches) are stubbed when they do not import."""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import SEED, syn  # noqa: E402

F, T, BS, NB = 181, 76, 2, 3
PROJ_DIM = 64


class Loader:
    """Stub text-to-motion dataloader: NB batches of BS clips with the y-dict keys the reference's collate produces."""
    batch_size = BS

    class _DS:
        w_vectorizer, mode = None, "gt"

        def __len__(self):
            return NB * BS

    def __init__(self):
        self.dataset = Loader._DS()

    def __len__(self):
        return NB

    def __iter__(self):
        for i in range(NB):
            motion = torch.from_numpy(syn.normal(SEED, f"gen/motion/{i}", (BS, F, 1, T)))
            texts = [f"clip {i} {b}" for b in range(BS)]
            yield motion, {"y": {"text": texts, "tokens": ["a/DET_person/NOUN_walks/VERB"] * BS,
                                 "lengths": torch.tensor([T, T - 7]), "mask": torch.ones(BS, 1, 1, T)}}


def projection():
    return syn.normal(SEED, "gen/projection", (PROJ_DIM, F * T))


def main():
    mg.install_shims()
    try:
        importlib.import_module("data_loaders.humanml.networks.modules")
        importlib.import_module("data_loaders.humanml.networks.trainers")
    except Exception as e:                       # evaluator networks: imported by the module, unused by the class
        print("stubbing data_loaders.humanml.networks (", type(e).__name__, e, ")")
        for name in ("data_loaders.humanml.networks", "data_loaders.humanml.networks.modules", "data_loaders.humanml.networks.trainers"):
            sys.modules[name] = types.ModuleType(name)
        sys.modules["data_loaders.humanml.networks.trainers"].CompTrainerV6 = object
    comp = importlib.import_module("data_loaders.humanml.motion_loaders.comp_v6_model_dataset")
    if not hasattr(comp, "np"):                  # the module gets `np` through `from ...networks.modules import *`
        comp.np = np
    mdm = importlib.import_module("model.mdm_forstyledataset")
    cfg = importlib.import_module("model.cfg_sampler")
    rs = importlib.import_module("diffusion.respace")
    igd = importlib.import_module("diffusion.inpainting_gaussian_diffusion")
    mu = importlib.import_module("utils.model_util")
    du = importlib.import_module("utils.dist_util")
    du.dev = lambda: torch.device("cpu")
    torch.set_num_threads(8)
    model = mg.build_reference_model(mdm, F)
    out = {}

    # ---------------------------------------------------------------- f-4: CompMDMGeneratedDataset
    d100 = mu.create_gaussian_diffusion(mg.args_for(), rs.SpacedDiffusion, "100")
    np.random.seed(5)
    with mg.recorded_noise("gen") as st:
        ds = comp.CompMDMGeneratedDataset(cfg.ClassifierFreeSampleModel(model), d100, Loader(), 2, 3, T, None, scale=2.5)
    out["gen|draws"] = np.array(st["k"])
    out["gen|motions"] = np.stack([d["motion"] for d in ds.generated_motion]).astype(np.float32)          # [6, T, F]
    out["gen|lengths"] = np.array([int(d["length"]) for d in ds.generated_motion])
    out["gen|captions"] = np.array([d["caption"] for d in ds.generated_motion])
    out["gen|cap_len"] = np.array([d["cap_len"] for d in ds.generated_motion])
    out["gen|mm_captions"] = np.array([d["caption"] for d in ds.mm_generated_motion])
    out["gen|mm_motions"] = np.stack([np.stack([r["motion"] for r in d["mm_motions"]]) for d in ds.mm_generated_motion]).astype(np.float32)
    out["gen|len"] = np.array(len(ds))
    print("generated", out["gen|motions"].shape, "mm", out["gen|mm_motions"].shape, "draws", int(out["gen|draws"]))

    # ---------------------------------------------------------------- f-2: neutralisation pre-pass, all 100 x0-hats
    d_full = mu.create_gaussian_diffusion(mg.args_for(), igd.InpaintingGaussianDiffusion, "")
    motion = torch.from_numpy(syn.normal(SEED, "xia/motion", (2, F, 1, T)))[:1]
    shp = (1, F, 1, T)
    y_n = {"y": {"text": ["a person walks proudly"], "mask": torch.ones(1, 1, 1, T),
                 "inpainting_mask": torch.zeros(shp), "inpainted_motion": motion}}
    with torch.no_grad(), mg.recorded_noise("xia/neutral900"):
        dump = d_full.p_sample_loop(model.motion_enc.mdm_model, shp, clip_denoised=False, model_kwargs=y_n, skip_timesteps=0,
                                    init_image=motion, progress=False, dump_steps=None, noise=None, const_noise=False,
                                    stop_timesteps=900, dump_all_xstart=True)
    assert len(dump) == 100
    allx = torch.cat(dump).numpy().reshape(100, -1)
    out["neutral900|n"] = np.array(len(dump))
    out["neutral900|sel"] = np.array([0, 9, 49, 99])
    out["neutral900|xstart_sel"] = allx[[0, 9, 49, 99]].reshape(4, F, 1, T)
    out["neutral900|norm"] = np.linalg.norm(allx.astype(np.float64), axis=1)
    out["neutral900|proj"] = (allx.astype(np.float64) @ projection().astype(np.float64).T).astype(np.float32)
    # ---------------------------------------------------------------- eps- / previous-x-predicting models (gaussian_diffusion.py:398-412)
    gd = importlib.import_module("diffusion.gaussian_diffusion")
    fake = torch.from_numpy(syn.normal(SEED, "fake/out", (2, F, 1, T)))
    xx = torch.from_numpy(syn.normal(SEED, "xia/x", (2, F, 1, T)))
    tt = torch.tensor([0, 19])

    class Fake(torch.nn.Module):
        def forward(self, x_, ts, **kw):
            return fake

    for tag, mean_type in (("eps", gd.ModelMeanType.EPSILON), ("prevx", gd.ModelMeanType.PREVIOUS_X)):
        d = rs.SpacedDiffusion(use_timesteps=rs.space_timesteps(1000, "ddim20"), betas=gd.get_named_beta_schedule("cosine", 1000),
                               model_mean_type=mean_type, model_var_type=gd.ModelVarType.FIXED_SMALL, loss_type=gd.LossType.MSE)
        with torch.no_grad(), mg.recorded_noise(f"mt/{tag}/p"):
            r = d.p_sample(Fake(), xx, tt, clip_denoised=False, model_kwargs={"y": {}})
        out[f"{tag}|p_sample|sample"], out[f"{tag}|p_sample|pred_xstart"] = r["sample"].numpy(), r["pred_xstart"].numpy()
        if mean_type == gd.ModelMeanType.EPSILON:
            with torch.no_grad(), mg.recorded_noise(f"mt/{tag}/d"):
                r = d.ddim_sample(Fake(), xx, tt, clip_denoised=False, model_kwargs={"y": {}}, eta=0.5)
            out[f"{tag}|ddim_sample|sample"] = r["sample"].numpy()
    np.savez_compressed(os.path.join(HERE, "gen.npz"), **out)
    print("gen.npz", os.path.getsize(os.path.join(HERE, "gen.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
