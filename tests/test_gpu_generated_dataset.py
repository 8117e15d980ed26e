"""the batched-generation caller (`CompMDMGeneratedDataset`,
data_loaders/humanml/motion_loaders/comp_v6_model_dataset.py:146-240 of the reference) on top of the native CFG
sampling loop: batch bookkeeping, multimodality repeats, the `scale` it adds to `y`, equality with a direct
`p_sample_loop` call under the same torch seed, and the rank-sharded walk."""
import types

import numpy as np
import pytest
import torch

import mst_amd  # noqa: F401
import mst_amd.synthetic as syn
from conftest import SEED
from test_gpu_boundary import F, T, build, dev, make_args

pytestmark = pytest.mark.gpu
BS, NB = 2, 3


class Loader:
    """A text-to-motion dataloader stand-in: NB batches of BS clips with the y-dict keys the reference's collate makes."""
    batch_size = BS

    def __init__(self):
        self.dataset = types.SimpleNamespace(mode="gt", __len__=lambda: NB * BS)
        self.dataset = list(range(NB * BS))

    def __len__(self):
        return NB

    def __iter__(self):
        for i in range(NB):
            motion = torch.from_numpy(syn.normal(SEED, f"gen/motion/{i}", (BS, F, 1, T)))
            texts = [f"clip {i} {b}" for b in range(BS)]
            yield motion, {"y": {"text": texts, "tokens": ["a/DET_person/NOUN_walks/VERB"] * BS,
                                 "text_embed": torch.from_numpy(syn.normal(SEED, f"gen/emb/{i}", (BS, 512))),
                                 "lengths": torch.tensor([T, T - 7]), "mask": torch.ones(BS, 1, 1, T)}}


_PLAIN = {}


def plain100():
    """A non-inpainting 100-step process (the evaluation scripts sample from a plain SpacedDiffusion; an
    InpaintingGaussianDiffusion demands y['inpainting_mask'], as in the reference)."""
    if "d" not in _PLAIN:
        from mst_amd.diffusion.respace import SpacedDiffusion
        from mst_amd.utils import model_util
        _PLAIN["d"] = model_util.create_gaussian_diffusion(make_args(), SpacedDiffusion, "100")
    return _PLAIN["d"]


def make(shard=None, mm=2, limit=None):
    from mst_amd.data_loaders.comp_v6_model_dataset import CompMDMGeneratedDataset
    from mst_amd.model.cfg_sampler import ClassifierFreeSampleModel
    c = build()
    np.random.seed(5)
    torch.manual_seed(11)
    return CompMDMGeneratedDataset(ClassifierFreeSampleModel(c["m"]), plain100(), Loader(), mm, 3, T, limit, scale=2.5, shard=shard)


def test_batches_repeats_and_direct_call_equality():
    from mst_amd.model.cfg_sampler import ClassifierFreeSampleModel
    ds = make()
    assert len(ds) == NB * BS and [d["batch"] for d in ds.generated_motion] == [0, 0, 1, 1, 2, 2]
    assert all(d["motion"].shape == (T, F) and d["cap_len"] == 3 for d in ds.generated_motion)
    np.random.seed(5)
    mm_idxs = np.sort(np.random.choice(NB, 2 // BS + 1, replace=False))
    assert sorted({d["batch"] for d in ds.mm_generated_motion}) == list(mm_idxs)
    assert len(ds.mm_generated_motion) == len(mm_idxs) * BS
    for d in ds.mm_generated_motion:
        assert len(d["mm_motions"]) == 3
        assert not np.array_equal(d["mm_motions"][0]["motion"], d["mm_motions"][1]["motion"])       # fresh noise per repeat
    # the first repeat of a multimodality batch IS the batch's generated clip
    first = ds.mm_generated_motion[0]
    ref = [d for d in ds.generated_motion if d["batch"] == first["batch"]][0]
    assert np.array_equal(first["mm_motions"][0]["motion"], ref["motion"])
    # batch 0 equals a direct call with the same torch seed and the scale the dataset adds
    c = build()
    motion, kw = next(iter(Loader()))
    kw["y"] = {k: v.to(dev()) if torch.is_tensor(v) else v for k, v in kw["y"].items()}
    kw["y"]["scale"] = torch.ones(BS, device=dev()) * 2.5
    torch.manual_seed(11)
    with torch.no_grad():
        direct = plain100().p_sample_loop(ClassifierFreeSampleModel(c["m"]), motion.shape, clip_denoised=False, model_kwargs=kw)
    assert np.array_equal(direct.squeeze(2).permute(0, 2, 1).cpu().numpy()[1], ds.generated_motion[1]["motion"])
    # same seeds -> same dataset
    again = make()
    assert all(np.array_equal(a["motion"], b["motion"]) for a, b in zip(ds.generated_motion, again.generated_motion))


def test_sample_limit_and_sharded_walk():
    ds = make(mm=0, limit=3)                        # limit checked before each batch: 2 batches -> 4 clips (reference semantics)
    assert len(ds) == 4 and ds.mm_generated_motion == []
    parts = [make(shard=(r, 2), mm=0) for r in range(2)]
    assert [d["batch"] for d in parts[0].generated_motion] == [0, 0, 2, 2]
    assert [d["batch"] for d in parts[1].generated_motion] == [1, 1]
    assert parts[0].all_generated.__self__.shard == (0, 2)


class GoldenLoader(Loader):
    """The loader tests/golden/make_golden_gen.py fed the REFERENCE's CompMDMGeneratedDataset: prompts only (the models
    encode them), no precomputed embedding."""

    def __iter__(self):
        for motion, kw in Loader.__iter__(self):
            del kw["y"]["text_embed"]
            yield motion, kw


def test_generated_dataset_matches_the_reference_class():
    """SURVEY 8 f-4 pinned to the reference: its `CompMDMGeneratedDataset` (comp_v6_model_dataset.py:146-240) was run on
    CPU over this loader with recorded noise (tests/golden/gen.npz); ours must produce the same clips, multimodality
    repeats, bookkeeping -- and consume the same number of RNG draws in the same order."""
    import os
    from conftest import GOLDEN, rel_l2
    from mst_amd.data_loaders.comp_v6_model_dataset import CompMDMGeneratedDataset
    from mst_amd.model.cfg_sampler import ClassifierFreeSampleModel
    from test_gpu_boundary import recorded_noise
    g = np.load(os.path.join(GOLDEN, "gen.npz"))
    c = build()
    np.random.seed(5)
    with recorded_noise("gen") as st:
        ds = CompMDMGeneratedDataset(ClassifierFreeSampleModel(c["m"]), plain100(), GoldenLoader(), 2, 3, T, None, scale=2.5)
    assert st["k"] == int(g["gen|draws"])                                   # same RNG call pattern as the reference
    assert len(ds) == int(g["gen|len"])
    assert [d["caption"] for d in ds.generated_motion] == list(g["gen|captions"])
    assert [int(d["length"]) for d in ds.generated_motion] == list(g["gen|lengths"])
    assert [d["cap_len"] for d in ds.generated_motion] == list(g["gen|cap_len"])
    assert [d["caption"] for d in ds.mm_generated_motion] == list(g["gen|mm_captions"])
    got = np.stack([d["motion"] for d in ds.generated_motion])
    errs = [rel_l2(got[i], g["gen|motions"][i]) for i in range(len(got))]
    mm = np.stack([np.stack([r["motion"] for r in d["mm_motions"]]) for d in ds.mm_generated_motion])
    assert mm.shape == g["gen|mm_motions"].shape
    mm_err = rel_l2(mm, g["gen|mm_motions"])
    print("generated dataset vs reference: per-clip", errs, "mm", mm_err)
    assert max(errs) < 1e-3 and mm_err < 1e-3
