"""Round 6's launch diet on the Python side changes no value (docs/LAB_NOTES.md R6.17): the two-launch `mask_cond` draws the same Bernoulli
mask from the same generator state and returns the same tensor as the reference's op sequence (model/mdm_forstyledataset.py:288-296); a cached
constant timestep batch and its cached respacing are what `th.full` + the index would give.  CPU: the code paths are device-agnostic."""
import types

import torch

import mst_amd  # noqa: F401
from mst_amd.diffusion import gaussian_diffusion as gd
from mst_amd.diffusion.respace import _WrappedModel
from mst_amd.model import mdm_forstyledataset as mdm


def test_mask_cond_fast_path_is_the_reference_sequence(monkeypatch):
    mod = types.SimpleNamespace(training=True, cond_mask_prob=0.3)
    cond = torch.randn(64, 512)
    torch.manual_seed(7)
    fast = mdm._mask_cond(mod, cond)
    state_after_fast = torch.get_rng_state()
    monkeypatch.setenv("MST_GLUE_CACHE", "0")
    torch.manual_seed(7)
    ref = mdm._mask_cond(mod, cond)
    assert torch.equal(torch.get_rng_state(), state_after_fast), "the fast path consumed the generator differently"
    assert torch.equal(fast == 0, ref == 0) and torch.allclose(fast, ref, rtol=0, atol=0)
    dropped = (ref.abs().sum(1) == 0).float().mean().item()
    assert 0.1 < dropped < 0.5
    # the mask handed over as drawn (mst_set_text_dropped's input) is the same draw
    monkeypatch.delenv("MST_GLUE_CACHE")
    torch.manual_seed(7)
    drop = mdm._cond_drop(mod, cond)
    assert torch.equal(drop.bool(), ref.abs().sum(1) == 0)
    # eval mode / p = 0: untouched; force_mask: zeros
    mod2 = types.SimpleNamespace(training=False, cond_mask_prob=0.3)
    assert mdm._mask_cond(mod2, cond) is cond
    assert float(mdm._mask_cond(mod, cond, force_mask=True).abs().max()) == 0.0


def test_constant_timesteps_and_their_respacing_from_the_cache(monkeypatch):
    holder = types.SimpleNamespace()
    t = gd.GaussianDiffusion._const_timesteps(holder, 5, 3, "cpu")
    assert torch.equal(t, torch.full((3,), 5, dtype=torch.long)) and t._mst_const == 5
    assert gd.GaussianDiffusion._const_timesteps(holder, 5, 3, "cpu") is t
    assert gd.GaussianDiffusion._const_timesteps(holder, 4, 3, "cpu") is not t
    seen = []
    wrapped = _WrappedModel(lambda x, ts, **kw: seen.append(ts) or x, [0, 50, 100, 150, 200, 250, 300], False, 1000)
    x = torch.zeros(3, 2)
    wrapped(x, t)
    wrapped(x, t)                                           # second call: served from the cache
    wrapped(x, torch.full((3,), 5, dtype=torch.long))       # an untagged tensor: the index path
    assert seen[0] is seen[1] and torch.equal(seen[0], seen[2]) and torch.equal(seen[2], torch.full((3,), 250, dtype=torch.long))
    resc = _WrappedModel(lambda x, ts, **kw: seen.append(ts) or x, [0, 50, 100, 150, 200, 250, 300], True, 1000)
    resc(x, t)
    assert torch.allclose(seen[-1], torch.full((3,), 250.0))
    monkeypatch.setenv("MST_GLUE_CACHE", "0")
    u = gd.GaussianDiffusion._const_timesteps(holder, 5, 3, "cpu")
    assert u is not t and not hasattr(u, "_mst_const") and torch.equal(u, t)
