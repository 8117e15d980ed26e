"""The resident-group trunk (csrc/mst_trunk.h): a sampling step's encoder stack (reference model/mdm_forstyledataset.py:539-546,
602-625: 8 x [self-attention block, feed-forward block]) as ONE launch in which the four workgroups of a clip hand their phase
outputs to each other through a per-clip arrival counter, against the two-launches-per-layer path.  The arithmetic is the same
code instruction for instruction, so every comparison here is BITWISE; parity of that path against the oracle and the reference's
goldens is what tests/test_gpu_parity.py / test_gpu_bench_path.py hold.  After every run `trunk_check()` must report that no
hand-off wait gave up."""
import numpy as np
import pytest
import torch

import mst_amd  # noqa: F401
from mst_amd import synthetic as syn
from conftest import SEED

pytestmark = pytest.mark.gpu
F, T = 263, 196


def dev():
    return torch.device("cuda:0")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


@pytest.fixture(scope="module")
def big():
    from mst_amd.engine import DenoiserEngine, Schedule
    from oracle import schedule
    eng = DenoiserEngine(F, T, 256, device=dev())
    w = syn.denoiser_state(SEED, F)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
    tab, tmap = schedule.make("cosine", 1000, "")
    return eng, Schedule(tab, tmap, dev())


def _both(eng, fn):
    eng.set_trunk_groups(False)
    a = fn()
    eng.set_trunk_groups(True)
    try:
        b = fn()
        torch.cuda.synchronize()
        eng.trunk_check()
    finally:
        eng.set_trunk_groups(False)
    return a, b


@pytest.mark.parametrize("B", [64, 48, 128, 1, 5, 12, 37])
def test_forward_is_the_two_kernel_path(big, B):
    """One model call: 64 clips = 256 resident workgroups (the headline), 128 clips = every group walks two clips, fewer clips than
    groups (grids of 48, 148, 192 workgroups: the group / member map of incomplete 32-block chunks).  Bitwise at every size: the
    resident launch runs 64- and 48-token tail tiles, a lone two-kernel launch of 12 or 37 clips 32-token tiles -- the three
    instantiations agree bit for bit (every contraction-sensitive multiply-add of the tail is spelled as an fma).  1 and 5 clips stay
    on the small-tile path whatever the switch says."""
    eng, _ = big
    x = cu(syn.normal(SEED, f"trunk/x{B}", (B, F, 1, T)))
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(B)).to(dev())
    eng.set_text(cu(syn.normal(SEED, f"trunk/txt{B}", (B, 512))))
    a, b = _both(eng, lambda: eng.forward(x, t).clone())
    assert torch.isfinite(a).all()
    assert torch.equal(a, b), float((a - b).abs().max())


def test_repeated_calls_and_loops_are_bitwise_the_two_kernel_path(big):
    """The counters run on from launch to launch (64 arrivals per clip and launch): 30 forward calls in a row, then a 12-step DDPM loop
    with recorded noise and one with in-kernel Philox noise, then classifier-free guidance (64 clips = 128 rows: two clips per group)."""
    from mst_amd.engine import SAMPLER_DDPM, SAMPLER_DDIM
    eng, sch = big
    B = 64
    x = cu(syn.normal(SEED, "trunk/loop/x", (B, F, 1, T)))
    t = torch.full((B,), 321, device=dev(), dtype=torch.long)
    txt = cu(syn.normal(SEED, "trunk/loop/txt", (B, 512)))
    eng.set_text(txt)
    eng.set_trunk_groups(False)
    ref = eng.forward(x, t).clone()
    eng.set_trunk_groups(True)
    try:
        for _ in range(30):
            assert torch.equal(eng.forward(x, t), ref)
        torch.cuda.synchronize()
        eng.trunk_check()
    finally:
        eng.set_trunk_groups(False)
    mask = cu(syn.root_horizontal_mask(B, F, T))
    motion = cu(syn.normal(SEED, "trunk/loop/motion", (B, F, 1, T)))
    nz = cu(np.random.default_rng(SEED).standard_normal((12, B, F, 1, T), dtype=np.float32))
    a, b = _both(eng, lambda: eng.sample_loop(sch, x.clone(), 11, 0, SAMPLER_DDPM, mask=mask, motion=motion, noise=nz).clone())
    assert torch.equal(a, b) and torch.equal(a[:, :3], motion[:, :3])
    a, b = _both(eng, lambda: eng.sample_loop(sch, x.clone(), 499, 480, SAMPLER_DDIM, mask=mask, motion=motion, seed=7).clone())
    assert torch.equal(a, b)
    # classifier-free guidance: cond + uncond twins = 128 rows through the stack
    scale = torch.full((B,), 2.5, device=dev())
    eng.set_text(txt, cfg=True)
    a, b = _both(eng, lambda: eng.sample_loop(sch, x.clone(), 5, 0, SAMPLER_DDPM, cfg=True, mask=mask, motion=motion, seed=3, scale=scale).clone())
    assert torch.equal(a, b)
    eng.set_text(txt)


def test_the_resident_launch_is_what_ran(big):
    """With the switch on, the 64-clip loop runs as ONE slice (every clip is a chain of its own inside the launch) and the profile of an
    instrumented step still shows the two kernels per layer (instrumented steps keep the launches apart)."""
    from mst_amd.engine import SAMPLER_DDPM
    eng, sch = big
    B = 64
    eng.set_text(cu(syn.normal(SEED, "trunk/loop/txt", (B, 512))))
    assert eng.loop_slices(B) == 3
    eng.set_trunk_groups(True)
    try:
        assert eng.loop_slices(B) == 1
    finally:
        eng.set_trunk_groups(False)
