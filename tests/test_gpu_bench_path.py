"""The code path `bench.py` times, held to numbers: BASELINE.json configs[1] (64 clips x (263,1,196)) runs as THREE clip
slices on three streams, and with `eng.profile(True, N)` every N-th step joins the slices into one instrumented 64-clip
launch sequence and forks again, with the frame rows chained from step to step (mst_engine.hip: mst_sample_loop, enqueue_step).
Reference loop being restated: /root/reference/diffusion/gaussian_diffusion.py:775-794 (one p_sample per index, no cross-clip op).

 (i)   profiling on == profiling off, bit for bit (the instrumented steps change the launch geometry, not the arithmetic);
 (ii)  the sliced 64-clip loop == clip-wise runs of one clip of every slice (same large-tile kernels: fp32 summation noise only);
 (iii) the reference's own golden `hml|tail8` (an 8-step DDPM tail of two clips) embedded as rows 0-1 of the 64-clip batch."""
import numpy as np
import pytest
import torch

import mst_amd  # noqa: F401
from mst_amd import synthetic as syn
from conftest import SEED, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-3
F, T, B = 263, 196, 64
PROMPTS = ["a person walks proudly", "an old man jumps"]


def dev():
    return torch.device("cuda:0")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


@pytest.fixture(scope="module")
def big():
    from mst_amd.engine import DenoiserEngine, Schedule
    from oracle import schedule
    eng = DenoiserEngine(F, T, B, device=dev())
    w = syn.denoiser_state(SEED, F)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
    tab, tmap = schedule.make("cosine", 1000, "")
    return eng, Schedule(tab, tmap, dev())


def _batch(nsteps, tag):
    txt = cu(syn.normal(SEED, f"bench/{tag}/txt", (B, 512)))
    x0 = cu(syn.normal(SEED, f"bench/{tag}/x", (B, F, 1, T)))
    mask = cu(syn.root_horizontal_mask(B, F, T))
    motion = cu(syn.normal(SEED, f"bench/{tag}/motion", (B, F, 1, T)))
    rng = np.random.default_rng(SEED + nsteps)
    nz = cu(rng.standard_normal((nsteps, B, F, 1, T), dtype=np.float32))
    return txt, x0, mask, motion, nz


def test_profiled_loop_equals_plain_loop_bitwise(big):
    """bench.py turns profiling on (every 50th step instrumented); here every 5th of 12 steps, i.e. steps 0, 5, 10 run as one
    64-clip slice between three-slice steps (join, fork, chained frame rows across the change of geometry)."""
    from mst_amd.engine import SAMPLER_DDPM
    eng, sch = big
    txt, x0, mask, motion, nz = _batch(12, "prof")
    eng.set_text(txt)
    assert eng.loop_slices(B) == 3
    plain = eng.sample_loop(sch, x0.clone(), 11, 0, SAMPLER_DDPM, mask=mask, motion=motion, noise=nz)
    eng.profile(True, 5)
    try:
        prof = eng.sample_loop(sch, x0.clone(), 11, 0, SAMPLER_DDPM, mask=mask, motion=motion, noise=nz)
        torch.cuda.synchronize()
        fams = eng.profile_read()
    finally:
        eng.profile(False)
    assert torch.equal(plain, prof)
    assert torch.equal(plain[:, :3], motion[:, :3])
    launched = {k: v[1] for k, v in fams.items() if v[1]}
    assert launched.get("layer_tail_fused") == 3 * 8 and launched.get("qkv_attention_fused") == 3 * 8, launched
    # the same with in-kernel Philox noise (the bench's noise source) and the seed fixed
    a = eng.sample_loop(sch, x0.clone(), 11, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=5)
    eng.profile(True, 5)
    try:
        b = eng.sample_loop(sch, x0.clone(), 11, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=5)
    finally:
        eng.profile(False)
    assert torch.equal(a, b)


def test_three_slice_loop_equals_clipwise_runs(big, monkeypatch):
    """One clip of every slice (0, 22, 63), run alone through the SAME large-tile kernels (MST_SMALL_M=0 on a second engine), must
    reproduce its rows of the three-slice 64-clip loop up to fp32 summation order."""
    from mst_amd.engine import DenoiserEngine, SAMPLER_DDPM
    eng, sch = big
    txt, x0, mask, motion, nz = _batch(6, "slice")
    eng.set_text(txt)
    assert eng.loop_slices(B) == 3
    full = eng.sample_loop(sch, x0.clone(), 5, 0, SAMPLER_DDPM, mask=mask, motion=motion, noise=nz)
    monkeypatch.setenv("MST_SMALL_M", "0")
    one = DenoiserEngine(F, T, 2, device=dev())
    w = syn.denoiser_state(SEED, F)
    one.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
    for i in (0, 22, 63):
        one.set_text(txt[i:i + 1])
        alone = one.sample_loop(sch, x0[i:i + 1].clone(), 5, 0, SAMPLER_DDPM, mask=mask[i:i + 1], motion=motion[i:i + 1],
                                noise=nz[:, i:i + 1].contiguous())
        e = rel_l2(alone.cpu().numpy(), full[i:i + 1].cpu().numpy())
        print("clip", i, e)
        assert e < 1e-6, (i, e)


def test_golden_clips_inside_the_headline_batch(big, golden):
    """The reference's `hml|tail8` (tests/golden/make_golden.py: p_sample_loop, skip_timesteps=992, two clips, recorded noise)
    as rows 0-1 of a 64-clip batch through the three-slice loop, with and without the instrumented steps."""
    from mst_amd.engine import SAMPLER_DDPM
    eng, sch = big
    g = golden["denoise"]["hml|tail8|sample"]
    shape2 = (2, F, 1, T)
    txt, x0, mask, motion, nz = _batch(9, "gold")
    txt[:2] = cu(np.stack([syn.normal(SEED, "text/" + p, (512,)) for p in PROMPTS]))
    motion[:2] = cu(syn.normal(SEED, "hml/motion", shape2))
    nz2 = np.stack([syn.normal(SEED, f"hml/tail8/noise/{k}", shape2) for k in range(9)])
    nz[:, :2] = cu(nz2)
    eng.set_text(txt)
    x7 = sch.q_sample(motion, torch.full((B,), 7, dtype=torch.int64, device=dev()), nz[0].contiguous(), mask)
    out = eng.sample_loop(sch, x7.clone(), 7, 0, SAMPLER_DDPM, mask=mask, motion=motion, noise=nz[1:].contiguous())
    e = rel_l2(out[:2].cpu().numpy(), g)
    print("tail8 inside batch 64", e)
    assert e < TOL
    eng.profile(True, 3)
    try:
        prof = eng.sample_loop(sch, x7.clone(), 7, 0, SAMPLER_DDPM, mask=mask, motion=motion, noise=nz[1:].contiguous())
    finally:
        eng.profile(False)
    assert torch.equal(out, prof)
    assert torch.equal(out[:, :3], motion[:, :3])
