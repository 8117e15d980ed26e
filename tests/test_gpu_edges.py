"""Edge shapes through the C ABI against the oracle: ragged frame counts (T % 4 != 0 takes the scalar
step-epilogue path), other feature widths (bandai 190, SMPL 150), batch 1 / odd batches split
over the two concurrent slices, the longest supported clip, DDIM with eta > 0 inside a loop,
clipping, and Philox-in-kernel vs the same numbers injected as a buffer."""
import numpy as np
import pytest
import torch

import mst_amd  # noqa: F401
import mst_amd.synthetic as syn
from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-3
SEED = 77


def dev():
    return torch.device("cuda:0")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def make(F, T, rows):
    from mst_amd.engine import DenoiserEngine
    eng = DenoiserEngine(F, T, rows, device=dev())
    w = syn.denoiser_state(SEED, F)
    pe = syn.positional_table(5000, 512)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(pe))
    return eng, w, pe


@pytest.mark.parametrize("F,T,B", [(190, 75, 3), (150, 61, 2), (263, 223, 1), (181, 5, 2), (24, 1, 2)])
def test_forward_and_loop_on_ragged_shapes(F, T, B):
    from mst_amd.engine import Schedule, SAMPLER_DDPM, SAMPLER_DDIM
    from oracle import denoiser, diffusion, schedule
    eng, w, pe = make(F, T, 2 * B)
    shape = (B, F, 1, T)
    x = syn.normal(SEED, "x", shape)
    txt = syn.normal(SEED, "txt", (B, 512))
    t = np.array([0, 999, 431][:B])
    eng.set_text(cu(txt))
    out = eng.forward(cu(x), cu(t)).cpu().numpy()
    e_fwd = rel_l2(out, denoiser.forward(w, pe, x, t, txt).numpy())
    assert e_fwd < TOL
    # CFG on the same shape
    eng.set_text(cu(txt), cfg=True)
    sc = np.linspace(1.5, 2.5, B).astype(np.float32)
    out = eng.forward(cu(x), cu(t), scale=cu(sc), cfg=True).cpu().numpy()
    e_cfg = rel_l2(out, denoiser.cfg_forward(w, pe, x, t, txt, sc).numpy())
    print(f"ragged ({F},{T},{B}): forward {e_fwd:.3e}, cfg {e_cfg:.3e}")
    assert e_cfg < TOL
    if T <= 16:                      # clips of <= 16 frames multiply every activation as hi + lo (engine: `precise`): margin, not luck, under the bar
        assert e_cfg < 8e-4
    # a short loop per sampler, with a mask that is not the root pattern (every third feature, frames 0..T/2)
    mask = np.zeros(shape, np.float32)
    mask[:, ::3, :, : max(1, T // 2)] = 1
    motion = syn.normal(SEED, "motion", shape)
    tab, tmap = schedule.make("cosine", 1000, "ddim20")
    sch = Schedule(tab, tmap, dev())
    eng.set_text(cu(txt))
    for sampler, name, eta in ((SAMPLER_DDPM, "ddpm", 0.0), (SAMPLER_DDIM, "ddim", 0.3)):
        nz = np.stack([syn.normal(SEED, f"nz{k}", shape) for k in range(5)])
        x4 = sch.q_sample(cu(motion), cu(np.full(B, 3)), cu(nz[0]), cu(mask))
        got, dump = eng.sample_loop(sch, x4, 3, 0, sampler, eta, mask=cu(mask), motion=cu(motion), noise=cu(nz[1:]), dump_xstart=True)
        ref = diffusion.sample_loop(lambda xx, tt: denoiser.forward(w, pe, xx, tt, txt), tab, tmap, shape,
                                    lambda k: torch.from_numpy(nz[k]), name, True, mask, motion, init_image=motion,
                                    skip_timesteps=16, eta=eta, dump_all_xstart=True)
        assert rel_l2(dump.cpu().numpy(), torch.stack(ref).numpy()) < TOL, (name, F, T)
        m = mask.astype(bool)
        assert np.array_equal(dump[-1].cpu().numpy()[m], motion[m])        # masked entries bit-exact
        assert np.array_equal(got.cpu().numpy()[m], motion[m])


@pytest.mark.parametrize("T", [15, 40, 100, 127, 150, 191, 196, 207, 208, 223])
def test_fused_qkv_attention_every_tile_count(T, monkeypatch):
    """The large-tile path's fused QKV + attention kernel is instantiated per token-tile count (16-token tiles: 2, 4, .. 12, 13 for
    S = T + 1 <= 208; S = 209..224 falls back to round 2's kernel, whose 32-row images fit those): one forward per instantiation
    against the oracle, with the small-tile path switched off so that two clips take it."""
    from oracle import denoiser
    monkeypatch.setenv("MST_SMALL_M", "0")
    F, B = 263, 2
    eng, w, pe = make(F, T, B)
    x = syn.normal(SEED, f"xq{T}", (B, F, 1, T))
    txt = syn.normal(SEED, "txtq", (B, 512))
    t = np.array([17, 803])
    eng.set_text(cu(txt))
    out = eng.forward(cu(x), cu(t)).cpu().numpy()
    err = rel_l2(out, denoiser.forward(w, pe, x, t, txt).numpy())
    print(f"fused QKV+attention at S = {T + 1}: {err:.3e}")
    assert err < TOL


def test_precise_mode_forward_cfg_and_loop():
    """Precise mode (hi + lo activations AND weights in every GEMM of the sampling path, any batch size on the small-tile kernels):
    forward, classifier-free guidance and a short DDPM / DDIM loop against the oracle -- with both operands split the error is the
    fp32 summation order plus the f16 roundings inside attention, well under the default path's 4e-4."""
    from mst_amd.engine import Schedule, SAMPLER_DDPM, SAMPLER_DDIM
    from oracle import denoiser, diffusion, schedule
    F, T, B = 263, 60, 3
    eng, w, pe = make(F, T, 2 * B)
    eng.set_precise(True)
    shape = (B, F, 1, T)
    x = syn.normal(SEED, "xp", shape)
    txt = syn.normal(SEED, "txtp", (B, 512))
    t = np.array([3, 998, 512])
    eng.set_text(cu(txt))
    e_fwd = rel_l2(eng.forward(cu(x), cu(t)).cpu().numpy(), denoiser.forward(w, pe, x, t, txt).numpy())
    eng.set_text(cu(txt), cfg=True)
    sc = np.array([1.5, 2.0, 2.5], np.float32)
    e_cfg = rel_l2(eng.forward(cu(x), cu(t), scale=cu(sc), cfg=True).cpu().numpy(), denoiser.cfg_forward(w, pe, x, t, txt, sc).numpy())
    print(f"precise mode: forward {e_fwd:.3e}, cfg {e_cfg:.3e}")
    assert e_fwd < 1e-4 and e_cfg < 2e-4                 # measured 4.7e-5 / 1.0e-4 (default path: 4.0e-4 / 5.9e-4)
    mask = syn.root_horizontal_mask(B, F, T)
    motion = syn.normal(SEED, "motionp", shape)
    tab, tmap = schedule.make("cosine", 1000, "ddim20")
    sch = Schedule(tab, tmap, dev())
    eng.set_text(cu(txt))
    for sampler, name, eta in ((SAMPLER_DDPM, "ddpm", 0.0), (SAMPLER_DDIM, "ddim", 0.3)):
        nz = np.stack([syn.normal(SEED, f"nzp{k}", shape) for k in range(5)])
        x4 = sch.q_sample(cu(motion), cu(np.full(B, 3)), cu(nz[0]), cu(mask))
        got, dump = eng.sample_loop(sch, x4, 3, 0, sampler, eta, mask=cu(mask), motion=cu(motion), noise=cu(nz[1:]), dump_xstart=True)
        ref = diffusion.sample_loop(lambda xx, tt: denoiser.forward(w, pe, xx, tt, txt), tab, tmap, shape,
                                    lambda k: torch.from_numpy(nz[k]), name, True, mask, motion, init_image=motion,
                                    skip_timesteps=16, eta=eta, dump_all_xstart=True)
        err = rel_l2(dump.cpu().numpy(), torch.stack(ref).numpy())
        print(f"precise mode {name} loop: {err:.3e}")
        assert err < 1e-4, (name, err)                    # measured 4.3e-5 / 4.4e-5
        m = mask.astype(bool)
        assert np.array_equal(got.cpu().numpy()[m], motion[m])            # masked entries bit-exact
    eng.set_precise(False)


def test_odd_batch_across_the_two_slices_equals_single_slice():
    """Batch 17 is split 9 + 8 over two streams (8-clip minimum per slice); results must not depend on the split."""
    from mst_amd.engine import Schedule, SAMPLER_DDPM
    from oracle import schedule
    F, T, B = 181, 76, 17
    eng, w, pe = make(F, T, B)
    # 17 clips x 77 tokens take the small-tile path: sliced (8-clip minimum); 64 clips = 4928 rows are past the hand-over (1900 rows:
    # the large-tile step wins from 10 full-length clips on) and fit the chip at once: one slice,
    # as does their CFG batch (9856 rows, 154 large tiles)
    assert eng.loop_slices(B) == 2 and eng.loop_slices(8) == 1 and eng.loop_slices(64) == 1 and eng.loop_slices(64, cfg=True) == 1
    shape = (B, F, 1, T)
    tab, tmap = schedule.make("cosine", 1000, "")
    sch = Schedule(tab, tmap, dev())
    txt = cu(syn.normal(SEED, "txt17", (B, 512)))
    x0 = cu(syn.normal(SEED, "x17", shape))
    mask = cu(syn.root_horizontal_mask(B, F, T))
    motion = cu(syn.normal(SEED, "m17", shape))
    eng.set_text(txt)
    whole = eng.sample_loop(sch, x0.clone(), 6, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=5)
    # the same clips one at a time (single slice): Philox counters carry the global clip index only
    # through the slice offset, so compare the buffer-noise path instead
    nz = cu(np.stack([syn.normal(SEED, f"nz17/{k}", shape) for k in range(7)]))
    a = eng.sample_loop(sch, x0.clone(), 6, 0, SAMPLER_DDPM, mask=mask, motion=motion, noise=nz)
    parts = []
    for lo, hi in ((0, 5), (5, 17)):
        eng.set_text(txt[lo:hi])
        parts.append(eng.sample_loop(sch, x0[lo:hi].clone(), 6, 0, SAMPLER_DDPM, mask=mask[lo:hi], motion=motion[lo:hi],
                                     noise=nz[:, lo:hi].contiguous()))
    assert rel_l2(torch.cat(parts).cpu().numpy(), a.cpu().numpy()) < 1e-6
    assert torch.isfinite(whole).all() and torch.equal(whole[:, :3], motion[:, :3])


def test_philox_in_kernel_equals_the_same_numbers_injected(monkeypatch):
    from mst_amd.engine import Schedule, SAMPLER_DDPM
    from oracle import schedule
    F, T, B = 263, 196, 18          # 2 slices of 9: also checks the slice offset in the Philox counter
    monkeypatch.setenv("MST_STREAMS", "3")          # (56 large tiles would run as one slice by default)
    eng, w, pe = make(F, T, B)
    assert eng.loop_slices(B) == 2
    tab, tmap = schedule.make("cosine", 1000, "")
    sch = Schedule(tab, tmap, dev())
    eng.set_text(cu(syn.normal(SEED, "txtp", (B, 512))))
    x0 = cu(syn.normal(SEED, "xp", (B, F, 1, T)))
    a = eng.sample_loop(sch, x0.clone(), 4, 0, SAMPLER_DDPM, mask_noise=False, seed=1234)
    nz = torch.stack([eng.philox_normal(B, T, 1234, j) for j in range(5)])
    b = eng.sample_loop(sch, x0.clone(), 4, 0, SAMPLER_DDPM, mask_noise=False, noise=nz)
    assert torch.equal(a, b)
    n = nz.flatten()
    assert abs(float(n.mean())) < 2e-3 and abs(float(n.std()) - 1) < 2e-3
    assert abs(float((n[:-1] * n[1:]).mean())) < 2e-3                     # neighbouring elements uncorrelated


def test_chained_frame_rows_equal_the_transpose_kernel(monkeypatch):
    """Host-enqueued loops let step j's epilogue write the f16 frame rows step j + 1 embeds (k_frames_f16 then runs for step 0
    only); MST_FUSE_FRAMES=0 runs the transpose kernel every step.  The rows are the same conversion of the same numbers:
    bit-identical loops, for both samplers, with inpainting, in-kernel noise and the x0-hat dump, over 3 concurrent slices."""
    from mst_amd.engine import Schedule, SAMPLER_DDPM, SAMPLER_DDIM
    from oracle import schedule
    F, T, B = 263, 196, 25
    tab, tmap = schedule.make("cosine", 1000, "")
    sch = Schedule(tab, tmap, dev())
    txt = cu(syn.normal(SEED, "txtc", (B, 512)))
    x0 = cu(syn.normal(SEED, "xc", (B, F, 1, T)))
    mask = cu(syn.root_horizontal_mask(B, F, T))
    motion = cu(syn.normal(SEED, "mc", (B, F, 1, T)))
    monkeypatch.setenv("MST_STREAMS", "3")
    monkeypatch.setenv("MST_FUSE_FRAMES", "1")
    chained, _, _ = make(F, T, B)
    assert chained.loop_slices(B) == 3
    monkeypatch.setenv("MST_FUSE_FRAMES", "0")
    plain, _, _ = make(F, T, B)
    outs = []
    for eng in (chained, plain):
        eng.set_text(txt)
        a = eng.sample_loop(sch, x0.clone(), 6, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=5)
        b, xs = eng.sample_loop(sch, x0.clone(), 999, 993, SAMPLER_DDIM, eta=0.5, seed=6, dump_xstart=True)
        outs.append((a, b, xs))
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u, v)
    assert torch.equal(outs[0][0][:, :3], motion[:, :3])


def test_small_launch_gemms_resident_tile_equals_slab_ring(monkeypatch):
    """A clip or two (<= MST_SMALL_M stream rows) runs the four layer GEMMs as k_rows_gemm (mst_small.h: token tile resident in
    LDS, weights streamed as fragments); MST_SMALL_FAST=0 keeps the round-1..3 slab ring.  Same f16 operands and fp32 accumulation,
    another order of the k sum: the two agree far inside the 1e-3 bar, forward and through a loop, and re-uploaded weights reach the
    packed fragments (weights of another seed -> another result, equal to the ring's again)."""
    from mst_amd.engine import DenoiserEngine, Schedule, SAMPLER_DDPM
    from oracle import denoiser, schedule
    F, T, B = 181, 76, 2
    shape = (B, F, 1, T)
    x, txt, t = syn.normal(SEED, "xs", shape), syn.normal(SEED, "ts", (B, 512)), np.array([3, 977])
    tab, tmap = schedule.make("cosine", 1000, "100")
    sch = Schedule(tab, tmap, dev())
    fast, w, pe = make(F, T, B)
    monkeypatch.setenv("MST_SMALL_FAST", "0")
    ring, _, _ = make(F, T, B)
    res = []
    for eng in (fast, ring):
        eng.set_text(cu(txt))
        res.append((eng.forward(cu(x), cu(t)).cpu().numpy(), eng.sample_loop(sch, cu(x), 99, 90, SAMPLER_DDPM, seed=4).cpu().numpy()))
    e_f, e_l = rel_l2(res[0][0], res[1][0]), rel_l2(res[0][1], res[1][1])
    e_o = rel_l2(res[0][0], denoiser.forward(w, pe, x, t, txt).numpy())
    print(f"resident tile vs ring: forward {e_f:.2e}, 10-step loop {e_l:.2e}; vs oracle {e_o:.2e}")
    assert e_f < 5e-4 and e_l < 2e-4 and e_o < TOL
    w2 = syn.denoiser_state(SEED + 1, F)
    for eng in (fast, ring):
        eng.load_state_dict({k: torch.from_numpy(v) for k, v in w2.items()}, pe=torch.from_numpy(pe))
        eng.set_text(cu(txt))
    o_f, o_r = fast.forward(cu(x), cu(t)).cpu().numpy(), ring.forward(cu(x), cu(t)).cpu().numpy()
    assert rel_l2(o_f, o_r) < 5e-4 and rel_l2(o_f, res[0][0]) > 1e-2


@pytest.mark.parametrize("F,T,B", [(181, 76, 1), (263, 196, 2), (190, 75, 5), (150, 61, 3)])
def test_layernorm_inside_the_consuming_gemm_is_bitwise_the_rows_kernel(monkeypatch, F, T, B):
    """Up to 512 stream rows the small-launch path makes each LayerNorm inside the GEMM that consumes it (k_rows_gemm<.., LNF = 1>,
    16-token tiles, the stream alternating between two buffers); MST_SMALL_LN=0 launches k_ln_rows between the GEMMs.  Same per-lane
    arithmetic and the same order of the k sum: bit-identical forwards and loops, also where the last tile is ragged."""
    from mst_amd.engine import Schedule, SAMPLER_DDPM
    from oracle import schedule
    assert B * (T + 1) <= 512
    shape = (B, F, 1, T)
    x, txt = syn.normal(SEED, "xl", shape), syn.normal(SEED, "tl", (B, 512))
    t = np.array([5, 400, 998, 77, 650][:B])
    tab, tmap = schedule.make("cosine", 1000, "")
    sch = Schedule(tab, tmap, dev())
    fused, _, _ = make(F, T, B)
    monkeypatch.setenv("MST_SMALL_LN", "0")
    rows, _, _ = make(F, T, B)
    res = []
    for eng in (fused, rows):
        eng.set_text(cu(txt))
        res.append((eng.forward(cu(x), cu(t)), eng.sample_loop(sch, cu(x), 999, 992, SAMPLER_DDPM, seed=9)))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])


def test_small_launch_tile_heights_agree_and_hold_the_oracle():
    """Eight clips of 196 frames (1576 stream rows: 32-token GEMM tiles), their halves (788 rows each: 16-token tiles) and five of them
    (985 rows: 64-token tiles) give the same loop up to fp32 summation order, and the 8-clip forward holds the bar against the oracle."""
    from mst_amd.engine import Schedule, SAMPLER_DDPM
    from oracle import denoiser, schedule
    F, T, B = 263, 196, 8
    eng, w, pe = make(F, T, B)
    assert eng.loop_slices(B) == 1
    shape = (B, F, 1, T)
    x, txt = syn.normal(SEED, "xh", shape), syn.normal(SEED, "th", (B, 512))
    t = np.array([0, 999, 431, 7, 650, 12, 300, 880])
    tab, tmap = schedule.make("cosine", 1000, "")
    sch = Schedule(tab, tmap, dev())
    nz = cu(np.stack([syn.normal(SEED, f"nzh/{k}", shape) for k in range(5)]))
    eng.set_text(cu(txt))
    out = eng.forward(cu(x), cu(t)).cpu().numpy()
    e = rel_l2(out, denoiser.forward(w, pe, x, t, txt).numpy())
    print("8 clips, 32-token tiles vs oracle", e)
    assert e < TOL
    whole = eng.sample_loop(sch, cu(x), 4, 0, SAMPLER_DDPM, noise=nz)
    for lo, hi in ((0, 4), (4, 8), (2, 7)):
        eng.set_text(cu(txt[lo:hi]))
        part = eng.sample_loop(sch, cu(x[lo:hi]), 4, 0, SAMPLER_DDPM, noise=nz[:, lo:hi].contiguous())
        assert rel_l2(part.cpu().numpy(), whole[lo:hi].cpu().numpy()) < 1e-6, (lo, hi)


def test_argument_errors_surface_as_exceptions():
    from mst_amd.engine import DenoiserEngine, Schedule
    from oracle import schedule
    eng, w, pe = make(181, 76, 2)
    x = torch.zeros(3, 181, 1, 76, device=dev())
    with pytest.raises(RuntimeError, match="max_rows"):
        eng.set_text(torch.zeros(3, 512, device=dev()))
    eng.set_text(torch.zeros(2, 512, device=dev()))
    with pytest.raises(RuntimeError, match="mst_set_text"):
        eng.forward(x[:1], torch.zeros(1, dtype=torch.long, device=dev()))     # batch differs from set_text
    with pytest.raises(RuntimeError, match="frames"):
        eng.forward(torch.zeros(2, 181, 1, 80, device=dev()), torch.zeros(2, dtype=torch.long, device=dev()))
    tab, tmap = schedule.make("cosine", 1000, "ddim20")
    sch = Schedule(tab, tmap, dev())
    with pytest.raises(RuntimeError, match="index range"):
        eng.sample_loop(sch, x[:2].clone(), 20, 0, seed=1)
    fresh = DenoiserEngine(181, 76, 2, device=dev())
    fresh.set_text(torch.zeros(2, 512, device=dev()))
    with pytest.raises(RuntimeError, match="tensors loaded"):
        fresh.forward(x[:2], torch.zeros(2, dtype=torch.long, device=dev()))


def test_cfg_loop_in_slices_equals_clipwise_runs():
    """CFG with the batch split over concurrent slices: every clip must come out as when run alone."""
    from mst_amd.engine import Schedule, SAMPLER_DDPM
    from oracle import schedule
    F, T, B = 181, 76, 12                          # 24 rows x 77 tokens = 1848: the small-tile path (hand-over at 1900 rows), three slices of 8 rows
    eng, w, pe = make(F, T, 2 * B)
    assert eng.loop_slices(B, cfg=True) == 3
    shape = (B, F, 1, T)
    tab, tmap = schedule.make("cosine", 1000, "")
    sch = Schedule(tab, tmap, dev())
    txt = cu(syn.normal(SEED, "txtc", (B, 512)))
    x0 = cu(syn.normal(SEED, "xc", shape))
    mask = cu(syn.root_horizontal_mask(B, F, T))
    motion = cu(syn.normal(SEED, "mc", shape))
    scale = cu(np.linspace(1.0, 3.0, B).astype(np.float32))
    nz = cu(np.stack([syn.normal(SEED, f"nzc/{k}", shape) for k in range(4)]))
    eng.set_text(txt, cfg=True)
    whole = eng.sample_loop(sch, x0.clone(), 3, 0, SAMPLER_DDPM, cfg=True, scale=scale, mask=mask, motion=motion, noise=nz)
    parts = []
    for i in (0, 6, 11):
        eng.set_text(txt[i:i + 1], cfg=True)
        parts.append(eng.sample_loop(sch, x0[i:i + 1].clone(), 3, 0, SAMPLER_DDPM, cfg=True, scale=scale[i:i + 1],
                                     mask=mask[i:i + 1], motion=motion[i:i + 1], noise=nz[:, i:i + 1].contiguous()))
    assert rel_l2(torch.cat(parts).cpu().numpy(), whole[[0, 6, 11]].cpu().numpy()) < 1e-6


def test_graph_replayed_loop_equals_host_enqueued_loop(monkeypatch):
    """MST_GRAPH=1: long loops replay a captured hipGraph of `MST_GRAPH_STEPS` denoise steps (tensors and the step index reach
    the kernels through device memory); the default enqueues every step from the host.  Same kernels, same arithmetic: bit-identical
    results, also when the instantiated graph is re-used by a second call with OTHER tensors and another seed, with recorded
    noise + x0-hat dump (per-step buffers indexed by the device-side step counter), and under CFG."""
    from mst_amd.engine import Schedule, SAMPLER_DDPM, SAMPLER_DDIM
    from oracle import schedule
    F, T, B, NS = 181, 76, 17, 53                  # 2 slices; 53 steps = 13 from the host + 2 replays of 20 (after the warm-up pass)
    tab, tmap = schedule.make("cosine", 1000, "")
    sch = Schedule(tab, tmap, dev())
    shape = (B, F, 1, T)
    txt = cu(syn.normal(SEED, "txtg", (B, 512)))
    mask = cu(syn.root_horizontal_mask(B, F, T))
    scale = cu(np.linspace(1.0, 3.0, B).astype(np.float32))
    monkeypatch.setenv("MST_GRAPH", "0")
    host, _, _ = make(F, T, 2 * B)
    monkeypatch.setenv("MST_GRAPH", "1")
    graph, _, _ = make(F, T, 2 * B)
    for rnd in range(3):                            # round 0 warms + captures, rounds 1-2 re-use the instantiated graph
        x0 = cu(syn.normal(SEED, f"xg/{rnd}", shape))
        motion = cu(syn.normal(SEED, f"mg/{rnd}", shape))
        outs = []
        for eng in (host, graph):
            eng.set_text(txt)
            outs.append(eng.sample_loop(sch, x0.clone(), NS - 1, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=100 + rnd))
        assert torch.equal(outs[0], outs[1]), rnd
        assert torch.equal(outs[1][:, :3], motion[:, :3])
    # recorded noise + x0-hat dump, DDIM with eta (other kernels, other flags -> a new capture)
    nz = cu(np.stack([syn.normal(SEED, f"nzg/{k}", (B, F, 1, T)) for k in range(NS)]))
    res = []
    for eng in (host, graph):
        eng.set_text(txt)
        res.append(eng.sample_loop(sch, x0.clone(), NS - 1, 0, SAMPLER_DDIM, eta=0.3, mask=mask, motion=motion, noise=nz, dump_xstart=True))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    # classifier-free guidance (doubled batch, per-clip scales read through the device block)
    res = []
    for eng in (host, graph):
        eng.set_text(txt, cfg=True)
        res.append(eng.sample_loop(sch, x0.clone(), NS - 1, 0, SAMPLER_DDPM, cfg=True, scale=scale, mask=mask, motion=motion, seed=9))
    assert torch.equal(res[0], res[1])


def test_fused_layer_upload_equals_tensor_by_tensor_upload():
    """mst_load_layers (all 96 layer tensors in one launch: what a fine-tune iteration does after the optimizer step) against 96
    mst_load_weight calls: the sampling forward (packed fragment streams) and the training node (plain [out][in] and [in][out]
    copies) give bit-identical results."""
    from mst_amd.engine import LAYER_TENSORS
    F, T, B = 181, 76, 2
    eng_a, w, pe = make(F, T, B)
    eng_b, _, _ = make(F, T, B)
    w2 = {k: (v * 1.25 + 0.01).astype(np.float32) if k.startswith("seqTransEncoder.layers.") else v for k, v in w.items()}
    layer_names = [f"seqTransEncoder.layers.{i}.{k}" for i in range(8) for k in LAYER_TENSORS]
    for k in layer_names:
        eng_a.load_tensor(k, torch.from_numpy(w2[k]))
    eng_b.load_layers([cu(w2[k]) for k in layer_names])
    x = cu(syn.normal(SEED, "up/x", (B, F, 1, T)))
    t = torch.tensor([10, 900], device=dev())
    txt = cu(syn.normal(SEED, "up/txt", (B, 512)))
    eng_a.set_text(txt)
    eng_b.set_text(txt)
    assert torch.equal(eng_a.forward(x, t), eng_b.forward(x, t))
    oa, ta = eng_a.train_model_forward(x, t, 0.1, 0.1, 4242)
    ob, tb = eng_b.train_model_forward(x, t, 0.1, 0.1, 4242)
    assert torch.equal(oa, ob)
    d = cu(syn.normal(SEED, "up/d", (B, F, 1, T)))
    ga = [torch.zeros(torch.from_numpy(w2[k]).shape, device=dev()) for k in layer_names]
    gb = [torch.zeros_like(g) for g in ga]
    dxa = eng_a.train_model_backward(ta, d, 0.1, 0.1, 4242, ga)
    dxb = eng_b.train_model_backward(tb, d, 0.1, 0.1, 4242, gb)
    assert torch.equal(dxa, dxb) and all(torch.equal(p, q) for p, q in zip(ga, gb))
