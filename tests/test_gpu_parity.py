"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs,
and against the golden vectors the reference itself produced (tests/golden/*.npz).

Tolerances: north_star states 1e-3 relative-L2 against the reference fp32 path for floating-point
outputs and bit-exact handling of the inpainting mask.  The engine computes GEMM/attention products
with f16 MFMA operands and fp32 accumulation; measured operand-rounding error is ~5e-4 per forward
(DESIGN.md), so the full-path bar below is the north_star's 1e-3, the elementwise kernels' 2e-6."""
import numpy as np
import pytest
import torch

import mst_amd  # noqa: F401
from mst_amd import synthetic as syn
from conftest import SEED, rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-3          # north_star: 1e-3 relative L2 vs the fp32 reference path
TOL_ELEM = 2e-6     # fp32 elementwise kernels (fma contraction / exp ulp differences only)
PROMPTS = ["a person walks proudly", "an old man jumps"]
SHAPES = {"xia": (181, 76), "hml": (263, 196)}


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


_ENGINES = {}


@pytest.fixture(autouse=True, params=["default", "0", "unfused", "ln128"],
                ids=["small-tiles", "large-tiles", "large-tiles-unfused-tail", "large-tiles-unfused-ln128"])
def tile_path(request, monkeypatch):
    """Every test of this module runs four times: launches of a few clips (all the golden comparisons) take the small-tile
    path by default; MST_SMALL_M=0 sends the same inputs through the batch-64 kernels (fused QKV+attention and the fused
    layer tail); MST_FUSE_TAIL=0 keeps out-proj+LN / FFN1 / FFN2+LN as three launches (the training path's kernels);
    MST_LN128_M=1 additionally takes the optional 128-token LayerNorm tiles on that path."""
    if request.param != "default":
        monkeypatch.setenv("MST_SMALL_M", "0")
    if request.param in ("unfused", "ln128"):
        monkeypatch.setenv("MST_FUSE_TAIL", "0")
    if request.param == "ln128":
        monkeypatch.setenv("MST_LN128_M", "1")
    return request.param


def engine_for(tag, prior=False, max_rows=4):
    import os
    from mst_amd.engine import DenoiserEngine
    key = (tag, prior, max_rows, os.environ.get("MST_SMALL_M"), os.environ.get("MST_LN128_M"), os.environ.get("MST_FUSE_TAIL"))
    if key not in _ENGINES:
        F, T = SHAPES[tag]
        eng = DenoiserEngine(F, T, max_rows, device=_dev())
        lp = "motion_enc.mdm_model.seqTransEncoder.layers." if prior else "seqTransEncoder.layers."
        w = syn.denoiser_state(SEED, F, layer_prefix=lp)
        eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, layer_prefix=lp,
                            pe=torch.from_numpy(syn.positional_table(5000, 512)))
        _ENGINES[key] = (eng, w)
    return _ENGINES[key]


def inputs(tag, B=2):
    F, T = SHAPES[tag]
    x = syn.normal(SEED, f"{tag}/x", (2, F, 1, T))[:B]
    t = np.array([3, 957])[:B]
    txt = np.stack([syn.normal(SEED, "text/" + p, (512,)) for p in PROMPTS])[:B]
    return F, T, x, t, txt


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_dev())


# ------------------------------------------------------------------------------ kernels, stage by stage
@pytest.mark.parametrize("tag", ["hml", "xia"])
def test_layer0_stage_by_stage(tag):
    """Every kernel family of layer 0 against the oracle's intermediates (localises a wrong kernel)."""
    from oracle import denoiser
    eng, w = engine_for(tag)
    F, T, x, t, txt = inputs(tag)
    S, B = T + 1, 2
    pe = syn.positional_table(5000, 512)
    seq = denoiser.token_stream(w, pe, x, t, txt)
    tr = {}
    denoiser.encoder_layer(seq, w, "seqTransEncoder.layers.0.", 4, trace=tr)
    eng.set_text(cu(txt))
    errs = {}
    try:
        for stage, (buf, cols, ref) in enumerate([("hs", 512, seq), ("qkv", 1536, tr["qkv"]), ("att", 512, tr["attn"]),
                                                  ("hs", 512, tr["x1"]), ("hid", 1024, tr["hid"]), ("hs", 512, tr["x2"])]):
            eng.debug_stop_after(0, stage)
            eng.forward(cu(x), cu(t))
            got = eng.debug_buffer(buf, B * S, cols).float().cpu().numpy()
            errs[stage] = rel_l2(got, ref.reshape(B * S, cols).numpy())
    finally:
        eng.debug_stop_after(-1, -1)
    print("stage errors", tag, errs)
    for stage, e in errs.items():
        assert e < TOL, (stage, errs)
    # the stream is kept as an f16 hi/lo pair: hi (the MFMA operand copy) must be the f16 rounding of the
    # stream value, i.e. within half an f16 ulp of hi + lo
    eng.debug_stop_after(0, 5)
    try:
        eng.forward(cu(x), cu(t))
        hs = eng.debug_buffer("hs", B * S, 512)
        hx = eng.debug_buffer("hx", B * S, 512)
    finally:
        eng.debug_stop_after(-1, -1)
    err = (hs - hx.float()).abs()
    assert bool((err <= hs.abs() * 2.0 ** -11 * 1.001 + 2.0 ** -25).all())
    assert float((hs.half() != hx).float().mean()) < 1e-3      # ties aside, it IS the rounding


# ------------------------------------------------------------------------------ model forward
@pytest.mark.parametrize("stress", ["ln_outliers", "big_weights", "everything"])
def test_forward_with_ill_conditioned_weights(stress):
    """The seeded weights are well-conditioned; trained transformers need not be.  With LayerNorm gains carrying x20
    outlier channels, 3x larger FFN / attention weights (peakier softmax), large biases and 5-sigma inputs, rounding the
    MFMA operands to f16 costs more than on the seeded weights -- the oracle's operand-rounding model (same arithmetic in
    torch, operands cast to f16 before every product) says 3.9e-3 / 1.2e-3 / 0.14 relative L2 for the three cases below.
    What this test pins: nothing overflows or goes non-finite, and the engine is no worse than that model predicts, i.e.
    the error is the declared numerics, not an implementation artefact."""
    from mst_amd.engine import DenoiserEngine
    from oracle import denoiser
    F, T, x, t, txt = inputs("xia")
    w = {k: v.copy() for k, v in syn.denoiser_state(SEED, F, layer_prefix="seqTransEncoder.layers.").items()}
    rng = np.random.default_rng(7)
    ln = stress in ("ln_outliers", "everything")
    big = stress in ("big_weights", "everything")
    for i in range(8):
        p = f"seqTransEncoder.layers.{i}."
        for k in ("norm1.weight", "norm2.weight"):
            idx = rng.choice(512, 4, replace=False)
            if ln:
                w[p + k][idx] *= 20.0
        if big:
            for k, f in (("linear1.weight", 3.0), ("linear2.weight", 1.5), ("self_attn.in_proj_weight", 3.0)):
                w[p + k] *= f
        b1, b2 = rng.normal(0, 1.0, 1024).astype(np.float32), rng.normal(0, 0.5, 1536).astype(np.float32)
        if stress == "everything":
            w[p + "linear1.bias"] += b1
            w[p + "self_attn.in_proj_bias"] += b2
    eng = DenoiserEngine(F, T, 4, device=_dev())
    pe = syn.positional_table(5000, 512)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, layer_prefix="seqTransEncoder.layers.", pe=torch.from_numpy(pe))
    xs = ((5.0 if stress == "everything" else 1.0) * x).astype(np.float32)
    eng.set_text(cu(txt))
    out = eng.forward(cu(xs), cu(t)).cpu().numpy()
    assert np.isfinite(out).all()
    ref = denoiser.forward(w, pe, xs, t, txt).numpy()
    model = denoiser.forward(w, pe, xs, t, txt, dt=torch.float16).numpy()
    e_eng, e_model = rel_l2(out, ref), rel_l2(model, ref)
    if stress == "everything":
        # This network amplifies ANY perturbation ~300x, so the rounding model's own figure depends on the rounding pattern: evaluated
        # on inputs that differ by one f16 rounding (3e-4 relative) it spreads over 0.12 .. 0.24.  The engine has to stay inside that spread.
        r2 = np.random.default_rng(1)
        for _ in range(4):
            xp = (xs * (1 + 3e-4 * r2.standard_normal(xs.shape))).astype(np.float32)
            e_model = max(e_model, rel_l2(denoiser.forward(w, pe, xp, t, txt, dt=torch.float16).numpy(), denoiser.forward(w, pe, xp, t, txt).numpy()))
    print("ill-conditioned", stress, "engine", e_eng, "operand-rounding model", e_model)
    assert e_eng <= 1.5 * e_model + 2e-4, (e_eng, e_model)
    # what a user gets: the default path IS above the bar on these checkpoints, and the engine says so when the weights are loaded
    # (outlier statistics over the LayerNorm gains and the weights' scale, DenoiserEngine.weight_diagnostics) and names the remedy
    d = eng.weight_diagnostics
    print("ill-conditioned", stress, "diagnostics", d)
    assert d["recommend_precise"] and e_eng > 8e-4, (e_eng, d)       # (measured 1.1e-3 / 1.2e-3 / 0.2: at or above the bar)
    assert (d["max_layernorm_gain_over_median"] >= 8.0) == ln and (d["max_weight_scale_over_init"] >= 2.0) == big, d
    good, _ = engine_for("xia")
    assert not good.weight_diagnostics["recommend_precise"], good.weight_diagnostics        # the seeded, well-conditioned weights: no warning
    # the remedy for such checkpoints: activations AND weights as f16 hi + lo pairs in every GEMM (engine.set_precise; measured 2.7e-4 /
    # 3.5e-4 / 0.066 for the three cases -- with the activations alone split it was 8.8e-4 / 1.01e-3: what is left is the weights' rounding)
    eng.set_precise(True)
    e_prec = rel_l2(eng.forward(cu(xs), cu(t)).cpu().numpy(), ref)
    eng.set_precise(False)
    print("ill-conditioned", stress, "precise mode", e_prec)
    if stress != "everything":
        assert e_prec < 5e-4, (stress, e_prec)      # the north_star bar (1e-3) with a factor of two to spare
    else:
        assert e_prec <= 0.5 * e_eng, (e_prec, e_eng)


@pytest.mark.parametrize("tag", ["hml", "xia"])
def test_forward_vs_oracle_and_golden(golden, tag):
    from oracle import denoiser
    eng, w = engine_for(tag)
    F, T, x, t, txt = inputs(tag)
    pe = syn.positional_table(5000, 512)
    eng.set_text(cu(txt))
    out = eng.forward(cu(x), cu(t)).cpu().numpy()
    ref = denoiser.forward(w, pe, x, t, txt).numpy()
    e_or, e_gold = rel_l2(out, ref), rel_l2(out, golden["denoise"][f"{tag}|fwd_cond"])
    print("forward", tag, e_or, e_gold)
    assert e_or < TOL and e_gold < TOL
    # unconditional rows (keep = 0) and the CFG doubled batch
    eng.set_text(cu(txt), keep=cu(np.zeros(2, np.float32)))
    out_u = eng.forward(cu(x), cu(t)).cpu().numpy()
    assert rel_l2(out_u, denoiser.forward(w, pe, x, t, txt, uncond=True).numpy()) < TOL
    eng.set_text(cu(txt), cfg=True)
    out_c = eng.forward(cu(x), cu(t), scale=cu(np.array([2.5, 1.5], np.float32)), cfg=True).cpu().numpy()
    e_cfg = rel_l2(out_c, golden["denoise"][f"{tag}|cfg"])
    print("cfg", tag, e_cfg)
    assert e_cfg < TOL       # guidance extrapolates (scale 2.5) and still has to meet the north_star bar


def test_forward_single_clip_and_odd_batch():
    from oracle import denoiser
    eng, w = engine_for("hml")
    F, T, x, t, txt = inputs("hml")
    pe = syn.positional_table(5000, 512)
    eng.set_text(cu(txt[:1]))
    out = eng.forward(cu(x[:1]), cu(t[:1])).cpu().numpy()
    assert rel_l2(out, denoiser.forward(w, pe, x[:1], t[:1], txt[:1]).numpy()) < TOL
    x3 = np.concatenate([x, x[:1] * 0.5]); t3 = np.array([3, 957, 40]); txt3 = np.concatenate([txt, txt[:1]])
    eng.set_text(cu(txt3))
    out3 = eng.forward(cu(x3), cu(t3)).cpu().numpy()
    assert rel_l2(out3, denoiser.forward(w, pe, x3, t3, txt3).numpy()) < TOL


def test_prior_as_denoiser(golden):
    eng, w = engine_for("xia", prior=True)
    F, T, x, t, txt = inputs("xia")
    eng.set_text(cu(txt))
    out = eng.forward(cu(x), cu(t)).cpu().numpy()
    assert rel_l2(out, golden["denoise"]["xia|prior_fwd"]) < TOL


# ------------------------------------------------------------------------------ stand-alone elementwise kernels
@pytest.mark.parametrize("tag", ["xia", "hml"])
def test_elementwise_kernels_vs_oracle(tag):
    from mst_amd.engine import Schedule, SAMPLER_DDPM, SAMPLER_DDIM
    from oracle import diffusion, schedule
    F, T = SHAPES[tag]
    shape = (2, F, 1, T)
    mask = syn.root_horizontal_mask(2, F, T)
    motion = syn.normal(SEED, f"{tag}/motion", shape)
    x = syn.normal(SEED, f"{tag}/x", shape)
    mo = syn.normal(SEED, f"{tag}/fake_model_out", shape)
    nz = syn.normal(SEED, f"{tag}/nz", shape)
    for resp in ("", "ddim20", "100"):
        tab, tmap = schedule.make("cosine", 1000, resp)
        sch = Schedule(tab, tmap, _dev())
        n = len(tmap)
        for tt in ([0, n - 1], [1, n // 2]):
            tt = np.array(tt)
            q = sch.q_sample(cu(motion), cu(tt), cu(nz), cu(mask)).cpu().numpy()
            assert rel_l2(q, diffusion.q_sample(tab, motion, tt, torch.from_numpy(nz), mask).numpy()) < TOL_ELEM
            q = sch.q_sample(cu(motion), cu(tt), cu(nz)).cpu().numpy()
            assert rel_l2(q, diffusion.q_sample(tab, motion, tt, torch.from_numpy(nz)).numpy()) < TOL_ELEM
            for inpaint in (True, False):
                s, p = sch.step(cu(mo), cu(x), cu(tt), cu(nz), SAMPLER_DDPM, mask=cu(mask), motion=cu(motion), mask_noise=inpaint)
                r = diffusion.p_sample(tab, mo, x, tt, torch.from_numpy(nz), inpaint, mask, motion)
                assert rel_l2(s.cpu().numpy(), r["sample"].numpy()) < TOL_ELEM
                assert np.array_equal(p.cpu().numpy(), r["pred_xstart"].numpy())      # blend is exact arithmetic
                assert np.array_equal(p.cpu().numpy()[:, :3], motion[:, :3])           # masked rows bit-exact
            for eta in (0.0, 0.5):
                s, p = sch.step(cu(mo), cu(x), cu(tt), cu(nz), SAMPLER_DDIM, eta=eta, mask=cu(mask), motion=cu(motion), mask_noise=True)
                r = diffusion.ddim_sample(tab, mo, x, tt, torch.from_numpy(nz), eta, True, mask, motion)
                assert rel_l2(s.cpu().numpy(), r["sample"].numpy()) < 2e-5, (resp, tt, eta)
            # no mask / no motion at all (plain GaussianDiffusion use) and clipping
            s, p = sch.step(cu(mo), cu(x), cu(tt), cu(nz), SAMPLER_DDPM, clip_denoised=True)
            r = diffusion.p_sample(tab, mo, x, tt, torch.from_numpy(nz), False, None, None, clip_denoised=True)
            assert rel_l2(s.cpu().numpy(), r["sample"].numpy()) < TOL_ELEM
            assert float(p.abs().max()) <= 1.0


# ------------------------------------------------------------------------------ fused steps and loops vs the reference's outputs
def _noise_stack(tag, n, shape):
    return np.stack([syn.normal(SEED, f"{tag}/noise/{k}", shape) for k in range(n)])


@pytest.mark.parametrize("tag", ["xia", "hml"])
def test_fused_single_steps_vs_golden(golden, tag):
    from mst_amd.engine import Schedule, SAMPLER_DDPM, SAMPLER_DDIM
    from oracle import schedule
    eng, w = engine_for(tag)
    F, T, x, t, txt = inputs(tag)
    g = golden["denoise"]
    shape = (2, F, 1, T)
    mask = syn.root_horizontal_mask(2, F, T)
    motion = syn.normal(SEED, f"{tag}/motion", shape)
    eng.set_text(cu(txt))
    sel = (lambda a: a[1:]) if tag == "hml" else (lambda a: a)
    # the reference ran p_sample with per-clip t = [t0, t1]; the fused loop takes one t per call,
    # so run clip i alone at its own t (engine batch 1) and compare clip-wise.
    for name, resp, tts in (("full", "", [0, 500]), ("ddim", "ddim20", [0, 19]), ("r100", "100", [1, 99])):
        tab, tmap = schedule.make("cosine", 1000, resp)
        sch = Schedule(tab, tmap, _dev())
        for kind, sampler, key, ntag, eta in (("p", SAMPLER_DDPM, f"{tag}|p_sample_{name}|sample", f"{tag}/ps_{name}", 0.0),
                                              ("d", SAMPLER_DDIM, f"{tag}|ddim_sample_{name}|sample", f"{tag}/dd_{name}", 0.0),
                                              ("e", SAMPLER_DDIM, f"{tag}|ddim_sample_eta_{name}|sample", f"{tag}/dd5_{name}", 0.5)):
            noise = syn.normal(SEED, f"{ntag}/noise/0", shape)
            outs, preds = [], []
            for i in range(2):
                eng.set_text(cu(txt[i:i + 1]))
                xi = cu(x[i:i + 1]).clone()
                xi, dump = eng.sample_loop(sch, xi, tts[i], tts[i], sampler, eta, mask=cu(mask[i:i + 1]),
                                           motion=cu(motion[i:i + 1]), mask_noise=True, noise=cu(noise[i:i + 1][None]),
                                           dump_xstart=True)
                outs.append(xi.cpu().numpy()); preds.append(dump[0].cpu().numpy())
            out, pred = np.concatenate(outs), np.concatenate(preds)
            e = rel_l2(sel(out), g[key])
            print("step", tag, name, kind, e)
            assert e < TOL, (name, kind, e)
            assert np.array_equal(pred[:, :3], motion[:, :3])          # inpainted rows of x0-hat bit-exact
            if kind == "p":
                assert rel_l2(sel(pred), g[f"{tag}|p_sample_{name}|pred_xstart"]) < TOL


def test_loops_vs_golden_xia(golden):
    from mst_amd.engine import Schedule, SAMPLER_DDPM, SAMPLER_DDIM
    from oracle import schedule
    g = golden["denoise"]
    eng, w = engine_for("xia")
    F, T, x, t, txt = inputs("xia")
    shape = (1, F, 1, T)
    mask = syn.root_horizontal_mask(1, F, T)
    motion = syn.normal(SEED, "xia/motion", (2, F, 1, T))[:1]
    eng.set_text(cu(txt[:1]))

    # BASELINE.json configs[0]: single clip, 100 respaced DDPM steps
    tab, tmap = schedule.make("cosine", 1000, "100")
    sch = Schedule(tab, tmap, _dev())
    nz = _noise_stack("xia/loop100", 101, shape)
    xT = cu(nz[0]).clone()
    out = eng.sample_loop(sch, xT, 99, 0, SAMPLER_DDPM, mask=cu(mask), motion=cu(motion), noise=cu(nz[1:])).cpu().numpy()
    e = rel_l2(out, g["xia|loop100|sample"])
    print("loop100", e)
    assert e < TOL
    assert np.array_equal(out[:, :3], motion[:, :3])                  # masked rows exact after the t=0 step

    # the demo: ddim20, skip 14 (indices 5..0), init image via q_sample, x0-hat dump
    tab, tmap = schedule.make("cosine", 1000, "ddim20")
    sch = Schedule(tab, tmap, _dev())
    nz = _noise_stack("xia/demo", 7, shape)
    x5 = sch.q_sample(cu(motion), cu(np.array([5])), cu(nz[0]), cu(mask))
    _, dump = eng.sample_loop(sch, x5, 5, 0, SAMPLER_DDIM, mask=cu(mask), motion=cu(motion), noise=cu(nz[1:]), dump_xstart=True)
    e = rel_l2(dump.reshape(6, F, 1, T).cpu().numpy(), g["xia|demo|xstart"])
    print("demo", e)
    assert e < TOL

    # classifier-free guidance inside the loop (doubled batch), indices 9..0 of the full process
    tab, tmap = schedule.make("cosine", 1000, "")
    sch = Schedule(tab, tmap, _dev())
    nz = _noise_stack("xia/cfgloop", 11, shape)
    x9 = sch.q_sample(cu(motion), cu(np.array([9])), cu(nz[0]), cu(mask))
    eng.set_text(cu(txt[:1]), cfg=True)
    out = eng.sample_loop(sch, x9, 9, 0, SAMPLER_DDPM, cfg=True, scale=cu(np.array([2.5], np.float32)), mask=cu(mask),
                          motion=cu(motion), noise=cu(nz[1:])).cpu().numpy()
    e = rel_l2(out, g["xia|cfgloop|sample"])
    print("cfgloop", e)
    assert e < TOL

    # neutralisation pre-pass: the frozen prior as denoiser, stop_timesteps = 990 (indices 999..990)
    engp, _ = engine_for("xia", prior=True)
    engp.set_text(cu(txt[:1]))
    nz = _noise_stack("xia/neutral", 11, shape)
    zero = np.zeros(shape, np.float32)
    x999 = sch.q_sample(cu(motion), cu(np.array([999])), cu(nz[0]), cu(zero))
    _, dump = engp.sample_loop(sch, x999, 999, 990, SAMPLER_DDPM, mask=cu(zero), motion=cu(motion), noise=cu(nz[1:]), dump_xstart=True)
    e = rel_l2(dump[-1].cpu().numpy(), g["xia|neutral|xstart_last"])
    print("neutral", e)
    assert e < TOL


def test_loop_tail_vs_golden_hml(golden):
    from mst_amd.engine import Schedule, SAMPLER_DDPM
    from oracle import schedule
    g = golden["denoise"]
    eng, w = engine_for("hml")
    F, T, x, t, txt = inputs("hml")
    shape = (2, F, 1, T)
    mask = syn.root_horizontal_mask(2, F, T)
    motion = syn.normal(SEED, "hml/motion", shape)
    tab, tmap = schedule.make("cosine", 1000, "")
    sch = Schedule(tab, tmap, _dev())
    nz = _noise_stack("hml/tail8", 9, shape)
    eng.set_text(cu(txt))
    x7 = sch.q_sample(cu(motion), cu(np.array([7, 7])), cu(nz[0]), cu(mask))
    out = eng.sample_loop(sch, x7, 7, 0, SAMPLER_DDPM, mask=cu(mask), motion=cu(motion), noise=cu(nz[1:])).cpu().numpy()
    e = rel_l2(out, g["hml|tail8|sample"])
    print("tail8", e)
    assert e < TOL
    assert np.array_equal(out[:, :3], motion[:, :3])


def test_full_length_loop_vs_oracle():
    """The headline configuration's LENGTH: all 1000 DDPM steps (no respacing) of one (181,1,76) clip with recorded noise,
    the engine's fused loop against the CPU oracle stepping the same 1000 indices (10-20 s of CPU): rounding does not
    accumulate over a full-length trajectory (x0-prediction contracts it), and the inpainted rows stay exact."""
    from mst_amd.engine import Schedule, SAMPLER_DDPM
    from oracle import denoiser, diffusion, schedule
    eng, w = engine_for("xia")
    F, T, x, t, txt = inputs("xia")
    shape = (1, F, 1, T)
    pe = syn.positional_table(5000, 512)
    mask = syn.root_horizontal_mask(1, F, T)
    motion = syn.normal(SEED, "xia/motion", (2, F, 1, T))[:1]
    tab, tmap = schedule.make("cosine", 1000, "")
    rng = np.random.default_rng(SEED)
    nz = rng.standard_normal((1001,) + shape, dtype=np.float32)
    torch.set_num_threads(16)
    ref = diffusion.sample_loop(lambda xx, tt: denoiser.forward(w, pe, xx, tt, txt[:1]), tab, tmap, shape, lambda k: torch.from_numpy(nz[k]),
                                "ddpm", True, torch.from_numpy(mask), torch.from_numpy(motion))
    ref = (ref["sample"] if isinstance(ref, dict) else ref).numpy()
    eng.set_text(cu(txt[:1]))
    sch = Schedule(tab, tmap, _dev())
    out = eng.sample_loop(sch, cu(nz[0]).clone(), 999, 0, SAMPLER_DDPM, mask=cu(mask), motion=cu(motion), noise=cu(nz[1:])).cpu().numpy()
    e = rel_l2(out, ref)
    print("1000-step loop", e)
    assert e < TOL, e
    assert np.array_equal(out[:, :3], motion[:, :3])


# ------------------------------------------------------------------------------ properties at the bench size
def test_full_size_properties(tile_path):
    """BASELINE.json configs[1] size (batch 64, 263 x 196): size-independent properties -- clips are
    independent (a batch-64 run equals the same clips run in batches of 2), masked rows come out
    exactly equal to the content clip, the Philox noise path is reproducible and has unit moments."""
    from mst_amd.engine import DenoiserEngine, Schedule, SAMPLER_DDPM
    from oracle import schedule
    F, T, B = 263, 196, 64
    eng = DenoiserEngine(F, T, B, device=_dev())
    w = syn.denoiser_state(SEED, F)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
    tab, tmap = schedule.make("cosine", 1000, "")
    sch = Schedule(tab, tmap, _dev())
    txt = cu(syn.normal(SEED, "big/txt", (B, 512)))
    x0 = cu(syn.normal(SEED, "big/x", (B, F, 1, T)))
    mask = cu(syn.root_horizontal_mask(B, F, T))
    motion = cu(syn.normal(SEED, "big/motion", (B, F, 1, T)))
    eng.set_text(txt)
    a = eng.sample_loop(sch, x0.clone(), 5, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=7)
    b = eng.sample_loop(sch, x0.clone(), 5, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=7)
    assert torch.equal(a, b)                                           # deterministic
    assert torch.equal(a[:, :3], motion[:, :3])                        # inpainting rows exact
    assert torch.isfinite(a).all()
    n = eng.philox_normal(B, T, 7, 3)
    assert abs(float(n.mean())) < 2e-3 and abs(float(n.std()) - 1.0) < 2e-3
    # clip independence: clips 10..11 alone reproduce their rows of the batch-64 forward
    t = torch.full((B,), 321, dtype=torch.int64, device=_dev())
    full = eng.forward(x0, t)
    eng.set_text(txt[10:12])
    part = eng.forward(x0[10:12], t[10:12])
    # same kernels (large-tile path forced): identical up to fp32 reduction noise; with the default small-tile path the two
    # clips run through differently tiled kernels -- two f16-operand evaluations of the same function
    assert rel_l2(part.cpu().numpy(), full[10:12].cpu().numpy()) < (1e-6 if tile_path == "0" else TOL)


# ------------------------------------------------------------------------------ BASELINE.json configs[4] and configs[2] at full size
def _big_batch(tag, B):
    """Clips 0..1 are the golden inputs, the rest seeded fill: the golden vectors pin two rows of a full-size launch."""
    F, T, x2, t2, txt2 = inputs(tag)
    x = np.concatenate([x2, syn.normal(SEED, f"{tag}/bigx/{B}", (B - 2, F, 1, T))])
    t = np.concatenate([t2, np.random.default_rng(B).integers(0, 1000, B - 2)])
    txt = np.concatenate([txt2, syn.normal(SEED, f"{tag}/bigtxt/{B}", (B - 2, 512))])
    return F, T, x, t, txt


def test_batch128_prior_as_denoiser_hml(golden, tile_path):
    """BASELINE.json configs[4]: 128 clips per GPU of (263,1,196) through the T2M prior as the denoiser
    (train/finetune_style_diffusion.py:195-212 drives MDM.forward this way).  M = 25 216 token rows: the only size that
    reaches the 128-token LayerNorm tiles by row count; run on the default tiles and with MST_LN128_M=1 (fixture).
    Rows 0..1 against the reference's golden `hml|prior_fwd`, then the size-independent properties at 128."""
    from mst_amd.engine import Schedule, SAMPLER_DDPM
    from oracle import schedule
    B = 128
    eng, w = engine_for("hml", prior=True, max_rows=B)
    F, T, x, t, txt = _big_batch("hml", B)
    eng.set_text(cu(txt))
    out = eng.forward(cu(x), cu(t))
    e = rel_l2(out[1:2].cpu().numpy(), golden["denoise"]["hml|prior_fwd"])       # the fixture keeps clip 1 of the hml shape
    print("batch128 prior_fwd", tile_path, e)
    assert e < TOL
    assert torch.isfinite(out).all()
    assert torch.equal(out, eng.forward(cu(x), cu(t)))                              # deterministic
    # clip independence: clips 0..1 and 77..78 alone reproduce their rows of the batch-128 launch
    for lo in (0, 77):
        eng.set_text(cu(txt[lo:lo + 2]))
        part = eng.forward(cu(x[lo:lo + 2]), cu(t[lo:lo + 2]))
        # forced large tiles: the very same kernels -> fp32 reduction noise only; default: two clips take the small-tile kernels
        # (two f16-operand evaluations of one function, each within TOL of the fp32 path)
        assert rel_l2(part.cpu().numpy(), out[lo:lo + 2].cpu().numpy()) < (1e-6 if tile_path != "default" else 1.5 * TOL)
    # a short sampling loop at 128 clips: deterministic, masked rows bit-exact, the same numbers whether Philox runs in the
    # kernel or is injected, slices (2 x 64 clips) consistent with clip-wise runs
    tab, tmap = schedule.make("cosine", 1000, "")
    sch = Schedule(tab, tmap, _dev())
    mask = cu(syn.root_horizontal_mask(B, F, T))
    motion = cu(syn.normal(SEED, "hml/bigmotion", (B, F, 1, T)))
    x0 = cu(x)
    eng.set_text(cu(txt))
    assert eng.loop_slices(B) == 2                          # 394 tiles: one slice per round of tiles over the 256 CUs
    a = eng.sample_loop(sch, x0.clone(), 3, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=11)
    b = eng.sample_loop(sch, x0.clone(), 3, 0, SAMPLER_DDPM, mask=mask, motion=motion, seed=11)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    assert torch.equal(a[:, :3], motion[:, :3])
    nz = torch.stack([eng.philox_normal(B, T, 11, j) for j in range(4)])
    c = eng.sample_loop(sch, x0.clone(), 3, 0, SAMPLER_DDPM, mask=mask, motion=motion, noise=nz)
    assert torch.equal(a, c)
    eng.set_text(cu(txt[100:102]))
    alone = eng.sample_loop(sch, x0[100:102].clone(), 3, 0, SAMPLER_DDPM, mask=mask[100:102], motion=motion[100:102],
                            noise=nz[:, 100:102].contiguous())
    assert rel_l2(alone.cpu().numpy(), a[100:102].cpu().numpy()) < (1e-6 if tile_path != "default" else 1.5 * TOL)


def test_cfg_at_headline_size_hml(golden, tile_path):
    """BASELINE.json configs[2] at the headline size: 64 clips = 128 rows through the transformer (cond | uncond doubled
    batch, model/cfg_sampler.py:36-43).  Rows 0..1 against the reference's golden `hml|cfg` (scales 2.5 / 1.5), then
    determinism, masked rows, and equality with clip-wise runs of a sliced CFG loop."""
    from mst_amd.engine import Schedule, SAMPLER_DDPM
    from oracle import schedule
    B = 64
    eng, w = engine_for("hml", max_rows=2 * B)
    F, T, x, t, txt = _big_batch("hml", B)
    scale = np.concatenate([np.array([2.5, 1.5], np.float32), np.linspace(1.0, 3.0, B - 2).astype(np.float32)])
    eng.set_text(cu(txt), cfg=True)
    out = eng.forward(cu(x), cu(t), scale=cu(scale), cfg=True)
    e = rel_l2(out[:2].cpu().numpy(), golden["denoise"]["hml|cfg"])
    print("cfg64", tile_path, e)
    assert e < TOL
    assert torch.equal(out, eng.forward(cu(x), cu(t), scale=cu(scale), cfg=True))
    eng.set_text(cu(txt[30:32]), cfg=True)
    part = eng.forward(cu(x[30:32]), cu(t[30:32]), scale=cu(scale[30:32]), cfg=True)
    assert rel_l2(part.cpu().numpy(), out[30:32].cpu().numpy()) < (1e-6 if tile_path != "default" else 1.5 * TOL)
    tab, tmap = schedule.make("cosine", 1000, "")
    sch = Schedule(tab, tmap, _dev())
    mask = cu(syn.root_horizontal_mask(B, F, T))
    motion = cu(syn.normal(SEED, "hml/cfgmotion", (B, F, 1, T)))
    nz = cu(np.stack([syn.normal(SEED, f"hml/cfgnz/{k}", (B, F, 1, T)) for k in range(3)]))
    x0 = cu(x)
    eng.set_text(cu(txt), cfg=True)
    assert eng.loop_slices(B, cfg=True) == 2
    a = eng.sample_loop(sch, x0.clone(), 2, 0, SAMPLER_DDPM, cfg=True, scale=cu(scale), mask=mask, motion=motion, noise=nz)
    b = eng.sample_loop(sch, x0.clone(), 2, 0, SAMPLER_DDPM, cfg=True, scale=cu(scale), mask=mask, motion=motion, noise=nz)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    assert torch.equal(a[:, :3], motion[:, :3])
    for i in (0, 22, 63):                                   # one clip of every slice, alone
        eng.set_text(cu(txt[i:i + 1]), cfg=True)
        one = eng.sample_loop(sch, x0[i:i + 1].clone(), 2, 0, SAMPLER_DDPM, cfg=True, scale=cu(scale[i:i + 1]),
                              mask=mask[i:i + 1], motion=motion[i:i + 1], noise=nz[:, i:i + 1].contiguous())
        # forced large tiles: the very same kernels.  Default: the single clip takes the small-tile kernels, i.e. this compares
        # TWO f16-operand evaluations, each within TOL of the fp32 path (checked against the golden above): sqrt(2) x TOL apart
        assert rel_l2(one.cpu().numpy(), a[i:i + 1].cpu().numpy()) < (1e-6 if tile_path != "default" else 1.5 * TOL)
