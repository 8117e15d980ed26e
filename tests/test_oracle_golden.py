"""Pin the CPU oracle to the reference: every oracle function against the golden vectors that
tests/golden/make_golden.py produced by running the reference itself (CPU only, no GPU)."""
import numpy as np
import pytest
import torch

import mst_amd  # noqa: F401
from mst_amd import synthetic as syn
from oracle import denoiser, diffusion, schedule
from conftest import SEED, rel_l2

TOL = 2e-5  # fp32 vs fp32, different summation order (SURVEY.md section 4 tier 3: <= 1e-5 .. 2e-5)
PROMPTS = ["a person walks proudly", "an old man jumps"]


# ------------------------------------------------------------------------------ schedules: exact
@pytest.mark.parametrize("sched", ["cosine", "linear"])
@pytest.mark.parametrize("resp", ["", "ddim20", "100", "10,20,30"])
def test_schedule_tables_exact(golden, sched, resp):
    tab, tmap = schedule.make(sched, 1000, resp)
    g = golden["schedules"]
    assert np.array_equal(np.array(tmap), g[f"{sched}|{resp}|timestep_map"])
    for name in schedule.TABLE_NAMES:
        ref = g[f"{sched}|{resp}|{name}"]
        assert ref.dtype == np.float64
        assert np.array_equal(tab[name], ref), name


def test_space_timesteps_known_answers():
    assert schedule.space_timesteps(1000, "ddim20") == set(range(0, 1000, 50))
    s = schedule.space_timesteps(1000, "100")
    assert len(s) == 100 and min(s) == 0 and max(s) == 999
    with pytest.raises(ValueError):
        schedule.space_timesteps(1000, "ddim600")
    with pytest.raises(ValueError):
        schedule.space_timesteps(10, [20])


# ------------------------------------------------------------------------------ masks: exact
def test_root_horizontal_pattern_matches_reference(golden):
    g = golden["masks"]
    for mod, F in (("stylexia_posrot_utils", 181), ("bandai_posrot_utils", 190), ("humanml_utils", 263)):
        row = g[f"{mod}|root_horizontal"]
        assert row.shape == (F,)
        mine = syn.root_horizontal_mask(1, F, 4)
        assert np.array_equal(mine[0, :, 0, 0].astype(np.uint8), row)


# ------------------------------------------------------------------------------ denoiser
def _setup(tag):
    F, T = {"xia": (181, 76), "hml": (263, 196)}[tag]
    w = syn.denoiser_state(SEED, F)
    pe = syn.positional_table(5000, 512)
    x = syn.normal(SEED, f"{tag}/x", (2, F, 1, T))
    t = np.array([3, 957])
    txt = np.stack([syn.normal(SEED, "text/" + p, (512,)) for p in PROMPTS])
    return F, T, w, pe, x, t, txt


def _sel(tag, a):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    return a[1:] if tag == "hml" else a


@pytest.mark.parametrize("tag", ["xia", "hml"])
def test_forward_matches_reference(golden, tag):
    F, T, w, pe, x, t, txt = _setup(tag)
    g = golden["denoise"]
    out = denoiser.forward(w, pe, x, t, txt)
    assert rel_l2(out.numpy(), g[f"{tag}|fwd_cond"]) < TOL
    out_u = denoiser.forward(w, pe, x, t, txt, uncond=True)
    assert rel_l2(_sel(tag, out_u), g[f"{tag}|fwd_uncond"]) < TOL
    c = denoiser.cfg_forward(w, pe, x, t, txt, np.array([2.5, 1.5], dtype=np.float32))
    assert rel_l2(c.numpy(), g[f"{tag}|cfg"]) < TOL


@pytest.mark.parametrize("tag", ["xia", "hml"])
def test_prior_as_denoiser_and_motion_encoder(golden, tag):
    F, T, w, pe, x, t, txt = _setup(tag)
    g = golden["denoise"]
    lp = "motion_enc.mdm_model.seqTransEncoder.layers."
    w.update(syn.denoiser_state(SEED, F, layer_prefix=lp))
    out = denoiser.forward(w, pe, x, t, txt, layer_prefix=lp)
    assert rel_l2(_sel(tag, out), g[f"{tag}|prior_fwd"]) < TOL
    ep = "motion_enc.seqTransEncoder.layers."
    w.update(syn.denoiser_state(SEED, F, layer_prefix=ep))
    for q in ("motion_enc.muQuery", "motion_enc.sigmaQuery"):
        w[q] = syn.tensor_for(SEED, q, (1, 512))
    fm = np.zeros((2, T), dtype=bool)
    fm[0, :T] = True
    fm[1, : T - 17] = True
    mu = denoiser.motion_encoder(w, pe, x, fm)
    assert rel_l2(mu.numpy(), g[f"{tag}|motion_enc_mu"]) < TOL


# ------------------------------------------------------------------------------ diffusion steps
def _noise(tag):
    return lambda k, shape: torch.from_numpy(syn.normal(SEED, f"{tag}/noise/{k}", shape))


@pytest.mark.parametrize("tag", ["xia", "hml"])
def test_single_steps_match_reference(golden, tag):
    F, T, w, pe, x, t, txt = _setup(tag)
    g = golden["denoise"]
    shape = (2, F, 1, T)
    mask = syn.root_horizontal_mask(2, F, T)
    motion = syn.normal(SEED, f"{tag}/motion", shape)
    full, map_full = schedule.make("cosine", 1000, "")
    ddim, map_ddim = schedule.make("cosine", 1000, "ddim20")
    r100, map_100 = schedule.make("cosine", 1000, "100")

    q = diffusion.q_sample(full, motion, np.array([10, 700]), _noise(f"{tag}/q")(0, shape), mask)
    assert rel_l2(q.numpy(), g[f"{tag}|q_sample"]) < 1e-6

    for name, tab, tmap, tt in (("full", full, map_full, [0, 500]), ("ddim", ddim, map_ddim, [0, 19]),
                                ("r100", r100, map_100, [1, 99])):
        tt = np.array(tt)
        mo = denoiser.forward(w, pe, x, np.array(tmap)[tt], txt)
        r = diffusion.p_sample(tab, mo, x, tt, _noise(f"{tag}/ps_{name}")(0, shape), True, mask, motion)
        assert rel_l2(_sel(tag, r["sample"]), g[f"{tag}|p_sample_{name}|sample"]) < TOL
        assert rel_l2(_sel(tag, r["pred_xstart"]), g[f"{tag}|p_sample_{name}|pred_xstart"]) < TOL
        # bit-exact inpainting rows of x0-hat
        assert np.array_equal(r["pred_xstart"].numpy()[:, :3], motion[:, :3])
        r = diffusion.ddim_sample(tab, mo, x, tt, _noise(f"{tag}/dd_{name}")(0, shape), 0.0, True, mask, motion)
        assert rel_l2(_sel(tag, r["sample"]), g[f"{tag}|ddim_sample_{name}|sample"]) < TOL
        r = diffusion.ddim_sample(tab, mo, x, tt, _noise(f"{tag}/dd5_{name}")(0, shape), 0.5, True, mask, motion)
        assert rel_l2(_sel(tag, r["sample"]), g[f"{tag}|ddim_sample_eta_{name}|sample"]) < TOL
    tt = np.array([7, 400])
    mo = denoiser.forward(w, pe, x, tt, txt)
    r = diffusion.p_sample(full, mo, x, tt, _noise(f"{tag}/ps_base")(0, shape), False, mask, motion)
    assert rel_l2(_sel(tag, r["sample"]), g[f"{tag}|p_sample_base|sample"]) < TOL


# ------------------------------------------------------------------------------ loops
def test_loops_match_reference_xia(golden):
    F, T, w, pe, x, t, txt = _setup("xia")
    g = golden["denoise"]
    shape = (1, F, 1, T)
    mask = syn.root_horizontal_mask(1, F, T)
    motion = syn.normal(SEED, "xia/motion", (2, F, 1, T))[:1]
    txt1 = txt[:1]

    def model(xx, tt):
        return denoiser.forward(w, pe, xx, tt, txt1)

    # BASELINE.json configs[0]: single clip, 100 respaced DDPM steps
    tab, tmap = schedule.make("cosine", 1000, "100")
    nz = _noise("xia/loop100")
    s = diffusion.sample_loop(model, tab, tmap, shape, lambda k: nz(k, shape), "ddpm", True, mask, motion)
    assert rel_l2(s.numpy(), g["xia|loop100|sample"]) < 2e-4   # 100 chained fp32 forwards
    assert np.array_equal(s.numpy()[:, :3], motion[:, :3])      # masked rows exact at t=0

    # demo setting: ddim20, skip 14, init image, dumped x0-hats
    tab, tmap = schedule.make("cosine", 1000, "ddim20")
    nz = _noise("xia/demo")
    dump = diffusion.sample_loop(model, tab, tmap, shape, lambda k: nz(k, shape), "ddim", True, mask, motion,
                                 init_image=motion, skip_timesteps=14, dump_all_xstart=True)
    assert len(dump) == 6
    assert rel_l2(torch.cat(dump).numpy(), g["xia|demo|xstart"]) < 5e-5

    # neutralisation pre-pass: frozen prior as denoiser, stop_timesteps
    lp = "motion_enc.mdm_model.seqTransEncoder.layers."
    w2 = dict(w)
    w2.update(syn.denoiser_state(SEED, F, layer_prefix=lp))
    tab, tmap = schedule.make("cosine", 1000, "")
    nz = _noise("xia/neutral")
    dump = diffusion.sample_loop(lambda xx, tt: denoiser.forward(w2, pe, xx, tt, txt1, layer_prefix=lp),
                                 tab, tmap, shape, lambda k: nz(k, shape), "ddpm", True,
                                 np.zeros(shape, np.float32), motion, init_image=motion,
                                 stop_timesteps=990, dump_all_xstart=True)
    assert len(dump) == int(g["xia|neutral|n"]) == 10
    assert rel_l2(dump[-1].numpy(), g["xia|neutral|xstart_last"]) < 5e-5

    # CFG model inside the loop
    nz = _noise("xia/cfgloop")
    s = diffusion.sample_loop(
        lambda xx, tt: denoiser.cfg_forward(w, pe, xx, tt, txt1, np.array([2.5], np.float32)),
        tab, tmap, shape, lambda k: nz(k, shape), "ddpm", True, mask, motion,
        init_image=motion, skip_timesteps=990)
    assert rel_l2(s.numpy(), g["xia|cfgloop|sample"]) < 5e-5


def test_loop_tail_hml(golden):
    F, T, w, pe, x, t, txt = _setup("hml")
    g = golden["denoise"]
    shape = (2, F, 1, T)
    mask = syn.root_horizontal_mask(2, F, T)
    motion = syn.normal(SEED, "hml/motion", shape)
    tab, tmap = schedule.make("cosine", 1000, "")
    nz = _noise("hml/tail8")
    s = diffusion.sample_loop(lambda xx, tt: denoiser.forward(w, pe, xx, tt, txt), tab, tmap, shape,
                              lambda k: nz(k, shape), "ddpm", True, mask, motion,
                              init_image=motion, skip_timesteps=992)
    assert rel_l2(s.numpy(), g["hml|tail8|sample"]) < 5e-5
    assert np.array_equal(s.numpy()[:, :3], motion[:, :3])
