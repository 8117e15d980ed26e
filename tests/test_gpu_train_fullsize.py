"""The native TRAINING path at the size BASELINE.json configs[3] runs it per GPU: 64 clips x (263, 1, 196) = 12 608 token rows.

tests/test_gpu_train.py holds the training kernels to torch autograd at 2-3 clips; the 64-clip pass takes other code: 197-tile
grids, split-K wgrads over 12 608 rows with an ordered second stage, a 1.6 GB activation tape, gradient scaling at scale.  Here:

  * forward and ALL 96 parameter gradients (+ dL/dh) against fp32 torch autograd of the same eight layers on the GPU, dropout 0
    and 0.1 with the engine's own keep masks (the tolerances of test_gpu_train.py: 1e-3 forward, 1.5e-3 per gradient tensor);
  * linearity over the batch: the 64-clip gradient = the sum of the gradients of its two 32-clip halves (size-independent property);
  * bit-determinism of EVERY gradient tensor across two runs -- the bias / LayerNorm-parameter reductions are two-stage sums in a
    fixed order, not float atomics, so data-parallel replicas stay bit-identical;
  * `few_shot_style_finetune_losses` (gaussian_diffusion.py:1317-1399) once at 64 clips through the model boundary: finite loss
    terms and gradients, identical for identical seeds, and AdamW's logged norms reproducible.
"""
import numpy as np
import pytest
import torch

import mst_amd  # noqa: F401
from mst_amd import synthetic as syn
from mst_amd.engine import LAYER_TENSORS
from conftest import SEED, rel_l2

from test_gpu_train import TOL_FWD, TOL_GRAD, D, L, engine_masks, layer_params, torch_stack

pytestmark = pytest.mark.gpu
B, FE, T = 64, 263, 196
S = T + 1


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


_ENG = {}


def big_engine():
    from mst_amd.engine import DenoiserEngine
    if "e" not in _ENG:
        eng = DenoiserEngine(FE, T, B, device=_dev())
        w = syn.denoiser_state(SEED, FE, layer_prefix="seqTransEncoder.layers.")
        eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, layer_prefix="seqTransEncoder.layers.",
                            pe=torch.from_numpy(syn.positional_table(5000, 512)))
        _ENG["e"] = (eng, w)
    return _ENG["e"]


def stream(rows, tag="full"):
    g = torch.Generator(device="cpu").manual_seed(SEED + 17)
    h = torch.randn(rows, S, D, generator=g).to(_dev())
    r = torch.randn(rows, S, D, generator=g).to(_dev())
    return h, r


def engine_grads(eng, w, h, r, p, seed):
    out, tape = eng.train_forward(h, p, seed)
    grads = [torch.zeros_like(q) for q in layer_params(w, False)]
    d_in = eng.train_backward(tape, r, p, seed, grads)
    return out, d_in, grads


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_stack_forward_backward_64_clips_vs_autograd(p):
    eng, w = big_engine()
    h, r = stream(B)
    seed = 0x5EED0F64C11F5
    params = layer_params(w, True)
    href = h.clone().requires_grad_(True)
    masks = engine_masks(eng, seed, p, B, S) if p > 0 else None
    ref = torch_stack(href, params, masks)
    (ref * r).sum().backward()
    del masks
    out, d_in, grads = engine_grads(eng, w, h, r, p, seed)
    errs = {"out": rel_l2(out.cpu().numpy(), ref.detach().cpu().numpy()), "d_in": rel_l2(d_in.cpu().numpy(), href.grad.cpu().numpy())}
    assert errs["out"] <= TOL_FWD, errs
    for i, (g, q) in enumerate(zip(grads, params)):
        errs[f"L{i // 12}.{LAYER_TENSORS[i % 12]}"] = rel_l2(g.cpu().numpy(), q.grad.cpu().numpy())
    worst = max(errs, key=errs.get)
    print(f"64 clips, dropout {p}: forward {errs['out']:.2e}, d_in {errs['d_in']:.2e}, worst gradient {worst} {errs[worst]:.2e}")
    assert errs[worst] <= TOL_GRAD, (worst, errs[worst])


def test_gradient_of_64_clips_is_the_sum_of_its_halves_and_bit_reproducible():
    eng, w = big_engine()
    h, r = stream(B)
    p, seed = 0.0, 0
    _, d_full, g_full = engine_grads(eng, w, h, r, p, seed)
    _, d_again, g_again = engine_grads(eng, w, h, r, p, seed)
    assert torch.equal(d_full, d_again)
    for i, (a, b) in enumerate(zip(g_full, g_again)):          # every tensor, wgrads and the ordered bias / LayerNorm reductions alike
        assert torch.equal(a, b), f"L{i // 12}.{LAYER_TENSORS[i % 12]} differs between two runs"
    _, d_a, g_a = engine_grads(eng, w, h[:32].contiguous(), r[:32].contiguous(), p, seed)
    _, d_b, g_b = engine_grads(eng, w, h[32:].contiguous(), r[32:].contiguous(), p, seed)
    assert rel_l2(torch.cat([d_a, d_b]).cpu().numpy(), d_full.cpu().numpy()) < 1e-5
    for i, (f, a, b) in enumerate(zip(g_full, g_a, g_b)):
        # same products, other summation orders (split-K partition, row blocks) and per-call gradient scales (powers of two)
        assert rel_l2((a + b).cpu().numpy(), f.cpu().numpy()) < 2e-5, f"L{i // 12}.{LAYER_TENSORS[i % 12]}"


def test_layernorm_backward_in_the_dgrad_epilogue_equals_the_two_launches(monkeypatch):
    """DEpiLnBwd (csrc/mst_train.h): LayerNorm1's backward behind the FFN1 dgrad product in one launch against the GEMM + k_ln_bwd pair
    (MST_FUSE_LN_BWD=0, read when the engine is created).  Same products and the same row arithmetic: dL/dh and every weight / bias
    gradient bit-identical; the LayerNorm-parameter and branch-bias sums are added tile by tile instead of block by block (1e-6)."""
    from mst_amd.engine import DenoiserEngine
    eng, w = big_engine()
    monkeypatch.setenv("MST_FUSE_LN_BWD", "0")
    two = DenoiserEngine(FE, T, B, device=_dev())
    two.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, layer_prefix="seqTransEncoder.layers.",
                        pe=torch.from_numpy(syn.positional_table(5000, 512)))
    h, r = stream(B)
    for p in (0.0, 0.1):
        out_a, d_a, g_a = engine_grads(eng, w, h, r, p, 77)
        out_b, d_b, g_b = engine_grads(two, w, h, r, p, 77)
        assert torch.equal(out_a, out_b) and torch.equal(d_a, d_b), p
        for i, (a, b) in enumerate(zip(g_a, g_b)):
            name = f"L{i // 12}.{LAYER_TENSORS[i % 12]}"
            if LAYER_TENSORS[i % 12] in ("norm1.weight", "norm1.bias", "self_attn.out_proj.bias"):
                assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-6, (name, p)
            else:
                assert torch.equal(a, b), (name, p)
    del two


@pytest.mark.parametrize("rows", [64, 37])
def test_frozen_stack_backward_fused_tail_vs_autograd(rows, monkeypatch):
    """Round 6: the backward pass of a FROZEN stack at batch size (grads=None: what the motion encoder's backward is in every fine-tune
    iteration) takes k_layer_tail_bwd<WG=false, LN2=true> (csrc/mst_tail_bwd.h): LayerNorm2 backward, FFN2 dgrad, GELU', FFN1 dgrad, LayerNorm1
    backward and the out-proj dgrad of a layer in ONE launch.  dL/dh against fp32 autograd with the engine's own dropout masks, with and
    without dropout, at 64 clips (197 full tiles) and at 37 clips (7289 rows: the last tile holds 57 of 64 tokens); bit-reproducible; and against
    engines built with the pieces switched off one by one (read when the engine is created): LayerNorm2 in a launch of its own
    (MST_TRAIN_FUSE_LN2_BWD=0), the three dgrad launches (MST_TRAIN_FUSE_BWD_TAIL=0), and the fused launch for the pass WITH parameter
    gradients too (=2, k_layer_tail_bwd<WG=true>): all 96 gradient tensors of that one against autograd."""
    from mst_amd.engine import DenoiserEngine
    eng, w = big_engine()
    h, r = stream(rows)

    def variant(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = DenoiserEngine(FE, T, B, device=_dev())
        for k in env:
            monkeypatch.delenv(k)
        e.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, layer_prefix="seqTransEncoder.layers.",
                          pe=torch.from_numpy(syn.positional_table(5000, 512)))
        return e

    others = {"LayerNorm2 backward in its own launch": variant(MST_TRAIN_FUSE_LN2_BWD="0"),
              "three dgrad launches": variant(MST_TRAIN_FUSE_BWD_TAIL="0"),
              "fused with parameter gradients": variant(MST_TRAIN_FUSE_BWD_TAIL="2")}
    for p in (0.0, 0.1):
        seed = 991 + rows
        params = layer_params(w, True)
        href = h.clone().requires_grad_(True)
        masks = engine_masks(eng, seed, p, rows, S) if p > 0 else None
        ref = torch_stack(href, params, masks)
        (ref * r).sum().backward()
        del masks, ref
        want = href.grad.cpu().numpy()
        out, tape = eng.train_forward(h, p, seed)
        d_frozen = eng.train_backward(tape, r, p, seed, None)
        err = rel_l2(d_frozen.cpu().numpy(), want)
        print(f"{rows} clips, dropout {p}: frozen backward (fused tail) vs autograd {err:.2e}")
        assert err <= TOL_GRAD, (rows, p, err)
        assert torch.equal(eng.train_backward(tape, r, p, seed, None), d_frozen), "not bit-reproducible"
        for name, e in others.items():
            out_v, tape_v = e.train_forward(h, p, seed)
            assert torch.equal(out_v, out), name
            d_v = e.train_backward(tape_v, r, p, seed, None)
            ev, ed = rel_l2(d_v.cpu().numpy(), want), rel_l2(d_v.cpu().numpy(), d_frozen.cpu().numpy())
            print(f"    {name}: vs autograd {ev:.2e}, vs the default {ed:.2e}")
            assert ev <= TOL_GRAD and ed < 1.5e-3, (name, rows, p, ev, ed)
        e = others["fused with parameter gradients"]
        out_v, tape_v = e.train_forward(h, p, seed)
        grads = [torch.zeros_like(q) for q in layer_params(w, False)]
        d_v = e.train_backward(tape_v, r, p, seed, grads)
        assert rel_l2(d_v.cpu().numpy(), want) <= TOL_GRAD
        errs = {f"L{i // 12}.{LAYER_TENSORS[i % 12]}": rel_l2(g.cpu().numpy(), q.grad.cpu().numpy()) for i, (g, q) in enumerate(zip(grads, params))}
        worst = max(errs, key=errs.get)
        print(f"    fused with parameter gradients: worst of 96 tensors {worst} {errs[worst]:.2e}")
        assert errs[worst] <= TOL_GRAD, (worst, errs[worst])
    del others


def test_finetune_objective_at_64_clips_is_finite_and_seeded():
    """One fine-tune iteration exactly as bench.py --mode finetune / train/training_loop.py:249-263 issue it: the 64-clip
    text-to-motion call, the 6 chained single-clip DDIM steps, the frozen motion encoder, backward, fused AdamW."""
    import contextlib
    import io
    import types
    from mst_amd.model.mdm_forstyledataset import StyleDiffusion
    from mst_amd.optim import FusedAdamW
    from mst_amd.utils import model_util
    dev, seed = _dev(), 20261003
    a = types.SimpleNamespace(dataset="humanml", latent_dim=512, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                              emb_trans_dec=False, diffusion_steps=1000, noise_schedule="cosine", sigma_small=True,
                              lambda_vel=0.0, lambda_rcxyz=0.0, lambda_fc=0.0)

    def run():
        with contextlib.redirect_stdout(io.StringIO()):
            model, d_ddim, _ = model_util.creat_serval_diffusion(a, StyleDiffusion, "ddim20")
        sd = {k: torch.from_numpy(np.ascontiguousarray(syn.tensor_for(seed, k, tuple(v.shape)))) for k, v in model.state_dict().items()
              if not k.endswith(".pe") and "clip_model" not in k}
        model.load_state_dict(sd, strict=False)
        model = model.to(dev).train()
        to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        t2m = to(syn.normal(seed, "ft/t2m", (B, FE, 1, T)))
        content, style = to(syn.normal(seed, "ft/content", (1, FE, 1, T))), to(syn.normal(seed, "ft/style", (1, FE, 1, T)))
        emb = to(syn.normal(seed, "ft/text", (1, 512)))
        y1 = {"y": {"text": ["a"], "text_embed": emb, "mask": torch.ones(1, 1, 1, T, device=dev),
                    "inpainting_mask": to(syn.root_horizontal_mask(1, FE, T)), "inpainted_motion": content}}
        yB = {"y": {"text": ["a"] * B, "text_embed": emb.expand(B, -1).contiguous(), "mask": torch.ones(B, 1, 1, T, device=dev),
                    "inpainting_mask": to(syn.root_horizontal_mask(B, FE, T)), "inpainted_motion": t2m}}
        opt = FusedAdamW(model.parameters_wo_enc(), lr=1e-5, weight_decay=0.0)
        torch.manual_seed(seed)
        np.random.seed(seed % (2 ** 31))
        tt = torch.randint(0, 6, (B,), generator=torch.Generator(device="cpu").manual_seed(seed)).to(dev)
        model.zero_grad()
        terms = d_ddim.few_shot_style_finetune_losses(model, t2m, tt, content, style, skip_steps=700, model_kwargs=y1,
                                                      model_t2m_kwargs=yB, semantic_guidance=1, use_ddim=1, Ls=10)
        terms["loss"].backward()
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        opt.step()
        norms = opt.last_sq_norms.clone()
        return {k: v.detach().clone() for k, v in terms.items()}, grads, norms

    t1, g1, n1 = run()
    t2, g2, n2 = run()
    assert len(g1) == 96
    for k in ("loss", "rot_mse", "text_cosine"):
        assert torch.isfinite(t1[k]).all(), k
        assert torch.equal(t1[k], t2[k]), k                       # same seeds -> same dropout, same noise, same reductions
    for n in g1:
        assert torch.isfinite(g1[n]).all() and float(g1[n].abs().max()) > 0, n
        assert torch.equal(g1[n], g2[n]), n
    assert torch.isfinite(n1).all() and torch.equal(n1, n2)
