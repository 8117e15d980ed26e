"""GPU parity of the native TRAINING path of the trainable encoder stack (mst_train_forward /
mst_train_backward, include/mst_engine.h) against a plain PyTorch fp32 implementation of the same
eight post-norm layers (nn.TransformerEncoderLayer semantics, model/mdm_forstyledataset.py:539-546 of
the reference) evaluated on the GPU with torch autograd.

Dropout is checked EXACTLY: the engine's counter-based keep masks are read back through
mst_dropout_mask and applied at the same four sites of the PyTorch layers, so forward and backward
can be compared value by value at p = 0.1 (the reference's training setting).

Tolerances: the engine multiplies with f16 MFMA operands (fp32 accumulation); the forward bar is the
north_star's 1e-3 relative L2; gradients pass through ~2x as many rounded products (recomputed
probabilities, dgrad and wgrad operands): measured <= 7e-4 on every one of the 96 tensors and on dL/dh
(tools/train_errs.py), bar 1.5e-3 relative L2 per tensor."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mst_amd  # noqa: F401
from mst_amd import synthetic as syn
from mst_amd.engine import LAYER_TENSORS
from conftest import SEED, rel_l2

from torch_reference import use_native, use_torch_ops

pytestmark = pytest.mark.gpu

TOL_FWD = 1e-3
TOL_GRAD = 1.5e-3
SHAPES = {"xia": (181, 76), "hml": (263, 196)}
L, D, H = 8, 512, 4


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


_ENGINES = {}


@pytest.fixture(autouse=True, params=["2048", "0"], ids=["small-tiles", "large-tiles"])
def tile_path(request, monkeypatch):
    """Every test here runs twice: these shapes (a few clips) take the small-launch path by default; MST_SMALL_M=0 (read when
    an engine is created) sends them through the kernels the batch-64 fine-tune pass uses."""
    monkeypatch.setenv("MST_SMALL_M", request.param)
    return request.param


def engine_for(tag, max_rows=4):
    import os
    from mst_amd.engine import DenoiserEngine
    tag_key = (tag, os.environ.get("MST_SMALL_M"))
    if tag_key not in _ENGINES:
        Fe, T = SHAPES[tag]
        eng = DenoiserEngine(Fe, T, max_rows, device=_dev())
        w = syn.denoiser_state(SEED, Fe, layer_prefix="seqTransEncoder.layers.")
        eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, layer_prefix="seqTransEncoder.layers.",
                            pe=torch.from_numpy(syn.positional_table(5000, 512)))
        _ENGINES[tag_key] = (eng, w)
    return _ENGINES[tag_key]


def layer_params(w, requires_grad):
    out = []
    for i in range(L):
        for k in LAYER_TENSORS:
            t = torch.from_numpy(w[f"seqTransEncoder.layers.{i}.{k}"]).to(_dev()).clone()
            out.append(t.requires_grad_(requires_grad))
    return out


def torch_stack(h, params, masks=None):
    """Eight post-norm encoder layers in fp32; h: [B, S, 512].  masks[l] = (m0 [B,4,S,S], m1 [B,S,512],
    m2 [B,S,1024], m3 [B,S,512]) keep-multipliers or None."""
    B, S, _ = h.shape
    x = h
    for l in range(L):
        win, bin_, wout, bout, w1, b1, w2, b2, g1, be1, g2, be2 = params[12 * l:12 * l + 12]
        m = masks[l] if masks is not None else (None,) * 4
        qkv = x @ win.t() + bin_
        q, k, v = qkv.view(B, S, 3, H, D // H).permute(2, 0, 3, 1, 4)          # [B, H, S, hd]
        p = torch.softmax((q * (D // H) ** -0.5) @ k.transpose(-1, -2), dim=-1)
        if m[0] is not None:
            p = p * m[0]
        att = (p @ v).permute(0, 2, 1, 3).reshape(B, S, D)
        o = att @ wout.t() + bout
        if m[1] is not None:
            o = o * m[1]
        x1 = F.layer_norm(x + o, (D,), g1, be1, 1e-5)
        hid = F.gelu(x1 @ w1.t() + b1)
        if m[2] is not None:
            hid = hid * m[2]
        f = hid @ w2.t() + b2
        if m[3] is not None:
            f = f * m[3]
        x = F.layer_norm(x1 + f, (D,), g2, be2, 1e-5)
    return x


def stream_input(tag, rows):
    Fe, T = SHAPES[tag]
    S = T + 1
    h = syn.normal(SEED, f"train/{tag}/h", (rows, S, D)).astype(np.float32)
    r = syn.normal(SEED, f"train/{tag}/r", (rows, S, D)).astype(np.float32)
    return S, torch.from_numpy(h).to(_dev()), torch.from_numpy(r).to(_dev())


def engine_masks(eng, seed, p, rows, S):
    out = []
    for l in range(L):
        out.append((eng.dropout_mask(seed, l, 0, p, rows * H * S * S).view(rows, H, S, S),
                    eng.dropout_mask(seed, l, 1, p, rows * S * D).view(rows, S, D),
                    eng.dropout_mask(seed, l, 2, p, rows * S * 1024).view(rows, S, 1024),
                    eng.dropout_mask(seed, l, 3, p, rows * S * D).view(rows, S, D)))
    return out


def test_dropout_mask_statistics():
    eng, _ = engine_for("xia")
    p, n = 0.1, 1 << 20
    m = eng.dropout_mask(1234, 0, 2, p, n)
    vals = torch.unique(m).cpu().numpy()
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.0 / (1.0 - p)) < 1e-6
    keep = float((m > 0).float().mean())
    assert abs(keep - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n), keep          # 4 sigma
    assert torch.equal(m, eng.dropout_mask(1234, 0, 2, p, n))                  # a pure function of (seed, site, index)
    for other in (eng.dropout_mask(1235, 0, 2, p, n), eng.dropout_mask(1234, 1, 2, p, n), eng.dropout_mask(1234, 0, 3, p, n)):
        agree = float(((m > 0) == (other > 0)).float().mean())
        assert abs(agree - (p * p + (1 - p) ** 2)) < 5e-3, agree              # independent streams
    assert float(eng.dropout_mask(7, 3, 1, 0.0, 4096).min()) == 1.0            # p = 0 keeps everything


@pytest.mark.parametrize("tag,rows", [("xia", 3), ("hml", 2)])
def test_train_forward_p0(tag, rows):
    eng, w = engine_for(tag)
    S, h, _ = stream_input(tag, rows)
    out, _ = eng.train_forward(h, 0.0, 0)
    with torch.no_grad():
        ref = torch_stack(h, layer_params(w, False))
    err = rel_l2(out.cpu().numpy(), ref.cpu().numpy())
    assert err <= TOL_FWD, err


@pytest.mark.parametrize("tag,rows,p", [("xia", 3, 0.0), ("hml", 2, 0.0), ("xia", 3, 0.1), ("hml", 2, 0.1)])
def test_train_backward_vs_autograd(tag, rows, p):
    eng, w = engine_for(tag)
    S, h, r = stream_input(tag, rows)
    seed = 0x1234ABCD5678
    # reference: autograd through the fp32 PyTorch layers with the engine's masks
    params = layer_params(w, True)
    href = h.clone().requires_grad_(True)
    masks = engine_masks(eng, seed, p, rows, S) if p > 0 else None
    ref = torch_stack(href, params, masks)
    (ref * r).sum().backward()
    # engine
    out, tape = eng.train_forward(h, p, seed)
    grads = [torch.zeros_like(q) for q in params]
    d_in = eng.train_backward(tape, r, p, seed, grads)
    errs = {"out": rel_l2(out.cpu().numpy(), ref.detach().cpu().numpy()),
            "d_in": rel_l2(d_in.cpu().numpy(), href.grad.cpu().numpy())}
    assert errs["out"] <= TOL_FWD, errs
    for i, (g, q) in enumerate(zip(grads, params)):
        errs[f"L{i // 12}.{LAYER_TENSORS[i % 12]}"] = rel_l2(g.cpu().numpy(), q.grad.cpu().numpy())
    worst = max(errs, key=errs.get)
    assert errs[worst] <= TOL_GRAD, (worst, errs[worst], {k: round(v, 5) for k, v in errs.items()})


def test_train_backward_accumulates_and_rescales():
    """Gradient buffers are accumulated into (+=), and the device-side rescaling makes tiny upstream
    gradients (2^-30 ~ 1e-9) exactly as accurate as O(1) ones: f16 operands never see the raw magnitude
    (a power-of-two factor changes no rounding, so the results agree to fp32 round-off)."""
    eng, w = engine_for("xia")
    S, h, r = stream_input("xia", 2)
    out, tape = eng.train_forward(h, 0.0, 0)
    params = layer_params(w, False)
    g1 = [torch.zeros_like(q) for q in params]
    d1 = eng.train_backward(tape, r, 0.0, 0, g1)
    g2 = [g.clone() for g in g1]
    tiny = 2.0 ** -30
    d2 = eng.train_backward(tape, r * tiny, 0.0, 0, g2)
    assert rel_l2((d2 / tiny).cpu().numpy(), d1.cpu().numpy()) < 1e-6
    for a, b in zip(g1, g2):
        assert rel_l2(b.cpu().numpy(), (a * (1 + tiny)).cpu().numpy()) < 1e-6
    g3 = [g.clone() for g in g1]
    eng.train_backward(tape, r, 0.0, 0, g3, need_input_grad=False)
    for a, b in zip(g1, g3):
        assert rel_l2(b.cpu().numpy(), (2 * a).cpu().numpy()) < 1e-5


def test_train_argument_errors():
    eng, _ = engine_for("xia")
    S, h, r = stream_input("xia", 2)
    with pytest.raises(RuntimeError, match="dropout probability"):
        eng.train_forward(h, 1.0, 0)
    with pytest.raises(RuntimeError, match="exceed the engine capacity"):
        eng.train_forward(torch.zeros(64, S, D, device=_dev()), 0.0, 0)
    out, tape = eng.train_forward(h, 0.0, 0)
    with pytest.raises(ValueError, match="gradient buffers"):
        eng.train_backward(tape, r, 0.0, 0, [torch.zeros(1, device=_dev())])


# ------------------------------------------------------------------------------ through the model boundary
def _style_model(dropout=0.1):
    from mst_amd.model.mdm_forstyledataset import StyleDiffusion
    m = StyleDiffusion("", 181, 1, 1, True, "rot6d", True, True, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4,
                       dropout=dropout, activation="gelu", data_rep="hml_vec", cond_mode="text", cond_mask_prob=0.0,
                       arch="trans_enc", dataset="stylexia_posrot")
    sd = {k: torch.from_numpy(syn.tensor_for(3, k, tuple(v.shape)).copy()) for k, v in m.state_dict().items()
          if not k.endswith(".pe") and "clip_model" not in k}
    m.load_state_dict(sd, strict=False)
    return m.to(_dev())


def _model_batch(B=3, T=76):
    x = torch.from_numpy(syn.normal(3, "tr/x", (B, 181, 1, T))).to(_dev())
    t = torch.tensor([5, 400, 900][:B], device=_dev())
    y = {"text_embed": torch.from_numpy(syn.normal(3, "tr/emb", (B, 512))).to(_dev())}
    tgt = torch.from_numpy(syn.normal(3, "tr/tgt", (B, 181, 1, T))).to(_dev())
    return x, t, y, tgt


def test_model_autograd_native_vs_torch_ops():
    """StyleDiffusion.forward inside an autograd graph: the native stack node (default) against the same
    module evaluated with torch ops (tests/torch_reference.py), eval mode (no dropout)."""
    m = _style_model().eval()
    x, t, y, tgt = _model_batch()
    res = {}
    for backend in ("native", "torch"):
        (use_torch_ops if backend == "torch" else use_native)(m)
        m.zero_grad()
        xin = x.clone().requires_grad_(True)
        out = m(xin, t, y=y)
        loss = ((out - tgt) ** 2).mean()
        loss.backward()
        res[backend] = (out.detach(), float(loss), xin.grad.clone(),
                        {n: p.grad.clone() for n, p in m.named_parameters() if p.requires_grad})
    use_native(m)
    assert rel_l2(res["native"][0].cpu().numpy(), res["torch"][0].cpu().numpy()) <= TOL_FWD
    assert abs(res["native"][1] - res["torch"][1]) <= 1e-3 * abs(res["torch"][1])
    assert rel_l2(res["native"][2].cpu().numpy(), res["torch"][2].cpu().numpy()) <= TOL_GRAD
    assert len(res["native"][3]) == 96
    for n, g in res["torch"][3].items():
        assert rel_l2(res["native"][3][n].cpu().numpy(), g.cpu().numpy()) <= TOL_GRAD, n


def test_model_training_mode_dropout_is_seeded():
    m = _style_model().train()
    x, t, y, tgt = _model_batch()

    def run(seed):
        torch.manual_seed(seed)
        m.zero_grad()
        loss = ((m(x, t, y=y) - tgt) ** 2).mean()
        loss.backward()
        g = m.seqTransEncoder.layers[3].linear1.weight.grad.clone()
        return float(loss), g

    a, b, c = run(5), run(5), run(6)
    assert a[0] == b[0] and torch.equal(a[1], b[1])                 # same torch seed -> same masks
    assert a[0] != c[0]
    assert torch.isfinite(a[1]).all() and float(a[1].abs().max()) > 0
    m.eval()
    with torch.no_grad():
        e1, e2 = m(x, t, y=y), m(x, t, y=y)
    assert torch.equal(e1, e2)                                       # inference path: no dropout
    with pytest.raises(RuntimeError, match="GPU only"):
        _style_model().cpu().train()(x.cpu(), t.cpu(), y={"text_embed": y["text_embed"].cpu()})


def test_native_gradients_reach_a_bucket_reducer():
    """GradSink adds the accumulated gradients into existing p.grad tensors (here: views of the reducer's per-layer
    buckets) at the end of backward and notifies the reducer, which fires its buckets last layer first."""
    from mst_amd.finetune_dp import LayerBucketReducer
    m = _style_model().eval()
    x, t, y, tgt = _model_batch()

    def step():
        loss = ((m(x, t, y=y) - tgt) ** 2).mean() + ((m(x * 0.5, t, y=y) - tgt) ** 2).mean()     # two passes, one backward
        loss.backward()

    m.zero_grad()
    step()
    plain = {n: p.grad.clone() for n, p in m.named_parameters() if p.requires_grad}
    red = LayerBucketReducer(m)
    red.zero_grad()
    step()
    red.finish()
    assert red.launch_order == list(range(7, -1, -1))
    assert red.launched_in == ["backward"] * 8
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert rel_l2(p.grad.cpu().numpy(), plain[n].cpu().numpy()) < 1e-6, n
    # a second backward before zero_grad() would add into buckets whose exchange has already been launched (ranks would silently
    # diverge): the reducer refuses it
    with pytest.raises(RuntimeError, match="second backward pass"):
        step()
    red.zero_grad()
    step()
    red.finish()
    g = m.seqTransEncoder.layers[5].linear2.weight.grad
    assert rel_l2(g.cpu().numpy(), plain["seqTransEncoder.layers.5.linear2.weight"].cpu().numpy()) < 1e-6


def test_motion_encoder_masked_stack_native_vs_torch_ops():
    """The frozen MotionEncoder (mdm_forstyledataset.py:90-124) inside an autograd graph: key-padding-masked stack
    through the native node (input gradient only) against torch ops with src_key_padding_mask."""
    m = _style_model().eval()
    enc = m.motion_enc
    x, t, y, tgt = _model_batch()
    T_ = x.shape[-1]
    fm = torch.ones(3, 1, 1, T_, device=_dev())
    fm[1, ..., T_ - 9:] = 0                                   # ragged lengths
    fm[2, ..., 40:] = 0
    yy = {"mask": fm, "text_embed": y["text_embed"]}
    w = torch.from_numpy(syn.normal(3, "tr/w", (3, 512))).to(_dev())
    res = {}
    for backend in ("native", "torch"):
        (use_torch_ops if backend == "torch" else use_native)(m)
        xin = x.clone().requires_grad_(True)
        mu, _ = enc(xin, y=yy)
        (mu * w).sum().backward()
        res[backend] = (mu.detach(), xin.grad.clone())
    use_native(m)
    assert rel_l2(res["native"][0].cpu().numpy(), res["torch"][0].cpu().numpy()) <= TOL_FWD
    assert rel_l2(res["native"][1].cpu().numpy(), res["torch"][1].cpu().numpy()) <= TOL_GRAD
    assert all(p.grad is None for p in enc.parameters())      # frozen: no parameter gradient is produced
    # frames behind the padding never influence mu: their input gradient is exactly zero on both paths
    assert float(res["native"][1][1, :, :, T_ - 9:].abs().max()) == 0.0
    assert float(res["native"][1][2, :, :, 40:].abs().max()) == 0.0


@pytest.mark.parametrize("rows,S,p", [(1, 2, 0.0), (1, 33, 0.3), (4, 64, 0.0), (2, 77, 0.5)])
def test_train_edge_shapes(rows, S, p):
    """Smallest sequence (2 tokens), tile-boundary lengths (33, 64), one clip, heavy dropout: forward, input gradient and a
    few parameter gradients against fp32 autograd with the engine's masks; frozen mode (grads=None) returns the same dL/dh."""
    eng, w = engine_for("xia")
    h = torch.from_numpy(syn.normal(SEED, f"edge/h/{rows}/{S}", (rows, S, D))).to(_dev())
    r = torch.from_numpy(syn.normal(SEED, f"edge/r/{rows}/{S}", (rows, S, D))).to(_dev())
    seed = 77 + S
    params = layer_params(w, True)
    href = h.clone().requires_grad_(True)
    masks = engine_masks(eng, seed, p, rows, S) if p > 0 else None
    ref = torch_stack(href, params, masks)
    (ref * r).sum().backward()
    out, tape = eng.train_forward(h, p, seed)
    grads = [torch.zeros_like(q) for q in params]
    d_in = eng.train_backward(tape, r, p, seed, grads)
    assert rel_l2(out.cpu().numpy(), ref.detach().cpu().numpy()) <= TOL_FWD
    assert rel_l2(d_in.cpu().numpy(), href.grad.cpu().numpy()) <= TOL_GRAD
    for i in (0, 1, 2, 4, 6, 7, 8, 11, 84 + 0, 84 + 5, 84 + 10):
        assert rel_l2(grads[i].cpu().numpy(), params[i].grad.cpu().numpy()) <= 2 * TOL_GRAD, (i, LAYER_TENSORS[i % 12])
    # frozen mode (grads=None: the motion encoder's backward).  On the large-tile path it takes the fused backward tail (round 6,
    # csrc/mst_tail_bwd.h) while the pass with parameter gradients keeps the three dgrad launches: d hid stays fp32 until GELU' is applied
    # there, the unfused epilogue rounds it to f16 first -- the two agree to f16 rounding, and each has to hold the autograd tolerance
    d_in2 = eng.train_backward(tape, r, p, seed, None)
    assert rel_l2(d_in2.cpu().numpy(), href.grad.cpu().numpy()) <= TOL_GRAD
    assert rel_l2(d_in2.cpu().numpy(), d_in.cpu().numpy()) < 1.5e-3


@pytest.mark.parametrize("rows,p", [(14, 0.1), (20, 0.1)])
def test_train_small_launch_tile_heights(rows, p):
    """The small-launch GEMMs pick their tile height by the row count (engine: launch_rows_gemm): 16 tokens up to 800 stream rows (the
    shapes above), 64 tokens up to 1300 (14 clips x 77 = 1078 rows), 32 tokens above (20 x 77 = 1540 rows) -- the training FFN1 mode
    (tape pre-activation + dropout) included.  Forward, input gradient and parameter gradients against fp32 autograd with the engine's masks."""
    from mst_amd.engine import DenoiserEngine
    S = 77
    eng = DenoiserEngine(181, 76, rows, device=_dev())
    w = syn.denoiser_state(SEED, 181, layer_prefix="seqTransEncoder.layers.")
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, layer_prefix="seqTransEncoder.layers.",
                        pe=torch.from_numpy(syn.positional_table(5000, 512)))
    h = torch.from_numpy(syn.normal(SEED, f"tiles/h/{rows}", (rows, S, D))).to(_dev())
    r = torch.from_numpy(syn.normal(SEED, f"tiles/r/{rows}", (rows, S, D))).to(_dev())
    seed = 1234 + rows
    params = layer_params(w, True)
    href = h.clone().requires_grad_(True)
    ref = torch_stack(href, params, engine_masks(eng, seed, p, rows, S))
    (ref * r).sum().backward()
    out, tape = eng.train_forward(h, p, seed)
    grads = [torch.zeros_like(q) for q in params]
    d_in = eng.train_backward(tape, r, p, seed, grads)
    assert rel_l2(out.cpu().numpy(), ref.detach().cpu().numpy()) <= TOL_FWD
    assert rel_l2(d_in.cpu().numpy(), href.grad.cpu().numpy()) <= TOL_GRAD
    for i in (0, 2, 4, 6, 8, 84 + 0, 84 + 4, 84 + 6):
        assert rel_l2(grads[i].cpu().numpy(), params[i].grad.cpu().numpy()) <= 2 * TOL_GRAD, (i, LAYER_TENSORS[i % 12])


def test_train_key_padding_mask_engine_level():
    """mst_train_forward / backward with key_keep: padded keys get no attention and no K/V gradient; vs torch autograd."""
    eng, w = engine_for("xia")
    rows, S = 3, 50
    h = torch.from_numpy(syn.normal(SEED, "kpm/h", (rows, S, D))).to(_dev())
    r = torch.from_numpy(syn.normal(SEED, "kpm/r", (rows, S, D))).to(_dev())
    keep = torch.ones(rows, S, dtype=torch.bool, device=_dev())
    keep[0, 31:] = False
    keep[2, 5:] = False
    params = layer_params(w, True)
    href = h.clone().requires_grad_(True)
    x = href
    import torch.nn as nn
    layer = nn.TransformerEncoderLayer(d_model=D, nhead=H, dim_feedforward=1024, dropout=0.0, activation="gelu")
    enc = nn.TransformerEncoder(layer, num_layers=L, enable_nested_tensor=False).to(_dev()).train()
    enc.load_state_dict({k[len("seqTransEncoder."):]: torch.from_numpy(v) for k, v in w.items() if k.startswith("seqTransEncoder.")})
    ref = enc(x.permute(1, 0, 2), src_key_padding_mask=~keep).permute(1, 0, 2)
    valid = keep[:, :, None].float()                      # outputs at padded positions are unconstrained: score the real ones
    (ref * r * valid).sum().backward()
    out, tape = eng.train_forward(h, 0.0, 0, key_keep=keep)
    grads = [torch.zeros_like(q) for q in params]
    d_in = eng.train_backward(tape, r * valid, 0.0, 0, grads, key_keep=keep)
    assert rel_l2((out * valid).cpu().numpy(), (ref.detach() * valid).cpu().numpy()) <= TOL_FWD
    assert rel_l2(d_in.cpu().numpy(), href.grad.cpu().numpy()) <= TOL_GRAD
    ref_g = dict(enc.named_parameters())
    for i, k in enumerate(LAYER_TENSORS):
        assert rel_l2(grads[i].cpu().numpy(), ref_g[f"layers.0.{k}"].grad.cpu().numpy()) <= 2 * TOL_GRAD, k
    with pytest.raises(ValueError, match="key_keep must be"):
        eng.train_forward(h, 0.0, 0, key_keep=keep[:, :10])


def test_second_backward_is_refused():
    m = _style_model().eval()
    x, t, y, tgt = _model_batch()
    loss = ((m(x, t, y=y) - tgt) ** 2).mean()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="a second time"):
        loss.backward()


def test_failed_backward_does_not_poison_the_gradient_sink():
    """the autograd engine drops queued callbacks when a node raises, so a backward pass that dies after the
    sink was opened must not leave it open: the next, normal backward on the SAME model has to deliver the gradients a
    fresh model gets."""
    x, t, y, tgt = _model_batch()

    def grads_of(m):
        m.zero_grad()
        ((m(x, t, y=y) - tgt) ** 2).mean().backward()
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    fresh = grads_of(_style_model().eval())
    m = _style_model().eval()
    # (1) the refused second backward
    loss = ((m(x, t, y=y) - tgt) ** 2).mean()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="a second time"):
        loss.backward()
    # (2) an error raised INSIDE the native backward call, after the sink was opened
    out = m(x, t, y=y)
    eng = m.mst_engine(x.shape[0], x.shape[-1])
    orig = eng.train_model_backward
    eng.train_model_backward = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("injected failure"))
    try:
        with pytest.raises(RuntimeError, match="injected failure"):
            ((out - tgt) ** 2).mean().backward()
    finally:
        eng.train_model_backward = orig
    sink = m.__dict__.get("_mst_grad_sink")
    assert sink is not None and not sink.active
    again = grads_of(m)
    assert set(again) == set(fresh) and len(fresh) >= 96
    for n in fresh:
        assert rel_l2(again[n].cpu().numpy(), fresh[n].cpu().numpy()) < 1e-6, n


def _cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_dev())


def test_chained_calls_share_one_backward_pass(monkeypatch):
    """The fine-tune objective's chained x0-hat steps (reference gaussian_diffusion.py:1364-1378: inputs cut with x.detach()) write the
    clips of ONE tape and are differentiated in ONE native pass (native_stack.ChainedCalls, mst_train_model_forward's clip0 /
    tape_clips).  (i) engine level, dropout 0.1: the pass over the shared tape == the sum of passes over each clip alone in a tape of
    the same layout (the same masks: the dropout counters are offset by the clip); (ii) through autograd, dropout 0: all 96 gradients
    with the chain == without it (MST_CHAIN=0)."""
    from mst_amd.engine import DenoiserEngine
    F, T, n = 181, 76, 3
    eng = DenoiserEngine(F, T, 4, device=_dev())
    w = syn.denoiser_state(SEED, F)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
    eng.set_text(_cu(syn.normal(SEED, "ch/txt", (1, 512))))
    xs = [_cu(syn.normal(SEED, f"ch/x{k}", (1, F, 1, T))) for k in range(n)]
    ts = [torch.tensor([t], device=_dev()) for t in (950, 500, 50)]
    dout = _cu(syn.normal(SEED, "ch/dout", (n, F, 1, T)))
    shapes = [tuple(torch.from_numpy(w[f"seqTransEncoder.layers.{l}.{k}"]).shape) for l in range(8) for k in LAYER_TENSORS]
    zeros = lambda: [torch.zeros(s, device=_dev()) for s in shapes]
    seed, p = 123456789, 0.1
    # one pass over the shared tape
    tape = eng.train_tape(n, T + 1, zero=True)
    outs = [eng.train_model_forward(xs[k], ts[k], p, p, seed, tape=tape, clip0=k, tape_clips=n)[0] for k in range(n)]
    g_all = zeros()
    eng.train_model_backward(tape, dout, p, p, seed, g_all, need_input_grad=False)
    # each clip alone in a tape of the same layout, the other clips' gradient zero
    g_sum = zeros()
    for k in range(n):
        tk = eng.train_tape(n, T + 1, zero=True)
        o, _ = eng.train_model_forward(xs[k], ts[k], p, p, seed, tape=tk, clip0=k, tape_clips=n)
        assert torch.equal(o, outs[k])
        dk = torch.zeros_like(dout)
        dk[k] = dout[k]
        eng.train_model_backward(tk, dk, p, p, seed, g_sum, need_input_grad=False)
    worst = max(rel_l2(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(g_all, g_sum))
    print("chained backward vs per-clip passes, worst tensor", worst)
    assert worst < 2e-3                    # (f16 wgrad operands scaled by ONE power of two for the pass vs one per clip)
    # a clip of the shared tape == the same call with a tape of its own when dropout is off
    o_own, _ = eng.train_model_forward(xs[1], ts[1], 0.0, 0.0, 0)
    t0 = eng.train_tape(n, T + 1, zero=True)
    o_slot, _ = eng.train_model_forward(xs[1], ts[1], 0.0, 0.0, 0, tape=t0, clip0=1, tape_clips=n)
    assert torch.equal(o_own, o_slot)


def test_training_call_head_variants_give_the_same_forward(monkeypatch):
    """Round 6 (docs/LAB_NOTES.md R6.17): the head of a training model call lost three launches -- the timestep MLP (its output for EVERY timestep
    is a table, MST_TEMB_TABLE), PositionalEncoding's dropout (inside the embedding kernel's epilogue, MST_TRAIN_FUSE_PE_DROP), and mask_cond's
    elementwise masking (the Bernoulli mask scales the row inside the text projection, mst_set_text_dropped).  Engines built with each switched
    off (read when the engine is created): the table changes NO bit (same kernels, same rows), the fused dropout only the rounding of the
    stream's lo halves (it now acts before the hi / lo split), and the dropped-text projection equals the projection of the masked embedding."""
    from mst_amd.engine import DenoiserEngine
    F, T, B = 181, 76, 3
    w = syn.denoiser_state(SEED, F)

    def build(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = DenoiserEngine(F, T, 4, device=_dev())
        for k in env:
            monkeypatch.delenv(k)
        e.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
        return e

    x = _cu(syn.normal(SEED, "head/x", (B, F, 1, T)))
    t = torch.tensor([950, 3, 412], device=_dev())
    txt = _cu(syn.normal(SEED, "head/txt", (B, 512)))
    drop = torch.tensor([0.0, 1.0, 0.0], device=_dev())
    seed, p = 20261005, 0.1
    res = {}
    for name, env in (("default", {}), ("no table", {"MST_TEMB_TABLE": "0"}), ("dropout launch", {"MST_TRAIN_FUSE_PE_DROP": "0"})):
        e = build(**env)
        e.set_text(txt, drop=drop)
        proj_dropped = e.debug_buffer("textproj", B, 512).clone()
        e.set_text(txt * (1.0 - drop).view(-1, 1))
        assert torch.equal(e.debug_buffer("textproj", B, 512), proj_dropped), name
        e.set_text(txt, drop=drop)
        out, tape = e.train_model_forward(x, t, p, p, seed)
        again, _ = e.train_model_forward(x, t, p, p, seed)
        assert torch.equal(out, again), name
        res[name] = out
    assert torch.equal(res["default"], res["no table"]), "the timestep-embedding table changed the forward pass"
    err = rel_l2(res["default"].cpu().numpy(), res["dropout launch"].cpu().numpy())
    print("PositionalEncoding dropout inside the embedding epilogue vs its own launch:", err)
    assert err < 2e-4, err
    # a reload of the (frozen) timestep MLP rebuilds the table
    e = build()
    e.set_text(txt)
    a, _ = e.train_model_forward(x, t, 0.0, 0.0, 0)
    w2 = dict(w)
    key = next(k for k in w if k.endswith("embed_timestep.time_embed.2.bias"))
    w2[key] = w[key] + 0.5
    e.load_state_dict({k: torch.from_numpy(v) for k, v in w2.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
    e.set_text(txt)
    b, _ = e.train_model_forward(x, t, 0.0, 0.0, 0)
    ref = build(MST_TEMB_TABLE="0")
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in w2.items()}, pe=torch.from_numpy(syn.positional_table(5000, 512)))
    ref.set_text(txt)
    c, _ = ref.train_model_forward(x, t, 0.0, 0.0, 0)
    assert not torch.equal(a, b) and torch.equal(b, c), "the table was not rebuilt behind a reload of the timestep MLP"
