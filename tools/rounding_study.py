"""Which f16 operand roundings carry the error on ill-conditioned weights?  CPU study with the oracle's arithmetic (oracle/denoiser.py
encoder_layer, restated here with one switch per product operand) on the stress weights of tests/test_gpu_parity.py
::test_forward_with_ill_conditioned_weights.  Test infrastructure only (imports oracle/).   python tools/rounding_study.py"""
import math, os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mst_amd  # noqa
from mst_amd import synthetic as syn
from oracle import denoiser

SEED = 20240917
H = torch.float16
OPS = ["qkv.x", "qkv.w", "qk.q", "qk.k", "pv.p", "pv.v", "out.x", "out.w", "ffn1.x", "ffn1.w", "ffn2.x", "ffn2.w"]


def rnd(x, on, cols=None):
    if not on:
        return x
    r = x.to(H).to(torch.float32)
    if cols is not None:           # these input channels keep full precision (the hi + lo remedy applied to them alone)
        r = r.clone(); r[..., cols] = x[..., cols]
    return r


def layer(x, w, p, on, keep):
    t = lambda k: torch.from_numpy(w[p + k])
    B, S, d = x.shape
    lin = lambda xx, W, b, tag, cols=None: rnd(xx, on[tag + ".x"], cols) @ rnd(t(W), on[tag + ".w"]).t() + t(b)
    qkv = lin(x, "self_attn.in_proj_weight", "self_attn.in_proj_bias", "qkv", keep.get("qkv"))
    q, k, v = [z.reshape(B, S, 4, 128).permute(0, 2, 1, 3) for z in qkv.split(d, dim=-1)]
    s = (rnd(q, on["qk.q"]) @ rnd(k, on["qk.k"]).transpose(-1, -2)) / math.sqrt(128)
    a = (rnd(torch.softmax(s, -1), on["pv.p"]) @ rnd(v, on["pv.v"])).permute(0, 2, 1, 3).reshape(B, S, d)
    a = lin(a, "self_attn.out_proj.weight", "self_attn.out_proj.bias", "out")
    x = F.layer_norm(x + a, (d,), t("norm1.weight"), t("norm1.bias"), 1e-5)
    hid = F.gelu(lin(x, "linear1.weight", "linear1.bias", "ffn1", keep.get("ffn1")))
    h = lin(hid, "linear2.weight", "linear2.bias", "ffn2")
    return F.layer_norm(x + h, (d,), t("norm2.weight"), t("norm2.bias"), 1e-5)


def forward(w, pe, x, t, txt, on, keep_fn=None, uncond=False):
    xs = denoiser.token_stream(w, pe, x, t, txt, uncond=uncond, dt=H)        # [B, S, d]; embedding GEMMs rounded as always
    for i in range(8):
        p = f"seqTransEncoder.layers.{i}."
        keep = keep_fn(w, i) if keep_fn else {}
        xs = layer(xs, w, p, on, keep)
    return denoiser._linear(xs[:, 1:], w[denoiser.PRIOR + "output_process.poseFinal.weight"], w[denoiser.PRIOR + "output_process.poseFinal.bias"], H)


def rel(a, b):
    return float((a - b).norm() / b.norm())


def stress(kind, F_):
    w = {k: v.copy() for k, v in syn.denoiser_state(SEED, F_, layer_prefix="seqTransEncoder.layers.").items()}
    rng = np.random.default_rng(7)
    for i in range(8):
        p = f"seqTransEncoder.layers.{i}."
        for k in ("norm1.weight", "norm2.weight"):
            idx = rng.choice(512, 4, replace=False)
            if kind == "ln_outliers":
                w[p + k][idx] *= 20.0
        if kind == "big_weights":
            for k, f in (("linear1.weight", 3.0), ("linear2.weight", 1.5), ("self_attn.in_proj_weight", 3.0)):
                w[p + k] *= f
        rng.normal(0, 1.0, 1024); rng.normal(0, 0.5, 1536)
    return w


def outlier_cols(w, i):
    """input channels of layer i's QKV GEMM (= norm2 of layer i - 1) and FFN1 GEMM (= norm1 of layer i) whose gain stands out"""
    def pick(g):
        g = np.abs(g); return np.nonzero(g > 4 * np.median(g))[0]
    keep = {"ffn1": pick(w[f"seqTransEncoder.layers.{i}.norm1.weight"])}
    if i > 0:
        keep["qkv"] = pick(w[f"seqTransEncoder.layers.{i - 1}.norm2.weight"])
    return keep


def guided(w, pe, x, t, txt, on, scale=2.5):
    """cfg_sampler.py:36-43: u + scale (c - u), both halves through the same rounding switches"""
    c = forward(w, pe, x, t, txt, on)
    u = forward(w, pe, x, t, txt, on, uncond=True)
    return u + scale * (c - u)


def cfg_study():
    """which operand sites carry the classifier-free-guidance error (scale 2.5 amplifies c - u)?  One guided
    forward at the HumanML shape, seeded weights: every site alone, all but it, and the cheapest candidates for a split (hi + lo)."""
    F_, T, B = 263, 196, 2
    x = syn.normal(SEED, "x/hml", (B, F_, 1, T)); txt = syn.normal(SEED, "txt/hml", (B, 512)); t = np.array([10, 900])
    pe = syn.positional_table(5000, 512)
    w = stress("seeded", F_)
    all_on = {k: True for k in OPS}; all_off = {k: False for k in OPS}
    ref = guided(w, pe, x, t, txt, all_off)
    one = rel(forward(w, pe, x, t, txt, all_on), forward(w, pe, x, t, txt, all_off))
    print(f"== classifier-free guidance (scale 2.5), HumanML shape: single forward all rounded {one:.2e}, guided all rounded {rel(guided(w, pe, x, t, txt, all_on), ref):.2e}")
    for k in OPS:
        only = dict(all_off); only[k] = True
        but = dict(all_on); but[k] = False
        print(f"   {k:7s} alone rounded {rel(guided(w, pe, x, t, txt, only), ref):.2e}    all but it {rel(guided(w, pe, x, t, txt, but), ref):.2e}")
    for name, sites in (("every weight exact (4 GEMMs' weights split: + 100 % of the layer MFMA work)", ["qkv.w", "out.w", "ffn1.w", "ffn2.w"]),
                        ("every GEMM activation exact (+ 100 %)", ["qkv.x", "out.x", "ffn1.x", "ffn2.x"]),
                        ("attention operands exact (+ 9 %)", ["qk.q", "qk.k", "pv.p", "pv.v"]),
                        ("out-proj both operands exact (+ 23 %)", ["out.x", "out.w"])):
        sw = dict(all_on)
        for k in sites:
            sw[k] = False
        print(f"   {name}: {rel(guided(w, pe, x, t, txt, sw), ref):.2e}")


if __name__ == "__main__":
    if "--cfg" in sys.argv:
        cfg_study()
        sys.exit(0)
    torch.manual_seed(0)
    F_, T, B = 181, 76, 2
    x = syn.normal(SEED, "x/xia", (B, F_, 1, T)); txt = syn.normal(SEED, "txt/xia", (B, 512)); t = np.array([10, 900])
    pe = syn.positional_table(5000, 512)
    for kind in ("seeded", "ln_outliers", "big_weights"):
        w = stress(kind, F_)
        X, TT = torch.from_numpy(x), torch.from_numpy(t)
        all_on = {k: True for k in OPS}; all_off = {k: False for k in OPS}
        ref = forward(w, pe, x, t, txt, all_off)
        print(f"== {kind}: all operands rounded {rel(forward(w, pe, x, t, txt, all_on), ref):.2e}")
        for k in OPS:
            only = dict(all_off); only[k] = True
            but = dict(all_on); but[k] = False
            print(f"   {k:7s} alone rounded {rel(forward(w, pe, x, t, txt, only), ref):.2e}    all but it {rel(forward(w, pe, x, t, txt, but), ref):.2e}")
        print(f"   all rounded, outlier input channels of QKV / FFN1 exact: {rel(forward(w, pe, x, t, txt, all_on, outlier_cols), ref):.2e}")
        wx = dict(all_on); wx["qkv.w"] = wx["ffn1.w"] = wx["ffn2.w"] = False
        print(f"   all rounded but the weights of QKV / FFN1 / FFN2: {rel(forward(w, pe, x, t, txt, wx), ref):.2e}")
        wx2 = dict(wx); wx2["qkv.x"] = wx2["ffn1.x"] = wx2["ffn2.x"] = False
        print(f"   ... and their activations: {rel(forward(w, pe, x, t, txt, wx2), ref):.2e}")
