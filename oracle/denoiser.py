"""Oracle: the StyleDiffusion / MDM denoiser forward on the CPU.  (test infrastructure)

Restates model/mdm_forstyledataset.py:602-625 (StyleDiffusion.forward), :315-364 (MDM.forward,
trans_enc branch), :387-404 (PositionalEncoding), :408-422 (TimestepEmbedder), :425-449
(InputProcess), :452-478 (OutputProcess), :592-600 (mask_cond) and :90-124 (MotionEncoder.forward),
plus the arithmetic of torch's post-norm nn.TransformerEncoderLayer as configured at :539-546
(4 heads, GELU(erf), LayerNorm eps 1e-5, norm_first=False), written out explicitly in batch-major
[B, S, d] layout.  CLIP is outside: the post-CLIP text embedding [B, clip_dim] is an input.

`operand_dtype` (None | torch.float16 | torch.bfloat16) rounds both operands of every dense
contraction to that type before an fp32 product -- a CPU model of MFMA operand rounding used to
budget the HIP path's error; None is the reference's fp32 arithmetic.
"""
import math

import torch
import torch.nn.functional as F

PRIOR = "motion_enc.mdm_model."
LAYERS = "seqTransEncoder.layers."


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(a)


def _rnd(x, dt):
    return x if dt is None else x.to(dt).to(torch.float32)


def _linear(x, w, b, dt=None):
    y = _rnd(x, dt) @ _rnd(_t(w), dt).t()
    return y + _t(b)


def encoder_layer(x, w, prefix, nhead, key_padding_mask=None, dt=None, trace=None):
    """One post-norm encoder layer on x [B, S, d].  `trace` (dict) receives the intermediates
    qkv / attn / x1 / hid / x2 for stage-by-stage kernel tests."""
    B, S, d = x.shape
    hd = d // nhead
    qkv = _linear(x, w[prefix + "self_attn.in_proj_weight"], w[prefix + "self_attn.in_proj_bias"], dt)
    q, k, v = qkv.split(d, dim=-1)

    def heads(z):
        return z.reshape(B, S, nhead, hd).permute(0, 2, 1, 3)

    q, k, v = heads(q), heads(k), heads(v)
    scores = (_rnd(q, dt) @ _rnd(k, dt).transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if key_padding_mask is not None:  # True = padded key
        scores = scores.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(scores, dim=-1)
    attn_heads = (_rnd(p, dt) @ _rnd(v, dt)).permute(0, 2, 1, 3).reshape(B, S, d)
    a = _linear(attn_heads, w[prefix + "self_attn.out_proj.weight"], w[prefix + "self_attn.out_proj.bias"], dt)
    x = F.layer_norm(x + a, (d,), _t(w[prefix + "norm1.weight"]), _t(w[prefix + "norm1.bias"]), 1e-5)
    hid = F.gelu(_linear(x, w[prefix + "linear1.weight"], w[prefix + "linear1.bias"], dt))
    h = _linear(hid, w[prefix + "linear2.weight"], w[prefix + "linear2.bias"], dt)
    x2 = F.layer_norm(x + h, (d,), _t(w[prefix + "norm2.weight"]), _t(w[prefix + "norm2.bias"]), 1e-5)
    if trace is not None:
        trace.update(qkv=qkv, attn=attn_heads, x1=x, hid=hid, x2=x2)
    return x2


def timestep_embedding(w, pe, t, prior=PRIOR):
    """time_embed(pe[t]) -> [B, d]   (:415-422).  `t` are ORIGINAL-process indices
    (after respace.py:129-131's timestep_map)."""
    e = _t(pe)[_t(t).long()]
    e = _linear(e, w[prior + "embed_timestep.time_embed.0.weight"], w[prior + "embed_timestep.time_embed.0.bias"])
    e = F.silu(e)
    return _linear(e, w[prior + "embed_timestep.time_embed.2.weight"], w[prior + "embed_timestep.time_embed.2.bias"])


def token_stream(w, pe, x, t, text_emb, uncond=False, cond_keep=None, prior=PRIOR, dt=None):
    """The [B, T+1, d] encoder input (conditioning token + embedded frames + positions)."""
    x = _t(x).float()
    B, Fe, one, T = x.shape
    pe = _t(pe)
    emb = timestep_embedding(w, pe, t, prior)
    te = _t(text_emb).float()
    if uncond:
        te = torch.zeros_like(te)
    elif cond_keep is not None:
        te = te * _t(cond_keep).float().view(B, 1)
    emb = emb + _linear(te, w[prior + "embed_text.weight"], w[prior + "embed_text.bias"])
    frames = x.permute(0, 3, 1, 2).reshape(B, T, Fe * one)
    h = _linear(frames, w[prior + "input_process.poseEmbedding.weight"],
                w[prior + "input_process.poseEmbedding.bias"], dt)
    return torch.cat([emb[:, None, :], h], dim=1) + pe[: T + 1][None]


def forward(w, pe, x, t, text_emb, uncond=False, cond_keep=None, nhead=4, num_layers=8,
            layer_prefix=LAYERS, prior=PRIOR, dt=None):
    """Denoiser forward.

    w         state dict (numpy or torch float32), reference key layout
    pe        [max_len, d] positional table
    x         [B, F, 1, T] float32
    t         [B] int, original-process timesteps
    text_emb  [B, clip_dim] post-CLIP embedding
    uncond    y.get('uncond'): zero the text embedding (mask_cond force_mask, :594-595)
    cond_keep optional [B] 0/1 keep-mask for the training-time Bernoulli cond mask (:596-598)
    returns   [B, F, 1, T]
    """
    x = _t(x).float()
    B, Fe, one, T = x.shape
    seq = token_stream(w, pe, x, t, text_emb, uncond, cond_keep, prior, dt)   # token 0 = conditioning
    for i in range(num_layers):
        seq = encoder_layer(seq, w, f"{layer_prefix}{i}.", nhead, None, dt)
    out = _linear(seq[:, 1:], w[prior + "output_process.poseFinal.weight"],
                  w[prior + "output_process.poseFinal.bias"], dt)       # [B, T, F]
    return out.reshape(B, T, Fe, one).permute(0, 2, 3, 1).contiguous()


def cfg_forward(w, pe, x, t, text_emb, scale, **kw):
    """ClassifierFreeSampleModel.forward (model/cfg_sampler.py:36-43)."""
    c = forward(w, pe, x, t, text_emb, uncond=False, **kw)
    u = forward(w, pe, x, t, text_emb, uncond=True, **kw)
    s = _t(scale).float().view(-1, 1, 1, 1)
    return u + s * (c - u)


def motion_encoder(w, pe, x, frame_mask, nhead=4, num_layers=8, enc_prefix="motion_enc.",
                   prior=PRIOR, dt=None):
    """MotionEncoder.forward (:90-124): mu token of the frozen 'semantic discriminator'.
    frame_mask [B, T] bool, True = real frame.  Returns mu [B, d]."""
    x = _t(x).float()
    B, Fe, one, T = x.shape
    frames = x.permute(0, 3, 1, 2).reshape(B, T, Fe * one)
    h = _linear(frames, w[prior + "input_process.poseEmbedding.weight"],
                w[prior + "input_process.poseEmbedding.bias"], dt)
    mu_q = _t(w[enc_prefix + "muQuery"])[:1][None].expand(B, 1, -1)
    sg_q = _t(w[enc_prefix + "sigmaQuery"])[:1][None].expand(B, 1, -1)
    seq = torch.cat([mu_q, sg_q, h], dim=1) + _t(pe)[: T + 2][None]
    keep = torch.cat([torch.ones(B, 2, dtype=torch.bool), _t(frame_mask).bool()], dim=1)
    for i in range(num_layers):
        seq = encoder_layer(seq, w, f"{enc_prefix}seqTransEncoder.layers.{i}.", nhead, ~keep, dt)
    return seq[:, 0]
