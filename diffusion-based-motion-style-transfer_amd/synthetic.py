"""Bit-stable synthetic data for benchmarks, smoke runs and parity tests.

There is no network for datasets or checkpoints, so every run uses random-init weights in the
reference's state-dict layout and random clips of the reference's tensor shape.  The generator is a
counter-based splitmix64 stream evaluated with numpy integer arithmetic only, so the same
(seed, name) pair yields the same array in the authoring container and on the GPU box, whatever
torch's RNG does.

State-dict key layout follows the reference:
  trainable denoiser layers   model/mdm_forstyledataset.py:539-546  (`seqTransEncoder.layers.{i}.*`)
  frozen prior projections    model/mdm_forstyledataset.py:223,254,258,267 under
                              `motion_enc.mdm_model.*` (StyleDiffusion.forward :602-625 borrows them)
"""
import math
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _stream_base(seed, name):
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    return np.uint64(((int(seed) & 0xFFFFFFFF) << 32) | h)


def uniform01(seed, name, n, lane=0):
    """n float64 values in [0, 1) from stream (seed, name, lane)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([_stream_base(seed, name) + np.uint64(lane)], dtype=np.uint64))[0]
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(2) + base
        z = _splitmix64(ctr)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed, name, shape, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    u = uniform01(seed, name, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, name, shape):
    """Standard normal float32 array (Box-Muller in float64, then cast)."""
    n = int(np.prod(shape))
    u1 = uniform01(seed, name, n, lane=0)
    u2 = uniform01(seed, name, n, lane=1)
    r = np.sqrt(-2.0 * np.log1p(-u1))
    return (r * np.cos(2.0 * math.pi * u2)).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------------
PRIOR = "motion_enc.mdm_model."


def layer_keys(i, prefix="seqTransEncoder.layers."):
    p = f"{prefix}{i}."
    return [p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias",
            p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias",
            p + "linear1.weight", p + "linear1.bias", p + "linear2.weight", p + "linear2.bias",
            p + "norm1.weight", p + "norm1.bias", p + "norm2.weight", p + "norm2.bias"]


def denoiser_shapes(njoints, latent_dim=512, ff_size=1024, num_layers=8, clip_dim=512,
                    layer_prefix="seqTransEncoder.layers.", prior_prefix=PRIOR):
    """name -> shape of every tensor the denoise path reads (17.88 M params at the defaults)."""
    d, ff = latent_dim, ff_size
    shapes = {}
    for i in range(num_layers):
        p = f"{layer_prefix}{i}."
        shapes[p + "self_attn.in_proj_weight"] = (3 * d, d)
        shapes[p + "self_attn.in_proj_bias"] = (3 * d,)
        shapes[p + "self_attn.out_proj.weight"] = (d, d)
        shapes[p + "self_attn.out_proj.bias"] = (d,)
        shapes[p + "linear1.weight"] = (ff, d)
        shapes[p + "linear1.bias"] = (ff,)
        shapes[p + "linear2.weight"] = (d, ff)
        shapes[p + "linear2.bias"] = (d,)
        for n in ("norm1", "norm2"):
            shapes[p + n + ".weight"] = (d,)
            shapes[p + n + ".bias"] = (d,)
    q = prior_prefix
    shapes[q + "input_process.poseEmbedding.weight"] = (d, njoints)
    shapes[q + "input_process.poseEmbedding.bias"] = (d,)
    shapes[q + "embed_timestep.time_embed.0.weight"] = (d, d)
    shapes[q + "embed_timestep.time_embed.0.bias"] = (d,)
    shapes[q + "embed_timestep.time_embed.2.weight"] = (d, d)
    shapes[q + "embed_timestep.time_embed.2.bias"] = (d,)
    shapes[q + "embed_text.weight"] = (d, clip_dim)
    shapes[q + "embed_text.bias"] = (d,)
    shapes[q + "output_process.poseFinal.weight"] = (njoints, d)
    shapes[q + "output_process.poseFinal.bias"] = (njoints,)
    return shapes


def tensor_for(seed, name, shape):
    """One seeded tensor with the scale torch's default init would give it
    (Linear: U(+-1/sqrt(fan_in)); in_proj: xavier-uniform; LayerNorm: 1 + small / small,
    deliberately not exactly 1/0 so gamma and beta are exercised)."""
    if name.endswith("norm1.weight") or name.endswith("norm2.weight"):
        return (1.0 + 0.1 * uniform(seed, name, shape)).astype(np.float32)
    if name.endswith("norm1.bias") or name.endswith("norm2.bias"):
        return (0.05 * uniform(seed, name, shape)).astype(np.float32)
    if name.endswith("Query"):          # MotionEncoder.muQuery / sigmaQuery are randn-initialised
        return normal(seed, name, shape)
    if name.endswith("in_proj_weight"):
        bound = math.sqrt(6.0 / (shape[0] + shape[1]))
        return bound * uniform(seed, name, shape)
    if len(shape) == 2:
        return uniform(seed, name, shape) / np.float32(math.sqrt(shape[1]))
    # biases: fan_in unknown from the bias alone; use a fixed small scale
    return 0.04 * uniform(seed, name, shape)


def denoiser_state(seed, njoints, **kw):
    """Seeded state dict (numpy float32) for the denoise path."""
    return {k: tensor_for(seed, k, s) for k, s in denoiser_shapes(njoints, **kw).items()}


def positional_table(max_len, d_model):
    """sin/cos table of model/mdm_forstyledataset.py:392-396, computed the same way in float32."""
    import torch
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.numpy()


def root_horizontal_mask(batch, njoints, nframes, dtype=np.float32):
    """`root_horizontal` inpainting pattern: features 0..2 kept from the content clip
    (data_loaders/stylexia_posrot_utils.py:69-71 applied to any feature count)."""
    m = np.zeros((batch, njoints, 1, nframes), dtype=dtype)
    m[:, :3] = 1
    return m
