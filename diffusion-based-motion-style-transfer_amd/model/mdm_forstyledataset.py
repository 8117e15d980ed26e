"""Drop-in counterpart of the reference's `model/mdm_forstyledataset.py` for the denoise path:
`StyleDiffusion` (:494-625), `MotionEncoder` (:11-124) and `MDM` (:183-385) with the same constructor
arguments, attributes and state-dict key layout (so reference checkpoints load and fine-tuned
checkpoints save in the reference's 96-tensor layout, train/training_loop.py:312-348).

The arithmetic has ONE implementation, the HIP library; the nn.Module containers below exist for the state-dict layout.
  * inference (no autograd graph needed): `forward` hands x, t and the text embedding to the native
    engine (csrc/, one HIP launch sequence); sampling loops bypass even that and run whole loops
    natively (see diffusion/gaussian_diffusion.py).  Weights are re-uploaded when parameters change.
  * autograd (fine-tuning, model.train()): the whole model call is ONE autograd node backed by the engine's training
    kernels (model/native_stack.py: activation tape + dropout forward, HIP backward).
There is no torch-op evaluation of the stacks here: CPU tensors raise.  (The tests carry their own fp32 reference,
tests/torch_reference.py, and install it on an instance when they need one.)
CLIP stays third-party: `encode_text` uses the `clip` package when it is installed, a callable set
with `set_text_encoder`, or a precomputed `y['text_embed']` ([B, clip_dim]).
"""
import contextlib
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine as _eng


def _encoder(latent_dim, num_heads, ff_size, dropout, activation, num_layers):
    layer = nn.TransformerEncoderLayer(d_model=latent_dim, nhead=num_heads, dim_feedforward=ff_size,
                                       dropout=dropout, activation=activation)
    return nn.TransformerEncoder(layer, num_layers=num_layers)


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        freq = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2] = torch.sin(pos * freq)
        pe[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer('pe', pe.unsqueeze(1))            # [max_len, 1, d] like the reference buffer

    def forward(self, x):                                      # x: [S, B, d]
        return self.dropout(x + self.pe[:x.shape[0], :])


class TimestepEmbedder(nn.Module):
    def __init__(self, latent_dim, sequence_pos_encoder):
        super().__init__()
        self.latent_dim = latent_dim
        self.sequence_pos_encoder = sequence_pos_encoder
        self.time_embed = nn.Sequential(nn.Linear(latent_dim, latent_dim), nn.SiLU(), nn.Linear(latent_dim, latent_dim))

    def forward(self, timesteps):
        return self.time_embed(self.sequence_pos_encoder.pe[timesteps]).permute(1, 0, 2)


class InputProcess(nn.Module):
    def __init__(self, data_rep, input_feats, latent_dim):
        super().__init__()
        self.data_rep, self.input_feats, self.latent_dim = data_rep, input_feats, latent_dim
        self.poseEmbedding = nn.Linear(input_feats, latent_dim)
        if data_rep == 'rot_vel':
            self.velEmbedding = nn.Linear(input_feats, latent_dim)

    def forward(self, x):
        bs, njoints, nfeats, nframes = x.shape
        x = x.permute((3, 0, 1, 2)).reshape(nframes, bs, njoints * nfeats)
        if self.data_rep in ('rot6d', 'xyz', 'hml_vec'):
            return self.poseEmbedding(x)
        if self.data_rep == 'rot_vel':
            return torch.cat((self.poseEmbedding(x[[0]]), self.velEmbedding(x[1:])), axis=0)
        raise ValueError


class OutputProcess(nn.Module):
    def __init__(self, data_rep, input_feats, latent_dim, njoints, nfeats):
        super().__init__()
        self.data_rep, self.input_feats, self.latent_dim = data_rep, input_feats, latent_dim
        self.njoints, self.nfeats = njoints, nfeats
        self.poseFinal = nn.Linear(latent_dim, input_feats)
        if data_rep == 'rot_vel':
            self.velFinal = nn.Linear(latent_dim, input_feats)

    def forward(self, output):
        nframes, bs, d = output.shape
        if self.data_rep in ('rot6d', 'xyz', 'hml_vec'):
            output = self.poseFinal(output)
        elif self.data_rep == 'rot_vel':
            output = torch.cat((self.poseFinal(output[[0]]), self.velFinal(output[1:])), axis=0)
        else:
            raise ValueError
        return output.reshape(nframes, bs, self.njoints, self.nfeats).permute(1, 2, 3, 0)


from os import environ as _environ
_os_environ_get = _environ.get


def _cond_drop(module, cond):
    """The Bernoulli drop mask of mask_cond, float32 [bs] (1 = drop the clip's text): the probability vector is a cached constant."""
    key = (cond.shape[0], cond.device, float(module.cond_mask_prob))
    ent = module.__dict__.get("_mst_mask_p")
    if ent is None or ent[0] != key:
        ent = module.__dict__["_mst_mask_p"] = (key, torch.full((cond.shape[0],), float(module.cond_mask_prob), device=cond.device))
    return torch.bernoulli(ent[1])


def _mask_cond(module, cond, force_mask=False):
    """mask_cond of the reference (:288-296 / :592-600)."""
    if force_mask:
        return torch.zeros_like(cond)
    if module.training and module.cond_mask_prob > 0. and _os_environ_get("MST_GLUE_CACHE", "1") == "0":      # the reference's five launches (A/B)
        drop = torch.bernoulli(torch.ones(cond.shape[0], device=cond.device) * module.cond_mask_prob).view(-1, 1)
        return cond * (1. - drop)
    if module.training and module.cond_mask_prob > 0.:
        # the reference: bernoulli(ones(bs) * p).view(-1, 1); cond * (1 - mask).  Same draw from the same generator state, two launches
        # instead of five: the probability vector is a cached constant, and cond - cond * mask is one fused multiply-add (mask is 0 or 1:
        # the same values).  Seven model calls per fine-tune iteration, six of them on the chained steps' critical path (LAB_NOTES R6.17).
        drop = _cond_drop(module, cond).view(-1, 1)
        return torch.addcmul(cond, cond, drop.to(cond.dtype), value=-1.0)
    return cond


class _EngineHost:
    """Mixin: lazily built native engines for a module that owns (or borrows) an MDM-shaped
    parameter set.  `_engine_sources()` -> (layer_prefix, prior_prefix, parameters to watch)."""

    def mst_engine(self, rows, frames, slot=None):
        """slot: None = the module's engine; a name = a SECOND engine instance with its own workspace and weight copies (the chained
        single-clip calls of the fine-tune objective run on a side stream beside the 64-clip call: native_stack.ChainedCalls)."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("the native denoiser runs on the GPU only; call .to('cuda') (there is no CPU fallback)")
        cache = self.__dict__.setdefault("_mst_engines", {})
        key = dev if slot is None else (dev, slot)
        ent = cache.get(key)
        if ent is None or ent["rows"] < rows or ent["frames"] < frames:
            rows_cap = max(rows, ent["rows"] if ent else 0)
            frames_cap = max(frames, ent["frames"] if ent else 0)
            eng = _eng.DenoiserEngine(self.input_feats, frames_cap, rows_cap, num_layers=self.num_layers, device=dev,
                                      latent_dim=self.latent_dim, num_heads=self.num_heads, ff_size=self.ff_size,
                                      clip_dim=self.clip_dim)
            ent = cache[key] = {"eng": eng, "rows": rows_cap, "frames": frames_cap, "version": None}
        src = self.__dict__.get("_mst_sources")          # (layer prefix, prior prefix, parameters to watch): built once;
        if src is None:                                  # Parameter objects survive .to() / load_state_dict / optimizer steps
            src = self.__dict__["_mst_sources"] = self._engine_sources()
        lp, pp, params = src
        version = tuple(p._version for p in params) + tuple(p.data_ptr() for p in params)
        if ent["version"] != version:        # first use, optimizer step, load_state_dict, .to()
            eng, old = ent["eng"], ent["version"]
            n = len(params)
            changed = None if old is None else [i for i in range(n) if old[i] != version[i] or old[n + i] != version[n + i]]
            layers = self._layer_param_index(params) if changed is not None else None
            if layers is not None and changed and set(changed) <= layers[0] and not getattr(eng, "_precise_on", False):
                # an optimizer step: only the stack's 96 tensors moved -- one launch instead of 107 uploads (the frozen projections,
                # timestep MLP and the 10 MB positional table stay as they are)
                eng.load_layers([p.detach() for p in layers[1]])
            else:
                sd = {k: v for k, v in self.state_dict().items() if 'clip_model.' not in k}
                eng.load_state_dict(sd, layer_prefix=lp, prior_prefix=pp)
            ent["version"] = version
        return ent["eng"]

    def _layer_param_index(self, params):
        """(indices of the encoder stack's tensors inside `params`, those tensors in the engine's layer order), or None when a stack
        tensor is not a float32 contiguous GPU parameter of this module's watch list."""
        cached = self.__dict__.get("_mst_layer_index")
        if cached is None:
            from .native_stack import stack_parameters
            pos = {id(p): i for i, p in enumerate(params)}
            stack = stack_parameters(self.seqTransEncoder)
            ok = all(id(p) in pos and p.dtype == torch.float32 for p in stack)
            cached = self.__dict__["_mst_layer_index"] = (frozenset(pos[id(p)] for p in stack), stack) if ok else False
        if cached is False or not all(p.is_contiguous() and p.is_cuda for p in cached[1]):
            return None
        return cached

    def mst_prepare(self, eng, y, cfg):
        """Upload the (constant over a loop) text conditioning: embed_text(mask_cond(encode_text))."""
        prior = self._prior()
        emb = y.get('text_embed')
        if emb is None:
            emb = prior.encode_text(y['text'])
        keep = None
        if y.get('uncond', False):
            keep = torch.zeros(emb.shape[0], device=emb.device)
        eng.set_text(emb, keep=keep, cfg=cfg)

    def _native_forward(self, x, timesteps, y):
        from .native_stack import ChainedCalls
        block = ChainedCalls.current        # inside a side-stream block of chained training calls: this call is not one of them
        with (block.foreign_call(x, timesteps) if block is not None else contextlib.nullcontext()):
            eng = self.mst_engine(x.shape[0], x.shape[-1])
            self.mst_prepare(eng, y, False)
            out = eng.forward(x, timesteps)
            if block is not None and block.side is not None:
                out.record_stream(block.side)
        return out

    def _encoder_stack(self, seq, key_keep=None):
        """seqTransEncoder(seq) as the native training node; seq: [S, B, d]; key_keep: None or bool [B, S].
        Inside an autograd graph it is one node (input gradient, parameter gradients when they are trainable); without one
        the same kernels run with dropout off and the activation tape is dropped."""
        if seq.device.type != "cuda":
            raise RuntimeError("the native encoder stack runs on the GPU only; call .to('cuda') (there is no CPU fallback)")
        from .native_stack import EncoderStackFn, stack_parameters
        if torch.is_grad_enabled() and (seq.requires_grad or any(p.requires_grad for p in self.seqTransEncoder.parameters())):
            p = self.seqTransEncoder.layers[0].dropout.p if self.seqTransEncoder.training else 0.0
            return EncoderStackFn.apply(seq, self, float(p), key_keep, *stack_parameters(self.seqTransEncoder))
        S, B, d = seq.shape
        eng = self.mst_engine(B, S - 1)
        keep = None if key_keep is None else key_keep.to(torch.uint8).contiguous()
        out, _tape = eng.train_forward(seq.detach().permute(1, 0, 2).contiguous(), 0.0, 0, key_keep=keep)
        return out.permute(1, 0, 2)

    def _native_train_call(self, x, timesteps, y):
        """model(x, t, y) inside an autograd graph as ONE native node (DenoiserTrainFn)."""
        if not x.is_cuda:
            raise RuntimeError("the native training path runs on the GPU only; call .to('cuda') (there is no CPU fallback)")
        if 'text' not in getattr(self, "cond_mode", ""):
            raise NotImplementedError("the native training node is built for the text-conditioned models the scripts create "
                                      "(utils/parser_util.py get_cond_mode)")
        from .native_stack import DenoiserTrainFn, stack_parameters
        prior = self._prior()
        enc = y['text_embed'] if y.get('text_embed') is not None else prior.encode_text(y['text'])
        force = y.get('uncond', False)
        self.__dict__.pop("_mst_cond_drop", None)          # (nothing left over from a call that raised before its node consumed it)
        if not force and self.training and self.cond_mask_prob > 0. and _os_environ_get("MST_GLUE_CACHE", "1") != "0":
            # mask_cond (:592-600) with the mask handed to the engine as drawn: cond * (1 - drop) happens inside the text projection's launch
            # (mst_set_text_dropped).  Same draw from the same generator state as `bernoulli(ones(bs) * p)`.
            self.__dict__["_mst_cond_drop"] = _cond_drop(self, enc)
        else:
            enc = self.mask_cond(enc, force_mask=force)
        stack, pe = self.seqTransEncoder, prior.sequence_pos_encoder
        p = stack.layers[0].dropout.p if stack.training else 0.0
        p_pe = pe.dropout.p if pe.training else 0.0
        return DenoiserTrainFn.apply(x, self, float(p), float(p_pe), timesteps, enc, *stack_parameters(stack))

    def _wants_autograd(self, x):
        return torch.is_grad_enabled() and (self.training or x.requires_grad
                                            or any(p.requires_grad for p in self.parameters()))


class MDM(nn.Module, _EngineHost):
    """The text-to-motion prior: owns input/output projections, timestep MLP, text projection,
    positional table and its own 8 encoder layers; also usable as a denoiser by itself
    (train/finetune_style_diffusion.py:195-212)."""

    def __init__(self, modeltype, njoints, nfeats, num_actions, translation, pose_rep, glob, glob_rot,
                 latent_dim=256, ff_size=1024, num_layers=8, num_heads=4, dropout=0.1,
                 ablation=None, activation="gelu", legacy=False, data_rep='rot6d', dataset='amass', clip_dim=512,
                 arch='trans_enc', emb_trans_dec=False, clip_version=None, **kargs):
        super().__init__()
        if arch != 'trans_enc':
            raise ValueError("only arch='trans_enc' is implemented (the shipped scripts use nothing else)")
        self.legacy, self.modeltype, self.njoints, self.nfeats = legacy, modeltype, njoints, nfeats
        self.num_actions, self.data_rep, self.dataset = num_actions, data_rep, dataset
        self.pose_rep, self.glob, self.glob_rot, self.translation = pose_rep, glob, glob_rot, translation
        self.latent_dim, self.ff_size, self.num_layers, self.num_heads = latent_dim, ff_size, num_layers, num_heads
        self.dropout, self.ablation, self.activation, self.clip_dim = dropout, ablation, activation, clip_dim
        self.action_emb = kargs.get('action_emb', None)
        self.input_feats = njoints * nfeats
        self.normalize_output = kargs.get('normalize_encoder_output', False)
        self.cond_mode = kargs.get('cond_mode', 'no_cond')
        self.cond_mask_prob = kargs.get('cond_mask_prob', 0.)
        self.arch, self.emb_trans_dec, self.gru_emb_dim = arch, emb_trans_dec, 0
        self.input_process = InputProcess(data_rep, self.input_feats, latent_dim)
        self.sequence_pos_encoder = PositionalEncoding(latent_dim, dropout)
        self.seqTransEncoder = _encoder(latent_dim, num_heads, ff_size, dropout, activation, num_layers)
        self.embed_timestep = TimestepEmbedder(latent_dim, self.sequence_pos_encoder)
        self.clip_version = clip_version
        self._text_encoder = None
        if 'text' in self.cond_mode:
            self.embed_text = nn.Linear(clip_dim, latent_dim)
            self.clip_model = self.load_and_freeze_clip(clip_version)
        self.output_process = OutputProcess(data_rep, self.input_feats, latent_dim, njoints, nfeats)
        self.rot2xyz = None    # SMPL forward kinematics: not on the denoise path

    def parameters_wo_clip(self):
        return [p for name, p in self.named_parameters() if not name.startswith('clip_model.')]

    def load_and_freeze_clip(self, clip_version):
        try:
            import clip
        except ImportError:
            return None        # encode_text then needs set_text_encoder(...) or y['text_embed']
        model, _ = clip.load(clip_version, device='cpu', jit=False)
        clip.model.convert_weights(model)
        model.eval()
        for p in model.parameters():
            p.requires_grad = False
        return model

    def set_text_encoder(self, fn):
        """fn(list[str]) -> float tensor [B, clip_dim]; replaces CLIP when it is not installed."""
        self._text_encoder = fn

    def mask_cond(self, cond, force_mask=False):
        return _mask_cond(self, cond, force_mask)

    def encode_text(self, raw_text):
        device = next(self.parameters()).device
        if self._text_encoder is not None:
            return self._text_encoder(raw_text).to(device).float()
        if self.clip_model is None:
            raise RuntimeError("CLIP is not installed: pass y['text_embed'] or call set_text_encoder(fn)")
        import clip
        if self.dataset in ('humanml', 'kit'):       # 20 tokens + start/end, zero-padded to CLIP's 77
            texts = clip.tokenize(raw_text, context_length=22, truncate=True).to(device)
            texts = torch.cat([texts, torch.zeros([texts.shape[0], 77 - 22], dtype=texts.dtype, device=device)], dim=1)
        else:
            texts = clip.tokenize(raw_text, truncate=True).to(device)
        return self.clip_model.encode_text(texts).float()

    # ---- engine plumbing
    def _prior(self):
        return self

    def _engine_sources(self):
        return "seqTransEncoder.layers.", "", [p for n, p in self.named_parameters() if not n.startswith('clip_model.')]

    def _condition(self, timesteps, y):
        emb = self.embed_timestep(timesteps)
        if 'text' in self.cond_mode:
            enc = y['text_embed'] if y.get('text_embed') is not None else self.encode_text(y['text'])
            emb = emb + self.embed_text(self.mask_cond(enc, force_mask=y.get('uncond', False)))
        return emb

    def forward(self, x, timesteps, y=None):
        if not self._wants_autograd(x):
            return self._native_forward(x, timesteps, y)
        return self._native_train_call(x, timesteps, y)

    def train(self, mode=True):
        return super().train(mode)


class MotionEncoder(nn.Module, _EngineHost):
    """The frozen 'semantic discriminator' (:11-124): mu/sigma query tokens + frames through 8 masked
    encoder layers; borrows the prior's input projection and positional table.  The masked stack runs natively with a
    key-padding mask: inside an autograd graph (the fine-tune objective needs its INPUT gradient) as the training node with
    no parameter gradients, without a graph through the same kernels with the tape dropped."""

    def __init__(self, modeltype, njoints, nfeats, num_actions, translation, pose_rep, glob, glob_rot,
                 latent_dim=256, ff_size=1024, num_layers=8, num_heads=4, dropout=0.1,
                 ablation=None, activation="gelu", legacy=False, data_rep='rot6d', dataset='amass', clip_dim=512,
                 arch='trans_enc', emb_trans_dec=False, clip_version=None, **kargs):
        super().__init__()
        self.modeltype, self.njoints, self.nfeats, self.num_actions = modeltype, njoints, nfeats, num_actions
        self.pose_rep, self.glob, self.glob_rot, self.translation = pose_rep, glob, glob_rot, translation
        self.latent_dim, self.ff_size, self.num_layers, self.num_heads = latent_dim, ff_size, num_layers, num_heads
        self.dropout, self.ablation, self.activation, self.clip_dim = dropout, ablation, activation, clip_dim
        self.input_feats = njoints * nfeats
        self.cond_mask_prob = kargs.get('cond_mask_prob', 0.)
        self.muQuery = nn.Parameter(torch.randn(1, latent_dim))
        self.sigmaQuery = nn.Parameter(torch.randn(1, latent_dim))
        self.seqTransEncoder = _encoder(latent_dim, num_heads, ff_size, dropout, activation, num_layers)
        self.mdm_model = self.load_and_freeze_mdm(modeltype, njoints, nfeats, num_actions, translation, pose_rep, glob,
                                                  glob_rot, latent_dim, ff_size, num_layers, num_heads, dropout, ablation,
                                                  activation, legacy, data_rep, dataset, clip_dim, arch, emb_trans_dec,
                                                  clip_version, **kargs)

    def parameters_wo_clip(self):
        return [p for name, p in self.named_parameters() if not name.startswith('mdm_model.')]

    def load_model_wo_clip(self, model, state_dict):
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        assert len(unexpected) == 0
        assert all(k.startswith('clip_model.') for k in missing)

    def load_and_freeze_mdm(self, modeltype, njoints, nfeats, num_actions, translation, pose_rep, glob, glob_rot,
                            latent_dim, ff_size, num_layers, num_heads, dropout, ablation, activation, legacy, data_rep,
                            dataset, clip_dim, arch, emb_trans_dec, clip_version, **kargs):
        mdm = MDM(modeltype, njoints, nfeats, num_actions, translation, pose_rep, glob, glob_rot, latent_dim, ff_size,
                  num_layers, num_heads, dropout, ablation, activation, legacy, data_rep, dataset, clip_dim, arch,
                  emb_trans_dec, clip_version, **kargs)
        path = kargs.get("mdm_path", "")
        if path:
            print("load mdm_model from checkpoint {}".format(path))
            self.load_model_wo_clip(mdm, torch.load(path, map_location='cpu'))
        mdm.eval()
        for p in mdm.parameters():
            p.requires_grad = False
        return mdm

    def forward(self, x, y=None):
        """-> (mu [bs, latent_dim], text feature or None).  One native call: frames -> pose embedding, [muQuery | sigmaQuery |
        frames] + positional rows, the masked stack, token 0 (native_stack.MotionEncoderFn inside an autograd graph, the same
        kernels without one)."""
        bs, njoints, nfeats, nframes = x.shape
        if y is not None:
            keep = y.get("mask").squeeze(1).squeeze(1).bool()
            enc_text = None
            if y.get('text_embed') is not None:
                enc_text = y['text_embed']
            elif y.get('text', None) is not None:
                enc_text = self.mdm_model.encode_text(y['text'])
        else:
            keep = torch.ones((bs, nframes), dtype=bool, device=x.device)
            enc_text = None
        if x.device.type != "cuda":
            raise RuntimeError("the native motion encoder runs on the GPU only; call .to('cuda') (there is no CPU fallback)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("the native motion encoder is the FROZEN semantic discriminator of the fine-tune objective "
                                      "(load_motion_enc freezes it); training it is not on this path")
        keep = torch.cat((torch.ones((bs, 2), dtype=bool, device=x.device), keep.to(x.device)), axis=1)
        stack, pe = self.seqTransEncoder, self.mdm_model.sequence_pos_encoder
        p = stack.layers[0].dropout.p if stack.training else 0.0
        p_pe = pe.dropout.p if pe.training else 0.0
        from .native_stack import MotionEncoderFn, _draw_seed
        if torch.is_grad_enabled() and x.requires_grad:
            return MotionEncoderFn.apply(x, self, float(p), float(p_pe), keep), enc_text
        eng = self.mst_engine(bs, nframes + 1)
        mu, _tape = eng.motion_encoder_forward(x.detach(), self.muQuery.detach(), self.sigmaQuery.detach(), keep, float(p), float(p_pe),
                                               _draw_seed(max(p, p_pe)))
        return mu, enc_text

    # ---- engine plumbing: own (frozen) encoder layers + the prior's projections
    def _prior(self):
        return self.mdm_model

    def _engine_sources(self):
        params = list(self.seqTransEncoder.parameters()) + [p for n, p in self.mdm_model.named_parameters()
                                                             if not n.startswith(('clip_model.', 'seqTransEncoder.'))]
        return "seqTransEncoder.layers.", "mdm_model.", params

    def mask_cond(self, cond, force_mask=False):
        return _mask_cond(self, cond, force_mask)


class StyleDiffusion(nn.Module, _EngineHost):
    """The style denoiser (:494-625): 8 trainable encoder layers of its own between the frozen prior's
    input/output projections, timestep MLP and text projection."""

    def __init__(self, modeltype, njoints, nfeats, num_actions, translation, pose_rep, glob, glob_rot,
                 latent_dim=256, ff_size=1024, num_layers=8, num_heads=4, dropout=0.1,
                 ablation=None, activation="gelu", legacy=False, data_rep='rot6d', dataset='amass', clip_dim=512,
                 arch='trans_enc', emb_trans_dec=False, clip_version=None, **kargs):
        super().__init__()
        if arch != 'trans_enc':
            raise ValueError("only arch='trans_enc' is implemented (the shipped scripts use nothing else)")
        self.legacy, self.modeltype, self.njoints, self.nfeats = legacy, modeltype, njoints, nfeats
        self.num_actions, self.data_rep, self.dataset = num_actions, data_rep, dataset
        self.pose_rep, self.glob, self.glob_rot, self.translation = pose_rep, glob, glob_rot, translation
        self.latent_dim, self.ff_size, self.num_layers, self.num_heads = latent_dim, ff_size, num_layers, num_heads
        self.dropout, self.ablation, self.activation, self.clip_dim = dropout, ablation, activation, clip_dim
        self.action_emb = kargs.get('action_emb', None)
        self.kargs = kargs
        self.input_feats = njoints * nfeats
        self.normalize_output = kargs.get('normalize_encoder_output', False)
        self.cond_mode = kargs.get('cond_mode', 'no_cond')
        self.cond_mask_prob = kargs.get('cond_mask_prob', 0.)
        self.arch, self.emb_trans_dec, self.gru_emb_dim = arch, emb_trans_dec, 0
        self.seqTransEncoder = _encoder(latent_dim, num_heads, ff_size, dropout, activation, num_layers)
        self.motion_enc = MotionEncoder(modeltype, njoints, nfeats, num_actions, translation, pose_rep, glob, glob_rot,
                                        latent_dim, ff_size, num_layers, num_heads, dropout, ablation, activation, legacy,
                                        data_rep, dataset, clip_dim, arch, emb_trans_dec, clip_version, **kargs)
        self.load_motion_enc()

    def load_motion_enc(self):
        path = self.kargs.get("semantic_discriminator_path", "")
        if path:
            print("load motion_enc from checkpoint {}".format(path))
            self.load_model(self.motion_enc, torch.load(path, map_location='cpu'))
        self.motion_enc = self.motion_enc.eval()
        for p in self.motion_enc.parameters():
            p.requires_grad = False

    def load_model(self, model, state_dict):
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        assert len(unexpected) == 0
        assert all(k.startswith('mdm_model.') for k in missing)

    def parameters_wo_enc(self):
        return [p for name, p in self.named_parameters() if not name.startswith('motion_enc.')]

    def mask_cond(self, cond, force_mask=False):
        return _mask_cond(self, cond, force_mask)

    # ---- engine plumbing
    def _prior(self):
        return self.motion_enc.mdm_model

    def _engine_sources(self):
        params = list(self.seqTransEncoder.parameters()) + [p for n, p in self._prior().named_parameters()
                                                             if not n.startswith(('clip_model.', 'seqTransEncoder.'))]
        return "seqTransEncoder.layers.", "motion_enc.mdm_model.", params

    def forward(self, x, timesteps, y=None):
        if not self._wants_autograd(x):
            return self._native_forward(x, timesteps, y)
        return self._native_train_call(x, timesteps, y)
