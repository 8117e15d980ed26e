"""Python handles on the native engine: `Schedule` (one diffusion process' tables on the device) and
`DenoiserEngine` (one MDM-shaped denoiser: weights + workspace + the fused sampling loop).

Everything here is plumbing -- tensor validation, pointer passing, stream selection; the arithmetic
is in csrc/.  Reference counterparts are cited in include/mst_engine.h."""
import ctypes as C

import numpy as np
import torch

from . import _native as N

SAMPLER_DDPM, SAMPLER_DDIM = 0, 1
NOISE_BUFFER, NOISE_PHILOX = 0, 1

# order of enum mst_table; names are the reference's attribute names (gaussian_diffusion.py:183-219)
TABLE_ORDER = ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "posterior_mean_coef1",
               "posterior_mean_coef2", "_log_variance", "sqrt_recip_alphas_cumprod",
               "sqrt_recipm1_alphas_cumprod", "alphas_cumprod", "alphas_cumprod_prev")

# engine tensor name -> where StyleDiffusion / MDM keep it (model/mdm_forstyledataset.py)
PRIOR_TENSORS = (
    "input_process.poseEmbedding.weight", "input_process.poseEmbedding.bias",
    "output_process.poseFinal.weight", "output_process.poseFinal.bias",
    "embed_timestep.time_embed.0.weight", "embed_timestep.time_embed.0.bias",
    "embed_timestep.time_embed.2.weight", "embed_timestep.time_embed.2.bias",
    "embed_text.weight", "embed_text.bias", "sequence_pos_encoder.pe")
LAYER_TENSORS = (
    "self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
    "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
    "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")


def _f32c(t, device, what):
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t))
    t = t.to(device=device, dtype=torch.float32)
    if not t.is_contiguous():
        t = t.contiguous()
    if t.device.type != "cuda":
        raise RuntimeError(f"{what}: the engine needs a GPU tensor")
    return t


class Schedule:
    """Device copy of one diffusion process (enum mst_table order), built from the float64 numpy
    tables of a GaussianDiffusion-like object or an `oracle.schedule`-style dict."""

    def __init__(self, tables, timestep_map, device, log_variance=None):
        get = (lambda k: tables[k]) if isinstance(tables, dict) else (lambda k: getattr(tables, k))
        if log_variance is None:
            log_variance = get("posterior_log_variance_clipped")      # FIXED_SMALL
        rows = []
        for name in TABLE_ORDER:
            a = log_variance if name == "_log_variance" else get(name)
            rows.append(np.asarray(a, dtype=np.float64).astype(np.float32))  # == `.float()` of :1615
        tab = np.ascontiguousarray(np.stack(rows))
        tmap = np.ascontiguousarray(np.asarray(timestep_map, dtype=np.int32))
        self.num_steps = tab.shape[1]
        assert tmap.shape == (self.num_steps,)
        self.device = torch.device(device)
        h = C.c_void_p()
        N.check(N.lib().mst_schedule_create(self.num_steps, tab.ctypes.data_as(C.c_void_p),
                                            tmap.ctypes.data_as(C.c_void_p), self.device.index or 0, C.byref(h)))
        self.handle = h

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and N is not None and N._lib is not None:      # None during interpreter shutdown
            N._lib.mst_schedule_destroy(h)
            self.handle = None

    # stand-alone elementwise kernels (any model callable) -------------------------------------
    def q_sample(self, x_start, t, noise, mask=None):
        dev = x_start.device
        x_start = _f32c(x_start, dev, "x_start")
        noise = _f32c(noise, dev, "noise")
        mask = None if mask is None else _f32c(mask, dev, "mask")
        t = t.to(device=dev, dtype=torch.int64).contiguous()
        out = torch.empty_like(x_start)
        B = x_start.shape[0]
        N.check(N.lib().mst_q_sample(self.handle, N.ptr(x_start), N.ptr(noise), N.ptr(mask), N.ptr(t), B,
                                     x_start.numel() // B, N.ptr(out), N.stream_ptr(dev)))
        return out

    def step(self, model_output, x, t, noise, sampler=SAMPLER_DDPM, eta=0.0, mask=None, motion=None,
             mask_noise=False, clip_denoised=False, mean_type=0):
        """(sample, pred_xstart) of one p_sample / ddim_sample step given the model output.
        mean_type: what the model predicts -- 0 x_start, 1 epsilon, 2 previous x (converted inside the kernel, reference :398-412)."""
        dev = x.device
        mo = _f32c(model_output, dev, "model_output")
        x = _f32c(x, dev, "x")
        noise = None if noise is None else _f32c(noise, dev, "noise")
        mask = None if mask is None else _f32c(mask, dev, "mask")
        motion = None if motion is None else _f32c(motion, dev, "motion")
        t = t.to(device=dev, dtype=torch.int64).contiguous()
        sample, xstart = torch.empty_like(x), torch.empty_like(x)
        B = x.shape[0]
        N.check(N.lib().mst_step_epilogue_mt(self.handle, N.ptr(mo), N.ptr(x), N.ptr(noise), N.ptr(mask), N.ptr(motion),
                                             N.ptr(t), B, x.numel() // B, int(sampler), int(mean_type), float(eta), int(bool(mask_noise)),
                                             int(bool(clip_denoised)), N.ptr(sample), N.ptr(xstart), N.stream_ptr(dev)))
        return sample, xstart


class DenoiserEngine:
    def __init__(self, feats, max_frames, max_rows, num_layers=8, device="cuda:0", latent_dim=512, num_heads=4,
                 ff_size=1024, clip_dim=512, pe_len=5000):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DenoiserEngine needs a GPU device; the HIP kernels are the only implementation")
        self.cfg = N.MstConfig(feats, max_frames, max_rows, latent_dim, num_heads, ff_size, num_layers, clip_dim,
                               pe_len, self.device.index or 0)
        self.feats, self.max_frames, self.max_rows, self.num_layers = feats, max_frames, max_rows, num_layers
        self.latent_dim = latent_dim
        self.weight_diagnostics = None
        h = C.c_void_p()
        N.check(N.lib().mst_engine_create(C.byref(self.cfg), C.byref(h)))
        self.handle = h
        import os
        self._precise_on = os.environ.get("MST_PRECISE", "0") not in ("", "0")

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and N is not None and N._lib is not None:
            N._lib.mst_engine_destroy(h)
            self.handle = None

    # ------------------------------------------------------------------------------ weights
    def _remember_source(self, name, tensor):
        """Weights go up WITHOUT their lo halves while precise mode is off (34 MB nobody reads); `set_precise(True)` later uploads them
        again from the caller's tensor, so a reference to it (not to the device temporary) is kept until then -- and only until then:
        with precise mode on every upload is complete and the entry is dropped.  The reference sees what the caller's tensor holds at
        THAT time: a tensor mutated in place since (an optimizer step) is uploaded with its current values, as any later load would."""
        self._sources = getattr(self, "_sources", {})
        if getattr(self, "_precise_on", False):
            self._sources.pop(name, None)
        else:
            self._sources[name] = tensor

    def load_tensor(self, name, tensor):
        self._remember_source(name, tensor)
        t = _f32c(tensor, self.device, name)
        shape = (C.c_int64 * t.dim())(*t.shape)
        N.check(N.lib().mst_load_weight(self.handle, name.encode(), N.ptr(t), shape, t.dim(), N.stream_ptr(self.device)))

    def load_layers(self, tensors):
        """All layer tensors (num_layers x LAYER_TENSORS order) in one launch per 8 layers: float32, contiguous, on this device."""
        assert len(tensors) == 12 * self.num_layers
        for t in tensors:
            if not (t.is_cuda and t.device == self.device and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError("load_layers wants float32 contiguous tensors on the engine's device")
        arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        N.check(N.lib().mst_load_layers(self.handle, arr, N.stream_ptr(self.device)))
        for i, t in enumerate(tensors):
            self._remember_source(f"seqTransEncoder.layers.{i // 12}.{LAYER_TENSORS[i % 12]}", t)

    def load_state_dict(self, sd, layer_prefix="seqTransEncoder.layers.", prior_prefix="motion_enc.mdm_model.",
                        pe=None):
        """Load from a reference-layout state dict: encoder layers under `layer_prefix`, the frozen
        prior's projections under `prior_prefix` (StyleDiffusion: 'seqTransEncoder.layers.' +
        'motion_enc.mdm_model.'; the prior itself: 'motion_enc.mdm_model.seqTransEncoder.layers.')."""
        for i in range(self.num_layers):
            for k in LAYER_TENSORS:
                self.load_tensor(f"seqTransEncoder.layers.{i}.{k}", sd[f"{layer_prefix}{i}.{k}"])
        for k in PRIOR_TENSORS:
            if k == "sequence_pos_encoder.pe" and pe is not None:
                self.load_tensor(k, pe)
            else:
                self.load_tensor(k, sd[prior_prefix + k])
        torch.cuda.current_stream(self.device).synchronize()   # sources may be temporaries
        N.check(N.lib().mst_weights_complete(self.handle))
        self.weight_diagnostics = self._diagnose({f"{i}.{k}": sd[f"{layer_prefix}{i}.{k}"] for i in range(self.num_layers) for k in LAYER_TENSORS})
        if self.weight_diagnostics["recommend_precise"] and not getattr(self, "_precise_on", False):
            import warnings
            warnings.warn("mst_amd: " + self.weight_diagnostics["why"] + " -- f16 MFMA operands may exceed 1e-3 relative L2 on this checkpoint; "
                          "DenoiserEngine.set_precise(True) / MST_PRECISE=1 splits every operand (hi + lo)", stacklevel=2)

    # Load-time check of the statistics that put ANY 16-bit operand type above the 1e-3 bar (measured on the stress
    # checkpoints of tests/test_gpu_parity.py::test_forward_with_ill_conditioned_weights: LayerNorm gains with x20 outlier channels 1.1e-3,
    # FFN / attention weights at 3x their initialisation scale 1.2e-3, against 4.0e-4 on well-conditioned weights).  A heuristic, not a
    # bound: it looks at what amplifies an operand rounding -- a LayerNorm gain far above its row's median (one channel dominates the next
    # product) and weight matrices far above the scale torch initialises them with (peaky softmax, large pre-activations).
    GAIN_OUTLIER, WEIGHT_SCALE = 8.0, 2.0

    def _diagnose(self, layer_tensors):
        """Per-tensor reductions (amax / median of the LayerNorm gains, rms of the matrices: no stacked copy of the 96 tensors) gathered
        into ONE small vector and ONE host synchronisation -- this runs inside the first call of a freshly loaded model.
        Both criteria are validated on the synthetic stress checkpoints only (tests/test_gpu_parity.py: x20 gain outliers 1.1e-3, x3
        weight scale 1.2e-3 on the default path).  The weight-scale ratio is relative to torch's INITIALISATION rms, which trained
        checkpoints may exceed without leaving the bar (not calibrated against a real trained checkpoint -- none is
        available offline), so `weight_diagnostics` says WHICH criterion fired (`gain_outlier`, `weight_scale_above_threshold`) and a
        caller who knows the checkpoint can tell the two apart."""
        init_rms = {"self_attn.in_proj_weight": (2.0 / (4 * self.latent_dim)) ** 0.5}            # xavier_uniform over [3d, d]
        stats, index = [], []
        for name, t in layer_tensors.items():
            layer, kind = name.split(".", 1)
            v = t.detach().float()
            if kind in ("norm1.weight", "norm2.weight"):
                a = v.abs()
                r = a.amax() / a.median().clamp_min(1e-12)
                what = "gain"
            elif kind.endswith("weight") and v.dim() == 2:
                ref = init_rms.get(kind, (1.0 / (3.0 * v.shape[1])) ** 0.5)                          # nn.Linear default: U(-1/sqrt(in), 1/sqrt(in))
                r = v.pow(2).mean().sqrt() / ref
                what = "scale"
            else:
                continue
            stats.append(r.reshape(1))
            index.append((what, f"{layer}.{kind}"))
        vals = torch.cat([x.to(stats[0].device) for x in stats]).tolist() if stats else []
        worst_gain, worst_w, where_g, where_w = 1.0, 1.0, "", ""
        for (what, name), r in zip(index, vals):
            if what == "gain" and r > worst_gain:
                worst_gain, where_g = r, name
            if what == "scale" and r > worst_w:
                worst_w, where_w = r, name
        why = []
        if worst_gain >= self.GAIN_OUTLIER:
            why.append(f"LayerNorm gain outlier x{worst_gain:.0f} over the median (layers.{where_g})")
        if worst_w >= self.WEIGHT_SCALE:
            why.append(f"weight scale x{worst_w:.1f} of the initialisation scale (layers.{where_w})")
        return {"max_layernorm_gain_over_median": worst_gain, "max_weight_scale_over_init": worst_w,
                "gain_outlier": worst_gain >= self.GAIN_OUTLIER, "weight_scale_above_threshold": worst_w >= self.WEIGHT_SCALE,
                "recommend_precise": bool(why), "why": "; ".join(why)}

    # ------------------------------------------------------------------------------ conditioning
    def set_text(self, text_emb, keep=None, cfg=False, drop=None):
        """drop: the Bernoulli mask of the training-mode mask_cond as drawn (1 = dropped): cond * (1 - drop) inside the projection launch."""
        te = _f32c(text_emb, self.device, "text_emb")
        if drop is not None:
            assert keep is None and not cfg
            dr = _f32c(drop.reshape(-1), self.device, "drop")
            if dr.numel() != te.shape[0]:
                raise ValueError(f"set_text: a drop mask of {dr.numel()} entries for {te.shape[0]} text embeddings")
            N.check(N.lib().mst_set_text_dropped(self.handle, N.ptr(te), N.ptr(dr), te.shape[0], N.stream_ptr(self.device)))
            self._text_keepalive = (te, dr)
            return
        kp = None if keep is None else _f32c(keep, self.device, "keep")
        N.check(N.lib().mst_set_text(self.handle, N.ptr(te), N.ptr(kp), te.shape[0], int(bool(cfg)),
                                     N.stream_ptr(self.device)))
        self._text_keepalive = (te, kp)

    # ------------------------------------------------------------------------------ model call
    def forward(self, x, t, scale=None, cfg=False):
        x = _f32c(x, self.device, "x")
        B, F, one, T = x.shape
        assert F * one == self.feats, (F, one, self.feats)
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        sc = None if scale is None else _f32c(scale, self.device, "scale")
        out = torch.empty_like(x)
        N.check(N.lib().mst_forward(self.handle, N.ptr(x), N.ptr(t), N.ptr(sc), B, T, int(bool(cfg)), N.ptr(out),
                                    N.stream_ptr(self.device)))
        return out

    # ------------------------------------------------------------------------------ training
    def train_tape(self, rows, S, zero=False):
        """Activation tape for one train_forward of `rows` clips x S tokens (caller-owned); uninitialised unless `zero` (a tape
        whose clips are written by several calls: a clip nobody wrote must differentiate to zeros, not to garbage)."""
        n = N.lib().mst_train_tape_bytes(self.handle, rows, S)
        if n <= 0:
            raise RuntimeError("mst_train_tape_bytes failed")
        return (torch.zeros if zero else torch.empty)(n, dtype=torch.uint8, device=self.device)

    def _key_keep(self, keep, rows, S):
        if keep is None:
            return None
        k = keep.to(device=self.device, dtype=torch.uint8).contiguous()
        if tuple(k.shape) != (rows, S):
            raise ValueError(f"key_keep must be [{rows}, {S}], got {tuple(k.shape)}")
        return k

    def train_forward(self, h, p_drop, seed, tape=None, key_keep=None):
        """h: [rows, S, 512] float32 -> (encoder-stack output [rows, S, 512], tape).
        key_keep: optional [rows, S] bool, False = padding key (src_key_padding_mask inverted)."""
        h = _f32c(h, self.device, "h")
        rows, S, d = h.shape
        if tape is None:
            tape = self.train_tape(rows, S)
        out = torch.empty_like(h)
        kk = self._key_keep(key_keep, rows, S)
        N.check(N.lib().mst_train_forward(self.handle, N.ptr(h), rows, S, float(p_drop), int(seed), N.ptr(kk), N.ptr(tape),
                                          N.ptr(out), N.stream_ptr(self.device)))
        return out, tape

    def train_backward(self, tape, d_out, p_drop, seed, grads, need_input_grad=True, key_keep=None):
        """grads: num_layers*12 float32 GPU tensors in LAYER_TENSORS order per layer, accumulated into;
        None skips every parameter gradient (frozen stack: input gradient only).
        Returns dL/dh [rows, S, 512] (or None)."""
        d_out = _f32c(d_out, self.device, "d_out")
        rows, S, d = d_out.shape
        arr = None
        if grads is not None:
            if len(grads) != self.num_layers * 12:
                raise ValueError(f"expected {self.num_layers * 12} gradient buffers, got {len(grads)}")
            for g in grads:
                if g.dtype != torch.float32 or not g.is_contiguous() or g.device != d_out.device:
                    raise ValueError("gradient buffers must be contiguous float32 tensors on the engine's device")
            arr = (C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        d_in = torch.empty_like(d_out) if need_input_grad else None
        kk = self._key_keep(key_keep, rows, S)
        N.check(N.lib().mst_train_backward(self.handle, N.ptr(tape), N.ptr(d_out), rows, S, float(p_drop), int(seed),
                                           N.ptr(kk), N.ptr(d_in), arr, N.stream_ptr(self.device)))
        return d_in

    def train_model_forward(self, x, t, p_drop, p_pe, seed, tape=None, clip0=0, tape_clips=0):
        """Whole denoiser in train mode: x [B, F, 1, T], t int64 [B] -> (model output [B, F, 1, T], tape).  set_text first.
        tape / clip0 / tape_clips: write clips [clip0, clip0 + B) of a tape made by train_tape(tape_clips, T + 1) (calls whose
        inputs are cut from each other's graphs share one tape and ONE backward pass over all tape_clips clips)."""
        x = _f32c(x, self.device, "x")
        B, F, one, T = x.shape
        assert F * one == self.feats, (F, one, self.feats)
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        if tape is None:
            tape, clip0, tape_clips = self.train_tape(B, T + 1), 0, 0
        out = torch.empty_like(x)
        N.check(N.lib().mst_train_model_forward(self.handle, N.ptr(x), N.ptr(t), B, T, float(p_drop), float(p_pe), int(seed),
                                                N.ptr(tape), N.ptr(out), int(clip0), int(tape_clips), N.stream_ptr(self.device)))
        return out, tape

    def train_model_backward(self, tape, d_out, p_drop, p_pe, seed, grads, need_input_grad=True):
        d_out = _f32c(d_out, self.device, "d_out")
        B, F, one, T = d_out.shape
        arr = None
        if grads is not None:
            if len(grads) != self.num_layers * 12:
                raise ValueError(f"expected {self.num_layers * 12} gradient buffers, got {len(grads)}")
            arr = (C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        d_x = torch.empty_like(d_out) if need_input_grad else None
        N.check(N.lib().mst_train_model_backward(self.handle, N.ptr(tape), N.ptr(d_out), B, T, float(p_drop), float(p_pe), int(seed),
                                                 N.ptr(d_x), arr, N.stream_ptr(self.device)))
        return d_x

    def motion_encoder_forward(self, x, mu_query, sigma_query, key_keep, p_drop, p_pe, seed):
        """MotionEncoder.forward natively: x [B, F, 1, T], queries [512] (or [1, 512]), key_keep bool [B, T + 2] -> (mu [B, 512], tape)."""
        x = _f32c(x, self.device, "x")
        B, F, one, T = x.shape
        assert F * one == self.feats, (F, one, self.feats)
        mq, sq = _f32c(mu_query.reshape(-1), self.device, "muQuery"), _f32c(sigma_query.reshape(-1), self.device, "sigmaQuery")
        kk = self._key_keep(key_keep, B, T + 2)
        tape = self.train_tape(B, T + 2)
        mu = torch.empty((B, mq.numel()), dtype=torch.float32, device=self.device)
        N.check(N.lib().mst_motion_encoder_forward(self.handle, N.ptr(x), N.ptr(mq), N.ptr(sq), N.ptr(kk), B, T, float(p_drop), float(p_pe),
                                                   int(seed), N.ptr(tape), N.ptr(mu), N.stream_ptr(self.device)))
        return mu, tape

    def motion_encoder_backward(self, tape, d_mu, key_keep, frames, p_drop, p_pe, seed):
        d_mu = _f32c(d_mu, self.device, "d_mu")
        B = d_mu.shape[0]
        kk = self._key_keep(key_keep, B, frames + 2)
        d_x = torch.empty((B, self.feats, 1, frames), dtype=torch.float32, device=self.device)
        N.check(N.lib().mst_motion_encoder_backward(self.handle, N.ptr(tape), N.ptr(d_mu), N.ptr(kk), B, frames, float(p_drop), float(p_pe),
                                                    int(seed), N.ptr(d_x), N.stream_ptr(self.device)))
        return d_x

    def wait_layer_grads(self, layer, stream):
        """Make the torch stream `stream` wait for layer `layer`'s parameter gradients of the most recent backward call."""
        N.check(N.lib().mst_train_wait_layer_grads(self.handle, int(layer), C.c_void_p(stream.cuda_stream)))

    def dropout_mask(self, seed, layer, site, p, n):
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        N.check(N.lib().mst_dropout_mask(int(seed), layer, site, float(p), n, N.ptr(out), N.stream_ptr(self.device)))
        return out

    # ------------------------------------------------------------------------------ sampling
    def sample_loop(self, schedule, x, t_start, t_end=0, sampler=SAMPLER_DDPM, eta=0.0, cfg=False, scale=None,
                    mask=None, motion=None, mask_noise=True, clip_denoised=False, noise=None, seed=None,
                    dump_xstart=False):
        """Run diffusion indices t_start..t_end in place on `x` ([B,F,1,T] float32 GPU tensor).
        noise: [nsteps,B,F,1,T] tensor (injected draws) or None -> in-kernel Philox with `seed`.
        Returns x (and the [nsteps,B,F,1,T] x0-hat dump when requested)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        B, F, one, T = x.shape
        nsteps = t_start - t_end + 1
        a = N.MstLoopArgs()
        a.batch, a.frames, a.cfg, a.sampler = B, T, int(bool(cfg)), int(sampler)
        a.mask_noise, a.clip_denoised = int(bool(mask_noise)), int(bool(clip_denoised))
        a.t_start, a.t_end, a.eta = int(t_start), int(t_end), float(eta)
        keep = []
        if noise is not None:
            noise = _f32c(noise, self.device, "noise")
            assert noise.numel() == nsteps * x.numel(), (noise.shape, nsteps, x.shape)
            a.noise_mode, a.noise_dev = NOISE_BUFFER, noise.data_ptr()
            keep.append(noise)
        else:
            a.noise_mode, a.seed = NOISE_PHILOX, int(seed or 0)
        for name, val in (("scale_dev", scale), ("inpainting_mask_dev", mask), ("inpainted_motion_dev", motion)):
            if val is not None:
                val = _f32c(val, self.device, name)
                keep.append(val)
                setattr(a, name, val.data_ptr())
        a.x_dev = x.data_ptr()
        dump = None
        if dump_xstart:
            dump = torch.empty((nsteps,) + tuple(x.shape), dtype=torch.float32, device=self.device)
            a.xstart_dump_dev = dump.data_ptr()
        N.check(N.lib().mst_sample_loop(self.handle, schedule.handle, C.byref(a), N.stream_ptr(self.device)))
        self._loop_keepalive = keep
        return (x, dump) if dump_xstart else x

    def loop_slices(self, batch, cfg=False, frames=None):
        """How many independent clip slices sample_loop runs concurrently for this batch (1..3) of `frames`-frame clips
        (default: the engine's frame cap)."""
        return int(N.lib().mst_loop_slices(self.handle, int(batch), int(bool(cfg)), int(frames or 0)))

    def philox_normal(self, batch, frames, seed, step):
        out = torch.empty((batch, self.feats, 1, frames), dtype=torch.float32, device=self.device)
        N.check(N.lib().mst_philox_normal(N.ptr(out), batch, self.feats, frames, int(seed), int(step),
                                          N.stream_ptr(self.device)))
        return out

    # ------------------------------------------------------------------------------ profiling / debug
    def profile(self, on=True, period=16):
        N.check(N.lib().mst_profile_enable(self.handle, (period if period > 1 else 1) if on else 0))

    def profile_event_overhead_us(self):
        """Median duration an EMPTY event pair reports on the loop stream (calibrated when profiling is switched on)."""
        return float(N.lib().mst_profile_event_overhead_us(self.handle))

    def profile_read(self):
        torch.cuda.current_stream(self.device).synchronize()
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = (C.c_int32 * 16)()
        k = N.lib().mst_profile_read(self.handle, names, ms, n, 16)
        return {names[i].decode(): (float(ms[i]), int(n[i])) for i in range(k)}

    def set_precise(self, on=True):
        """Every layer GEMM of the sampling path multiplies its activation as an f16 hi + lo pair (~22 bits): for checkpoints
        whose outlier statistics put plain f16 operands above the 1e-3 bar; about half the default throughput at 64 clips."""
        self._precise_on = bool(on)
        rc = N.lib().mst_set_precise(self.handle, int(bool(on)))
        if rc == 2:                                        # weights went up without their lo halves: upload again, now with them
            for name, tensor in list(getattr(self, "_sources", {}).items()):
                self.load_tensor(name, tensor)             # (precise is on now: each entry is dropped as it goes up)
            torch.cuda.current_stream(self.device).synchronize()
            rc = 0
        N.check(rc)

    def set_trunk_groups(self, on=True):
        """The encoder stack of a sampling step as ONE launch of resident workgroup groups (csrc/mst_trunk.h); bit-identical results."""
        N.check(N.lib().mst_set_trunk_groups(self.handle, int(bool(on))))

    def trunk_check(self):
        """After a synchronisation: raises when a hand-off wait of a resident-group launch gave up."""
        N.check(N.lib().mst_trunk_check(self.handle))

    def debug_stop_after(self, layer=-1, stage=-1):
        N.check(N.lib().mst_debug_stop_after(self.handle, layer, stage))

    def debug_buffer(self, which, rows, cols):
        dt = torch.float32 if which in ("hs", "temb", "textproj") else torch.float16
        out = torch.empty((rows, cols), dtype=dt, device=self.device)
        N.check(N.lib().mst_debug_copy(self.handle, which.encode(), N.ptr(out), out.numel() * out.element_size(),
                                       N.stream_ptr(self.device)))
        return out
