"""Drop-in counterpart of the reference's `diffusion/respace.py` (space_timesteps :8-61,
SpacedDiffusion :64-115, _WrappedModel :118-134): the kept-timestep subset, the betas re-derived for
it and the index remap.  The remap itself is applied on the device by the engine
(`mst_schedule` carries timestep_map); `_WrappedModel` serves models called from Python."""
import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """Set of original-process steps to keep: "ddimN" = fixed integer stride giving exactly N
    steps; otherwise per-section counts spread evenly (rounded) inside equal sections."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                steps = range(0, num_timesteps, stride)
                if len(steps) == want:
                    return set(steps)
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    kept, start = [], 0
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(start + round(pos))
            pos += stride
        start += size
    return set(kept)


class SpacedDiffusion(GaussianDiffusion):
    """A process over a subset of the base process' timesteps."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        base = GaussianDiffusion(**kwargs)
        self.timestep_map, new_betas, prev = [], [], 1.0
        for i, ac in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / prev)
                prev = ac
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def _model_output(self, model, x, t, model_kwargs):
        return super()._model_output(self._wrap_model(model), x, t, model_kwargs)

    def training_losses(self, model, *args, **kwargs):
        # the reference forwards to a base method that does not exist (respace.py:94-97); keep the
        # failure mode explicit instead of an AttributeError deep in super()
        raise AttributeError("GaussianDiffusion has no training_losses in this code base; the fine-tune "
                             "objective is few_shot_style_finetune_losses")

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        # one device copy of the map per (device, dtype) for the life of the diffusion object: the wrapper is rebuilt on every call,
        # and an upload per call is a blocking host-to-device copy (it drains the GPU once per chained step of the fine-tune objective)
        maps = self.__dict__.setdefault("_device_maps", {})
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps, maps)

    def _scale_timesteps(self, t):
        return t    # scaling is the wrapped model's job


class _WrappedModel:
    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps, maps=None):
        self.model, self.timestep_map = model, timestep_map
        self.rescale_timesteps, self.original_num_steps = rescale_timesteps, original_num_steps
        self._maps = {} if maps is None else maps

    def __call__(self, x, ts, **kwargs):
        key = (ts.device, ts.dtype)
        if key not in self._maps:        # one upload per device instead of one per step
            self._maps[key] = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        const = getattr(ts, "_mst_const", None)           # a cached constant batch of the *_with_grad loops (GaussianDiffusion._const_timesteps)
        ckey = (key, const, ts.shape[0]) if const is not None else None
        new_ts = self._maps.get(ckey) if ckey is not None else None
        if new_ts is None:
            new_ts = self._maps[key][ts]
            if self.rescale_timesteps:
                new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
            if ckey is not None:
                self._maps[ckey] = new_ts
        return self.model(x, new_ts, **kwargs)
