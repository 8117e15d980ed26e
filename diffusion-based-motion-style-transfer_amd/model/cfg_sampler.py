"""Drop-in counterpart of the reference's `model/cfg_sampler.py` (:8-43): classifier-free guidance
for SAMPLING.  The reference runs the model twice (conditional, then `uncond=True`) and blends
u + scale * (c - u).  With an engine-backed model both halves go through the transformer as ONE
doubled batch and the blend happens in the output-projection kernel's registers (K12 fused into
K9); inside a sampling loop the diffusion objects detect this wrapper and run the whole loop that
way (diffusion/gaussian_diffusion.py `_unwrap`)."""
from copy import deepcopy

import torch.nn as nn


class ClassifierFreeSampleModel(nn.Module):
    is_cfg_sampler = True

    def __init__(self, model):
        super().__init__()
        self.model = model
        assert self.model.cond_mask_prob > 0, \
            'Cannot run a guided diffusion on a model that has not been trained with no conditions'
        self.rot2xyz = getattr(model, 'rot2xyz', None)
        self.translation = model.translation
        self.njoints = model.njoints
        self.nfeats = model.nfeats
        self.data_rep = model.data_rep
        self.cond_mode = model.cond_mode

    def forward(self, x, timesteps, y=None):
        assert self.model.cond_mode in ['text', 'action']
        m = self.model
        if hasattr(m, "mst_engine") and not m._wants_autograd(x):
            eng = m.mst_engine(2 * x.shape[0], x.shape[-1])
            m.mst_prepare(eng, y, True)
            return eng.forward(x, timesteps, scale=y['scale'], cfg=True)
        y_uncond = deepcopy(y)
        y_uncond['uncond'] = True
        out = m(x, timesteps, y)
        out_uncond = m(x, timesteps, y_uncond)
        return out_uncond + (y['scale'].view(-1, 1, 1, 1) * (out - out_uncond))
