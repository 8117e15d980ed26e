"""Drop-in counterpart of the reference's `diffusion/inpainting_gaussian_diffusion.py`: the
inpainting variant multiplies every noise draw by (1 - inpainting_mask) -- in q_sample (:6-23),
p_sample (:25-64), ddim_sample (:125-177) and their with-grad forms (:66-123, :179-239).
Here that is a flag the fused HIP step kernels read (`mask_noise`), not extra elementwise passes."""
from .respace import SpacedDiffusion


class InpaintingGaussianDiffusion(SpacedDiffusion):
    inpainting_noise = True

    def few_shot_style_finetune_losses(self, model, *args, **kwargs):
        return super().few_shot_style_finetune_losses(self._wrap_model(model), *args, **kwargs)
