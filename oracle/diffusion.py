"""Oracle: q_sample, p_mean_variance, p_sample, ddim_sample and the sampling loops.  (test infra)

torch-CPU float32 tensor math with float64 tables, like the reference
(`_extract_into_tensor`, gaussian_diffusion.py:1605-1618: gather in f64, cast to f32, broadcast).
Random draws are never made here: every noise tensor is injected by the caller so the HIP path and
this restatement can be compared on identical inputs (SURVEY.md section 7 hard part iii).

`inpainting=True` selects the InpaintingGaussianDiffusion overrides
(inpainting_gaussian_diffusion.py:6-64,125-177: noise is multiplied by 1 - inpainting_mask);
`inpainting=False` is the base class (gaussian_diffusion.py:267-285,532-585).
"""
import torch


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(a)


def extract(arr, t, ndim):
    """gaussian_diffusion.py:1605-1618."""
    return torch.from_numpy(arr)[_t(t).long()].float().view(-1, *([1] * (ndim - 1)))


def _masked_noise(noise, mask):
    # `noise *= 1. - mask` (igd.py:18,54,168); exact for the 0/1 masks the callers pass
    return noise * (1.0 - _t(mask)).to(noise.dtype)


def q_sample(tab, x_start, t, noise, inpainting_mask=None):
    """gaussian_diffusion.py:267-285 / inpainting_gaussian_diffusion.py:6-23."""
    x_start, noise = _t(x_start), _t(noise)
    if inpainting_mask is not None:
        noise = _masked_noise(noise, inpainting_mask)
    n = x_start.dim()
    return (extract(tab["sqrt_alphas_cumprod"], t, n) * x_start
            + extract(tab["sqrt_one_minus_alphas_cumprod"], t, n) * noise)


def p_mean_variance(tab, model_output, x, t, inpainting_mask=None, inpainted_motion=None,
                    clip_denoised=False, var_type="fixed_small"):
    """gaussian_diffusion.py:311-424 for START_X prediction and a fixed variance."""
    x, out = _t(x), _t(model_output)
    n = x.dim()
    if inpainting_mask is not None and inpainted_motion is not None:
        m = torch.ones_like(_t(inpainting_mask), dtype=torch.float) * _t(inpainting_mask)
        out = out * (1 - m) + _t(inpainted_motion) * m                       # :341-349
    if var_type == "fixed_small":
        var, logvar = tab["posterior_variance"], tab["posterior_log_variance_clipped"]
    else:  # fixed_large (:371-374)
        import numpy as np
        var = np.append(tab["posterior_variance"][1], tab["betas"][1:])
        logvar = np.log(var)
    pred = out.clamp(-1, 1) if clip_denoised else out
    mean = (extract(tab["posterior_mean_coef1"], t, n) * pred
            + extract(tab["posterior_mean_coef2"], t, n) * x)               # :295-298
    shape = x.shape
    return {"mean": mean, "variance": extract(var, t, n).expand(shape),
            "log_variance": extract(logvar, t, n).expand(shape), "pred_xstart": pred}


def p_sample(tab, model_output, x, t, noise, inpainting=True, inpainting_mask=None,
             inpainted_motion=None, clip_denoised=False, var_type="fixed_small"):
    """One ancestral step (gaussian_diffusion.py:561-585 / igd.py:42-64)."""
    x = _t(x)
    o = p_mean_variance(tab, model_output, x, t, inpainting_mask, inpainted_motion,
                        clip_denoised, var_type)
    noise = _t(noise)
    if inpainting:
        noise = _masked_noise(noise, inpainting_mask)
    nz = (_t(t) != 0).float().view(-1, *([1] * (x.dim() - 1)))
    sample = o["mean"] + nz * torch.exp(0.5 * o["log_variance"]) * noise
    return {"sample": sample, "pred_xstart": o["pred_xstart"]}


def ddim_sample(tab, model_output, x, t, noise, eta=0.0, inpainting=True, inpainting_mask=None,
                inpainted_motion=None, clip_denoised=False, var_type="fixed_small"):
    """One DDIM step (igd.py:141-177 / gaussian_diffusion.py:813-860)."""
    x = _t(x)
    n = x.dim()
    o = p_mean_variance(tab, model_output, x, t, inpainting_mask, inpainted_motion,
                        clip_denoised, var_type)
    pred = o["pred_xstart"]
    eps = ((extract(tab["sqrt_recip_alphas_cumprod"], t, n) * x - pred)
           / extract(tab["sqrt_recipm1_alphas_cumprod"], t, n))             # :443-447
    ab = extract(tab["alphas_cumprod"], t, n)
    abp = extract(tab["alphas_cumprod_prev"], t, n)
    sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
    noise = _t(noise)
    if inpainting:
        noise = _masked_noise(noise, inpainting_mask)
    mean_pred = pred * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps
    nz = (_t(t) != 0).float().view(-1, *([1] * (n - 1)))
    return {"sample": mean_pred + nz * sigma * noise, "pred_xstart": pred}


def loop_indices(num_timesteps, skip_timesteps=0, stop_timesteps=None):
    """gaussian_diffusion.py:759-762 / :1047-1050."""
    if stop_timesteps is not None:
        return list(range(stop_timesteps, num_timesteps - skip_timesteps))[::-1]
    return list(range(num_timesteps - skip_timesteps))[::-1]


def sample_loop(model_fn, tab, timestep_map, shape, noise_fn, sampler="ddpm", inpainting=True,
                inpainting_mask=None, inpainted_motion=None, init_image=None, skip_timesteps=0,
                stop_timesteps=None, eta=0.0, clip_denoised=False, var_type="fixed_small",
                dump_all_xstart=False):
    """p_sample_loop / ddim_sample_loop (gaussian_diffusion.py:644-794, :948-1082).

    model_fn(x, t_original) -> model output; the index remap of respace.py:129-134 is done here.
    noise_fn(k) -> float32 tensor of `shape`: k = 0 is the initial image draw (:754), k = 1 + j the
    draw of the j-th executed step (:569 / igd.py:51,167; drawn even when DDIM eta = 0).
    """
    B = shape[0]
    T = len(tab["betas"])
    img = _t(noise_fn(0)).clone()
    if skip_timesteps and init_image is None:
        init_image = torch.zeros_like(img)
    idx = loop_indices(T, skip_timesteps, stop_timesteps)
    if init_image is not None:
        t0 = torch.full((B,), idx[0], dtype=torch.long)
        img = q_sample(tab, init_image, t0, img, inpainting_mask if inpainting else None)
    tmap = torch.tensor(timestep_map, dtype=torch.long)
    dump, out = [], None
    for j, i in enumerate(idx):
        t = torch.full((B,), i, dtype=torch.long)
        mo = model_fn(img, tmap[t])
        step = p_sample if sampler == "ddpm" else ddim_sample
        kw = dict(eta=eta) if sampler == "ddim" else {}
        out = step(tab, mo, img, t, noise_fn(1 + j), inpainting=inpainting,
                   inpainting_mask=inpainting_mask, inpainted_motion=inpainted_motion,
                   clip_denoised=clip_denoised, var_type=var_type, **kw)
        if dump_all_xstart:
            dump.append(out["pred_xstart"])
        img = out["sample"]
    return dump if dump_all_xstart else out["sample"]


def masked_l2(a, b, mask):
    """gaussian_diffusion.py:223-235: sum((a-b)^2 * mask) / (sum(mask) * F * nfeats) per row."""
    a, b, mask = _t(a), _t(b), _t(mask)
    loss = ((a - b) ** 2 * mask.float()).flatten(1).sum(1)
    n_entries = a.shape[1] * a.shape[2]
    return loss / (mask.flatten(1).sum(1) * n_entries)
