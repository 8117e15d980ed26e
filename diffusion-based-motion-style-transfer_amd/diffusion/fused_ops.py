"""Autograd nodes of the fine-tune objective's glue, one HIP launch each way:

  FusedStepFn     p_sample_with_grad / ddim_sample_with_grad's step algebra (inpainting_gaussian_diffusion.py:66-123, :179-239):
                  forward = the fused step kernel (mst_step_epilogue: blend, x0-hat, posterior mean / DDIM update, masked noise),
                  backward = mst_step_backward (both outputs are affine in the model output).
  MaskedL2Fn      masked_l2 (gaussian_diffusion.py:223-235) with broadcast operands (the `.expand(num_step, ...)` of :1380).
  TextCosineFn    (1 - cosine_similarity(f / |f|, m / |m|)).mean() (:1384-1388).

They replace ~20 + ~10 + ~10 elementwise / reduction launches of torch ops per use; the tensors must be float32 on the GPU
(there is no other implementation: CPU tensors raise)."""
import ctypes as C

import torch

from .. import _native as N


def _cuda_f32(t, what):
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError(f"{what}: float32 GPU tensor required (the HIP kernels are the only implementation)")
    return t


class FusedStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model_out, x, t, noise, mask, motion, schedule, sampler, eta, mask_noise, clip_denoised):
        sample, pred = schedule.step(model_out.detach(), x, t, noise, sampler, eta, mask=mask, motion=motion,
                                     mask_noise=mask_noise, clip_denoised=clip_denoised)
        ctx.schedule, ctx.sampler, ctx.eta = schedule, int(sampler), float(eta)
        ctx.has_blend = mask is not None and motion is not None
        # an output without a gradient arrives as None, not as a zero tensor: in the chained steps `sample` is cut from the graph (x.detach()),
        # and autograd would otherwise fill a zero tensor per step in front of k_step_backward (six launches on the chain's backward pass)
        ctx.set_materialize_grads(False)
        # clip_denoised (the reference signature's default): x0-hat = clamp(blend, -1, 1); its gradient mask is read off the output
        ctx.save_for_backward(t, mask if ctx.has_blend else None, pred if clip_denoised else None)
        return sample, pred

    @staticmethod
    def backward(ctx, g_sample, g_pred):
        t, mask, pred_clipped = ctx.saved_tensors
        if g_pred is None and g_sample is None:
            return (None,) * 11
        ref = g_pred if g_pred is not None else g_sample
        gs = None if g_sample is None else _cuda_f32(g_sample.contiguous(), "g_sample")
        gp = None if g_pred is None else _cuda_f32(g_pred.contiguous(), "g_pred")
        d = torch.empty_like(ref, memory_format=torch.contiguous_format)
        B = d.shape[0]
        tt = t.to(device=d.device, dtype=torch.int64).contiguous()
        mk = None if mask is None else mask.to(device=d.device, dtype=torch.float32).contiguous()
        N.check(N.lib().mst_step_backward(ctx.schedule.handle, N.ptr(gs), N.ptr(gp), N.ptr(mk), int(ctx.has_blend), N.ptr(tt), B,
                                          d.numel() // B, ctx.sampler, ctx.eta, N.ptr(pred_clipped), N.ptr(d), N.stream_ptr(d.device)))
        return (d,) + (None,) * 10


def _rows(t, n, inner, what):
    """(tensor, element stride between samples) of a [n, ...] operand that may be an expand()ed view of one sample."""
    t = _cuda_f32(t, what)
    if t.shape[0] == n and t.stride(0) == 0:
        t = t[0]
        if not t.is_contiguous():
            t = t.contiguous()
        return t, 0
    if t.shape[0] == 1 and n > 1:
        return t.contiguous(), 0
    t = t.contiguous()
    assert t.numel() == n * inner, (what, tuple(t.shape), n, inner)
    return t, inner


class MaskedL2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mask):
        n, F, one, T = b.shape
        b_c = _cuda_f32(b.contiguous(), "masked_l2 b")
        a_c, a_stride = _rows(a, n, F * one * T, "masked_l2 a")
        m_c, m_stride = _rows(mask.float() if mask.dtype != torch.float32 else mask, n, T, "masked_l2 mask")
        loss = torch.empty(n, dtype=torch.float32, device=b.device)
        with torch.cuda.device(b.device):      # the entry point launches on the current device (no device argument in the ABI)
            N.check(N.lib().mst_masked_l2(N.ptr(a_c), a_stride, N.ptr(b_c), N.ptr(m_c), m_stride, n, F * one, T, None, N.ptr(loss),
                                          N.stream_ptr(b.device)))
        ctx.save_for_backward(a_c, b_c, m_c)
        ctx.meta = (a_stride, m_stride, n, F * one, T, a.shape, a.requires_grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        a_c, b_c, m_c = ctx.saved_tensors
        a_stride, m_stride, n, F, T, a_shape, a_needs = ctx.meta
        d_b = torch.empty_like(b_c)
        g = _cuda_f32(g.contiguous(), "masked_l2 grad")
        with torch.cuda.device(d_b.device):
            N.check(N.lib().mst_masked_l2(N.ptr(a_c), a_stride, N.ptr(b_c), N.ptr(m_c), m_stride, n, F, T, N.ptr(g), N.ptr(d_b),
                                          N.stream_ptr(d_b.device)))
        d_a = None
        if ctx.needs_input_grad[0]:
            d_a = -d_b if a_stride else (-d_b).sum(0, keepdim=True).expand(a_shape)
        return d_a, (d_b if ctx.needs_input_grad[1] else None), None


class JoinLossesFn(torch.autograd.Function):
    """a + b where `a` was produced on a SIDE stream and `b` on the caller's: the sum is evaluated on the side stream (which waits for
    b; the caller's stream waits for nothing), the node itself belongs to the caller's stream -- autograd files a node under the stream
    that is current when its forward returns -- so the backward pass hands the incoming gradient to b's branch on the caller's stream at
    once and to a's branch on the side stream behind an event.  The result must not be read on the caller's stream before the backward
    pass has joined the streams (GaussianDiffusion.few_shot_style_finetune_losses, overlap_backward)."""

    @staticmethod
    def forward(ctx, a, b, side):
        main = torch.cuda.current_stream(b.device)
        ready = torch.cuda.Event()
        ready.record(main)
        b.record_stream(side)                  # b's memory may go back to the caller's pool only behind the side stream's read
        with torch.cuda.stream(side):
            side.wait_event(ready)
            out = a + b
        out.record_stream(main)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g, None


class TextCosineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, m):
        f_c, m_c = _cuda_f32(f.contiguous(), "text_features"), _cuda_f32(m.contiguous(), "mu")
        B, D = m_c.shape
        out = torch.empty(1, dtype=torch.float32, device=m.device)
        with torch.cuda.device(m.device):
            N.check(N.lib().mst_text_cosine(N.ptr(f_c), N.ptr(m_c), B, D, None, N.ptr(out), N.stream_ptr(m.device)))
        ctx.save_for_backward(f_c, m_c)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        f_c, m_c = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("the text feature of the fine-tune objective is a constant (frozen CLIP)")
        B, D = m_c.shape
        d_m = torch.empty_like(m_c)
        gg = _cuda_f32(g.reshape(1).contiguous(), "grad")
        with torch.cuda.device(d_m.device):
            N.check(N.lib().mst_text_cosine(N.ptr(f_c), N.ptr(m_c), B, D, N.ptr(gg), N.ptr(d_m), N.stream_ptr(d_m.device)))
        return None, d_m
