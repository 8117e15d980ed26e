"""The trainable encoder stack as ONE autograd node backed by the native engine.

`EncoderStackFn.apply(seq, host, p_drop, *params)` replaces `self.seqTransEncoder(seq)`
(model/mdm_forstyledataset.py:622 of the reference) inside the graph that
few_shot_style_finetune_losses (diffusion/gaussian_diffusion.py:1317-1399) back-propagates through:
forward = mst_train_forward (activation tape + dropout), backward = mst_train_backward (dgrad / wgrad
GEMMs, attention / LayerNorm / GELU backward).  The 96 parameters are passed as inputs so autograd
routes their gradients; the arithmetic uses the engine's f16 copies of them (re-uploaded by
`_EngineHost.mst_engine` whenever a parameter's version changes)."""
import torch

from ..engine import LAYER_TENSORS


def stack_parameters(encoder):
    """The 12 tensors of every layer in LAYER_TENSORS order."""
    out = []
    for layer in encoder.layers:
        named = dict(layer.named_parameters())
        out.extend(named[k] for k in LAYER_TENSORS)
    return out


class EncoderStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seq, host, p_drop, *params):
        S, B, d = seq.shape
        eng = host.mst_engine(B, S - 1)
        # one 63-bit seed per call from torch's generator: torch.manual_seed reproduces the masks
        seed = int(torch.randint(0, 2 ** 62, (1,), device="cpu").item()) if p_drop > 0 else 0
        h = seq.detach().permute(1, 0, 2).contiguous()
        out, tape = eng.train_forward(h, p_drop, seed)
        ctx.eng, ctx.tape, ctx.p_drop, ctx.seed = eng, tape, p_drop, seed
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.host = host
        return out.permute(1, 0, 2).contiguous()

    @staticmethod
    def backward(ctx, grad_out):
        need_in = ctx.needs_input_grad[0]
        need_p = [ctx.needs_input_grad[3 + i] for i in range(len(ctx.shapes))]
        grads = [torch.zeros(s, dtype=torch.float32, device=grad_out.device) for s in ctx.shapes]
        d_out = grad_out.permute(1, 0, 2).contiguous()
        d_in = ctx.eng.train_backward(ctx.tape, d_out, ctx.p_drop, ctx.seed, grads, need_input_grad=need_in)
        ctx.tape = None
        gi = d_in.permute(1, 0, 2).contiguous() if need_in else None
        return (gi, None, None) + tuple(g if n else None for g, n in zip(grads, need_p))
