"""Data-parallel fine-tuning: the one exchange step the path has.

The reference is single-device (`global_batch = batch_size # * dist.get_world_size()`,
train/training_loop.py:73; `setup_dist` is a no-op, utils/dist_util.py:18-41).  Sharding the
text-to-motion batch of `few_shot_style_finetune_losses` over ranks needs exactly one collective per
iteration: the mean of the 96 trainable gradient tensors (`seqTransEncoder.layers.{0..7}.*`,
16,822,272 fp32 = 67.3 MB).  They are packed into 8 per-layer buckets (2,102,784 params = 8.4 MB
each); a bucket's all-reduce is launched from the autograd hook of its LAST gradient -- backward
visits layer 7 first -- so the collectives overlap the remaining backward work.  With the native training
node (model/native_stack.py) the gradients of all passes through the stack arrive together when the backward
pass ends (GradSink.flush adds them into the bucket views in one fused op); the reducer is told through
`model._native_grads_ready` and launches the eight bucket all-reduces back to back, layer 7 first.  On MI355X that is
RCCL over point-to-point xGMI (`backend="nccl"`); 8.4 MB buckets keep every link busy without
serialising behind one 67 MB ring.  AdamW state stays replicated.

Loss semantics: `loss = rot_mse.mean() + Ls * text_cosine` where text_cosine is a mean over the
LOCAL batch (gaussian_diffusion.py:1388), so averaging gradients over ranks equals the gradient of the
global-batch mean -- what `LayerBucketReducer` produces."""
import re

import torch
import torch.distributed as dist

_LAYER = re.compile(r"seqTransEncoder\.layers\.(\d+)\.")


class LayerBucketReducer:
    def __init__(self, model, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        # native training node (GPU, train_backend "native"): gradients arrive through GradSink.flush -> _native_ready;
        # autograd still runs (empty) AccumulateGrad nodes for the parameters, so per-parameter hooks must NOT be used
        self.native = getattr(model, "train_backend", "torch") == "native" and all(p.is_cuda for _, p in named)
        by_layer = {}
        for n, p in named:
            m = _LAYER.match(n)
            by_layer.setdefault(int(m.group(1)) if m else -1, []).append(p)
        self.buckets = []
        for layer in sorted(by_layer, reverse=True):            # launch order = backward order
            params = by_layer[layer]
            flat = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype, device=params[0].device)
            off = 0
            for p in params:                                     # .grad becomes a view of the bucket
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            b = {"layer": layer, "params": params, "flat": flat, "pending": len(params), "work": None}
            self.buckets.append(b)
            if not self.native:
                for p in params:
                    p.register_post_accumulate_grad_hook(self._hook(b))
        self.launch_order = []
        if self.native:
            model.__dict__["_native_grads_ready"] = self._native_ready

    def _native_ready(self):
        for b in self.buckets:
            if b["layer"] < 0 or b["pending"] != len(b["params"]):
                continue                                    # not a stack bucket / already handled by hooks
            b["pending"] = 0
            if self.world > 1:
                b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.launch_order.append(b["layer"])

    def _hook(self, bucket):
        def fire(_param):
            bucket["pending"] -= 1
            if bucket["pending"] == 0 and self.world > 1:
                bucket["work"] = dist.all_reduce(bucket["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.launch_order.append(bucket["layer"])
        return fire

    def zero_grad(self):
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"] = len(b["params"])
            b["work"] = None
        self.launch_order = []

    def finish(self):
        """Wait for every bucket and turn the sums into means; call between backward() and step()."""
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["flat"].div_(self.world)
            elif self.world > 1 and b["pending"] != len(b["params"]):
                raise RuntimeError(f"layer {b['layer']}: only part of the bucket received gradients")

    def bucket_bytes(self):
        return [b["flat"].numel() * b["flat"].element_size() for b in self.buckets]
