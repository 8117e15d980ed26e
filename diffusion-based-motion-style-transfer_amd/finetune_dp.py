"""Data-parallel fine-tuning: the one exchange step the path has.

The reference is single-device (`global_batch = batch_size # * dist.get_world_size()`,
train/training_loop.py:73; `setup_dist` is a no-op, utils/dist_util.py:18-41).  Sharding the
text-to-motion batch of `few_shot_style_finetune_losses` over ranks needs exactly one collective per
iteration: the mean of the 96 trainable gradient tensors (`seqTransEncoder.layers.{0..7}.*`,
16,822,272 fp32 = 67.3 MB).  They are packed into 8 per-layer buckets (2,102,784 params = 8.4 MB
each) whose views ARE the parameters' `.grad`, so nothing is copied into or out of a bucket.

When a bucket's all-reduce starts.
  * native training node (model/native_stack.py; GPU): the kernels accumulate straight into the bucket views.
    The node whose backward runs last (the 64-clip text-to-motion call) reports as soon as its native backward
    call has returned -- that call only ENQUEUES the GPU work -- and `_native_layer_ready` then walks the layers
    7..0: the communication stream waits for that layer's gradient event (mst_train_wait_layer_grads, recorded by
    the engine behind the layer's last wgrad) and the bucket's all-reduce is launched behind it.  On the GPU the
    collective of layer l therefore runs while layers l-1..0 are still being differentiated.
  * torch-op models (anything with `seqTransEncoder.layers.N.` parameters; the CPU gloo tests): a bucket's
    all-reduce is launched from the post-accumulate hook of its LAST gradient -- backward visits layer 7 first.
On MI355X the collective is RCCL over point-to-point xGMI (`backend="nccl"`); 8.4 MB buckets keep every link busy
without serialising behind one 67 MB ring.  AdamW state stays replicated.

Loss semantics: `loss = rot_mse.mean() + Ls * text_cosine` where text_cosine is a mean over the
LOCAL batch (gaussian_diffusion.py:1388), so averaging gradients over ranks equals the gradient of the
global-batch mean -- what `LayerBucketReducer` produces."""
import re

import torch
import torch.distributed as dist

_LAYER = re.compile(r"seqTransEncoder\.layers\.(\d+)\.")


class LayerBucketReducer:
    def __init__(self, model, process_group=None, native=None):
        """native: None = decide (an engine-backed model on the GPU), False = per-parameter hooks (any torch-op model)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.enabled = True                  # False: fill the buckets but skip the collective (timing the exchange away)
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("LayerBucketReducer: the model has no trainable parameter")
        # native training node (GPU): gradients are written by the kernels, autograd still runs (empty) AccumulateGrad nodes
        # for the parameters, so per-parameter hooks must NOT be used there
        self.native = (hasattr(model, "mst_engine") and all(p.is_cuda for _, p in named)) if native is None else bool(native)
        by_layer = {}
        for n, p in named:
            m = _LAYER.match(n)
            by_layer.setdefault(int(m.group(1)) if m else -1, []).append(p)
        if self.native and -1 in by_layer:
            raise ValueError("LayerBucketReducer: trainable parameters outside seqTransEncoder.layers.* are not produced by the "
                             "native training node and would never be reduced; freeze them")
        self.buckets = []
        for layer in sorted(by_layer, reverse=True):            # launch order = backward order
            params = by_layer[layer]
            flat = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype, device=params[0].device)
            off = 0
            for p in params:                                     # .grad becomes a view of the bucket
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            b = {"layer": layer, "params": params, "flat": flat, "pending": len(params), "work": None, "launched": False}
            self.buckets.append(b)
            if not self.native:
                for p in params:
                    p.register_post_accumulate_grad_hook(self._hook(b))
        self.launch_order, self.launched_in = [], []
        self.comm = None
        if self.native:
            self.comm = torch.cuda.Stream(device=named[0][1].device)
            model.__dict__["_native_layer_ready"] = self._native_layer_ready
            model.__dict__["_native_grads_ready"] = self._native_grads_ready

    # The exchange of a bucket starts while the backward pass is still running and the kernels add straight into the bucket: a
    # second backward before zero_grad() (gradient accumulation, two losses) would add into memory the collective is reading or has
    # already reduced, and the ranks would silently diverge.  One backward per iteration is the protocol; anything else raises.
    _TWICE = ("LayerBucketReducer: a second backward pass reached a bucket whose all-reduce was already launched; call "
              "zero_grad() between backward passes (accumulate losses into ONE backward instead)")

    # ------------------------------------------------------------------------------ launching
    def _launch(self, b, where):
        b["launched"], b["pending"] = True, 0
        if self.world > 1 and self.enabled:
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.launch_order.append(b["layer"])
        self.launched_in.append(where)

    def _native_layer_ready(self, eng, after=None, chain=None):
        """Called by the LAST native node of a backward pass right after its backward call returned (GPU work enqueued, not
        finished): start every layer's exchange behind that layer's gradient event.  after: a CUDA event behind gradient sums that
        were added OUTSIDE the engine's own streams (GradSink.join_chain on the caller's stream); the exchange waits for it too.
        chain: gradient sums of a pass that ran on a SIDE stream into accumulators of its own (GradSink.hand_over_chain): each layer's
        share is added into its bucket HERE, on the communication stream, behind the layer's gradient event and in front of the
        bucket's exchange, so the caller's stream never waits for that pass."""
        if all(b["launched"] for b in self.buckets):
            raise RuntimeError(self._TWICE)
        if after is not None:
            self.comm.wait_event(after)
        if chain is not None:
            self.comm.wait_stream(chain["side"])
            extra = chain["views"]
            if any(id(p) not in extra for b in self.buckets for p in b["params"]):
                raise RuntimeError("LayerBucketReducer: the chained pass holds no accumulator for a bucket's parameter")
        for b in self.buckets:
            if b["launched"]:
                if chain is not None:
                    raise RuntimeError(self._TWICE)
                continue
            with torch.cuda.stream(self.comm):
                eng.wait_layer_grads(b["layer"], self.comm)          # comm stream: wait for layer's wgrads, nothing else
                b["flat"].record_stream(self.comm)
                if chain is not None:
                    torch._foreach_add_([p.grad for p in b["params"]], [extra[id(p)] for p in b["params"]])
                self._launch(b, "backward")
        if chain is not None:
            done = torch.cuda.Event()
            done.record(self.comm)
            chain["release"](done)
    # (capability flag read by native_stack._node_done: this callback accepts the `chain` argument and adds those sums itself; a callback
    # without it -- another reducer written against round 5's two-argument protocol -- gets the sums joined on the caller's stream first)
    _native_layer_ready.adds_chain_sums = True

    def _native_grads_ready(self):
        """End of the backward pass (GradSink.flush): whatever was not started from inside the pass starts now."""
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b, "flush")

    def _hook(self, bucket):
        def fire(_param):
            if bucket["launched"]:
                raise RuntimeError(self._TWICE)
            bucket["pending"] -= 1
            if bucket["pending"] == 0:
                self._launch(bucket, "backward")
        return fire

    # ------------------------------------------------------------------------------ per-iteration protocol
    def zero_grad(self):
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"] = len(b["params"])
            b["work"], b["launched"] = None, False
        self.launch_order, self.launched_in = [], []

    def finish(self):
        """Wait for every bucket and turn the sums into means; call between backward() and step()."""
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["flat"].div_(self.world)
            elif self.world > 1 and self.enabled and not b["launched"]:
                if b["pending"] != len(b["params"]):
                    raise RuntimeError(f"layer {b['layer']}: only part of the bucket received gradients")
                raise RuntimeError(f"layer {b['layer']}: the bucket was never reduced (no gradient reached it); ranks would diverge")
        if self.comm is not None:
            torch.cuda.current_stream(self.comm.device).wait_stream(self.comm)

    def allreduce_only(self):
        """The eight collectives alone, back to back (bench: what the exchange costs when nothing hides it)."""
        works = [dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets] \
            if self.world > 1 else []
        for w in works:
            w.wait()
        if self.buckets[0]["flat"].is_cuda:
            torch.cuda.synchronize(self.buckets[0]["flat"].device)

    def bucket_bytes(self):
        return [b["flat"].numel() * b["flat"].element_size() for b in self.buckets]
