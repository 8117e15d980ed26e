"""The trainable encoder stack as ONE autograd node backed by the native engine.

`EncoderStackFn.apply(seq, host, p_drop, key_keep, *params)` replaces `self.seqTransEncoder(seq)`
(model/mdm_forstyledataset.py:622 of the reference) inside the graph that
few_shot_style_finetune_losses (diffusion/gaussian_diffusion.py:1317-1399) back-propagates through:
forward = mst_train_forward (activation tape + dropout), backward = mst_train_backward (dgrad / wgrad
GEMMs, attention / LayerNorm / GELU backward).  The arithmetic uses the engine's f16 copies of the
parameters (re-uploaded by `_EngineHost.mst_engine` whenever a parameter's version changes).

Parameter gradients.  One fine-tune iteration passes through the stack 7 times (one 64-clip call, six
chained single-clip steps); handing 96 fresh gradient tensors per pass to autograd would cost 96 fills +
96 adds per pass (~1300 tiny launches per iteration, more host time than the kernels).  Instead every
node of one backward pass accumulates -- inside the kernels, which add into their output -- into the
module's `GradSink`: straight into the existing `p.grad` tensors when every parameter has one (the
data-parallel reducer's bucket views, or gradients left from an earlier backward), else into ONE fresh flat
buffer that a callback queued on the autograd engine hands to `p.grad` when the backward pass ends (the
mechanism DDP-style reducers use).  The parameters are still inputs of the node, so autograd knows the node
needs to run; it receives None for them.

Data-parallel overlap.  The sink counts the nodes the forward pass created; the node whose backward runs LAST
(in the fine-tune objective: the 64-clip text-to-motion call, by far the longest) tells the reducer
(`host._native_layer_ready`) as soon as its native backward call has returned -- the call only enqueues GPU
work -- and the reducer starts every layer's all-reduce behind that layer's gradient event
(mst_train_wait_layer_grads), layer 7 first, while the GPU is still differentiating the layers below."""
import contextlib

import torch
from torch.autograd import Variable

from ..engine import LAYER_TENSORS


def stack_parameters(encoder):
    """The 12 tensors of every layer in LAYER_TENSORS order (cached on the encoder: the Parameter objects are stable)."""
    cached = encoder.__dict__.get("_mst_stack_params")
    if cached is not None:
        return cached
    out = []
    for layer in encoder.layers:
        named = dict(layer.named_parameters())
        out.extend(named[k] for k in LAYER_TENSORS)
    encoder.__dict__["_mst_stack_params"] = out
    return out


class GradSink:
    """Per-module accumulator of the stack's parameter gradients over one backward pass."""

    def __init__(self, host, params):
        self.host, self.params = host, params
        self.ids = tuple(id(p) for p in params)
        self.flat, self.views, self.active, self.task = None, None, False, -1
        self.open_nodes = 0            # native nodes with parameter gradients created since the last backward pass ended
        self.pending = None            # (side stream, private accumulators, caller's stream) of a chained backward pass still running beside this one

    def abort(self):
        """A backward pass died (the autograd engine drops queued callbacks when a node raises): forget its partial sums."""
        self.flat = self.views = None
        self.active, self.task = False, -1
        self.open_nodes = 0
        self.pending = None

    def join_chain(self):
        """The chained single-clip calls were differentiated on a side stream into accumulators of their own (ChainedCalls.finish): make
        the CALLER's stream (the one the chain was opened on; named explicitly: this runs from inside the chain node's own backward too,
        where the current stream IS the side stream -- ADVICE round 5) wait for that pass and add its sums into this pass's accumulators
        there.  Called before the next native node of the pass accumulates (in the fine-tune objective: the 64-clip call, ~2 ms after
        the chain's pass started, so nothing waits), before a gradient reducer is told that layers are ready, and at the end of the
        pass.  x + 0 is exact, so the gradients are bit for bit those of the in-line pass."""
        if self.pending is None:
            return None
        side, cviews, main = self.pending
        self.pending = None
        if main is None:
            main = torch.cuda.current_stream(cviews[0].device)
        main.wait_stream(side)
        with torch.cuda.stream(main):
            if self.views is not None:
                torch._foreach_add_(list(self.views), list(cviews))
            # the private accumulators may be zeroed (on the side stream) only behind this add
            free = torch.cuda.Event()
            free.record(main)
        self.host.__dict__["_mst_chain_acc_free"] = free
        cur = torch.cuda.current_stream(cviews[0].device)
        if cur != main:
            cur.wait_stream(main)          # whoever asked for the join reads the sums on ITS stream
        return free                        # recorded on the caller's stream behind the add (a reducer told "ready" right now waits for it)

    def hand_over_chain(self):
        """Late join with a gradient reducer (round 6): instead of adding the chain's sums on the caller's stream, give them to the
        reducer, which adds each layer's share into its bucket on the COMMUNICATION stream, behind that layer's gradient event and in
        front of the bucket's exchange -- the caller's stream never waits for the side stream's pass.  Returns None when nothing is
        pending, else {"side": stream, "views": {id(param): its private accumulator}, "release": fn(event)}; `release` must be called
        with an event recorded behind the last add (the accumulators may be zeroed only behind it)."""
        if self.pending is None:
            return None
        side, cviews, _main = self.pending
        self.pending = None
        host = self.host

        def release(event):
            host.__dict__["_mst_chain_acc_free"] = event
        return {"side": side, "views": {id(p): v for p, v in zip(self.params, cviews)}, "release": release}

    def begin(self, device):
        grads = [p.grad for p in self.params]
        if all(g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.device == device for g in grads):
            self.flat, self.views = None, grads          # accumulate in place (kernels add into their output)
        else:
            total = sum(p.numel() for p in self.params)
            self.flat = torch.zeros(total, dtype=torch.float32, device=device)      # fresh: views may become p.grad
            self.views, off = [], 0
            for p in self.params:
                self.views.append(self.flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
        self.active, self.task = True, _graph_task()
        Variable._execution_engine.queue_callback(self.flush)

    def flush(self):
        """End of the backward pass: p.grad (+)= accumulated gradient, then tell a gradient reducer."""
        if not self.active:
            return
        self.join_chain()
        self.active, self.task = False, -1
        self.open_nodes = 0
        if self.flat is not None:
            have, add = [], []
            for p, v in zip(self.params, self.views):
                if not p.requires_grad:
                    continue
                if p.grad is None:
                    p.grad = v
                else:
                    have.append(p.grad)
                    add.append(v)
            if have:
                torch._foreach_add_(have, add)
        self.flat = self.views = None
        ready = getattr(self.host, "_native_grads_ready", None)
        if ready is not None:
            ready()


def _graph_task():
    """Id of the running backward pass (-1 outside one): a sink stamped with another id was left behind by a pass that failed."""
    fn = getattr(torch._C, "_current_graph_task_id", None)
    return int(fn()) if fn is not None else -1


def _draw_seed(p):
    """One 62-bit seed per call from torch's CPU generator: torch.manual_seed reproduces the dropout masks."""
    return int(torch.randint(0, 2 ** 62, (1,), device="cpu").item()) if p > 0 else 0


def _take_tape(ctx):
    if ctx.tape is None:
        raise RuntimeError("backward through a native training node a second time: its activation tape is released by the "
                           "first backward (retain_graph / double backward are not supported)")
    tape, ctx.tape = ctx.tape, None
    return tape


def _sink_of(host, params):
    sink = host.__dict__.get("_mst_grad_sink")
    if sink is None or sink.ids != tuple(id(p) for p in params):
        sink = host.__dict__["_mst_grad_sink"] = GradSink(host, list(params))
    return sink


def _sink_views(ctx, params_need_grad, device, join=True):
    """The gradient accumulators of this backward pass (None when no stack parameter needs a gradient)."""
    if not params_need_grad:
        return None
    sink = _sink_of(ctx.host, ctx.params)
    if sink.active and sink.task != _graph_task():
        sink.abort()                      # stale: its pass raised before the flush callback could run
    if not sink.active:
        sink.begin(device)
    if join and not _JOIN_LATE():
        sink.join_chain()
    return sink.views


def _abort_sink(ctx):
    sink = ctx.host.__dict__.get("_mst_grad_sink")
    if sink is not None:
        sink.abort()


def _node_done(ctx, had_param_grads):
    """After a node's native backward call returned (its GPU work is enqueued): the LAST node of the pass hands the layers to
    a data-parallel reducer right away instead of at the end of the pass."""
    if not had_param_grads:
        return
    sink = ctx.host.__dict__.get("_mst_grad_sink")
    if sink is None:
        return
    sink.open_nodes -= 1
    ready = getattr(ctx.host, "_native_layer_ready", None)
    if sink.open_nodes == 0 and ready is not None and sink.flat is None:      # in-place mode: p.grad IS the reducer's bucket
        # (the chain as the pass's LAST node: its sums are added on the caller's stream HERE, behind everything the earlier nodes
        # accumulated there; the engine's per-layer events do not cover that add, so the reducer also waits for `after`)
        # (late join: only when THIS node is a call of its own on the caller's stream -- its engine's per-layer events then cover
        # everything but the chain's sums; with the chain as the last node the events are the chain engine's, and `after` is needed)
        if _JOIN_LATE() and getattr(ready, "adds_chain_sums", False) and getattr(ctx, "chain", None) is None:
            ready(ctx.eng, None, sink.hand_over_chain())
        else:
            ready(ctx.eng, sink.join_chain())


def _CHAIN_ON():
    import os
    return os.environ.get("MST_CHAIN", "1") != "0"      # MST_CHAIN=0: every model call differentiates alone (A/B, tests)


def _JOIN_LATE():
    """MST_CHAIN_JOIN_LATE=1 (round 6, measured, OFF by default): the chain's gradient sums join the pass's accumulators at its END
    (GradSink.flush, or layer by layer on a reducer's communication stream), not in front of the next native node: a + b = b + a bit for
    bit (tests/test_gpu_boundary.py), and the 64-clip call's backward pass does not wait for the side stream's pass.  Same-box A/B of the
    fine-tune line: 9.75 / 9.77 / 9.93 ms per iteration with the early join, 9.79 / 9.84 / 10.51 ms with the late one -- the caller's
    stream was not waiting there (LAB_NOTES R6.10), so the default stays the order of round 5."""
    import os
    return os.environ.get("MST_CHAIN_JOIN_LATE", "0") == "1"


def _CHAIN_BWD_SIDE_ON():
    import os
    return os.environ.get("MST_CHAIN_BWD_SIDE", "1") != "0"     # MST_CHAIN_BWD_SIDE=0: the chain's ONE backward pass runs in line on the caller's stream


def _chain_accumulators(host, params, device):
    """Private gradient accumulators of the chain's backward pass (one flat fp32 buffer, views in parameter order), cached on the module."""
    ent = host.__dict__.get("_mst_chain_acc")
    ids = tuple(id(p) for p in params)
    if ent is None or ent[0] != ids or ent[1].device != device:
        flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=device)
        views, off = [], 0
        for p in params:
            views.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        ent = host.__dict__["_mst_chain_acc"] = (ids, flat, views)
    return ent[1], ent[2]


def _CHAIN_STREAM_ON():
    import os
    return os.environ.get("MST_CHAIN_STREAM", "1") != "0"      # MST_CHAIN_STREAM=0: the chain's forward calls stay on the caller's stream


class ChainedCalls:
    """n single-clip model calls whose inputs are cut from each other's graphs -- the chained x0-hat steps of the fine-tune objective
    (reference gaussian_diffusion.py:1364-1378 / inpainting_gaussian_diffusion.py:96, :197: `x = x.detach()`).  Their forward passes
    are sequential (a step's input is the previous step's sample), their backward passes are independent: they write the clips of ONE
    activation tape (mst_train_model_forward's clip0 / tape_clips) and are differentiated by ONE native backward pass over the n
    clips, instead of n passes of ~160 launches each on a single clip's 197 token rows (6 x 0.94 ms of the fine-tune iteration).
    The pass runs when the last of the n nodes has received its gradient, or at the end of the autograd pass with zeros for nodes
    that received none."""
    current = None
    _side = {}

    def __init__(self, n, start_event=None, device=None, defer_join=False):
        """start_event: a CUDA event recorded (on the caller's stream) where everything the chain reads was ready -- the chain's forward
        calls then run on a SIDE stream behind that event, on a second engine instance, beside whatever the caller's stream has been
        given since (the fine-tune objective: the 64-clip text-to-motion call and the frozen motion encoder, neither of which the
        chain depends on; six launch-bound single-clip calls use a few CUs each).  The chain's ONE backward pass runs on the caller's
        stream (parameter gradients are accumulated in place: the passes of one iteration stay ordered on one stream).
        device: the tensors' device (the streams are taken there, not on torch's current device).
        The side stream is taken only when the calls are really chained (MST_CHAIN on): an unchained single-clip call works on the
        module's ONE engine, whose workspace and gradient accumulators the caller's stream is using."""
        self.n, self.k, self.reported, self.done = int(n), 0, 0, False
        self.tape = self.seed = self.dbuf = self.ctx0 = None
        self.key = None
        self.start_event = start_event if (_CHAIN_STREAM_ON() and _CHAIN_ON()) else None
        self.device = device
        self.main = self.side = self._sctx = None
        self.prev = None
        self._depth = 0
        # defer_join (round 6): the caller's stream is NOT made to wait for the side stream when a step's block ends; whoever consumes the
        # steps' outputs does so on `self.side` (GaussianDiffusion.few_shot_style_finetune_losses(overlap_backward=True): the masked-L2
        # terms and the sum of the losses), so that the caller's stream can start the text branch's backward pass while the chain is still
        # in its forward calls.  Only honoured when the chain really runs on a side stream.
        self.defer_join = bool(defer_join)

    # The block is entered PER STEP of the *_with_grad loop (`with chain: model call`), never across a generator's yield: between two
    # steps the consumer runs with its own current stream and no chain installed.
    def __enter__(self):
        self._depth += 1
        if self._depth > 1:
            return self
        self.prev, ChainedCalls.current = ChainedCalls.current, self
        if self.start_event is not None and torch.cuda.is_available():
            if self.side is None:
                dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
                dev = torch.device(dev)
                idx = dev.index if dev.index is not None else torch.cuda.current_device()
                self.main = torch.cuda.current_stream(idx)
                self.side = ChainedCalls._side.get(idx)
                if self.side is None:
                    self.side = ChainedCalls._side[idx] = torch.cuda.Stream(idx)
                self.side.wait_event(self.start_event)
            self._sctx = torch.cuda.stream(self.side)
            self._sctx.__enter__()
        return self

    def __exit__(self, *exc):
        self._depth -= 1
        if self._depth > 0:
            return False
        ChainedCalls.current = self.prev
        if self._sctx is not None:
            self._sctx.__exit__(*exc)
            self._sctx = None
            if not self.defer_join:
                self.main.wait_stream(self.side)           # what follows on the caller's stream may read the step's outputs
        return False

    @property
    def deferred(self):
        """True when the steps' outputs live on the side stream and the caller's stream has not been told to wait for them."""
        return self.defer_join and self.side is not None

    @contextlib.contextmanager
    def foreign_call(self, *tensors):
        """A native call inside the block that does NOT join the chain (frozen stack, other shape, MST_CHAIN off ...): it works on the
        module's one engine, so it goes to the CALLER's stream, ordered behind what the side stream produced and in front of what the
        side stream does next; autograd then runs its backward on the caller's stream as well."""
        if self.side is None or self._sctx is None:
            yield
            return
        self.main.wait_stream(self.side)
        for t in tensors:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(self.main)
        with torch.cuda.stream(self.main):
            yield
        self.side.wait_stream(self.main)

    def accepts(self, key):
        """Same engine, shape and dropout rates as the chain's first call, and a free slot."""
        return not self.done and self.k < self.n and (self.key is None or self.key == key)

    def report(self, ctx, k, grad_out):
        if self.done:
            raise RuntimeError("backward through a chained native training node a second time (its tape is released by the first)")
        if self.dbuf is None:
            self.dbuf = torch.zeros((self.n,) + tuple(grad_out.shape[1:]), dtype=torch.float32, device=grad_out.device)
            self.ctx0 = ctx
            Variable._execution_engine.queue_callback(self.finish)      # nodes that never receive a gradient: zeros
        self.dbuf[k:k + 1].copy_(grad_out)
        self.reported += 1
        if self.reported == self.k:
            self.finish()

    def finish(self):
        if self.done or self.dbuf is None:
            return
        self.done = True
        ctx = self.ctx0
        tape, self.tape = self.tape, None
        try:
            views = _sink_views(ctx, True, self.dbuf.device, join=False)
            if self.side is not None and _CHAIN_BWD_SIDE_ON() and self.reported == self.k:      # (not through the end-of-pass callback: the sink may have been flushed)
                # Round 5: the pass stays on the SIDE stream, beside the backward passes the caller's stream runs next (the frozen motion
                # encoder, then the 64-clip call), and accumulates into buffers of its own; GradSink.join_chain adds them in before the
                # next native node accumulates.  ~160 launch-bound kernels on six clips' token rows no longer sit in front of those passes.
                sink = _sink_of(ctx.host, ctx.params)
                cflat, cviews = _chain_accumulators(ctx.host, ctx.params, self.dbuf.device)
                cur = torch.cuda.current_stream(self.dbuf.device)
                if cur != self.side:
                    self.side.wait_stream(cur)       # (through the end-of-pass callback this runs on the caller's stream: the copies into dbuf)
                # A second chain of the SAME backward pass (two objective evaluations summed into one loss): the first
                # chain's sums are still waiting in the private accumulators for join_chain -- keep adding on the same side stream.
                again = sink.pending is not None and sink.pending[0] is self.side and sink.pending[1] is cviews
                if sink.pending is not None and not again:
                    sink.join_chain()
                with torch.cuda.stream(self.side):
                    if not again:
                        free = ctx.host.__dict__.get("_mst_chain_acc_free")
                        if free is not None:
                            self.side.wait_event(free)       # the previous join's add (on the caller's stream) has read them
                        cflat.zero_()
                    ctx.eng.train_model_backward(tape, self.dbuf, ctx.p_drop, ctx.p_pe, self.seed, cviews, need_input_grad=False)
                self.dbuf.record_stream(self.side)
                tape.record_stream(self.side)
                sink.pending = (self.side, cviews, self.main)
            elif self.main is not None:
                # autograd runs this node on the side stream (its forward's); the pass itself belongs on the caller's stream, behind
                # the gradients the side stream has just copied in (report() ran there; through the end-of-pass callback this code is on
                # the caller's stream already, so the side stream is named) and in line with the iteration's other backward passes
                self.main.wait_stream(self.side)
                self.main.wait_stream(torch.cuda.current_stream(self.dbuf.device))
                with torch.cuda.stream(self.main):
                    ctx.eng.train_model_backward(tape, self.dbuf, ctx.p_drop, ctx.p_pe, self.seed, views, need_input_grad=False)
                self.dbuf.record_stream(self.main)
                tape.record_stream(self.main)
            else:
                ctx.eng.train_model_backward(tape, self.dbuf, ctx.p_drop, ctx.p_pe, self.seed, views, need_input_grad=False)
            _node_done(ctx, True)
        except BaseException:
            _abort_sink(ctx)
            raise
        finally:
            self.dbuf = None


class DenoiserTrainFn(torch.autograd.Function):
    """The WHOLE denoiser call as one node: conditioning token, pose embedding, positional-encoding dropout, the trainable
    stack and the output projection (mst_train_model_forward / _backward).  Used when the module conditions on text (every
    shipped script); gradients: dL/dx and the 96 stack tensors -- projections, timestep MLP and text projection are frozen."""

    @staticmethod
    def forward(ctx, x, host, p_drop, p_pe, timesteps, text_emb, *params):
        B, F, one, T = x.shape
        cond_drop = host.__dict__.pop("_mst_cond_drop", None)      # mask_cond's Bernoulli mask, applied inside the projection (_native_train_call)
        block = ChainedCalls.current
        chain = block if _CHAIN_ON() else None
        if chain is not None and not (B == 1 and not ctx.needs_input_grad[0] and any(ctx.needs_input_grad[6:])):
            chain = None
        seed = _draw_seed(max(p_drop, p_pe))
        ctx.p_drop, ctx.p_pe, ctx.host, ctx.params, ctx.chain = p_drop, p_pe, host, params, None
        if chain is not None:
            # (a chain on a side stream works on an engine instance of its own: workspace, text projection and weight copies)
            eng = host.mst_engine(max(B, chain.n), T, slot="chain" if chain.side is not None else None)
            key = (id(eng), B, F, T, float(p_drop), float(p_pe))
            if chain.accepts(key):
                eng.set_text(text_emb.detach(), drop=cond_drop)
                ctx.eng = eng
                if chain.k == 0:
                    chain.key, chain.seed, chain.tape = key, seed, eng.train_tape(chain.n, T + 1, zero=True)
                    _sink_of(host, params).open_nodes += 1          # the chain is ONE native backward call
                out, _ = eng.train_model_forward(x.detach(), timesteps, p_drop, p_pe, chain.seed, tape=chain.tape, clip0=chain.k,
                                                 tape_clips=chain.n)
                ctx.chain, ctx.slot, ctx.tape, ctx.seed = chain, chain.k, None, chain.seed
                chain.k += 1
                return out
        # a call of its own: the module's one engine, on the caller's stream (inside a side-stream block too: foreign_call)
        with (block.foreign_call(x, timesteps, text_emb) if block is not None else contextlib.nullcontext()):
            eng = host.mst_engine(B, T)
            eng.set_text(text_emb.detach(), drop=cond_drop)
            ctx.eng = eng
            out, tape = eng.train_model_forward(x.detach(), timesteps, p_drop, p_pe, seed)
            if block is not None and block.side is not None:
                out.record_stream(block.side)
        ctx.tape, ctx.seed = tape, seed
        if any(ctx.needs_input_grad[6:]):
            _sink_of(host, params).open_nodes += 1
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.chain is not None:         # one slot of a shared tape: differentiated with the chain's other calls in one pass
            ctx.chain.report(ctx, ctx.slot, grad_out)
            return (None,) * (6 + len(ctx.params))
        need_in = ctx.needs_input_grad[0]
        tape = _take_tape(ctx)            # raises on a second backward BEFORE the sink is touched
        try:
            views = _sink_views(ctx, any(ctx.needs_input_grad[6:]), grad_out.device)
            d_x = ctx.eng.train_model_backward(tape, grad_out.contiguous(), ctx.p_drop, ctx.p_pe, ctx.seed, views,
                                               need_input_grad=need_in)
            _node_done(ctx, views is not None)
        except BaseException:
            _abort_sink(ctx)
            raise
        return (d_x, None, None, None, None, None) + (None,) * len(ctx.params)


class EncoderStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seq, host, p_drop, key_keep, *params):
        """key_keep: None or bool [B, S], False = padding key (the motion encoder's ~src_key_padding_mask)."""
        S, B, d = seq.shape
        eng = host.mst_engine(B, S - 1)
        seed = _draw_seed(p_drop)
        h = seq.detach().permute(1, 0, 2).contiguous()
        if key_keep is not None:
            key_keep = key_keep.to(torch.uint8).contiguous()
        out, tape = eng.train_forward(h, p_drop, seed, key_keep=key_keep)
        ctx.eng, ctx.tape, ctx.p_drop, ctx.seed, ctx.host, ctx.keep = eng, tape, p_drop, seed, host, key_keep
        ctx.params = params
        if any(ctx.needs_input_grad[4:]):
            _sink_of(host, params).open_nodes += 1
        return out.permute(1, 0, 2).contiguous()

    @staticmethod
    def backward(ctx, grad_out):
        need_in = ctx.needs_input_grad[0]
        tape = _take_tape(ctx)
        try:
            views = _sink_views(ctx, any(ctx.needs_input_grad[4:]), grad_out.device)
            d_out = grad_out.permute(1, 0, 2).contiguous()
            d_in = ctx.eng.train_backward(tape, d_out, ctx.p_drop, ctx.seed, views, need_input_grad=need_in, key_keep=ctx.keep)
            _node_done(ctx, views is not None)
        except BaseException:
            _abort_sink(ctx)
            raise
        gi = d_in.permute(1, 0, 2).contiguous() if need_in else None
        return (gi, None, None, None) + (None,) * len(ctx.params)


class MotionEncoderFn(torch.autograd.Function):
    """MotionEncoder.forward (reference mdm_forstyledataset.py:90-124) as one node: pose embedding of the frames, the mu / sigma
    query tokens, positional rows (+ dropout), the 8 key-padding-masked layers and the pick of token 0
    (mst_motion_encoder_forward / _backward).  Every parameter on this path is frozen in the fine-tune objective, so the node
    returns dL/dx only."""

    @staticmethod
    def forward(ctx, x, host, p_drop, p_pe, key_keep):
        B, F, one, T = x.shape
        eng = host.mst_engine(B, T + 1)
        seed = _draw_seed(max(p_drop, p_pe))
        mu, tape = eng.motion_encoder_forward(x.detach(), host.muQuery.detach(), host.sigmaQuery.detach(), key_keep, p_drop, p_pe, seed)
        ctx.eng, ctx.tape, ctx.keep, ctx.T, ctx.p, ctx.p_pe, ctx.seed = eng, tape, key_keep, T, p_drop, p_pe, seed
        return mu

    @staticmethod
    def backward(ctx, d_mu):
        d_x = ctx.eng.motion_encoder_backward(_take_tape(ctx), d_mu.contiguous(), ctx.keep, ctx.T, ctx.p, ctx.p_pe, ctx.seed)
        return d_x, None, None, None, None
