"""Fine-tune iteration, un-synchronised: (1) does the per-iteration host->device copy of the timestep batch (bench.py --mode finetune,
train/training_loop.py) tie the host to the GPU?  N iterations with it and with a resident timestep tensor: wall time per iteration and
how far ahead the host ends.  (2) CUDA events on the caller's stream at the objective's seams (no synchronisation inside the loop):
where the caller's stream spends the iteration in steady state."""
import os, sys, time
os.environ["FB_NATIVE_ONLY"] = "1"
os.environ["FB_ITERS"] = "2"
sys.argv = ["finetune_bench.py"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "tools", "finetune_bench.py")).read().replace("print(json.dumps(", "(lambda *a: None)((")
g = {"__name__": "bench", "__file__": os.path.join(ROOT, "tools", "finetune_bench.py")}
try:
    exec(compile(src, "finetune_bench.py", "exec"), g)
except SystemExit:
    pass
import torch
model, d_ddim, opt, dev = g["model"], g["d_ddim"], g["opts"]["native"], g["dev"]
t2m, tt_dev, content, style, y1, yB = (g[k] for k in ("t2m", "tt", "content", "style", "y1", "yB"))
gen = torch.Generator(device="cpu").manual_seed(1)
B = t2m.shape[0]
marks = []


def mark(name):
    if marks is not None and collecting[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e))


collecting = [False]
orig_call = model._native_train_call
def call(x, t, y):
    r = orig_call(x, t, y)
    if x.shape[0] > 1:
        mark("64-clip call enqueued behind")
    return r
model._native_train_call = call
me = model.motion_enc
me_fwd = me.forward
def mef(*a, **k):
    r = me_fwd(*a, **k)
    mark("motion encoder forward")
    return r
me.forward = mef


# the chained single-clip steps on their side stream: an event where a step's block starts and one where it ends (both on the side stream)
from mst_amd.model import native_stack as _ns
chain_marks = []
_enter, _exit = _ns.ChainedCalls.__enter__, _ns.ChainedCalls.__exit__
def _en(self):
    r = _enter(self)
    if collecting[0] and self._depth == 1 and self.side is not None:
        e = torch.cuda.Event(enable_timing=True); e.record(self.side); chain_marks.append(("in", e))
    return r
def _ex(self, *exc):
    if collecting[0] and self._depth == 1 and self.side is not None:
        e = torch.cuda.Event(enable_timing=True); e.record(self.side); chain_marks.append(("out", e))
    return _exit(self, *exc)
_ns.ChainedCalls.__enter__, _ns.ChainedCalls.__exit__ = _en, _ex


OVERLAP = os.environ.get("FT_OVERLAP", "0") == "1"      # FT_OVERLAP=1: few_shot_style_finetune_losses(overlap_backward=True)
QUICK = os.environ.get("FT_QUICK", "0") == "1"           # FT_QUICK=1: only the resident-timestep variants
side_done = []


def iteration(h2d):
    mark("start")
    tt = torch.randint(0, 6, (B,), generator=gen).to(dev) if h2d else tt_dev
    opt.zero_grad(set_to_none=True)
    terms = d_ddim.few_shot_style_finetune_losses(model, t2m, tt, content, style, skip_steps=700, model_kwargs=y1, model_t2m_kwargs=yB,
                                                  semantic_guidance=1, use_ddim=1, Ls=10, overlap_backward=OVERLAP)
    mark("objective (forward) complete on the caller's stream")
    terms["loss"].backward()
    mark("backward")
    side = _ns.ChainedCalls._side.get(dev.index if dev.index is not None else torch.cuda.current_device())
    if collecting[0] and side is not None:                   # where the side stream's last work of the iteration (the chain's backward pass) ends
        e = torch.cuda.Event(enable_timing=True); e.record(side); side_done.append(e)
    opt.step()
    mark("optimizer step")
    return terms["loss"]


for h2d in ((False,) if QUICK else (True, False, True, False)):
    for _ in range(3):
        iteration(h2d)
    torch.cuda.synchronize()
    N = 20
    t0 = time.perf_counter()
    for _ in range(N):
        iteration(h2d)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"timestep batch {'copied from the host every iteration' if h2d else 'resident on the device'}: {1e3 * (t2 - t0) / N:.2f} ms per iteration, "
          f"host enqueue {1e3 * (t1 - t0) / N:.2f} ms per iteration (host {1e3 * (t2 - t1):.1f} ms ahead at the end)")
for h2d in ((False,) if QUICK else (True, False)):
    for _ in range(3):
        iteration(h2d)
    torch.cuda.synchronize()
    side_done.clear()
    collecting[0] = True
    marks.clear()
    for _ in range(10):
        iteration(h2d)
    collecting[0] = False
    torch.cuda.synchronize()
    acc, prev = {}, None
    for name, e in marks:
        if name != "start" and prev is not None:
            acc[name] = acc.get(name, 0.0) + prev.elapsed_time(e)
        prev = e
    print(f"caller's stream, {'with' if h2d else 'without'} the per-iteration copy (events, no synchronisation; ms per iteration):")
    for k, v in acc.items():
        print(f"  {k:55s} {v / 10:7.2f}")
    # side stream: per iteration 7 blocks (loop set-up + six steps)
    starts = [e for k, e in chain_marks if k == "in"]
    ends = [e for k, e in chain_marks if k == "out"]
    n_it = 10
    per = len(starts) // n_it
    its = [m for m in marks if m[0] == "start"]
    if per:
        dur = [0.0] * per
        gap = [0.0] * per
        for it in range(n_it):
            for j in range(per):
                dur[j] += starts[it * per + j].elapsed_time(ends[it * per + j])
                gap[j] += (its[it][1].elapsed_time(starts[it * per + j]) if j == 0 else ends[it * per + j - 1].elapsed_time(starts[it * per + j]))
        print("  side stream blocks (ms): " + ", ".join(f"{d / n_it:.2f}" for d in dur) + f"   sum {sum(dur) / n_it:.2f}")
        print("  gaps in front of them (first: from the iteration's start): " + ", ".join(f"{g_ / n_it:.2f}" for g_ in gap))
        print(f"  iteration start -> last block's end: {sum(its[it][1].elapsed_time(ends[it * per + per - 1]) for it in range(n_it)) / n_it:.2f}")
    if side_done:
        print(f"  iteration start -> end of the side stream's work (chain backward): {sum(its[it][1].elapsed_time(side_done[it]) for it in range(n_it)) / n_it:.2f}")
        bw = [e for k, e in marks if k == "backward"]
        print(f"  iteration start -> end of the backward pass on the caller's stream: {sum(its[it][1].elapsed_time(bw[it]) for it in range(n_it)) / n_it:.2f}")
    chain_marks.clear()
    tot = marks[0][1].elapsed_time(marks[-1][1]) / 10
    print(f"  {'first start -> last optimizer step, per iteration':55s} {tot:7.2f}")
