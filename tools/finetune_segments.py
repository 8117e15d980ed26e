"""Where one fine-tune iteration spends its GPU time: the objective's model calls (64-clip text-to-motion call, frozen motion encoder, the
six chained single-clip steps), the backward pass, the optimizer step and the weight re-upload, each bracketed by a device
synchronisation (so the parts add up to a little more than the un-instrumented iteration)."""
import os, sys, time
os.environ["FB_NATIVE_ONLY"] = "1"; os.environ["FB_ITERS"] = "3"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "tools", "finetune_bench.py")).read().replace("print(json.dumps(", "(lambda *a: None)((")
g = {"__name__": "bench", "__file__": os.path.join(ROOT, "tools", "finetune_bench.py")}
exec(compile(src, "finetune_bench.py", "exec"), g)
import torch
model, d, opts = g["model"], g["d_ddim"], g["opts"]
acc = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
    return w
orig_call = model._native_train_call
def call(x, t, y):
    return timed("model call, %d clip(s)" % x.shape[0], orig_call)(x, t, y)
model._native_train_call = call
me = model.motion_enc
me_fwd = me.forward
me.forward = timed("motion encoder forward (64 clips)", me_fwd)
orig_engine = model.mst_engine
model.mst_engine = timed("engine lookup + weight re-upload", orig_engine)
def iteration():
    opt = opts["native"]
    opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    terms = d.few_shot_style_finetune_losses(model, g["t2m"], g["tt"], g["content"], g["style"], skip_steps=700, model_kwargs=g["y1"],
                                             model_t2m_kwargs=g["yB"], semantic_guidance=1, use_ddim=1, Ls=10)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    terms["loss"].backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    acc["objective (forward), all"] = acc.get("objective (forward), all", 0) + (t1 - t0) * 1e3
    acc["backward, all"] = acc.get("backward, all", 0) + (t2 - t1) * 1e3
    acc["optimizer step"] = acc.get("optimizer step", 0) + (t3 - t2) * 1e3
for _ in range(3): iteration()
acc.clear()
N = 5
for _ in range(N): iteration()
for k, v in acc.items():
    print(f"  {k:42s} {v / N:7.2f} ms per iteration")
