"""Is the fine-tune iteration host-bound or GPU-bound?  N iterations enqueued back to back WITHOUT a per-iteration synchronisation (as bench.py
--mode finetune runs them): the time the host needs to enqueue them against the time until the GPU has finished."""
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
model, d_ddim, opt = g["model"], g["d_ddim"], g["opts"]["native"]
t2m, tt, content, style, y1, yB = (g[k] for k in ("t2m", "tt", "content", "style", "y1", "yB"))


def iteration():
    opt.zero_grad(set_to_none=True)
    terms = d_ddim.few_shot_style_finetune_losses(model, t2m, tt, content, style, skip_steps=700, model_kwargs=y1, model_t2m_kwargs=yB,
                                                  semantic_guidance=1, use_ddim=1, Ls=10)
    terms["loss"].backward()
    opt.step()
    return terms["loss"]


for _ in range(3):
    iteration()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    iteration()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{N} iterations: host enqueue {1e3 * (t1 - t0) / N:.2f} ms per iteration, GPU done after {1e3 * (t2 - t0) / N:.2f} ms per iteration "
      f"(the host was {1e3 * (t2 - t1):.1f} ms ahead at the end)")
