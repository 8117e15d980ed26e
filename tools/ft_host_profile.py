"""Where the HOST spends a fine-tune iteration (bench.py --mode finetune enqueues 8.3 ms of Python + launches per 9.6-ms iteration: on a
busy host the loop turns host-bound).  cProfile over N un-synchronised iterations, top functions by own time and by cumulative time."""
import cProfile, io, os, pstats, sys
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
import time
import torch
model, d_ddim, opt, dev = g["model"], g["d_ddim"], g["opts"]["native"], g["dev"]
t2m, tt, content, style, y1, yB = (g[k] for k in ("t2m", "tt", "content", "style", "y1", "yB"))


def iteration():
    opt.zero_grad(set_to_none=True)
    terms = d_ddim.few_shot_style_finetune_losses(model, t2m, tt, content, style, skip_steps=700, model_kwargs=y1, model_t2m_kwargs=yB,
                                                  semantic_guidance=1, use_ddim=1, Ls=10)
    terms["loss"].backward()
    opt.step()


for _ in range(5):
    iteration()
torch.cuda.synchronize()
N = int(os.environ.get("FT_HOST_ITERS", "40"))
t0 = time.perf_counter()
for _ in range(N):
    iteration()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"un-profiled: host enqueue {1e3 * (t1 - t0) / N:.2f} ms per iteration, wall {1e3 * (t2 - t0) / N:.2f} ms per iteration")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    iteration()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print(f"== by {key} (totals over {N} iterations; /{N} = per iteration)")
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[6:]))
