"""Host-side profile of one fine-tune iteration (cProfile, top cumulative entries) + enqueue vs total time."""
import cProfile, io, os, pstats, sys, time
os.environ["FB_NATIVE_ONLY"] = "1"
os.environ["FB_ITERS"] = "2"
sys.argv = ["finetune_bench.py"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "tools", "finetune_bench.py")).read().replace('res["native_ms"] = timed(', 'res["native_ms"] = 0 and timed(').replace('res["native_loss"] = iteration()', 'pass').replace('res["clips_per_s_native"]', 'res["x"] = 0 #').replace("print(json.dumps(", "(lambda *a: None)((")
g = {"__name__": "bench", "__file__": os.path.join(ROOT, "tools", "finetune_bench.py")}
try:
    exec(compile(src, "finetune_bench.py", "exec"), g)
except SystemExit:
    pass
import torch
it = g["iteration"]
for _ in range(2):
    it()
torch.cuda.synchronize()
t0 = time.perf_counter(); it(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"iteration returns after {1e3*(t1-t0):.1f} ms (includes the loss .item() sync), GPU idle at {1e3*(t2-t0):.1f} ms")
pr = cProfile.Profile(); pr.enable(); it(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
