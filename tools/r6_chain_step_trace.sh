# Round 6: the launches of ONE chained single-clip step of the fine-tune iteration, in order, with their durations and the gaps between them
# (rocprofv3 --kernel-trace of bench.py --mode finetune; the queue that carries k_rows_gemm; one step = from one k_step_epilogue to the next)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_chain
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_chain -- python3 bench.py --mode finetune --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r6_chain_trace.log 2>&1 || { tail -5 gpurun_out/r6_chain_trace.log; exit 1; }
python3 - <<'PY' | tee gpurun_out/r6_chain_step_trace.txt
import csv, glob, collections
f = glob.glob("gpurun_out/prof_chain/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
q = max(byq, key=lambda k: sum("k_rows_gemm" in r["Kernel_Name"] for r in byq[k]))
lst = byq[q]
def short(n):
    n = n.replace("void ", "")
    for p in ("at::native::", "(anonymous namespace)::", "_ZN3mst"):
        n = n.replace(p, "")
    return n[:70]
idx = [i for i, r in enumerate(lst) if "k_step_epilogue" in r["Kernel_Name"]]
# a step in the middle of the run: between the (len - 8)th and the next epilogue
a, b = idx[-9], idx[-8]
prev_end = int(lst[a]["End_Timestamp"])
tot = 0
print(f"one chained step on queue {q}: {b - a} launches")
for r in lst[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  gap {(s - prev_end) / 1e3:6.1f} us  run {(e - s) / 1e3:6.1f} us  {short(r['Kernel_Name'])}")
    prev_end = e
print(f"step length {(int(lst[b]['End_Timestamp']) - int(lst[a]['End_Timestamp'])) / 1e3:.1f} us; sum of runs {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in lst[a + 1:b + 1]) / 1e3:.1f} us")
PY
find gpurun_out/prof_chain -name "*kernel_trace.csv" -delete
