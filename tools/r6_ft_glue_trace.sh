# Round 6: who launches the buffer fills / copies of a fine-tune iteration: kernel trace of bench.py --mode finetune, the engine kernels
# in front of and behind every fillBufferAligned / copyBuffer / FillFunctor on the same queue, counted
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_glue
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_glue -- python3 bench.py --mode finetune --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r6_ft_glue.log 2>&1 || { tail -5 gpurun_out/r6_ft_glue.log; exit 1; }
python3 - <<'PY' | tee gpurun_out/r6_ft_glue_trace.txt
import csv, glob, collections
f = glob.glob("gpurun_out/prof_glue/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
def short(n):
    n = n.replace("void ", "")
    for p in ("at::native::", "(anonymous namespace)::", "_ZN3mst"):
        n = n.replace(p, "")
    return n[:48]
for key in ("fillBufferAligned", "copyBuffer", "FillFunctor<float>"):
    cnt = collections.Counter()
    for q, lst in byq.items():
        for i, r in enumerate(lst):
            if key in r["Kernel_Name"]:
                prev = short(lst[i - 1]["Kernel_Name"]) if i else "-"
                nxt = short(lst[i + 1]["Kernel_Name"]) if i + 1 < len(lst) else "-"
                size = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
                cnt[(prev, nxt, size)] += 1
    print("==", key, sum(cnt.values()), "launches; (previous kernel on the queue, next kernel, grid) x count, /15 = per iteration")
    for (p, n, s), c in cnt.most_common(14):
        print(f"  {c:5d}  grid {s:>9}  after {p:48s} before {n}")
PY
find gpurun_out/prof_glue -name "*kernel_trace.csv" -delete
