# Round 6: the chained calls' ONE backward pass on the side stream: its launches grouped by kernel (count, run time) and the gaps between
# them (rocprofv3 --kernel-trace of bench.py --mode finetune; the queue that carries k_rows_gemm; from a k_masked_l2 backward to the next bernoulli)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_chain
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_chain -- python3 bench.py --mode finetune --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r6_chain_trace.log 2>&1 || { tail -5 gpurun_out/r6_chain_trace.log; exit 1; }
python3 - <<'PY' | tee gpurun_out/r6_chain_bwd_trace.txt
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
    return n[:84]
ep = [i for i, r in enumerate(lst) if "k_step_epilogue" in r["Kernel_Name"]]
# the last epilogue of an iteration is followed (somewhere) by backward kernels, then by the next iteration's bernoulli
starts = [i for j, i in enumerate(ep) if j + 1 == len(ep) or not any("k_rows_gemm" in r["Kernel_Name"] and "FfnTrain" in r["Kernel_Name"] for r in lst[i:ep[j + 1]][:3])]
a = ep[-7]                                   # last step of the second-to-last iteration
b = next(i for i in range(a + 1, len(lst)) if "bernoulli" in lst[i]["Kernel_Name"])
seg = lst[a + 1:b]
agg = collections.OrderedDict()
prev_end = int(lst[a]["End_Timestamp"]); gaps = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = short(r["Kernel_Name"])
    c = agg.setdefault(k, [0, 0.0])
    c[0] += 1; c[1] += (e - s) / 1e3
    gaps += max(0, s - prev_end) / 1e3
    prev_end = e
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
print(f"side queue {q}: {len(seg)} launches between the last chained step and the next iteration; span {span:.0f} us, sum of runs {sum(v[1] for v in agg.values()):.0f} us, sum of gaps {gaps:.0f} us")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {n:4d} x {t / n:6.1f} us = {t:7.1f} us  {k}")
PY
find gpurun_out/prof_chain -name "*kernel_trace.csv" -delete
