# Round 6: the caller's stream over ONE fine-tune iteration: launches grouped by kernel (count, run time), gaps, and the small ones in order
# around the backward pass's seams (rocprofv3 --kernel-trace of bench.py --mode finetune; the queue that carries k_layer_tail_train)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_chain
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_chain -- python3 bench.py --mode finetune --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r6_chain_trace.log 2>&1 || { tail -5 gpurun_out/r6_chain_trace.log; exit 1; }
python3 - <<'PY' | tee gpurun_out/r6_main_queue_trace.txt
import csv, glob, collections
f = glob.glob("gpurun_out/prof_chain/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
q = max(byq, key=lambda k: sum("k_layer_tail_train" in r["Kernel_Name"] for r in byq[k]))
lst = byq[q]
def short(n):
    n = n.replace("void ", "")
    for p in ("at::native::", "(anonymous namespace)::", "_ZN3mst"):
        n = n.replace(p, "")
    return n[:84]
ad = [i for i, r in enumerate(lst) if "k_adamw_multi" in r["Kernel_Name"]]
a, b = ad[-3], ad[-2]                        # one whole iteration: behind one optimizer step up to and including the next
seg = lst[a + 1:b + 1]
agg = collections.OrderedDict()
prev_end = int(lst[a]["End_Timestamp"]); gaps = 0; biggaps = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = short(r["Kernel_Name"])
    c = agg.setdefault(k, [0, 0.0])
    c[0] += 1; c[1] += (e - s) / 1e3
    g = max(0, s - prev_end) / 1e3
    gaps += g
    if g > 40: biggaps.append((g, k))
    prev_end = e
span = (int(seg[-1]["End_Timestamp"]) - int(lst[a]["End_Timestamp"])) / 1e3
print(f"caller's queue {q}: {len(seg)} launches per iteration; span {span:.0f} us, sum of runs {sum(v[1] for v in agg.values()):.0f} us, sum of gaps {gaps:.0f} us")
print("gaps above 40 us (in front of):", ", ".join(f"{g:.0f} us {k[:40]}" for g, k in biggaps[:12]))
small = [(n, t, k) for k, (n, t) in agg.items() if t / n < 12]
print(f"launches shorter than 12 us on average: {sum(n for n, _, _ in small)} of {len(seg)}, {sum(t for _, t, _ in small):.0f} us of run time")
for n, t, k in sorted(small, key=lambda x: -x[0])[:30]:
    print(f"  {n:4d} x {t / n:6.1f} us  {k}")
PY
find gpurun_out/prof_chain -name "*kernel_trace.csv" -delete
