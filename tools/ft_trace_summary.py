"""Timeline of ONE fine-tune iteration from a rocprofv3 kernel trace (tools/ft_trace.sh): per queue busy time, time with no kernel at all
on the GPU, and the iteration cut into segments at the AdamW launch with the dominant kernels of each queue.
    python tools/ft_trace_summary.py <kernel_trace.csv> [iteration-from-the-end, default 2]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda e: e[0])
adam = [i for i, e in enumerate(ev) if "k_adamw_multi" in e[2]]
if len(adam) < back + 1:
    sys.exit("not enough iterations in the trace")
lo, hi = adam[-back - 1] + 1, adam[-back] + 1          # the kernels between two optimizer steps
it = ev[lo:hi]
t0, t1 = it[0][0], max(e[1] for e in it)
print(f"iteration: {(t1 - t0) / 1e6:.3f} ms, {len(it)} kernels on {len(set(e[3] for e in it))} queues")
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in it:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"some kernel running: {busy / 1e6:.3f} ms; GPU empty: {(t1 - t0 - busy) / 1e6:.3f} ms")
per_q = defaultdict(list)
for e in it:
    per_q[e[3]].append(e)
for q, es in sorted(per_q.items(), key=lambda kv: -sum(e[1] - e[0] for e in kv[1])):
    tot = sum(e[1] - e[0] for e in es)
    print(f"queue {q}: {len(es)} kernels, {tot / 1e6:.3f} ms busy, first at +{(es[0][0] - t0) / 1e6:.3f} ms, last ends +{(max(e[1] for e in es) - t0) / 1e6:.3f} ms")
# gaps > 20 us with nothing running
print("empty stretches > 15 us (offset ms, length us, kernel before -> kernel after):")
cur_e, last = None, None
for s, e, n, q in it:
    if cur_e is not None and s - cur_e > 15000:
        print(f"  +{(cur_e - t0) / 1e6:7.3f}  {(s - cur_e) / 1e3:7.1f}   {last[:50]} -> {n[:50]}")
    if cur_e is None or e > cur_e:
        cur_e, last = e, n
# 0.5-ms slices: which queues are active and their top kernel
print("per 0.5 ms: busy fraction per queue (top kernel)")
step = 500000
qs = sorted(per_q)
t = t0
while t < t1:
    line = f"  +{(t - t0) / 1e6:5.1f}:"
    for q in qs:
        acc = defaultdict(int)
        for s, e, n, _ in per_q[q]:
            o = min(e, t + step) - max(s, t)
            if o > 0:
                acc[n] += o
        tot = sum(acc.values())
        top = max(acc, key=acc.get)[:28] if acc else "-"
        line += f"  q{q} {tot / step:4.2f} {top:28s}"
    print(line)
    t += step
