"""Per-stream view of a rocprofv3 --kernel-trace CSV of the sampling loop: for every stream (queue) the kernels in start order,
their durations and the gaps between consecutive kernels of the same queue; summed per denoise step."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
def short(n):
    if "k_embed_out" in n:
        return "embed_out" if "ELi0EEEv" in n else "embed_step"        # embed_step: the output projection that also embeds the next step
    for k, v in (("k_layer_tail", "tail"), ("k_qkv_attention", "attn"), ("k_embed_in", "embed_in"), ("DEpiEmbedIn", "embed_in"), ("DEpiEmbedOut", "embed_out"), ("k_frames_f16", "frames")):
        if k in n:
            return v
    return "other"
by_q = collections.defaultdict(list)
for r in rows:
    by_q[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
for q, ks in sorted(by_q.items()):
    ks.sort()
    # keep the second half of the trace (the timed pass)
    ks = ks[len(ks) // 2:]
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    for (s0, e0, n0, g0), (s1, e1, n1, g1) in zip(ks, ks[1:]):
        dur[(n0, g0)].append((e0 - s0) * 1e-3)
        gap[(n0, n1)].append((s1 - e0) * 1e-3)
    steps = max(1, sum(1 for k in ks if k[2] in ("embed_out", "embed_step")))
    print(f"queue {q}: {len(ks)} kernels, {steps} steps, span {(ks[-1][1] - ks[0][0]) * 1e-3 / steps:.1f} us per step")
    tot_d = sum(sum(v) for v in dur.values()) / steps; tot_g = sum(sum(v) for v in gap.values()) / steps
    print(f"   per step: kernel durations {tot_d:.1f} us, gaps {tot_g:.1f} us")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        print(f"   dur  {k[0]:10s} grid {k[1]:8d}: n {len(v):5d}  median {v2[len(v2)//2]:7.2f}  mean {sum(v)/len(v):7.2f}  p90 {v2[int(len(v2)*0.9)]:7.2f}")
    for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        print(f"   gap  {k[0]:10s}->{k[1]:10s}: n {len(v):5d}  median {v2[len(v2)//2]:7.2f}  mean {sum(v)/len(v):7.2f}  p90 {v2[int(len(v2)*0.9)]:7.2f}")
