# Round 6: what the fine-tune iteration launches that is NOT an engine kernel (torch fills, copies, elementwise glue): rocprofv3 --stats of
# bench.py --mode finetune (15 iterations in all: 2 warm-up + 10 timed + 3 with the wgrad events), calls and time per iteration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_glue
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_glue -- python3 bench.py --mode finetune --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r6_ft_glue.log 2>&1 || { tail -5 gpurun_out/r6_ft_glue.log; exit 1; }
python3 - <<'PY' | tee gpurun_out/r6_ft_glue.txt
import csv, glob
f = glob.glob("gpurun_out/prof_glue/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
IT = 15.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
eng = sum(float(r["TotalDurationNs"]) for r in rows if "mst" in r["Name"])
print(f"all kernels {tot / IT / 1e6:.2f} ms of launch time per iteration, engine kernels {eng / IT / 1e6:.2f} ms, everything else {(tot - eng) / IT / 1e6:.2f} ms")
print("not engine kernels, per iteration:")
for r in rows:
    if "mst" in r["Name"]:
        continue
    c, t = float(r["Calls"]) / IT, float(r["TotalDurationNs"]) / IT / 1e3
    if t >= 3.0:
        print(f"  {c:7.1f} calls {t:8.1f} us  {r['Name'][:110]}")
PY
find gpurun_out/prof_glue -name "*kernel_trace.csv" -delete
