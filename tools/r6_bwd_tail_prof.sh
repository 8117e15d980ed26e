# Round 6: per-kernel averages of the stack backward with the fused backward tail (rocprofv3 --stats of tools/train_bench.py): with
# parameter gradients, and frozen (TB_FROZEN=1: input gradient only, what the motion encoder's backward is), fused off / on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for fr in "" 1; do for v in 0 2; do
rm -rf gpurun_out/prof_tb
TB_FROZEN=$fr MST_TRAIN_FUSE_BWD_TAIL=$v TB_NATIVE_ONLY=1 TB_ITERS=3 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tb -- python3 tools/train_bench.py > gpurun_out/r6_tb_prof.log 2>&1 || { tail -5 gpurun_out/r6_tb_prof.log; exit 1; }
echo "== frozen='$fr' MST_TRAIN_FUSE_BWD_TAIL=$v"; grep native_bwd_ms gpurun_out/r6_tb_prof.log | cut -c1-120
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_tb/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(f'{float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]:>5}  {r["Percentage"]:>6}%  {r["Name"][:100]}')
PY
done; done 2>&1 | tee gpurun_out/r6_bwd_tail_prof.txt
find gpurun_out/prof_tb -name "*kernel_trace.csv" -delete
