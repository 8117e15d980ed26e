# Round 6: what the fused backward tail's time is made of (frozen stack): rocprofv3 averages with the product build and three timing-only
# probe builds (-DMST_TB_PROBE=1: no LayerNorm-stage global loads / stores; 2: no out-proj phase; 3: no pre loads / GELU'); results wrong
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=diffusion-based-motion-style-transfer_amd/csrc/ab_libs
for v in default $L/lib_tb_probe1.so $L/lib_tb_probe2.so $L/lib_tb_probe3.so; do
  if [ "$v" = default ]; then unset MST_ENGINE_LIB; else export MST_ENGINE_LIB=$PWD/$v; fi
  rm -rf gpurun_out/prof_tb
  TB_FROZEN=1 TB_NATIVE_ONLY=1 TB_ITERS=3 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tb -- python3 tools/train_bench.py > gpurun_out/r6_tb_prof.log 2>&1 || { tail -5 gpurun_out/r6_tb_prof.log; exit 1; }
  echo "== $v"
  python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_tb/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if "tail_bwd" in r["Name"]: print(f'{float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]:>5}  {r["Name"][:60]}')
PY
done 2>&1 | tee gpurun_out/r6_tb_probe2.txt
find gpurun_out/prof_tb -name "*kernel_trace.csv" -delete
