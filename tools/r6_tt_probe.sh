# Round 6: what the fused training tail's 104 us are made of: rocprofv3 per-kernel averages of the stack forward with the product build and
# three probe builds (no keep-mask hash / no accumulator-layout tape stores / neither; timing only, their results are wrong)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=diffusion-based-motion-style-transfer_amd/csrc/ab_libs
for v in default $L/lib_tt_nohash.so $L/lib_tt_nostore.so $L/lib_tt_neither.so; do
  if [ "$v" = default ]; then unset MST_ENGINE_LIB; else export MST_ENGINE_LIB=$PWD/$v; fi
  rm -rf gpurun_out/prof_tt
  TB_NATIVE_ONLY=1 TB_ITERS=3 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tt -- python3 tools/train_bench.py > gpurun_out/r6_tt_probe.log 2>&1 || { tail -5 gpurun_out/r6_tt_probe.log; exit 1; }
  echo "== $v"; grep native_fwd_ms gpurun_out/r6_tt_probe.log
  python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_tt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(f'{float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]:>5}  {r["Percentage"]:>6}%  {r["Name"][:80]}')
PY
done 2>&1 | tee gpurun_out/r6_tt_probe.txt
