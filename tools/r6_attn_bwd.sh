# Round 6: k_attention_bwd's pass 1 on the forward's row statistics (lse from the tape; no max / sum sweeps, one score tile alive at a time):
# training-side tests, kernel averages of the stack's backward pass (previous commit's library / this one), fine-tune line alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OLD=diffusion-based-motion-style-transfer_amd/csrc/ab_libs/lib_prev.so
timeout -k 10 1000 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fullsize.py tests/test_gpu_boundary.py tests/test_gpu_training_loop.py -x -q -m gpu > gpurun_out/r6_attn_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r6_attn_tests.log
[ $rc = 0 ] || { grep -E "^E |Error|assert" gpurun_out/r6_attn_tests.log | head -30; exit $rc; }
for v in $OLD default; do
  if [ "$v" = default ]; then unset MST_ENGINE_LIB; else export MST_ENGINE_LIB=$PWD/$v; fi
  for fr in "" 1; do
  rm -rf gpurun_out/prof_tb
  TB_FROZEN=$fr TB_NATIVE_ONLY=1 TB_ITERS=3 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tb -- python3 tools/train_bench.py > gpurun_out/r6_tb_prof.log 2>&1 || { tail -5 gpurun_out/r6_tb_prof.log; exit 1; }
  echo "== $v frozen='$fr'"; grep native_bwd_ms gpurun_out/r6_tb_prof.log | cut -c1-110
  python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_tb/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attention_bwd" in r["Name"]:
        print(f'  {float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]:>5}  min {float(r["MinNs"])/1e3:6.1f}  {r["Name"][:50]}')
PY
  done
done 2>&1 | tee gpurun_out/r6_attn_bwd_kernels.txt
unset MST_ENGINE_LIB
find gpurun_out/prof_tb -name "*kernel_trace.csv" -delete
bash tools/ft_lib_ab.sh $OLD default 2>&1 | tee gpurun_out/r6_attn_bwd_ab.txt
