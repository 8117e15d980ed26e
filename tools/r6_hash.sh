# Round 6: the keep-mask hash on 24-bit multiplies (keep_hash, csrc/mst_train.h): every training-side GPU test with it, the kernels it touches
# (rocprofv3 averages of tools/finetune_bench.py, old library / new), then the fine-tune line old / new alternating (tools/ft_lib_ab.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OLD=diffusion-based-motion-style-transfer_amd/csrc/ab_libs/lib_oldhash.so
timeout -k 10 1000 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fullsize.py tests/test_gpu_boundary.py tests/test_gpu_training_loop.py -x -q -m gpu > gpurun_out/r6_hash_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r6_hash_tests.log
[ $rc = 0 ] || { grep -E "^E |Error|assert" gpurun_out/r6_hash_tests.log | head -30; exit $rc; }
for v in $OLD default; do
  if [ "$v" = default ]; then unset MST_ENGINE_LIB; else export MST_ENGINE_LIB=$PWD/$v; fi
  rm -rf gpurun_out/prof_ft
  FB_ITERS=3 FB_NATIVE_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ft -- python3 tools/finetune_bench.py > gpurun_out/r6_hash_prof.log 2>&1 || { tail -5 gpurun_out/r6_hash_prof.log; exit 1; }
  echo "== $v"
  python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_ft/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(f"  all kernels: {sum(float(r['TotalDurationNs']) for r in rows) / 5e6:.2f} ms of launch time per iteration")
for r in rows[:14]:
    print(f'  {float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]:>5}  {r["Percentage"]:>6}%  {r["Name"][:96]}')
PY
done 2>&1 | tee gpurun_out/r6_hash_kernels.txt
unset MST_ENGINE_LIB
find gpurun_out/prof_ft -name "*kernel_trace.csv" -delete
bash tools/ft_lib_ab.sh $OLD default 2>&1 | tee gpurun_out/r6_hash_ab.txt
