# Round 6: k_wgrad_tr with a deeper slab ring (MST_WG_NSTAGE builds under csrc/ab_libs/) and with every split's tiles on one XCD
# (MST_WGRAD_XCD=1): training tests with the XCD map, then the stack backward alone (train_bench.py under rocprofv3: wgrad and reduce averages)
# for each variant, then the fine-tune line alternating over the variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
A=diffusion-based-motion-style-transfer_amd/csrc/ab_libs
MST_WGRAD_XCD=1 timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fullsize.py -x -q -m gpu > gpurun_out/r6_wgrad_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r6_wgrad_tests.log
[ $rc = 0 ] || { grep -E "^E |Error|assert" gpurun_out/r6_wgrad_tests.log | head -30; exit $rc; }
for v in "default 0" "default 1" "$A/lib_wg_nstage5.so 0" "$A/lib_wg_nstage6.so 0" "$A/lib_wg_nstage6.so 1"; do
  set -- $v
  if [ "$1" = default ]; then unset MST_ENGINE_LIB; else export MST_ENGINE_LIB=$PWD/$1; fi
  rm -rf gpurun_out/prof_tb
  MST_WGRAD_XCD=$2 TB_NATIVE_ONLY=1 TB_ITERS=3 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tb -- python3 tools/train_bench.py > gpurun_out/r6_tb_prof.log 2>&1 || { tail -5 gpurun_out/r6_tb_prof.log; exit 1; }
  echo "== lib $1 MST_WGRAD_XCD=$2"; grep native_bwd_ms gpurun_out/r6_tb_prof.log | cut -c1-110
  python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_tb/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("wgrad_tr", "splitk_reduce", "attention_bwd")):
        print(f'  {float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]:>5}  {r["Percentage"]:>6}%  {r["Name"][:60]}')
PY
done 2>&1 | tee gpurun_out/r6_wgrad_ring.txt
unset MST_ENGINE_LIB
find gpurun_out/prof_tb -name "*kernel_trace.csv" -delete
for r in 1 2 3; do for v in "default 0" "default 1" "$A/lib_wg_nstage6.so 0" "$A/lib_wg_nstage6.so 1"; do
  set -- $v
  if [ "$1" = default ]; then unset MST_ENGINE_LIB; else export MST_ENGINE_LIB=$PWD/$1; fi
  MST_WGRAD_XCD=$2 timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/ftab.log 2>&1 || { tail -5 gpurun_out/ftab.log; exit 1; }
  tail -1 gpurun_out/ftab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib $1 xcd=$2', d['ms_per_step'], 'ms/iteration')"
done; done 2>&1 | tee gpurun_out/r6_wgrad_ring_ab.txt
