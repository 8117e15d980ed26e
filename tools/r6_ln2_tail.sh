# Round 6: LayerNorm2's backward at the head of the fused backward tail (frozen stacks; MST_TRAIN_FUSE_LN2_BWD=1): the training tests with it,
# the frozen stack backward's kernel averages off / on, then the fine-tune line off / on alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MST_TRAIN_FUSE_LN2_BWD=1 timeout -k 10 900 python -m pytest tests/test_gpu_train_fullsize.py tests/test_gpu_train.py tests/test_gpu_boundary.py tests/test_gpu_training_loop.py -x -q -m gpu > gpurun_out/r6_gpu_tests8.log 2>&1; rc=$?
tail -2 gpurun_out/r6_gpu_tests8.log
[ $rc = 0 ] || { grep -E "^E |Error|assert" gpurun_out/r6_gpu_tests8.log | head -30; exit $rc; }
for v in 0 1; do
rm -rf gpurun_out/prof_tb
TB_FROZEN=1 MST_TRAIN_FUSE_LN2_BWD=$v TB_NATIVE_ONLY=1 TB_ITERS=3 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tb -- python3 tools/train_bench.py > gpurun_out/r6_tb_prof.log 2>&1 || { tail -5 gpurun_out/r6_tb_prof.log; exit 1; }
echo "== frozen MST_TRAIN_FUSE_LN2_BWD=$v"; grep native_bwd_ms gpurun_out/r6_tb_prof.log | cut -c1-120
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_tb/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(f'{float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]:>5}  {r["Percentage"]:>6}%  {r["Name"][:100]}')
PY
done 2>&1 | tee gpurun_out/r6_ln2_tail_prof.txt
find gpurun_out/prof_tb -name "*kernel_trace.csv" -delete
for r in 1 2 3; do for v in 0 1; do
  MST_TRAIN_FUSE_LN2_BWD=$v timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r6_ft.log 2>&1 || { tail -5 gpurun_out/r6_ft.log; exit 1; }
  tail -1 gpurun_out/r6_ft.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MST_TRAIN_FUSE_LN2_BWD=$v', d['ms_per_step'], 'ms/iteration', 'loss', d.get('final_loss'))"
done; done 2>&1 | tee gpurun_out/r6_ln2_tail_ab.txt
