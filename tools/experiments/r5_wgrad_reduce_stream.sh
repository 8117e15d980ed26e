# MST_WGRAD_STREAM = 1 (wgrads beside the dgrad chain) against 2 (their reduces and bias sums on a third stream): training tests with 2, then
# the fine-tune iteration alternating, one box
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MST_WGRAD_STREAM=2 timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fullsize.py tests/test_gpu_boundary.py tests/test_gpu_training_loop.py -x -q > gpurun_out/wgr_tests.txt 2>&1 || { tail -30 gpurun_out/wgr_tests.txt; exit 1; }
tail -2 gpurun_out/wgr_tests.txt
for r in 1 2 3; do for v in 1 2; do
  MST_WGRAD_STREAM=$v timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/wgr.log 2>&1 || { tail -5 gpurun_out/wgr.log; exit 1; }
  tail -1 gpurun_out/wgr.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MST_WGRAD_STREAM=$v', d['ms_per_step'], 'ms/iteration')"
done; done
