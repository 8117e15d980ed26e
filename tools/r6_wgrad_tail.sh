# Round 6: the last-differentiated layer's weight gradients on twice the workgroups (MST_WGRAD_TAIL_WIDE=0 / 1 alternating); tests first
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fullsize.py tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/r6_gpu_tests4.log 2>&1; rc=$?
tail -3 gpurun_out/r6_gpu_tests4.log
[ $rc = 0 ] || exit $rc
for r in 1 2 3; do for v in 0 1; do
  MST_WGRAD_TAIL_WIDE=$v timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r6_ft.log 2>&1 || { tail -5 gpurun_out/r6_ft.log; exit 1; }
  tail -1 gpurun_out/r6_ft.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MST_WGRAD_TAIL_WIDE=$v', d['ms_per_step'], 'ms/iteration', 'loss', d.get('final_loss'))"
done; done 2>&1 | tee gpurun_out/r6_wgrad_tail_ab.txt
