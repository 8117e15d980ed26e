# Round 6: the text branch's backward pass started without waiting for the chain's forward calls (overlap_backward) and the chain's sums
# joined at the end of the pass (MST_CHAIN_JOIN_LATE): the training-side GPU tests, then the fine-tune line with the four switch combinations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_training_loop.py tests/test_gpu_train.py tests/test_gpu_train_fullsize.py -x -q -m gpu > gpurun_out/r6_gpu_tests9.log 2>&1; rc=$?
tail -2 gpurun_out/r6_gpu_tests9.log
[ $rc = 0 ] || { grep -E "^E |Error|assert" gpurun_out/r6_gpu_tests9.log | head -40; exit $rc; }
for r in 1 2 3; do for v in "0 0" "0 1" "1 0" "1 1"; do
  set -- $v
  MST_FT_OVERLAP_BACKWARD=$1 MST_CHAIN_JOIN_LATE=$2 timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r6_ft.log 2>&1 || { tail -5 gpurun_out/r6_ft.log; exit 1; }
  tail -1 gpurun_out/r6_ft.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap_backward=$1 join_late=$2', d['ms_per_step'], 'ms/iteration, host enqueue', d['host_enqueue_ms_per_step'], 'ms, loss', d.get('final_loss'))"
done; done 2>&1 | tee gpurun_out/r6_overlap_ab.txt
