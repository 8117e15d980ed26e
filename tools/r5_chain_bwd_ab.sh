# Round 5: the chained calls' ONE backward pass on the side stream into accumulators of its own (default) against in line on the caller's stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_train.py tests/test_gpu_training_loop.py tests/test_gpu_train_fullsize.py -x -q -m gpu > gpurun_out/r5_t2.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r5_t2.log
[ $rc -eq 0 ] || exit 1
for r in 1 2 3; do for v in 0 1; do
  MST_CHAIN_BWD_SIDE=$v timeout -k 10 300 python bench.py --mode finetune --steps 20 --warmup 3 > gpurun_out/ftab.log 2>&1; tail -1 gpurun_out/ftab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chain backward on the side stream=$v', d['value'], d['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r5_chain_bwd_ab.txt
