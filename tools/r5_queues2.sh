# Round 5: (1) is a queue that only WAITS one of the four the hardware serves?  (2) the GPU tests the round has touched so far
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin/queues
OUT=gpurun_out/r5_queues_blocked.txt
: > $OUT
for b in 0 1 2; do
  echo "== probe blocked=$b (default queue count)" >> $OUT
  timeout -k 10 120 $P 30 40 400 0 $b >> $OUT 2>&1 || exit 1
done
cat $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_train.py tests/test_gpu_training_loop.py tests/test_gpu_train_fullsize.py -x -q -s -m gpu > gpurun_out/r5_t1.log 2>&1; echo "pytest rc=$?"
grep -E "worst|passed|failed|error|Error" gpurun_out/r5_t1.log | tail -30
