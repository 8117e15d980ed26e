# Round 5: the hand-off cost of a per-clip persistent trunk (csrc/probes/group_chain.hip), then the GPU tests the round has touched
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin/group_chain
OUT=gpurun_out/r5_group_chain.txt
: > $OUT
for cfg in "8 21.6 38.0 33.0 0 0 0" "8 21.6 38.0 33.0 0 0 1" "8 21.6 38.0 33.0 1 0 0" "8 21.6 38.0 33.0 0 1 0" "8 21.6 38.0 33.0 0 1 1" "8 21.6 38.0 33.0 1 1 1" "8 21.6 38.0 38.0 0 1 1" "8 5.0 5.0 5.0 0 0 0" "8 5.0 5.0 5.0 0 0 1"; do
  timeout -k 10 120 $P $cfg >> $OUT 2>&1 || { echo "probe failed: $cfg" >> $OUT; cat $OUT; exit 1; }
done
cat $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_train.py tests/test_gpu_training_loop.py tests/test_gpu_train_fullsize.py -x -q -s -m gpu > gpurun_out/r5_t1.log 2>&1; echo "pytest rc=$?"
grep -E "worst|passed|failed|error|Error" gpurun_out/r5_t1.log | tail -30
