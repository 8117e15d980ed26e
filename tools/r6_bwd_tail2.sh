# Round 6: the fused backward tail, iterate: training tests with it on, then stack backward frozen / with gradients, fused off / on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MST_TRAIN_FUSE_BWD_TAIL=2 timeout -k 10 900 python -m pytest tests/test_gpu_train_fullsize.py tests/test_gpu_train.py tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/r6_gpu_tests6.log 2>&1; rc=$?
tail -3 gpurun_out/r6_gpu_tests6.log
[ $rc = 0 ] || { grep -E "^E |Error|assert" gpurun_out/r6_gpu_tests6.log | head -30; exit $rc; }
bash tools/r6_bwd_tail_prof.sh 2>&1 | grep -E "^==|native_bwd|k_layer_tail_bwd"
