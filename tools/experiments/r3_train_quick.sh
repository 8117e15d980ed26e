# Round 3: training-path check: full-size + small training tests, then the fine-tune bench line.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train_fullsize.py tests/test_gpu_train.py tests/test_gpu_training_loop.py -x -q -m gpu > gpurun_out/r3_train_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3_train_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --mode finetune --steps 10 --warmup 3 > gpurun_out/r3_ft.log 2>&1 || exit 1
tail -1 gpurun_out/r3_ft.log | cut -c1-900
