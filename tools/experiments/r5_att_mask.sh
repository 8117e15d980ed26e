# attention-site keep masks from one hash per PAIR of keys (att_hash): every test that touches dropout, the ABI tests, then the fine-tune
# iteration against the previous library, alternating
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fullsize.py tests/test_gpu_boundary.py tests/test_gpu_training_loop.py tests/test_gpu_fused_ops.py -x -q > gpurun_out/am_tests.txt 2>&1 || { tail -40 gpurun_out/am_tests.txt; exit 1; }
tail -2 gpurun_out/am_tests.txt
bash tools/ft_lib_ab.sh build/ab/prev.so default
