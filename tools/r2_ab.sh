cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/r2_tests_ab.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r2_tests_ab.log
bash tools/lib_ab.sh lib_prev_attn.so default
