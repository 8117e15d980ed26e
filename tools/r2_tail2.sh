# tail v2 (read-ahead fragments, x1 parked in the tile's stream rows): phase stamps, parity, same-box A/B against the previous library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PROBES="tail_clock tail_clock_NODMA" bash tools/r2_clock2.sh | grep "rep 2\|==" &&
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_boundary.py -m gpu -x -q > gpurun_out/r2_tests_tail2.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2_tests_tail2.log
bash tools/lib_ab.sh lib_prev_tail.so default
