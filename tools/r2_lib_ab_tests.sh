cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests_ab.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2_tests_ab.log
bash tools/lib_ab.sh lib_prev_tail.so default
