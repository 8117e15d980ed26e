# Selected GPU tests with output: tools/experiments/gpu_tests_k.sh "<pytest -k expression>" [file]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest ${2:-tests} -x -q -m gpu -s -k "$1" > gpurun_out/gpu_tests_k.log 2>&1; rc=$?
grep -v "^$" gpurun_out/gpu_tests_k.log | tail -40
exit $rc
