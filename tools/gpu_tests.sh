# Full GPU test suite in one process (what the driver runs at round end); log under gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -25 gpurun_out/gpu_tests.log
exit $rc
