# Round 5: the whole GPU suite, then the record run (tools/r5_profiles.sh), one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5_gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r5_gpu_tests.log
[ $rc -eq 0 ] || exit 1
bash tools/r5_profiles.sh
