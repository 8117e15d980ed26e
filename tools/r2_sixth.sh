cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests6.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r2_tests6.log
timeout -k 10 300 python bench.py --mode finetune --steps 5 --warmup 2 > gpurun_out/r2_ft3.log 2>&1
echo "ft rc=$?"; tail -1 gpurun_out/r2_ft3.log | cut -c1-260
