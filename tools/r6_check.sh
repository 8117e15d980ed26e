# Round 6 check: the whole GPU suite in one process, then the fine-tune line and the headline line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6_gpu_tests.log 2>&1; rc=$?
tail -6 gpurun_out/r6_gpu_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r6_ft.log 2>&1 || { tail -5 gpurun_out/r6_ft.log; exit 1; }
tail -1 gpurun_out/r6_ft.log | cut -c1-400
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-boundary > gpurun_out/r6_bench.log 2>&1 || { tail -5 gpurun_out/r6_bench.log; exit 1; }
tail -1 gpurun_out/r6_bench.log | cut -c1-300
