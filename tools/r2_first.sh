# round-2 first GPU call: full GPU test suite, default bench, rocprof kernel stats with one stream (isolated 64-clip launches)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests.log 2>&1
echo "tests rc=$?"
tail -5 gpurun_out/r2_tests.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/r2_bench0.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/r2_bench0.log | cut -c1-600
MST_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof_s1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2_prof_s1.log 2>&1
echo "prof rc=$?"
find gpurun_out/r2_prof_s1 -name "*kernel_trace.csv" -delete
find gpurun_out/r2_prof_s1 -name "*_kernel_stats.csv" | head -3
