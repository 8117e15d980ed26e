cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_boundary.py -m gpu -x -q > gpurun_out/r2_tests8.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r2_tests8.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-boundary --no-cpu-baseline > gpurun_out/r2_bench5.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/r2_bench5.log | cut -c1-200
tail -1 gpurun_out/r2_bench5.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ev', r['event_pair_overhead_us_subtracted'], r['kernel'], r['frac'], r['avg_launch_us'])
for k,v in r['families'].items(): print('  ', k, v['avg_launch_us'], 'us', v['tflops'], 'TF', v['hbm_gbps'], 'GB/s')"
HB_BATCHES=64,1 timeout -k 10 200 python tools/host_bound.py 2>&1 | tail -2
HB_BATCHES=64,1 MST_GRAPH=1 timeout -k 10 200 python tools/host_bound.py 2>&1 | tail -2
