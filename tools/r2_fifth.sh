cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/r2_tests5.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r2_tests5.log
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-boundary --no-cpu-baseline > gpurun_out/r2_bench4.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/r2_bench4.log | cut -c1-200
tail -1 gpurun_out/r2_bench4.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
for k,v in r['families'].items(): print('  ', k, v['avg_launch_us'], 'us', v['tflops'], 'TF')"
