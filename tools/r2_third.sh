cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_boundary.py -m gpu -x -q > gpurun_out/r2_tests3.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r2_tests3.log
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-boundary --no-cpu-baseline > gpurun_out/r2_bench2.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/r2_bench2.log | cut -c1-300
MST_FUSE_TAIL=0 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-boundary --no-cpu-baseline > gpurun_out/r2_bench2u.log 2>&1
echo "bench unfused rc=$?"; tail -1 gpurun_out/r2_bench2u.log | cut -c1-300
