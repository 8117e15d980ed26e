cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests7.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r2_tests7.log
grep -h "generated dataset vs reference\|neutral900\|motion_enc_mu" gpurun_out/r2_tests7.log | cut -c1-400
for g in 1 0; do
MST_GRAPH=$g timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-boundary --no-cpu-baseline > gpurun_out/r2_bench_graph$g.log 2>&1
echo "bench graph=$g rc=$?: $(tail -1 gpurun_out/r2_bench_graph$g.log | cut -c60-100)"
done
timeout -k 10 200 python tools/latency_b1.py > gpurun_out/r2_latency_b1.log 2>&1; tail -3 gpurun_out/r2_latency_b1.log
MST_GRAPH=0 timeout -k 10 200 python tools/latency_b1.py > gpurun_out/r2_latency_b1_nograph.log 2>&1; tail -3 gpurun_out/r2_latency_b1_nograph.log
