# Round 4: kernel trace of the default three-slice loop (short run) -> per-stream gaps and durations (tools/trace_summary.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace3
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace3 -- python3 bench.py --steps 1 --warmup 1 --denoise-steps 40 --no-cpu-baseline --no-boundary > gpurun_out/trace3.log 2>&1
echo rc=$?
python3 tools/trace_summary.py gpurun_out/trace3 > gpurun_out/r4_trace_summary.txt
cat gpurun_out/r4_trace_summary.txt
find gpurun_out/trace3 -name "*kernel_trace.csv" -size +20M -delete
