cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FB_ITERS=3 FB_NATIVE_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ft -- python3 tools/finetune_bench.py > gpurun_out/prof_ft.log 2>&1
echo rc=$?
find gpurun_out/prof_ft -name "*kernel_trace.csv" -delete
