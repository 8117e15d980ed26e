# kernel trace of the fine-tune iteration (bench.py --mode finetune) -> gpurun_out/fttrace/*_kernel_trace.csv, then tools/ft_trace_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/fttrace && mkdir -p $R/gpurun_out/fttrace
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fttrace -o ft -- python3 bench.py --mode finetune --steps 8 --warmup 3 --no-cpu-baseline > $R/gpurun_out/fttrace/run.txt 2>&1 || { tail -20 $R/gpurun_out/fttrace/run.txt; exit 1; }
tail -1 $R/gpurun_out/fttrace/run.txt | cut -c1-300
python3 tools/ft_trace_summary.py $(find $R/gpurun_out/fttrace -name "*kernel_trace.csv" | head -1) | tee $R/gpurun_out/fttrace/summary.txt
