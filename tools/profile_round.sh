# rocprofv3 kernel stats of a bench run + PMC passes; summaries only are kept (traces are large)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_final_bench.log 2>&1
echo rc=$?
find gpurun_out/prof_final -name "*kernel_trace.csv" -delete
bash tools/pmc_passes.sh > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC gpurun_out/pmcD gpurun_out/pmcE > gpurun_out/pmc_summary.txt 2>&1
find gpurun_out/pmc? -name "*kernel_trace.csv" -delete
ls gpurun_out/prof_final/*/
tail -1 gpurun_out/prof_final_bench.log | cut -c1-300
