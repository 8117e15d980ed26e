# round-2 record run: full GPU suite, the default bench line (as the driver runs it), kernel stats at one stream, PMC traffic / MFMA-busy passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests_final.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r2_tests_final.log
timeout -k 10 500 python bench.py --steps 5 --warmup 2 > gpurun_out/r2_bench_final.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/r2_bench_final.log | cut -c1-200
export MST_STREAMS=1
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-boundary"
rm -rf gpurun_out/r2_prof_final
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof_final -- $B > gpurun_out/r2_prof_final.log 2>&1
echo "prof rc=$?"
find gpurun_out/r2_prof_final -name "*kernel_trace.csv" -delete
B12="python3 bench.py --steps 1 --warmup 0 --denoise-steps 12 --no-cpu-baseline --no-boundary"
rm -rf gpurun_out/pmcF gpurun_out/pmcW gpurun_out/pmcM
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcF -- $B12 > gpurun_out/pmcF.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcW -- $B12 > gpurun_out/pmcW.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcM -- $B12 > gpurun_out/pmcM.log 2>&1
echo "pmc rc=$?"
python3 tools/pmc_traffic.py gpurun_out/pmcF gpurun_out/pmcW > gpurun_out/r2_pmc_traffic.json
python3 tools/pmc_summary.py gpurun_out/pmcM > gpurun_out/r2_pmc_mfma.txt 2>&1
find gpurun_out/pmcF gpurun_out/pmcW gpurun_out/pmcM -name "*kernel_trace.csv" -delete
