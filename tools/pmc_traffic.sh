# HBM traffic of full-batch launches (one clip slice: MST_STREAMS=1) from two separate --pmc passes, as the microarch
# guide prescribes: FETCH_SIZE and WRITE_SIZE are reported in KiB; FETCH_SIZE under-counts wide coalesced reads 2x on gfx950.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MST_STREAMS=1
B="python3 bench.py --steps 1 --warmup 0 --denoise-steps 12 --no-cpu-baseline"
rm -rf gpurun_out/pmcF gpurun_out/pmcW
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcF -- $B > gpurun_out/pmcF.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcW -- $B > gpurun_out/pmcW.log 2>&1
echo rc=$?
python3 tools/pmc_traffic.py gpurun_out/pmcF gpurun_out/pmcW > gpurun_out/pmc_traffic.json
cat gpurun_out/pmc_traffic.json
find gpurun_out/pmcF gpurun_out/pmcW -name "*kernel_trace.csv" -delete
