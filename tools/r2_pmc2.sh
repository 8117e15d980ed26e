cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MST_STREAMS=1
B12="python3 bench.py --steps 1 --warmup 0 --denoise-steps 12 --no-cpu-baseline --no-boundary"
rm -rf gpurun_out/pmcS gpurun_out/pmcL
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcS -- $B12 > gpurun_out/pmcS.log 2>&1
echo "pmcS rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/pmcL -- $B12 > gpurun_out/pmcL.log 2>&1
echo "pmcL rc=$?"
python3 tools/pmc_summary.py gpurun_out/pmcS gpurun_out/pmcL 2>&1 | grep -E "layer_tail|qkv_attention|embed" > gpurun_out/r2_pmc_waves.txt
find gpurun_out/pmcS gpurun_out/pmcL -name "*kernel_trace.csv" -delete
cat gpurun_out/r2_pmc_waves.txt
tail -3 gpurun_out/pmcL.log | cut -c1-300
