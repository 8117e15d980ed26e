# round-2 record run (after the VALU / bias-staging work): full GPU suite, default bench, kernel stats at one stream, PMC passes,
# CFG and batch-128 lines, fine-tune line, in-kernel phase stamps with their ablations, board power beside the loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2f_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r2f_tests.log
timeout -k 10 500 python bench.py > gpurun_out/r2f_bench_default.json 2> gpurun_out/r2f_bench_default.err
echo "bench rc=$?"; cut -c1-160 gpurun_out/r2f_bench_default.json
timeout -k 10 300 python bench.py --steps 1 --warmup 1 --cfg --no-cpu-baseline > gpurun_out/r2f_bench_cfg.json 2>/dev/null; echo "cfg $(cut -c60-130 gpurun_out/r2f_bench_cfg.json)"
timeout -k 10 300 python bench.py --steps 1 --warmup 1 --batch 128 --no-cpu-baseline > gpurun_out/r2f_bench_batch128.json 2>/dev/null; echo "b128 $(cut -c60-130 gpurun_out/r2f_bench_batch128.json)"
timeout -k 10 300 python bench.py --mode finetune --steps 10 --warmup 3 > gpurun_out/r2f_bench_finetune.json 2>/dev/null; echo "ft $(cut -c1-200 gpurun_out/r2f_bench_finetune.json)"
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin
( for b in attn_clock attn_clock_NODMA attn_clock_NOMMA attn_clock_NOREAD attn_clock_READONLY attn_clock_NODMA_NOREAD attn_clock_NODMA_READONLY attn_clock_PRODUCER tail_clock tail_clock_NODMA tail_clock_NOMMA tail_clock_NODMA_NOREAD tail_clock_NODMA_READONLY; do echo "== $b"; timeout -k 10 100 $P/$b | tail -1 || exit 1; done ) > gpurun_out/r2f_phase_stamps.txt 2>&1
echo "stamps rc=$?"
export MST_STREAMS=1
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-boundary"
rm -rf gpurun_out/r2f_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f_prof -- $B > gpurun_out/r2f_bench_under_rocprof_streams1.json 2> gpurun_out/r2f_prof.err
echo "prof rc=$?"
find gpurun_out/r2f_prof -name "*kernel_trace.csv" -delete
B12="python3 bench.py --steps 1 --warmup 0 --denoise-steps 12 --no-cpu-baseline --no-boundary"
rm -rf gpurun_out/pmcF gpurun_out/pmcW gpurun_out/pmcM gpurun_out/pmcS gpurun_out/pmcL
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcF -- $B12 > gpurun_out/pmcF.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcW -- $B12 > gpurun_out/pmcW.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcM -- $B12 > gpurun_out/pmcM.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcS -- $B12 > gpurun_out/pmcS.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/pmcL -- $B12 > gpurun_out/pmcL.log 2>&1
echo "pmc rc=$?"
python3 tools/pmc_traffic.py gpurun_out/pmcF gpurun_out/pmcW > gpurun_out/r2f_pmc_traffic.json
python3 tools/pmc_summary.py gpurun_out/pmcM > gpurun_out/r2f_pmc_mfma_busy.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/pmcS gpurun_out/pmcL 2>&1 | grep -E "layer_tail|qkv_attention|embed" > gpurun_out/r2f_pmc_wave_states_lds.txt
find gpurun_out/pmcF gpurun_out/pmcW gpurun_out/pmcM gpurun_out/pmcS gpurun_out/pmcL -name "*kernel_trace.csv" -delete
ls gpurun_out/r2f_prof/*/ 2>/dev/null | head
