# Round 6 record run: every file of profiles/r06_* that this script names comes from ONE box and ONE build (the same-box A/Bs of the round have
# scripts of their own: r6_train_tail*.sh, r6_tt_probe.sh, r6_tt_nt.sh, r6_small_ln.sh, r6_ft_events.sh).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; rm -rf $O; mkdir -p $O
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin
step() { echo "== $1"; }
step "pmc traffic (first: bench.py quotes roofline.traffic only from a file whose source hash is this build's)"
B="python3 bench.py --steps 1 --warmup 0 --denoise-steps 12 --no-cpu-baseline --no-boundary"
MST_STREAMS=1 MST_FUSE_EMBED=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcF -- $B > $O/pmcF.log 2>&1 &&
MST_STREAMS=1 MST_FUSE_EMBED=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcW -- $B > $O/pmcW.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $O/pmcF $O/pmcW > $O/r06_pmc_traffic.json; head -30 $O/r06_pmc_traffic.json
cp $O/r06_pmc_traffic.json profiles/r06_pmc_traffic.json
find $O/pmc?* -name "*kernel_trace.csv" -delete
step "bench default";  timeout -k 10 600 python bench.py > $O/bench_default.log 2>&1 || exit 1; tail -1 $O/bench_default.log > $O/r06_bench_default.json; cut -c1-400 $O/r06_bench_default.json
step "bench, driver form"; timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.log 2>&1 || exit 1; tail -1 $O/bench_driver.log > $O/r06_bench_driver_form_steps20.json; cut -c1-200 $O/r06_bench_driver_form_steps20.json
step "bench cfg";      timeout -k 10 300 python bench.py --cfg --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_cfg.log 2>&1 || exit 1; tail -1 $O/bench_cfg.log > $O/r06_bench_cfg.json; cut -c1-200 $O/r06_bench_cfg.json
step "bench batch 128"; timeout -k 10 300 python bench.py --batch 128 --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_b128.log 2>&1 || exit 1; tail -1 $O/bench_b128.log > $O/r06_bench_batch128.json; cut -c1-200 $O/r06_bench_batch128.json
step "bench batch 32"; timeout -k 10 300 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_b32.log 2>&1 || exit 1; tail -1 $O/bench_b32.log > $O/r06_bench_batch32.json; cut -c1-200 $O/r06_bench_batch32.json
step "bench, resident-group trunk (MST_TRUNK=1)"; MST_TRUNK=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_trunk.log 2>&1 || exit 1; tail -1 $O/bench_trunk.log > $O/r06_bench_resident_trunk.json; cut -c1-200 $O/r06_bench_resident_trunk.json
step "pmc traffic of the training kernels (bench.py --mode finetune quotes roofline.traffic from it for this build)"
bash tools/r6_train_pmc.sh > $O/train_pmc.log 2>&1 && cp gpurun_out/r06_train_pmc_traffic.json $O/ && cp gpurun_out/r06_train_pmc_traffic.json profiles/ && cp gpurun_out/r6_train_pmc.txt $O/r06_train_kernels_hbm_traffic.txt; head -5 $O/r06_train_kernels_hbm_traffic.txt | cut -c1-120
step "finetune bench"; timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 > $O/bench_ft.log 2>&1 || exit 1; tail -1 $O/bench_ft.log > $O/r06_finetune_bench_1gpu.json; cut -c1-300 $O/r06_finetune_bench_1gpu.json
step "finetune bench, MST_CHAIN=0 MST_CHAIN_STREAM=0 (every model call differentiated alone, one stream)"; MST_CHAIN=0 MST_CHAIN_STREAM=0 timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 > $O/bench_ft0.log 2>&1 || exit 1; tail -1 $O/bench_ft0.log > $O/r06_finetune_bench_1gpu_unchained.json; cut -c1-200 $O/r06_finetune_bench_1gpu_unchained.json
step "finetune timeline, events, no synchronisation (round 5's stream protocol, then overlap_backward: LAB_NOTES R6.10)"
for v in 0 1; do echo "== overlap_backward=$v"; FT_QUICK=1 FT_OVERLAP=$v timeout -k 10 300 python tools/ft_events.py 2>&1 | grep -E "^timestep|^caller|^  " | grep -v "it/s\|return"; done > $O/r06_finetune_events.txt; cat $O/r06_finetune_events.txt
step "finetune segments"; timeout -k 10 300 python tools/finetune_segments.py 2>&1 | tail -7 > $O/r06_finetune_segments.txt; cat $O/r06_finetune_segments.txt
step "kernel stats, one slice"
MST_STREAMS=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-boundary > $O/prof_bench.log 2>&1 || exit 1
grep "^{\"metric\"" $O/prof_bench.log | tail -1 > $O/r06_bench_under_rocprof_streams1.json
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/r06_kernel_stats_bench_steps1_streams1.csv; find $O/prof -name "*kernel_trace.csv" -delete
head -6 $O/r06_kernel_stats_bench_steps1_streams1.csv | cut -c1-160
step "kernel trace, three slices (the timed path)"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace3 -- python3 bench.py --steps 1 --warmup 1 --denoise-steps 40 --no-cpu-baseline --no-boundary > $O/trace3.log 2>&1 || exit 1
python3 tools/trace_summary.py $O/trace3 > $O/r06_three_slice_trace_summary.txt; grep -E "^queue|dur  (tail|attn|embed)" $O/r06_three_slice_trace_summary.txt | head -20
export MST_STREAMS=1
step "pmc mfma / lds"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmcA -- $B > $O/pmcA.log 2>&1 || exit 1
python3 tools/pmc_summary.py $O/pmcA > $O/r06_pmc_mfma_lds.txt 2>&1; grep -E "embed|qkv_att|layer_tail" $O/r06_pmc_mfma_lds.txt | cut -c1-300
find $O/pmc? -name "*kernel_trace.csv" -delete
unset MST_STREAMS
step "finetune kernel stats"
FB_ITERS=3 FB_NATIVE_ONLY=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ft -- python3 tools/finetune_bench.py > $O/prof_ft.log 2>&1 || exit 1
cp $(find $O/prof_ft -name "*kernel_stats.csv" | head -1) $O/r06_finetune_kernel_stats_streams1.csv; find $O/prof_ft -name "*kernel_trace.csv" -delete
head -8 $O/r06_finetune_kernel_stats_streams1.csv | cut -c1-200
step "phase stamps"; bash tools/phase_stamps.sh > /dev/null 2>&1; cp gpurun_out/phase_stamps.txt $O/r06_phase_stamps.txt; cut -c1-300 $O/r06_phase_stamps.txt
step "host enqueue share"; timeout -k 10 300 python tools/host_bound.py 2>&1 | grep "^B=" > $O/r06_host_bound.txt; cat $O/r06_host_bound.txt
step "latency batch 1"; timeout -k 10 300 python tools/latency_b1.py 2>&1 | grep "^F=" > $O/r06_latency_batch1.txt; cat $O/r06_latency_batch1.txt
step "train stack"; timeout -k 10 300 python tools/train_bench.py > $O/train.log 2>&1; tail -1 $O/train.log > $O/r06_train_stack_bench.json; cut -c1-300 $O/r06_train_stack_bench.json
step "train stack, frozen (input gradient only: the motion encoder's backward)"; TB_FROZEN=1 timeout -k 10 300 python tools/train_bench.py > $O/train_fr.log 2>&1; tail -1 $O/train_fr.log > $O/r06_train_stack_bench_frozen.json; cut -c1-300 $O/r06_train_stack_bench_frozen.json
rm -rf $O/prof $O/prof_ft $O/pmcF $O/pmcW $O/pmcA $O/trace3
ls $O
