# Round 3 record run: every file of profiles/r03_* from one box and one build.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; rm -rf $O; mkdir -p $O
step() { echo "== $1"; }
step "bench default";  timeout -k 10 600 python bench.py > $O/bench_default.log 2>&1 || exit 1; tail -1 $O/bench_default.log > $O/r03_bench_default.json; cut -c1-400 $O/r03_bench_default.json
step "bench cfg";      timeout -k 10 300 python bench.py --cfg --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_cfg.log 2>&1 || exit 1; tail -1 $O/bench_cfg.log > $O/r03_bench_cfg.json; cut -c1-200 $O/r03_bench_cfg.json
step "bench batch 128"; timeout -k 10 300 python bench.py --batch 128 --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_b128.log 2>&1 || exit 1; tail -1 $O/bench_b128.log > $O/r03_bench_batch128.json; cut -c1-200 $O/r03_bench_batch128.json
step "bench precise mode"; MST_PRECISE=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-boundary > $O/bench_precise.log 2>&1 || exit 1; tail -1 $O/bench_precise.log > $O/r03_bench_precise_mode.json; cut -c1-200 $O/r03_bench_precise_mode.json
step "finetune bench"; timeout -k 10 300 python bench.py --mode finetune --steps 10 --warmup 3 > $O/bench_ft.log 2>&1 || exit 1; tail -1 $O/bench_ft.log > $O/r03_finetune_bench_1gpu.json; cut -c1-300 $O/r03_finetune_bench_1gpu.json
step "kernel stats, one slice"
MST_STREAMS=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-boundary > $O/prof_bench.log 2>&1 || exit 1
grep "^{\"metric\"" $O/prof_bench.log | tail -1 > $O/r03_bench_under_rocprof_streams1.json
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/r03_kernel_stats_bench_steps1_streams1.csv; find $O/prof -name "*kernel_trace.csv" -delete
head -6 $O/r03_kernel_stats_bench_steps1_streams1.csv | cut -c1-160
step "pmc traffic"
export MST_STREAMS=1
B="python3 bench.py --steps 1 --warmup 0 --denoise-steps 12 --no-cpu-baseline --no-boundary"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcF -- $B > $O/pmcF.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcW -- $B > $O/pmcW.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $O/pmcF $O/pmcW > $O/r03_pmc_traffic.json; cat $O/r03_pmc_traffic.json | head -30
step "pmc mfma / lds"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmcA -- $B > $O/pmcA.log 2>&1 || exit 1
python3 tools/pmc_summary.py $O/pmcA > $O/r03_pmc_mfma_lds.txt 2>&1; head -40 $O/r03_pmc_mfma_lds.txt
find $O/pmc? -name "*kernel_trace.csv" -delete
unset MST_STREAMS
step "finetune kernel stats"
FB_ITERS=3 FB_NATIVE_ONLY=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ft -- python3 tools/finetune_bench.py > $O/prof_ft.log 2>&1 || exit 1
cp $(find $O/prof_ft -name "*kernel_stats.csv" | head -1) $O/r03_finetune_kernel_stats_streams1.csv; find $O/prof_ft -name "*kernel_trace.csv" -delete
head -12 $O/r03_finetune_kernel_stats_streams1.csv | cut -c1-200
step "phase stamps"; bash tools/phase_stamps.sh > /dev/null 2>&1; cp gpurun_out/phase_stamps.txt $O/r03_phase_stamps.txt; cut -c1-300 $O/r03_phase_stamps.txt
step "launch boundary probe"; timeout -k 10 120 diffusion-based-motion-style-transfer_amd/csrc/probes/bin/boundary > $O/r03_launch_boundary_probe.txt 2>&1; head -10 $O/r03_launch_boundary_probe.txt
step "latency batch 1"; timeout -k 10 300 python tools/latency_b1.py > $O/r03_latency_batch1.txt 2>&1; tail -5 $O/r03_latency_batch1.txt
step "train stack"; timeout -k 10 300 python tools/train_bench.py > $O/train.log 2>&1; tail -1 $O/train.log > $O/r03_train_stack_bench.json; cut -c1-300 $O/r03_train_stack_bench.json
rm -rf $O/prof $O/prof_ft $O/pmcF $O/pmcW $O/pmcA
ls $O
