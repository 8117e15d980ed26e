# Round 3: the fine-tune / training files of profiles/r03_* (re-collected after the weight re-upload change; same recipe as tools/experiments/r3_profiles.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; rm -rf $O; mkdir -p $O
timeout -k 10 300 python bench.py --mode finetune --steps 10 --warmup 3 > $O/bench_ft.log 2>&1 || exit 1; tail -1 $O/bench_ft.log > $O/r03_finetune_bench_1gpu.json; cut -c1-300 $O/r03_finetune_bench_1gpu.json
FB_ITERS=3 FB_NATIVE_ONLY=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ft -- python3 tools/finetune_bench.py > $O/prof_ft.log 2>&1 || exit 1
cp $(find $O/prof_ft -name "*kernel_stats.csv" | head -1) $O/r03_finetune_kernel_stats_streams1.csv; rm -rf $O/prof_ft
head -8 $O/r03_finetune_kernel_stats_streams1.csv | cut -c1-160
timeout -k 10 300 python tools/train_bench.py > $O/train.log 2>&1; tail -1 $O/train.log > $O/r03_train_stack_bench.json; cut -c1-300 $O/r03_train_stack_bench.json
