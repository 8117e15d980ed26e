cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# plain multi-rank invocation (bench.py spawns its ranks itself): rehearsal with two gloo ranks sharing the one card
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --shared-device --batch 16 --denoise-steps 100 --steps 1 --warmup 1 --no-boundary > gpurun_out/r2_spawn2.log 2>&1
echo "spawn rc=$?"; tail -1 gpurun_out/r2_spawn2.log | cut -c1-330
timeout -k 10 300 python bench.py --cfg --steps 2 --warmup 1 --no-boundary --no-cpu-baseline > gpurun_out/r2_bench_cfg.log 2>&1
echo "cfg rc=$?"; tail -1 gpurun_out/r2_bench_cfg.log | cut -c1-200
timeout -k 10 300 python bench.py --batch 128 --steps 2 --warmup 1 --no-boundary --no-cpu-baseline > gpurun_out/r2_bench_b128.log 2>&1
echo "b128 rc=$?"; tail -1 gpurun_out/r2_bench_b128.log | cut -c1-200
timeout -k 10 200 python tools/train_bench.py > gpurun_out/r2_train_bench.log 2>&1; tail -2 gpurun_out/r2_train_bench.log | cut -c1-600
timeout -k 10 300 python tools/finetune_bench.py > gpurun_out/r2_finetune_bench.log 2>&1; tail -1 gpurun_out/r2_finetune_bench.log | cut -c1-600
