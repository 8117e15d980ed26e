# GPU clock / power while the sampling loop runs (read-only queries)
python bench.py --steps 2 --warmup 1 --denoise-steps 1000 --no-cpu-baseline > gpurun_out/clock_bench.log 2>&1 &
BP=$!
sleep 25
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | head -6; echo --; sleep 1; done
wait $BP
tail -c 300 gpurun_out/clock_bench.log
