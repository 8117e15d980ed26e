for b in 16 32 64 128; do
  timeout -k 10 300 python bench.py --steps 1 --warmup 1 --denoise-steps 100 --batch $b --no-cpu-baseline > gpurun_out/sweep_$b.log 2>&1
  tail -1 gpurun_out/sweep_$b.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=$b', 'ms/denoise-step', round(d['ms_per_step']/100,3), d['roofline']['kernel_avg_us'])"
done
