for n in 2 3 4 2 3 4; do
  MST_STREAMS=$n timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/streams_$n.log 2>&1
  tail -1 gpurun_out/streams_$n.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$n', d['value'], d['roofline']['whole_path_tflops'], d['roofline']['kernel_avg_us'])"
done
