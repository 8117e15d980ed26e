for n in 0 1 0 1; do
  MST_FUSE_QKV_ATTN=$n timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/fuse_$n.log 2>&1
  tail -1 gpurun_out/fuse_$n.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse=$n', d['value'], d['roofline']['whole_path_tflops'], {k:v for k,v in d['roofline']['kernel_avg_us'].items() if v>0})"
done
