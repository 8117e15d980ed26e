# A/B variant libraries in one box: tools/ab.sh <lib-suffix> ...   (libs built as libmst_engine_<suffix>.so)
P=diffusion-based-motion-style-transfer_amd
for v in "$@"; do
  MST_ENGINE_LIB=$PWD/$P/libmst_engine_$v.so timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1
  tail -1 gpurun_out/ab_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['kernel_avg_us'])"
done
