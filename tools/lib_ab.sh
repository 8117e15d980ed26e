# A/B two full library builds in one box: tools/lib_ab.sh <pathA> <pathB>
for r in 1 2; do for v in "$@"; do
  MST_ENGINE_LIB=$PWD/$v timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/libab.log 2>&1
  tail -1 gpurun_out/libab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], {k:v for k,v in d['roofline']['kernel_avg_us'].items() if v>0})"
done; done
