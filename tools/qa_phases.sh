# phase ablations of the fused QKV+attention kernel: build libqa<N>.so with -DQA_STOP=N first
for v in "$@"; do
  MST_ENGINE_LIB=$PWD/$v timeout -k 10 200 python bench.py --steps 1 --warmup 0 --denoise-steps 64 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['roofline']['kernel_avg_us']['qkv_attention_fused'])"
done
