# Round 4: tail tile height in the real loop (64 clips, 1000 steps), interleaved
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > gpurun_out/ab.log 2>&1; tail -1 gpurun_out/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], {k:v['avg_launch_us'] for k,v in d['roofline']['families'].items() if v['avg_launch_us']>10})"; }
timeout -k 10 600 python -m pytest tests/test_gpu_bench_path.py tests/test_gpu_parity.py -x -q -m gpu -k "bench_path or forward or tail or full_size" > gpurun_out/r4_t3.log 2>&1; tail -3 gpurun_out/r4_t3.log
for r in 1 2 3; do
  MST_TAIL_NTB=4 run "ntb=4 slices=3"
  run "ntb=auto(3) slices=3"
  MST_TAIL_NTB=2 run "ntb=2 slices=3"
  MST_STREAMS=2 run "ntb=auto slices=2"
  MST_STREAMS=2 MST_TAIL_NTB=3 run "ntb=3 slices=2"
done
