# Round 4: tile height of the fused layer tail for lone (single-slice) launches: clips/s over the batch size, 200-step loops
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --denoise-steps 200 --batch $1 --no-cpu-baseline --no-boundary > gpurun_out/ab.log 2>&1; tail -1 gpurun_out/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 $2', d['value'], round(1e3*$1/d['value']/200*1e3/1e3,1), 'us/step')"; }
for r in 1 2; do
for b in 4 8 12 16 24 32 48; do
  MST_SMALL_M=0 MST_TAIL_NTB=4 run $b "large tiles, 64-token tail"
  MST_SMALL_M=0 run $b "large tiles, auto tail      "
  run $b "default                     "
done; done
