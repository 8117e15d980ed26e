# Round 4: mid-size batches as slices of <= 9 clips through the (now faster) small-launch path against one large-tile slice.  200-step loops.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --denoise-steps 200 --batch $1 --no-cpu-baseline --no-boundary > gpurun_out/ab.log 2>&1; tail -1 gpurun_out/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 $2', d['value'], 'clips/s', round(1e6*$1/d['value']/200,1), 'us/step')"; }
for b in 12 16 18; do
  run $b "default          "
  MST_STREAMS=2 run $b "two slices       "
done
for b in 24 27; do
  run $b "default          "
  MST_STREAMS=3 run $b "three slices     "
done
