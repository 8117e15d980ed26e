# Round 4: 32-token tiles for the plain small-launch GEMMs between 800 and 1900 stream rows?  200-step loops, clips of 196 frames
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --denoise-steps 200 --batch $1 --no-cpu-baseline --no-boundary > gpurun_out/ab.log 2>&1; tail -1 gpurun_out/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 $2', round(1e6*$1/d['value']/200,1), 'us/step')"; }
for b in 4 5 6 8 9; do
  run $b "default (16-token tiles up to 800 rows, 64 above)"
  MST_SMALL_NTB2_M=100000 run $b "32-token tiles above 800 rows                    "
done
