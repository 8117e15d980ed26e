# Round 4: up to how many stream rows do 16-token tiles pay for the plain small-launch GEMMs?  200-step loops, clips of 196 frames
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --denoise-steps 200 --batch $1 --no-cpu-baseline --no-boundary > gpurun_out/ab.log 2>&1; tail -1 gpurun_out/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 $2', round(1e6*$1/d['value']/200,1), 'us/step')"; }
for b in 1 2 3 4 6 8; do
  MST_SMALL_NTB1_M=0 run $b "64-token tiles            "
  MST_SMALL_NTB1_M=100000 run $b "16-token tiles            "
  MST_SMALL_NTB1_M=100000 MST_SMALL_LN_M=100000 run $b "16-token tiles, LN inside "
done
