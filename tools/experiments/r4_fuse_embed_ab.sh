# Round 4: the output projection of step j + the pose embedding of step j + 1 in one launch, against two launches; interleaved
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary $2 > gpurun_out/ab.log 2>&1; tail -1 gpurun_out/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], {k:v['avg_launch_us'] for k,v in d['roofline']['families'].items() if v['avg_launch_us']>5})"; }
for r in 1 2 3; do
  MST_FUSE_EMBED=0 run "two launches "
  run "one launch   "
done
MST_FUSE_EMBED=0 run "two launches, batch 32" "--batch 32"
run "one launch, batch 32  " "--batch 32"
