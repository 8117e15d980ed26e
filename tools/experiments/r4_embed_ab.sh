# Round 4: the latency embed kernels (mst_embed.h) against the ring GEMMs, interleaved, 64 clips x 1000 steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary $2 > gpurun_out/ab.log 2>&1; tail -1 gpurun_out/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], {k:v['avg_launch_us'] for k,v in d['roofline']['families'].items() if v['avg_launch_us']>5})"; }
for r in 1 2 3; do
  MST_EMBED_FAST=0 run "ring embed "
  run "fast embed "
done
MST_EMBED_FAST=0 run "ring embed cfg" --cfg
run "fast embed cfg" --cfg
