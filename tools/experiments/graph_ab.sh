cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2 3; do for g in 0 1; do
  MST_GRAPH=$g timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > gpurun_out/g.log 2>&1
  tail -1 gpurun_out/g.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph=$g', d['value'])"
done; done
