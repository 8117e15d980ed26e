cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do for sm in 2048 1600; do
  for b in 8 9 10 20 30; do
  MST_SMALL_M=$sm timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-boundary --batch $b > gpurun_out/s5.log 2>&1
  tail -1 gpurun_out/s5.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('small_m=$sm batch=$b', d['value'])"
done; done; done
