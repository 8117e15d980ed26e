cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for sm in 2048 8192; do
  for b in 11 12 14 16 20 24; do
  MST_STREAMS=1 MST_SMALL_M=$sm timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-boundary --batch $b > gpurun_out/s6.log 2>&1
  tail -1 gpurun_out/s6.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one launch, small_m=$sm batch=$b', d['value'], round($b/d['value']*1000,1),'us/step')"
done; done
