# Small-tile / large-tile hand-over: clips/s of one loop launch at batches around the crossover with the small path off (0),
# forced (8192) and the default policy.  gpurun -- 'bash tools/experiments/small_m_ab2.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 8 11 14 16 20 24 30; do
  for sm in 0 8192 default; do
  if [ $sm = default ]; then unset MST_SMALL_M; else export MST_SMALL_M=$sm; fi
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary --batch $b > gpurun_out/s6.log 2>&1 || exit 1
  tail -1 gpurun_out/s6.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch=$b small_m=$sm', d['value'], 'clips/s', round($b/d['value']*1000,1),'us/step', d['config'].get('slices'))"
done; done
