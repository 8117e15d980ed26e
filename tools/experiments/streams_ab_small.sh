cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do for n in auto 1 3; do
  for b in 16 24 30 40 48; do
  if [ $n = auto ]; then unset MST_STREAMS; else export MST_STREAMS=$n; fi
  timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-boundary --batch $b > gpurun_out/s4.log 2>&1
  tail -1 gpurun_out/s4.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$n batch=$b', d['value'])"
done; done; done
