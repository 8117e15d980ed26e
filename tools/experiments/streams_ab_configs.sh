cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do for n in 1 3; do
  for extra in "--cfg" "--batch 128" "--batch 16" "--batch 32"; do
  MST_STREAMS=$n timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-boundary $extra > gpurun_out/s2.log 2>&1
  tail -1 gpurun_out/s2.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$n $extra', d['value'])"
done; done; done
