cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do for cfg in "4 3" "8 3" "8 4"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$1 MST_STREAMS=$2 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > gpurun_out/q.log 2>&1
  tail -1 gpurun_out/q.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues=$1 streams=$2', d['value'])"
done; done
