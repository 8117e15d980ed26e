# A/B: priority of the chained calls' side stream in the fine-tune iteration (MST_CHAIN_PRIORITY = 0 default / -1 high), one box, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in 0 -1; do
  MST_CHAIN_PRIORITY=$v timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/ftprio.log 2>&1 || { tail -5 gpurun_out/ftprio.log; exit 1; }
  tail -1 gpurun_out/ftprio.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MST_CHAIN_PRIORITY=$v', d['ms_per_step'], 'ms/iteration')"
done; done
