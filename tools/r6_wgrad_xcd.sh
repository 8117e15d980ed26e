# Round 6: k_wgrad_tr with every split's tiles on one XCD (MST_WGRAD_XCD): workgroup targets swept with the map on, against the map off;
# the fine-tune line, three alternating rounds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2 3; do for v in "0 128" "1 64" "1 128" "1 192" "1 256"; do
  set -- $v
  MST_WGRAD_XCD=$1 MST_WGRAD_WGS=$2 timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/ftab.log 2>&1 || { tail -5 gpurun_out/ftab.log; exit 1; }
  tail -1 gpurun_out/ftab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MST_WGRAD_XCD=$1 MST_WGRAD_WGS=$2', d['ms_per_step'], 'ms/iteration')"
done; done 2>&1 | tee gpurun_out/r6_wgrad_xcd_ab.txt
