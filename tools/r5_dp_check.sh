# Round 5: two ranks of bench.py --mode finetune sharing one card over gloo, with the chain's backward pass in line / on the side stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  MST_CHAIN_BWD_SIDE=$v timeout -k 10 600 python bench.py --mode finetune --gpus 2 --backend gloo --shared-device --steps 3 --warmup 2 > gpurun_out/r5_ft_dp2_$v.log 2>&1; echo "side=$v rc=$?"
  tail -1 gpurun_out/r5_ft_dp2_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['allreduce'])"
done
