# Round 6 timing probe (a library built with MST_PROBE_SKIP support, not the product): what the fine-tune iteration would gain if the split-K
# reduce launches (bit 0) and the bias-gradient column sums (bit 1) of the weight-gradient stream cost nothing -- the upper bound of folding them
# into k_wgrad_tr
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export MST_ENGINE_LIB=$PWD/diffusion-based-motion-style-transfer_amd/csrc/ab_libs/lib_probe_skip.so
for r in 1 2 3; do for v in 0 1 2 3; do
  MST_PROBE_SKIP=$v timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/ftab.log 2>&1 || { tail -5 gpurun_out/ftab.log; exit 1; }
  tail -1 gpurun_out/ftab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MST_PROBE_SKIP=$v (1: no split-K reduce, 2: no column sums)', d['ms_per_step'], 'ms/iteration')"
done; done 2>&1 | tee gpurun_out/r6_wgrad_probe.txt
