# Round 6: both LayerNorms' second-stage sums of a layer in one launch (MST_LN_FINISH_MERGE): training-side tests, fine-tune line off / on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fullsize.py tests/test_gpu_boundary.py tests/test_gpu_training_loop.py -x -q -m gpu > gpurun_out/r6_lnfin_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r6_lnfin_tests.log
[ $rc = 0 ] || { grep -E "^E |Error|assert" gpurun_out/r6_lnfin_tests.log | head -30; exit $rc; }
for r in 1 2 3; do for v in 0 1; do
  MST_LN_FINISH_MERGE=$v timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/ftab.log 2>&1 || { tail -5 gpurun_out/ftab.log; exit 1; }
  tail -1 gpurun_out/ftab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MST_LN_FINISH_MERGE=$v', d['ms_per_step'], 'ms/iteration')"
done; done 2>&1 | tee gpurun_out/r6_ln_finish_merge_ab.txt
