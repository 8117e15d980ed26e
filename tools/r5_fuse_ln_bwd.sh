# LayerNorm1's backward in the dgrad GEMM's epilogue (DEpiLnBwd): training tests, then MST_FUSE_LN_BWD=0 / 1 on the fine-tune iteration, alternating
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fullsize.py tests/test_gpu_boundary.py tests/test_gpu_training_loop.py -x -q > gpurun_out/fln_tests.txt 2>&1 || { tail -30 gpurun_out/fln_tests.txt; exit 1; }
tail -2 gpurun_out/fln_tests.txt
for r in 1 2 3; do for v in 0 1; do
  MST_FUSE_LN_BWD=$v timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/fln.log 2>&1 || { tail -5 gpurun_out/fln.log; exit 1; }
  tail -1 gpurun_out/fln.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MST_FUSE_LN_BWD=$v', d['ms_per_step'], 'ms/iteration')"
done; done
