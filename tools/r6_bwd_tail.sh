# Round 6: the fused backward tail (k_layer_tail_bwd, MST_TRAIN_FUSE_BWD_TAIL=1): training tests with it on, stack forward / backward, fine-tune line 0 / 1 alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MST_TRAIN_FUSE_BWD_TAIL=2 timeout -k 10 900 python -m pytest tests/test_gpu_train_fullsize.py tests/test_gpu_train.py tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/r6_gpu_tests6.log 2>&1; rc=$?
tail -3 gpurun_out/r6_gpu_tests6.log
[ $rc = 0 ] || { grep -E "^E |Error|assert" gpurun_out/r6_gpu_tests6.log | head -30; exit $rc; }
for v in 0 2; do MST_TRAIN_FUSE_BWD_TAIL=$v TB_NATIVE_ONLY=1 timeout -k 10 200 python tools/train_bench.py 2>&1 | tail -1 | cut -c1-200; done
for r in 1 2 3; do for v in 0 2; do
  MST_TRAIN_FUSE_BWD_TAIL=$v timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r6_ft.log 2>&1 || { tail -5 gpurun_out/r6_ft.log; exit 1; }
  tail -1 gpurun_out/r6_ft.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MST_TRAIN_FUSE_BWD_TAIL=$v', d['ms_per_step'], 'ms/iteration', 'loss', d.get('final_loss'))"
done; done 2>&1 | tee gpurun_out/r6_bwd_tail_ab.txt
