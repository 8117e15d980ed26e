# Round 6: the training forward on the fused layer tail (k_layer_tail_train).  Training tests, then the fine-tune iteration with
# MST_TRAIN_FUSE_TAIL=0 / 1 alternating (three rounds of 50 iterations), then the headline A/B of the await poll (ds_read vs flat).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_train_fullsize.py tests/test_gpu_train.py -x -q -m gpu > gpurun_out/r6_train_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r6_train_tests.log
[ $rc = 0 ] || exit $rc
for r in 1 2 3; do for v in 0 1; do
  MST_TRAIN_FUSE_TAIL=$v timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r6_ft.log 2>&1 || { tail -5 gpurun_out/r6_ft.log; exit 1; }
  tail -1 gpurun_out/r6_ft.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MST_TRAIN_FUSE_TAIL=$v', d['ms_per_step'], 'ms/iteration', 'loss', d.get('final_loss'))"
done; done 2>&1 | tee gpurun_out/r6_train_tail_ab.txt
bash tools/lib_ab.sh default diffusion-based-motion-style-transfer_amd/csrc/ab_libs/lib_await_flat.so 2>&1 | tee gpurun_out/r6_await_ab.txt
