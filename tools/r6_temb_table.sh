# Round 6: the timestep-embedding table for training calls (MST_TEMB_TABLE; two dependent launches less at the head of every model call):
# training-side tests, then the fine-tune line off / on, three alternating rounds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_boundary.py tests/test_gpu_training_loop.py tests/test_gpu_train_fullsize.py -x -q -m gpu > gpurun_out/r6_temb_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r6_temb_tests.log
[ $rc = 0 ] || { grep -E "^E |Error|assert" gpurun_out/r6_temb_tests.log | head -30; exit $rc; }
for r in 1 2 3; do for v in 0 1; do
  MST_TEMB_TABLE=$v timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/ftab.log 2>&1 || { tail -5 gpurun_out/ftab.log; exit 1; }
  tail -1 gpurun_out/ftab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MST_TEMB_TABLE=$v', d['ms_per_step'], 'ms/iteration, host enqueue', d['host_enqueue_ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r6_temb_ab.txt
