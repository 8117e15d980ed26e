# Round 6: iterate on the fused training tail: training tests, phase stamps of the TRAIN instantiation, stack forward / backward, fine-tune line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fullsize.py tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/r6_gpu_tests5.log 2>&1; rc=$?
tail -3 gpurun_out/r6_gpu_tests5.log
[ $rc = 0 ] || { tail -40 gpurun_out/r6_gpu_tests5.log; exit $rc; }
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin
timeout -k 10 100 $P/tail_train_clock | tail -3
TB_NATIVE_ONLY=1 timeout -k 10 200 python tools/train_bench.py 2>&1 | tail -1
for r in 1 2; do
timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r6_ft.log 2>&1 || { tail -5 gpurun_out/r6_ft.log; exit 1; }
tail -1 gpurun_out/r6_ft.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/iteration', 'loss', d.get('final_loss'))"
done
