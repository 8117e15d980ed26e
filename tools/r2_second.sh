# round-2 second GPU call: full GPU suite, new bench line (boundary legs), fine-tune mode single GPU + 2-rank gloo rehearsal on one card
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests2.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r2_tests2.log
timeout -k 10 400 python bench.py --steps 2 --warmup 1 > gpurun_out/r2_bench1.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/r2_bench1.log | cut -c1-400
timeout -k 10 300 python bench.py --mode finetune --steps 5 --warmup 2 > gpurun_out/r2_ft1.log 2>&1
echo "ft rc=$?"; tail -1 gpurun_out/r2_ft1.log | cut -c1-700
timeout -k 10 300 python bench.py --mode finetune --gpus 2 --backend gloo --shared-device --batch 16 --steps 3 --warmup 1 > gpurun_out/r2_ft2.log 2>&1
echo "ft2 rc=$?"; tail -1 gpurun_out/r2_ft2.log | cut -c1-900
