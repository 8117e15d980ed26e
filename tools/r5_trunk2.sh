# Round 5: the whole GPU suite on the refactored kernels, then trunk on / off at the two-round configs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5_gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r5_gpu_tests.log
[ $rc -eq 0 ] || exit 1
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary $2 > gpurun_out/tr.log 2>&1 || { tail -5 gpurun_out/tr.log; exit 1; }; tail -1 gpurun_out/tr.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for t in 0 1; do
  MST_TRUNK=$t run "cfg trunk=$t" "--cfg"
  MST_TRUNK=$t run "batch128 trunk=$t" "--batch 128"
done
