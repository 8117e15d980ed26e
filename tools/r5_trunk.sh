# Round 5: the resident-group trunk -- bitwise tests against the two-kernel path, then a same-box A/B of the headline bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_trunk.py -x -q -s -m gpu > gpurun_out/r5_trunk_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"
grep -E "passed|failed|error|Error|assert" gpurun_out/r5_trunk_tests.log | tail -15
[ $rc -eq 0 ] || exit 1
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > gpurun_out/tr.log 2>&1 || { tail -5 gpurun_out/tr.log; exit 1; }; tail -1 gpurun_out/tr.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  MST_TRUNK=0 run "trunk=0"
  MST_TRUNK=1 run "trunk=1"
done
