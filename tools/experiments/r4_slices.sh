# Round 4: more than three clip slices (needs more hardware queues than ROCm's default four): clips/s at 64 clips
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MST_SMALL_M=1500
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > gpurun_out/sl.log 2>&1; tail -1 gpurun_out/sl.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'])"; }
for r in 1 2; do
  unset GPU_MAX_HW_QUEUES
  for n in 3 4; do export MST_STREAMS=$n; run "queues=default slices=$n"; done
  export GPU_MAX_HW_QUEUES=8
  for n in 3 4 5 6; do export MST_STREAMS=$n; run "queues=8 slices=$n"; done
done
