# Round 5: why does a 4th clip slice serialise?  (1) the runtime alone: csrc/probes/queues.hip -- K chains of idle kernels on K streams;
# (2) the engine: 3 / 4 / 6 slices under the same settings.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin/queues
OUT=gpurun_out/r5_queues.txt
: > $OUT
for q in unset 8 16; do
  if [ $q = unset ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for prio in 0 1; do
    echo "== probe queues=$q prio=$prio" >> $OUT
    timeout -k 10 120 $P 30 40 400 $prio >> $OUT 2>&1 || exit 1
  done
done
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > gpurun_out/sl.log 2>&1 || { tail -5 gpurun_out/sl.log; exit 1; }; tail -1 gpurun_out/sl.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" >> $OUT; }
unset GPU_MAX_HW_QUEUES
for n in 3 4; do export MST_STREAMS=$n; run "engine queues=default slices=$n"; done
export GPU_MAX_HW_QUEUES=8
for n in 3 4 6; do export MST_STREAMS=$n; run "engine queues=8 slices=$n"; done
cat $OUT
