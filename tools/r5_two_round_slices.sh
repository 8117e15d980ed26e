# Round 5: the two-round configurations (CFG at 64 clips = 128 rows, batch 128 = 394 tail tiles on 256 CUs) at 1 / 2 / 3 clip slices and with
# 48-token tail tiles, same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary $2 > gpurun_out/ts.log 2>&1 || { tail -5 gpurun_out/ts.log; exit 1; }; tail -1 gpurun_out/ts.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for cfgname in "--cfg" "--batch 128"; do
  for n in 2 3 1; do MST_STREAMS=$n run "$cfgname slices=$n" "$cfgname"; done
  MST_STREAMS=2 MST_TAIL_NTB=3 run "$cfgname slices=2 48-token tiles" "$cfgname"
  MST_STREAMS=3 MST_TAIL_NTB=3 run "$cfgname slices=3 48-token tiles" "$cfgname"
done 2>&1 | tee gpurun_out/r5_two_round_slices.txt
