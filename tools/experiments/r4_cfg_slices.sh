# Round 4: slices for the two-round configurations (CFG at 64 clips, batch 128) with this round's kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary $2 > gpurun_out/ab.log 2>&1; tail -1 gpurun_out/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'])"; }
for r in 1 2; do
for n in 1 2 3; do MST_STREAMS=$n run "cfg slices=$n" --cfg; done
for n in 1 2 3; do MST_STREAMS=$n run "batch128 slices=$n" "--batch 128"; done
for n in 2 3; do MST_STREAMS=$n MST_TAIL_NTB=3 run "cfg slices=$n ntb=3" --cfg; done
done
