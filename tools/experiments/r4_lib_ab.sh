# Round 4: A/B of library builds in the real loop, interleaved: tools/experiments/r4_lib_ab.sh <lib|default> ... [-- bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "$@"; do
  if [ "$v" = default ]; then unset MST_ENGINE_LIB; else export MST_ENGINE_LIB=$PWD/$v; fi
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > gpurun_out/libab.log 2>&1
  tail -1 gpurun_out/libab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], {k:v['avg_launch_us'] for k,v in d['roofline']['families'].items() if v['avg_launch_us']>5})"
done; done
