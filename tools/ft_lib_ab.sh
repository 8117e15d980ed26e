# A/B two library builds on the fine-tune iteration, one box, alternating: tools/ft_lib_ab.sh <pathA> <pathB> ...  ("default" = the in-tree library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "$@"; do
  if [ "$v" = default ]; then unset MST_ENGINE_LIB; else export MST_ENGINE_LIB=$PWD/$v; fi
  timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/ftab.log 2>&1 || { tail -5 gpurun_out/ftab.log; exit 1; }
  tail -1 gpurun_out/ftab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], 'ms/iteration')"
done; done
