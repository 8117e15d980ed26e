# Round 6: where the caller's stream and the side stream spend the fine-tune iteration, round 5's protocol against overlap_backward
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1; do
  echo "== overlap_backward=$v"
  FT_QUICK=1 FT_OVERLAP=$v timeout -k 10 300 python tools/ft_events.py 2>&1 | grep -v Warning
done > gpurun_out/r6_overlap_events.txt 2>&1
cat gpurun_out/r6_overlap_events.txt
