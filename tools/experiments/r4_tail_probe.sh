# Round 4: back-to-back probe of tail-kernel variants (csrc/probes/bin/tail_v*: tail_clock.hip built with -DTAIL_X_* knobs).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin
: > gpurun_out/r4_tail_probe.txt
for round in 1 2; do
for v in "$@"; do
  echo "== $v (round $round)" >> gpurun_out/r4_tail_probe.txt
  timeout -k 10 120 $P/tail_$v >> gpurun_out/r4_tail_probe.txt 2>&1 || { echo "FAILED $v" >> gpurun_out/r4_tail_probe.txt; exit 1; }
done; done
grep -E "^==|rep 2|FFN per" gpurun_out/r4_tail_probe.txt | cut -c1-330
