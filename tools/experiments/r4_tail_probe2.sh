# Round 4: tail tile height (NTB = 16-token blocks per tile) and timing-only ablations, back to back in the probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin
O=gpurun_out/r4_tail_probe2.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 120 $P/tail_$1 64 $2 >> $O 2>&1 || { echo FAILED >> $O; exit 1; }; }
for round in 1 2; do
run n4 12608; run n3 12288; run n3 12608; run n4 4334; run n3 4334; run n2 4334; run n2 8192
run a_nogelu 12608; run a_nostream 12608; run a_both 12608; run n3_nogelu 12288
done
grep -E "^==|rep 2|FFN per" $O | cut -c1-330
