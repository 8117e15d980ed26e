# Round 4: the tail with the slab-major att image (out-proj starts on plane 0) against the previous build, back to back in the probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin
O=gpurun_out/r4_tail_probe3.txt; : > $O
run() { echo "== $*" >> $O; timeout -k 10 120 $P/tail_$1 64 $2 >> $O 2>&1 || { echo FAILED >> $O; exit 1; }; }
for round in 1 2; do run base4 12608; run s4 12608; run base4 4334; run s4 4334; run base2 4334; run s2 4334; done
grep -E "^==|rep 2" $O | cut -c1-330
