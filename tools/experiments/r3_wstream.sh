# Round 3 probe: per-CU weight stream L2 -> VGPR (csrc/probes/wstream.hip) at 197 and 256 workgroups, next to round 2's LDS-DMA tail ablations.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin
( timeout -k 10 120 $P/wstream 197 && timeout -k 10 120 $P/wstream 256 ) > gpurun_out/r3_wstream.txt 2>&1
cat gpurun_out/r3_wstream.txt
