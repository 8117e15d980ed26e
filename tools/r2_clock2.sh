cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin
( for b in ${PROBES:-attn_clock attn_clock_NODMA attn_clock_NOMMA tail_clock tail_clock_NODMA tail_clock_NOMMA}; do echo "== $b"; timeout -k 10 100 $P/$b || exit 1; done ) > gpurun_out/r2_phase_clock.log 2>&1
cat gpurun_out/r2_phase_clock.log
