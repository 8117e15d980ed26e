# In-kernel phase stamps of the two dominant kernels at the headline shape (csrc/probes/attn_clock.hip, tail_clock.hip: diagnostic builds
# of the product headers with MST_PROBE_BUILD; the library itself carries no stamp).  Build the probes first (see their headers).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin
( for b in attn_clock tail_clock; do echo "== $b"; timeout -k 10 100 $P/$b | tail -3 || exit 1; done ) > gpurun_out/phase_stamps.txt 2>&1
cat gpurun_out/phase_stamps.txt
