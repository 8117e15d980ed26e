# In-kernel phase stamps of the two dominant kernels with their ablations (probes/attn_clock.hip, probes/tail_clock.hip; build:
# hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DQA_NODMA|-DQA_NOMMA|-DQA_NOREAD|-DQA_READONLY|-DQA_PRODUCER=1 | -DTAIL_...] probes/X.hip -o probes/bin/X_...)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin
( for b in attn_clock attn_clock_NODMA attn_clock_NOMMA attn_clock_NOREAD attn_clock_READONLY attn_clock_NODMA_NOREAD attn_clock_NODMA_READONLY attn_clock_PRODUCER tail_clock tail_clock_NODMA tail_clock_NOMMA tail_clock_NODMA_NOREAD tail_clock_NODMA_READONLY; do echo "== $b"; timeout -k 10 100 $P/$b | tail -1 || exit 1; done ) > gpurun_out/phase_stamps.txt 2>&1
cat gpurun_out/phase_stamps.txt
