# Round 3: quick look at an attention-kernel edit: parity (forward tests), phase stamps, interleaved A/B of two libraries ("default" = in-tree).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward or cfg" > gpurun_out/r3_parity.log 2>&1; rc=$?
tail -3 gpurun_out/r3_parity.log
[ $rc -eq 0 ] || exit $rc
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin
timeout -k 10 120 $P/attn_clock > gpurun_out/r3_attn_clock.txt 2>&1 || exit 1
tail -1 gpurun_out/r3_attn_clock.txt | cut -c1-400
bash tools/lib_ab.sh ${1:-lib_a.so} default
