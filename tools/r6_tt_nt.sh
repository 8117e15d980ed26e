# Round 6: non-temporal tape stores of the fused training tail against plain stores (library A/B on the fine-tune iteration), then the
# chained steps' side-stream timing with the product build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ft_lib_ab.sh default diffusion-based-motion-style-transfer_amd/csrc/ab_libs/lib_tt_plain.so 2>&1 | tee gpurun_out/r6_tt_nt_ab.txt
timeout -k 10 300 python tools/ft_events.py 2>&1 | grep -v "it/s" | tail -9
