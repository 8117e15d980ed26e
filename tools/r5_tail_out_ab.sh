# Round 5: k_layer_tail's LayerNorm2 stores: 8 bytes per lane plain (default) | 16 bytes per lane plain (2) | 16 bytes write-through (3); same-box A/B, interleaved rounds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=diffusion-based-motion-style-transfer_amd/csrc/ab_libs
bash tools/lib_ab.sh default $L/lib_tail_out2.so $L/lib_tail_out3.so 2>&1 | tee -a gpurun_out/r5_tail_out_ab.txt
