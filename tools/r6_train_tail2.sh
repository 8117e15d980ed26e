# Round 6: where the fused training tail's time goes: synced segments with the switch off / on, then rocprofv3 per-kernel averages with it on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1; do echo "== MST_TRAIN_FUSE_TAIL=$v"; MST_TRAIN_FUSE_TAIL=$v timeout -k 10 300 python tools/finetune_segments.py 2>&1 | tail -8; done | tee gpurun_out/r6_ft_segments.txt
MST_TRAIN_FUSE_TAIL=1 bash tools/ft_kernel_stats.sh 2>&1 | tee gpurun_out/r6_ft_kernel_stats.txt
