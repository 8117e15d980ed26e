# Round 5: k_qkv_attention2's output stores -- 8-byte stores from the accumulator layout (default) against 16-byte whole-line stores through
# the wave's Q rows, plain (1) and write-through (2): same-box A/B of three library builds, interleaved rounds; then the finetune bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=diffusion-based-motion-style-transfer_amd/csrc/ab_libs
bash tools/lib_ab.sh default $L/lib_qa_out1.so $L/lib_qa_out2.so 2>&1 | tee gpurun_out/r5_qa_out_ab.txt
timeout -k 10 300 python bench.py --mode finetune --steps 10 --warmup 3 > gpurun_out/r5_ft.log 2>&1; tail -1 gpurun_out/r5_ft.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('finetune', d['value'], d['ms_per_step'], json.dumps(d['roofline'].get('dominant_kernel')))"
