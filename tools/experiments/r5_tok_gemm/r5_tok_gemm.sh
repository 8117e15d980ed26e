# same-box A/B of the fine-tune iteration: which of the eight training GEMMs run as k_tok_gemm (MST_TOK_MASK; 0 = the slab ring everywhere)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 0x7F 0 0x7F 0 0x7F; do
  MST_TOK_MASK=$v timeout -k 10 300 python bench.py --mode finetune --steps 50 --warmup 5 --no-cpu-baseline 2> gpurun_out/ft_tok.err | tail -1 > gpurun_out/ft_tok_$v.json
  python -c "import json;d=json.load(open('gpurun_out/ft_tok_$v.json'));print('MST_TOK_MASK=$v', d['ms_per_step'], 'ms/iteration', d['value'], 'clips/s')"
done
