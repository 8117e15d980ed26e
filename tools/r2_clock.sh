# In-kernel clock + phase times of the fused QKV+attention kernel, then the full GPU suite and the default bench on HEAD.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 diffusion-based-motion-style-transfer_amd/csrc/probes/bin/attn_clock > gpurun_out/r2_attn_clock.log 2>&1 &&
cat gpurun_out/r2_attn_clock.log &&
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests_head.log 2>&1 &&
tail -2 gpurun_out/r2_tests_head.log &&
timeout -k 10 400 python bench.py > gpurun_out/r2_bench_head.json 2> gpurun_out/r2_bench_head.err &&
cat gpurun_out/r2_bench_head.json
