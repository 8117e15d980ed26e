# Round 5: today's fine-tune baseline on one box: bench line, host profile, synced segments
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_trunk.py -q -s -m gpu -k "forward" 2>&1 | grep -E "clips: resident|passed|failed"
timeout -k 10 300 python bench.py --mode finetune --steps 10 --warmup 3 > gpurun_out/r5_ft_base.log 2>&1; tail -1 gpurun_out/r5_ft_base.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('finetune', d['value'], d['ms_per_step'])"
timeout -k 10 300 python tools/finetune_hostprof.py > gpurun_out/r5_ft_hostprof.txt 2>&1; head -60 gpurun_out/r5_ft_hostprof.txt
timeout -k 10 300 python tools/finetune_segments.py > gpurun_out/r5_ft_segments.txt 2>&1; tail -12 gpurun_out/r5_ft_segments.txt
