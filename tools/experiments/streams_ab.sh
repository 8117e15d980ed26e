# same-box A/B of the number of concurrent clip slices (MST_STREAMS), interleaved rounds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2 3; do for n in 1 2 3 4; do
  MST_STREAMS=$n timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary > gpurun_out/streams_$n.log 2>&1
  tail -1 gpurun_out/streams_$n.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$n', d['value'], d['roofline']['whole_path_tflops'])"
done; done
