# Board power and shader clock while the default bench loop runs (rocm-smi sampled twice a second beside it).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocm-smi --showpower --showclocks --showmaxpower > gpurun_out/power_idle.txt 2>&1
timeout -k 10 300 python bench.py --steps 30 --warmup 1 --no-cpu-baseline --no-boundary > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BPID=$!
sleep 12
for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | tr '\n' ' '; echo; sleep 0.5; done > gpurun_out/power_samples.txt
wait $BPID
echo "bench rc=$?"; tail -c 300 gpurun_out/power_bench.json | head -c 300; echo
cat gpurun_out/power_idle.txt | grep -i "power\|sclk\|max" | head; echo ---; head -12 gpurun_out/power_samples.txt
