# Round 4: PMC traffic (families from the stand-alone embed kernels: MST_FUSE_EMBED=0; plus one pass with the fused step) and the three-slice trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; rm -rf $O; mkdir -p $O
export MST_STREAMS=1
B="python3 bench.py --steps 1 --warmup 0 --denoise-steps 12 --no-cpu-baseline --no-boundary"
MST_FUSE_EMBED=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcF -- $B > $O/pmcF.log 2>&1 &&
MST_FUSE_EMBED=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcW -- $B > $O/pmcW.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcF2 -- $B > $O/pmcF2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcW2 -- $B > $O/pmcW2.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $O/pmcF $O/pmcW > $O/r04_pmc_traffic.json
python3 tools/pmc_traffic.py $O/pmcF2 $O/pmcW2 > $O/r04_pmc_traffic_fused_step.json
unset MST_STREAMS
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace3 -- python3 bench.py --steps 1 --warmup 1 --denoise-steps 40 --no-cpu-baseline --no-boundary > $O/trace3.log 2>&1 || exit 1
python3 tools/trace_summary.py $O/trace3 > $O/r04_three_slice_trace_summary.txt
rm -rf $O/pmc* $O/trace3
cp $O/r04_pmc_traffic.json profiles/r04_pmc_traffic.json
python bench.py --steps 3 --warmup 1 --no-boundary 2>&1 | tail -1 > $O/r04_bench_default_with_traffic.json
python3 -c "
import json
print(json.load(open('$O/r04_pmc_traffic.json'))['kernels'])
print(json.load(open('$O/r04_pmc_traffic_fused_step.json'))['kernels'].get('embed_step_fused'))
d=json.load(open('$O/r04_bench_default_with_traffic.json')); print(d['value'], d['roofline']['traffic'], d['roofline']['traffic_source'])"
grep -E "^queue|dur  " $O/r04_three_slice_trace_summary.txt | head -12
