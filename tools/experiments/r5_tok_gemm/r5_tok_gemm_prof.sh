set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tokprof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tokprof -o tok -- python3 $R/tools/experiments/r5_tok_gemm/r5_tok_gemm_check.py 64 > $R/gpurun_out/tokprof/run.txt 2>&1 || { tail -20 $R/gpurun_out/tokprof/run.txt; exit 1; }
grep -E "bit-identical|forward" $R/gpurun_out/tokprof/run.txt
F=$(find $R/gpurun_out/tokprof -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: r["Name"]):
    n = r["Name"]
    if "tok_gemm" in n or "gemm_dma" in n:
        print(f'{float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]:>5}  {n[10:150]}')
PY
