# rocprofv3 --stats of three fine-tune iterations (tools/finetune_bench.py): the top kernels' average launch times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ft
bash tools/finetune_profile.sh > /dev/null
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_ft/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f'{float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]:>5}  {r["Percentage"]:>6}%  {r["Name"][:90]}')
PY
