# k_ln_bwd with batched row loads: training tests (bit-reproducibility, gradients vs autograd / reference goldens), per-kernel averages, iteration time
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fullsize.py tests/test_gpu_boundary.py -x -q > gpurun_out/lnbwd_tests.txt 2>&1 || { tail -30 gpurun_out/lnbwd_tests.txt; exit 1; }
tail -3 gpurun_out/lnbwd_tests.txt
bash tools/finetune_profile.sh
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_ft/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f'{float(r["AverageNs"])/1e3:8.1f} us x {r["Calls"]:>5}  {r["Percentage"]:>6}%  {r["Name"][:90]}')
PY
bash tools/ft_lib_ab.sh default
