# Round 6: HBM traffic of the training kernels at 64 clips (FETCH_SIZE / WRITE_SIZE in separate --pmc passes over tools/train_bench.py, one
# stack forward + backward; KiB -> bytes, FETCH_SIZE x 2 per the gfx950 note), per launch, against what each kernel has to move
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  MST_WGRAD_STREAM=0 TB_NATIVE_ONLY=1 TB_ITERS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 tools/train_bench.py > gpurun_out/r6_train_pmc.log 2>&1 || { tail -5 gpurun_out/r6_train_pmc.log; exit 1; }
done
python3 - <<'PY' | tee gpurun_out/r6_train_pmc.txt
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, cs in acc.items():
    if "mst" not in k or not cs["FETCH_SIZE"] or not cs["WRITE_SIZE"]:
        continue
    fe = 2 * 1024 * sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]); wr = 1024 * sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
    rows.append((fe + wr, fe, wr, len(cs["FETCH_SIZE"]), k))
rows.sort(reverse=True)
print("per launch: MB fetched (x2-corrected) + MB written = MB; launches; kernel")
for t, fe, wr, n, k in rows[:22]:
    print(f"  {fe / 1e6:8.1f} + {wr / 1e6:8.1f} = {t / 1e6:8.1f} MB  x {n:4d}  {k[:110]}")
PY
find gpurun_out/pmc_* -name "*kernel_trace.csv" -delete; find gpurun_out/pmc_* -name "*counter_collection.csv" -delete
