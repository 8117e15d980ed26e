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
python3 - <<'PY'
# the same numbers as JSON with the library's source hash (bench.py --mode finetune quotes roofline.traffic from it for that build only)
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.getcwd())
import mst_amd  # noqa: F401
from mst_amd import _native
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
fam = collections.defaultdict(lambda: {"fetch": [], "write": []})
for k, cs in acc.items():
    if "mst" not in k or not cs["FETCH_SIZE"] or not cs["WRITE_SIZE"]:
        continue
    name = k.split("(")[0].replace("mst::", "")
    for key in ("k_wgrad_tr", "k_attention_bwd", "k_attention_train", "k_layer_tail_train", "k_layer_tail_bwd", "k_ln_bwd", "k_splitk_reduce", "k_colsum_f16",
                "DEpiLnBwd", "OpGeluBwd", "DEpiF32", "DEpiBiasF16"):
        if key in k:
            name = key
            break
    fam[name]["fetch"] += cs["FETCH_SIZE"]
    fam[name]["write"] += cs["WRITE_SIZE"]
out = {"source_hash": _native.built_hash(), "clips": 64,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/train_bench.py (one stack forward + backward at 64 clips, "
                 "MST_WGRAD_STREAM=0: one stream); KiB -> bytes; FETCH_SIZE x2 (gfx950 correction); per launch", "kernels": {}}
for name, v in fam.items():
    fe = 2 * 1024 * sum(v["fetch"]) / len(v["fetch"]); wr = 1024 * sum(v["write"]) / len(v["write"])
    out["kernels"][name] = {"fetch_bytes": round(fe), "write_bytes": round(wr), "hbm_bytes": round(fe + wr), "launches": len(v["fetch"])}
json.dump(out, open("gpurun_out/r06_train_pmc_traffic.json", "w"), indent=1)
print("k_wgrad_tr", out["kernels"].get("k_wgrad_tr"))
PY
find gpurun_out/pmc_* -name "*kernel_trace.csv" -delete; find gpurun_out/pmc_* -name "*counter_collection.csv" -delete
