"""FETCH_SIZE / WRITE_SIZE counter_collection.csv (two separate rocprofv3 --pmc passes, MST_STREAMS=1: full-batch launches)
-> per-launch HBM bytes of each kernel family, in the names bench.py's roofline uses.  Units and corrections per
/opt/skills/guides/MI355X_MICROARCH.md: both counters are in KiB; FETCH_SIZE under-counts wide coalesced reads 2x on gfx950."""
import collections, csv, glob, json, sys

# (kernel-name fragment, family): first match wins.  The fused step kernel (k_embed_out<.., KSN> with KSN != 0: output projection of step j +
# pose embedding of step j + 1) is a family of its own; "embed_out_step" / "embed_in" are the two stand-alone kernels the bench's event-timed families are.
FAMILIES = (("k_qkv_attention", "qkv_attention_fused"), ("k_layer_tail", "layer_tail_fused"), ("DEpiResidLNE", "ln_gemm"), ("DEpiBiasF16ILb1E", "ffn1_gelu_gemm"),
            ("k_embed_in", "embed_in"), ("k_embed_out", "embed_out_step"), ("DEpiEmbedInE", "embed_in"), ("DEpiEmbedOut", "embed_out_step"))
def family_of(name):
    for key, fam in FAMILIES:
        if key in name:
            if key == "k_embed_out" and "ELi0EEEv" not in name:
                return "embed_step_fused"
            return fam
    return None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            fam = family_of(r["Kernel_Name"])
            if fam:
                acc[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mst_amd  # noqa: E402,F401
from mst_amd import _native  # noqa: E402
out = {"source_hash": _native.built_hash(), "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --denoise-steps 12` "
                 "with MST_STREAMS=1 (64-clip launches); KiB -> bytes; FETCH_SIZE x2 (gfx950 correction)", "kernels": {}}
for fam, cs in acc.items():
    fetch = 2 * 1024 * sum(cs["FETCH_SIZE"]) / max(len(cs["FETCH_SIZE"]), 1)
    write = 1024 * sum(cs["WRITE_SIZE"]) / max(len(cs["WRITE_SIZE"]), 1)
    out["kernels"][fam] = {"fetch_bytes": round(fetch), "write_bytes": round(write), "hbm_bytes": round(fetch + write),
                           "launches": len(cs["FETCH_SIZE"])}
print(json.dumps(out, indent=1))
