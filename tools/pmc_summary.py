"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel mean of each counter."""
import csv, glob, sys, collections, re
def short(n):
    n = re.sub(r"^_ZN3mst\d+", "", n)
    for a, b in (("k_qkv_attention", "qkv_attention_fused"), ("k_layer_tail", "layer_tail_fused"), ("k_gemm_dmaILi64ELi512ELi2ELi2ELi4ELi1ENS_10RowsDirectENS_11DEpiResidLNE", "gemm_ln(outproj/ffn2)"), ("DEpiBiasF16ILb1E", "ffn1_gelu_gemm"),
                 ("DEpiEmbedInE", "embed_in_gemm"), ("DEpiEmbedOutILi1E", "embed_out_ddpm_step")):
        if a in n: return b
    return n[:40]
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(f"{k:28s}", "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())), f"(n={len(next(iter(cs.values())))})")
