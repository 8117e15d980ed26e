"""Per-phase LDS counters of the two dominant kernels from one rocprofv3 --pmc run of csrc/probes/bin/phase_pmc (dispatches in a
fixed order, each kernel launched "up to" a phase mark; see the probe's header).  Prints cumulative counters per stop and the
differences = each phase's own share.   python tools/phase_pmc.py <rocprofv3 output dir>"""
import collections, csv, glob, sys

A_STOPS = [1, 1, 2, 2, 3, 3, 4, 4]
T_STOPS = [1, 1, 2, 2, 3, 3, 12, 12, 13, 13, 4, 4, 5, 5]
A_NAMES = {1: "projection (token-slab reads, ring)", 2: "+ q / k / v images written", 3: "+ Q / K reads, scores, softmax", 4: "+ P.V (transposed V reads), store"}
T_NAMES = {1: "start burst (no LDS instruction)", 2: "+ out-proj loop (att fragments, lo residual reads)", 3: "+ LayerNorm1 (hi residual reads, statistics exchange, x1 image)",
           12: "+ FFN1 of chunk 0 (x1 fragments)", 13: "+ GELU of chunk 0 (table gathers, H image stores)", 4: "+ the rest of the FFN", 5: "+ LayerNorm2 (scratch) and store"}

rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
disp = collections.OrderedDict()
for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
    d = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
att = [d for d in disp.values() if "k_qkv_attention2" in d["kernel"]]
tail = [d for d in disp.values() if "k_layer_tail" in d["kernel"]]
for name, ds, stops, names in (("k_qkv_attention2<13>", att, A_STOPS, A_NAMES), ("k_layer_tail", tail, T_STOPS, T_NAMES)):
    if len(ds) != len(stops):
        print(name, "unexpected number of dispatches", len(ds)); continue
    print(f"== {name}: 64-clip launch, counters summed over the chip")
    prev = {"SQ_LDS_BANK_CONFLICT": 0.0, "SQ_LDS_IDX_ACTIVE": 0.0}
    for i in range(1, len(stops), 2):                      # the second dispatch of every stop value
        d, st = ds[i], stops[i]
        c, a = d["SQ_LDS_BANK_CONFLICT"], d["SQ_LDS_IDX_ACTIVE"]
        dc, da = c - prev["SQ_LDS_BANK_CONFLICT"], a - prev["SQ_LDS_IDX_ACTIVE"]
        print(f"  up to mark {st:2d}  {names[st]:72s} conflict {c:10.0f}  active {a:10.0f} | this phase: conflict {dc:10.0f}  active {da:10.0f}  ratio {dc / da if da > 0 else 0:6.3f}")
        prev = {"SQ_LDS_BANK_CONFLICT": c, "SQ_LDS_IDX_ACTIVE": a}
