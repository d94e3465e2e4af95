"""Per-kernel SQ counter table from one or more rocprofv3 --pmc passes (counter_collection.csv files of the SAME command):
    python tools/pmc_sq_summary.py pass1.csv pass2.csv ... [--top 14]
Kernels are keyed by (name, grid); counters summed over dispatches, durations from the first pass.  Ratios printed when their
operands are present: MFMA busy / SQ busy, LDS bank-conflict cycles / LDS active cycles, wait-on-LDS share of wave cycles."""
import csv
import sys
from collections import defaultdict

args = sys.argv[1:]
top = 14
if "--top" in args:
    i = args.index("--top")
    top = int(args[i + 1])
    del args[i:i + 2]
paths = args
val = defaultdict(lambda: defaultdict(float))
dur = defaultdict(float)
calls = defaultdict(int)
for pi, path in enumerate(paths):
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        for pre in ("void q3::", "q3::"):
            if k.startswith(pre):
                k = k[len(pre):]
        key = (k, r.get("Grid_Size", ""))
        val[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if pi == 0 and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            calls[key] += 1
keys = sorted(dur, key=lambda k: -dur[k])[:top]
names = sorted({c for k in keys for c in val[k]})
print("counters:", " ".join(names))
for k in keys:
    v = val[k]
    def ratio(a, b):
        return f"{v[a] / v[b]:.3f}" if v.get(b) else "-"
    print(f"{k[0][:58]:58s} grid {k[1]:>9s} x{calls[k]:<3d} {dur[k] / calls[k]:9.1f} us  "
          f"mfma_busy/busy {ratio('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES')}  lds_conflict/lds_active {ratio('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE')}  "
          f"wait_lds/wave {ratio('SQ_WAIT_INST_LDS', 'SQ_WAVE_CYCLES')}  wait_any/wave {ratio('SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES')}  "
          f"active_lds/wave {ratio('SQ_ACTIVE_INST_LDS', 'SQ_WAVE_CYCLES')}  active_valu/wave {ratio('SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES')}  "
          f"vmem_cyc/wave {ratio('SQ_INST_CYCLES_VMEM', 'SQ_WAVE_CYCLES')}")
    print("    " + "  ".join(f"{c}={v[c]:.3g}" for c in names if c in v))
