"""Summarise a rocprofv3 kernel_trace.csv by (kernel, grid size): calls, avg/min/max us."""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0].replace("void q3::", "").replace("q3::", "")
        grid = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        rows[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print(f"{'kernel':58s} {'grid(WGs)':>16s} {'calls':>7s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'share':>6s}")
for (name, grid), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name[:58]:58s} {str(grid):>16s} {len(v):7d} {sum(v)/len(v):8.2f} {min(v):8.2f} {max(v):8.2f} {100*sum(v)/tot:5.1f}%")
