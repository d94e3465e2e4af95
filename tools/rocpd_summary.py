"""Summarise a rocprofv3 rocpd database (…_results.db) by (kernel, grid): calls, avg us, share.  python tools/rocpd_summary.py X.db [N] [--pct]
(--pct adds min / median / p90 / max of the per-dispatch durations: data-dependent paths show up as a wide spread)"""
import sqlite3
import sys
from collections import defaultdict

pct = "--pct" in sys.argv
if pct:
    sys.argv.remove("--pct")
db = sqlite3.connect(sys.argv[1])
rows = defaultdict(list)
for name, gx, gy, gz, wx, s, e in db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, start, end from kernels"):
    n = name.split("(")[0].replace("void q3::", "").replace("q3::", "")
    rows[(n, (gx // max(wx, 1), gy, gz))].append((e - s) / 1e3)
tot = sum(sum(v) for v in rows.values())
print(f"{'kernel':58s} {'grid(WGs)':>16s} {'calls':>7s} {'avg_us':>9s} {'share':>6s}")
for (n, g), v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    extra = ""
    if pct:
        w = sorted(v)
        extra = f"   min {w[0]:.2f}  p50 {w[len(w) // 2]:.2f}  p90 {w[int(len(w) * 0.9)]:.2f}  max {w[-1]:.2f}"
    print(f"{n[:58]:58s} {str(g):>16s} {len(v):7d} {sum(v) / len(v):9.2f} {100 * sum(v) / tot:5.1f}%{extra}")
