"""Per-kernel table of ONE stage from a rocprofv3 kernel trace (tools/prof_stage.sh): only the dispatches after the
marker kernel of tools/stage_times.py --only (a torch flip) are counted, so engine set-up, calibration and warm-up
stay out.  usage: prof_trace.py <kernel_trace.csv> <passes> [top]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
passes = int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cut = max((i for i, r in enumerate(rows) if "flip" in r["Kernel_Name"]), default=-1)
rows = rows[cut + 1:]
agg = defaultdict(lambda: [0, 0])
for r in rows:
    a = agg[r["Kernel_Name"]]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) if rows else 0
print(f"{len(rows) / passes:.0f} launches per pass, kernel time {tot / 1e6 / passes:.3f} ms per pass, "
      f"first start to last end {span / 1e6 / passes:.3f} ms per pass")
for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{name[:110]:110s} {n / passes:7.1f} x {ns / n / 1e3:8.1f} us = {ns / 1e6 / passes:7.3f} ms")
