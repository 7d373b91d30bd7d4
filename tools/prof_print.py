"""Print rows of a rocprofv3 kernel_stats CSV (gpurun_out/prof_<stage>_kernel_stats.csv) whose kernel name contains one of
the given substrings (all rows if none): calls per step (the stage tools run 7 passes), average us, ms per step."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / 1e6 / 7:.2f} ms per pass")
for r in rows:
    if not pats or any(p in r["Name"] for p in pats):
        print(f"{r['Name'][:90]:90s} calls/pass {int(r['Calls']) / 7:6.1f} avg {float(r['AverageNs']) / 1e3:8.1f} us "
              f"{float(r['TotalDurationNs']) / 1e6 / 7:7.3f} ms/pass")
