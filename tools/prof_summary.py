"""Summarise rocprofv3 CSV output on the GPU box (the raw per-dispatch CSVs are too big to copy back).
usage: prof_summary.py <rocprof_dir> <out_json>"""
import collections, csv, glob, json, sys

def bucket(name):
    if "gemm_f16_nt_pp<4, 5" in name: return "gemm_pp320"
    if "gemm_f16_nt<256, 256, 64, 4, 4, 2" in name: return "gemm256"
    if "gemm_f16_nt" in name: return "gemm_other"
    if "flash_attn_kernel<80, 1" in name or "glob4_attn_kernel" in name: return "attn_global"
    if "flash_attn_kernel<80, 2" in name or "win4_attn_kernel" in name: return "attn_window"
    if "flash_attn" in name or "attn_few" in name: return "attn_other"
    if "layernorm" in name or "ln_merge" in name or "groupnorm" in name: return "norm"
    return "other"

d, out = sys.argv[1], sys.argv[2]
res = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (bucket(r["Kernel_Name"]), r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for (b, c), (s, n) in agg.items():
        res.setdefault(b, {})[c] = {"sum": s, "launches": n, "avg": s / n}
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = [{"name": r["Name"][:140], "calls": int(r["Calls"]), "total_ns": float(r["TotalDurationNs"]),
                            "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])} for r in rows[:45]]
json.dump(res, open(out, "w"), indent=1)
print("wrote", out)
