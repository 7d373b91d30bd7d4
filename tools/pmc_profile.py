"""gpurun_out/prof_summary.json (tools/prof_round.sh on the GPU box) -> profiles/rNN_gemm_pmc.json, the file bench.py
reads `roofline.traffic` from.  HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: both counters are in KB and
FETCH_SIZE is doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950 (16 B/lane streaming reads are tallied
at 64 B per 128-B request).
usage: pmc_profile.py <prof_summary.json> <out.json>"""
import json
import sys


def traffic(e):
    return (2.0 * e["FETCH_SIZE"]["avg"] + e["WRITE_SIZE"]["avg"]) * 1024.0


def hit_rate(e):
    h, m = e["TCC_HIT_sum"]["sum"], e["TCC_MISS_sum"]["sum"]
    return h / (h + m)


def main(src, dst):
    d = json.load(open(src))
    g = d["gemm_pp320"]
    sam_tokens, dim, heads = 8 * 4096, 1280, 16
    qkvo = 4 * sam_tokens * dim * 2                      # q, k, v read + o written, f16
    out = {
        "kernel": "gemm_f16_nt_pp<4,5> (ping-pong 256x320x32 tile)",
        "variant": 45,
        "launches": g["FETCH_SIZE"]["launches"],
        "fetch_kb_avg_raw": g["FETCH_SIZE"]["avg"],
        "write_kb_avg": g["WRITE_SIZE"]["avg"],
        "l2_hit_rate": hit_rate(g),
        "traffic_bytes_per_launch": traffic(g),
        "note": "rocprofv3 --pmc passes (one counter group per pass, tools/prof_round.sh) over `python3 bench.py "
                "--steps 1 --warmup 1 --no-cpu-baseline`; FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md HBM section, "
                "WRITE_SIZE (KB) as read; average over the four SAM ViT-H GEMM shapes (qkv, proj, lin1, lin2); "
                "made by tools/pmc_profile.py",
        "other_kernels": {k: {c: v[c]["avg"] for c in v} for k, v in d.items()
                          if k not in ("gemm_pp320", "kernel_stats")},
        "attention": {
            "attn_global": {"traffic_bytes_per_launch": traffic(d["attn_global"]), "l2_hit_rate": hit_rate(d["attn_global"]),
                            "algorithmic_bytes_qkvo": qkvo,
                            "rel_pos_tables_f32_bytes": 2 * sam_tokens * heads * 64 * 4,
                            "round1_traffic_bytes_per_launch": 4.4e9},
            "attn_window": {"traffic_bytes_per_launch": traffic(d["attn_window"]), "l2_hit_rate": hit_rate(d["attn_window"]),
                            "algorithmic_bytes_qkvo": qkvo},
        },
    }
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst, "gemm traffic/launch %.1f MB" % (out["traffic_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
