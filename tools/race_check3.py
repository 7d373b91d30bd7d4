"""Failure rate of the second of two in-flight host-to-host steps under different stream arrangements (counts over
many repetitions; the first step and synchronous steps have never been seen wrong).  Development aid."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import bench
    from inklayer_amd import pipeline, synthetic
    dev = torch.device("cuda:0")
    det, seg, _ = bench.build_engines(dev, 0, 1, 8)
    imgs = [synthetic.synthetic_sketch(i) for i in range(8)]
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    cur = torch.cuda.current_stream(dev)

    def grab(res):
        return [np.asarray(x[0]).copy() for x in res]

    configs = [
        ("two compute streams, copies on their own streams (default)", dict()),
        ("two compute streams, D2H on the caller's stream", dict(d2h_cur=True)),
        ("two compute streams, both copies on the caller's stream", dict(d2h_cur=True, h2d_cur=True)),
        ("one compute stream, copies on their own streams", dict(overlap=False)),
        ("two compute streams, fuse_ffn off", dict(fuse_ffn=False)),
        ("two compute streams, fold_fusion off", dict(fold_fusion=False)),
    ]
    for label, kw in configs:
        det.fuse_ffn = kw.get("fuse_ffn", True)
        det.fold_fusion = kw.get("fold_fusion", True)
        p = pipeline.InkLayerPipeline(det, seg, overlap=kw.get("overlap", True))
        if kw.get("d2h_cur"):
            p.s_d2h = cur
        if kw.get("h2d_cur"):
            p.s_h2d = cur
        host = p.pinned_like(imgs)
        torch.cuda.synchronize()
        ref = grab(p.collect_host(p.submit_host(host, top_n=16)))
        torch.cuda.synchronize()
        bad1 = bad2 = 0
        for rep in range(reps):
            t1 = p.submit_host(host, top_n=16)
            t2 = p.submit_host(host, top_n=16)
            a = grab(p.collect_host(t1))
            b = grab(p.collect_host(t2))
            bad1 += int(any(not np.array_equal(x, y) for x, y in zip(a, ref)))
            bad2 += int(any(not np.array_equal(x, y) for x, y in zip(b, ref)))
            torch.cuda.synchronize()
        print(f"{label:66s} first step wrong {bad1}/{reps}, second step wrong {bad2}/{reps}", flush=True)


if __name__ == "__main__":
    with torch.no_grad():
        main()
