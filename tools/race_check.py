"""Two host-to-host steps in flight (submit_host x 2) against a synchronous reference, under a few pipeline settings:
which step goes wrong, and with what.  Development aid."""
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

    def grab(res):
        return [(np.asarray(x[0]).copy(), x[3].copy()) for x in res]

    def diff(name, a, b):
        nb = sum(int(not np.array_equal(x[0], y[0])) for x, y in zip(a, b))
        nm = sum(int(not np.array_equal(x[1], y[1])) for x, y in zip(a, b))
        print(f"{name:60s} images with different boxes {nb}/8, different masks {nm}/8", flush=True)

    for label, kw in (("encoder_first=True (default)", dict(encoder_first=True)),
                      ("encoder_first=False", dict(encoder_first=False)),
                      ("fuse_ffn=False", dict(encoder_first=True, fuse_ffn=False)),
                      ("fold_fusion=False", dict(encoder_first=True, fold_fusion=False)),
                      ("single stream (overlap=False)", dict(overlap=False))):
        det.fuse_ffn = kw.get("fuse_ffn", True)
        det.fold_fusion = kw.get("fold_fusion", True)
        p = pipeline.InkLayerPipeline(det, seg, overlap=kw.get("overlap", True))
        p.encoder_first = kw.get("encoder_first", True)
        host = p.pinned_like(imgs)
        torch.cuda.synchronize()
        ref = grab(p.collect_host(p.submit_host(host, top_n=16)))
        torch.cuda.synchronize()
        ref2 = grab(p.collect_host(p.submit_host(host, top_n=16)))
        diff(f"[{label}] synchronous step twice", ref, ref2)
        for rep in range(2):
            t1 = p.submit_host(host, top_n=16)
            t2 = p.submit_host(host, top_n=16)
            a = grab(p.collect_host(t1))
            b = grab(p.collect_host(t2))
            diff(f"[{label}] in flight, rep {rep}: FIRST vs reference", a, ref)
            diff(f"[{label}] in flight, rep {rep}: SECOND vs reference", b, ref)
            torch.cuda.synchronize()


if __name__ == "__main__":
    with torch.no_grad():
        main()
