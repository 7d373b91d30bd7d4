"""Run each stage of the config-3 step twice (and the whole overlapped step several times) and compare bit for bit.
Development aid for hunting races / uninitialised reads."""
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
    pipe = pipeline.InkLayerPipeline(det, seg, overlap=False)
    imgs = [synthetic.synthetic_sketch(i) for i in range(8)]
    raw = pipe.upload(imgs)
    det_in, sam_in, sizes = pipe.preprocess(raw)
    torch.cuda.synchronize()

    def same(name, a, b):
        ok = all(torch.equal(x, y) for x, y in zip(a, b))
        worst = max(((x.float() - y.float()).abs().max().item() for x, y in zip(a, b)), default=0.0)
        print(f"{name:34s} {'identical' if ok else 'DIFFERENT'}   max abs diff {worst:.3e}", flush=True)
        return ok

    for flags in ((True, True), (False, True), (True, False)):
        det.fuse_ffn, det.fold_fusion = flags
        outs = []
        for _ in range(3):
            lg, bx = det.forward(det_in, allow_graph=False)
            outs.append((lg.clone(), bx.clone()))
            torch.cuda.synchronize()
        same(f"detector fuse_ffn={flags[0]} fold={flags[1]} run 0/1", outs[0], outs[1])
        same(f"detector fuse_ffn={flags[0]} fold={flags[1]} run 1/2", outs[1], outs[2])
    det.fuse_ffn, det.fold_fusion = True, True
    e = [seg.encode(sam_in, chan_reverse=True).clone() for _ in range(3)]
    same("encoder run 0/1", [e[0]], [e[1]])
    same("encoder run 1/2", [e[1]], [e[2]])
    dets = det.detect(det_in, top_n=16)
    boxes, iob = [], []
    for b, (bc, sc) in enumerate(dets):
        bx = bc.double().numpy()
        xyxy = np.stack([bx[:, 0] - bx[:, 2] / 2, bx[:, 1] - bx[:, 3] / 2, bx[:, 0] + bx[:, 2] / 2, bx[:, 1] + bx[:, 3] / 2], -1)
        boxes.append(pipeline.boxes_to_pixels(xyxy, 1024, 1024))
        iob += [b] * len(bx)
    boxes = torch.cat(boxes, 0)
    lows = [seg.decode_low_res(e[0], boxes, iob)[0].clone() for _ in range(3)]
    same("decoder run 0/1", [lows[0]], [lows[1]])
    same("decoder run 1/2", [lows[1]], [lows[2]])
    for overlap in (False, True):
        p = pipeline.InkLayerPipeline(det, seg, overlap=overlap)
        res = []
        for _ in range(4):
            r = p.run_uploaded(raw, top_n=16)
            torch.cuda.synchronize()
            res.append([x.masks.clone() for x in r] + [torch.from_numpy(np.asarray(x.boxes_xyxy_norm)) for x in r])
        for i in range(3):
            same(f"whole step overlap={overlap} run {i}/{i + 1}", res[i], res[i + 1])
    host = p.pinned_like(imgs)
    t1 = p.submit_host(host, top_n=16)
    t2 = p.submit_host(host, top_n=16)
    a = [(torch.from_numpy(x[3].copy()), torch.from_numpy(np.asarray(x[0]).copy())) for x in p.collect_host(t1)]
    b = [(torch.from_numpy(x[3].copy()), torch.from_numpy(np.asarray(x[0]).copy())) for x in p.collect_host(t2)]
    same("submit_host x2: masks", [x[0] for x in a], [x[0] for x in b])
    same("submit_host x2: boxes", [x[1] for x in a], [x[1] for x in b])


if __name__ == "__main__":
    with torch.no_grad():
        main()
