"""Does any stage read memory it did not write?  Each stage of the config-3 step is run normally, then again after the
caching allocator's free memory has been filled with a poison pattern (NaN bytes, then large finite values): a stage
whose result changes reads uninitialised (or out-of-bounds) memory.  Development aid."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def poison(dev, byte, gib=24):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    t = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
    t.fill_(byte)
    torch.cuda.synchronize()
    del t                      # stays cached: the next allocations are carved out of it


def main():
    import bench
    from inklayer_amd import pipeline, synthetic
    dev = torch.device("cuda:0")
    det, seg, _ = bench.build_engines(dev, 0, 1, 8)
    pipe = pipeline.InkLayerPipeline(det, seg, overlap=False)
    imgs = [synthetic.synthetic_sketch(i) for i in range(8)]
    raw = pipe.upload(imgs)
    det_in, sam_in, sizes = pipe.preprocess(raw)
    dets = det.detect(det_in, top_n=16)
    emb = seg.encode(sam_in, chan_reverse=True).clone()
    boxes, iob = [], []
    for b, (bc, sc) in enumerate(dets):
        bx = bc.double().numpy()
        xyxy = np.stack([bx[:, 0] - bx[:, 2] / 2, bx[:, 1] - bx[:, 3] / 2, bx[:, 0] + bx[:, 2] / 2, bx[:, 1] + bx[:, 3] / 2], -1)
        boxes.append(pipeline.boxes_to_pixels(xyxy, 1024, 1024))
        iob += [b] * len(bx)
    boxes = torch.cat(boxes, 0)
    stages = {
        "preprocess": lambda: [t for pair in zip(*pipe.preprocess(raw)[:2]) for t in pair],
        "detector (fused ffn, folded fusion)": lambda: list(det.forward(det_in, allow_graph=False)),
        "encoder": lambda: [seg.encode(sam_in, chan_reverse=True)],
        "decoder": lambda: [seg.decode_low_res(emb, boxes, iob)[0]],
    }
    variants = {"detector, two-GEMM ffn": ("fuse_ffn", False), "detector, unfolded fusion": ("fold_fusion", False)}
    for name, fn in list(stages.items()) + [(k, None) for k in variants]:
        if fn is None:
            attr, val = variants[name]
            setattr(det, attr, val)
            fn = stages["detector (fused ffn, folded fusion)"]
        want = [t.clone() for t in fn()]
        torch.cuda.synchronize()
        for byte in (0xFF, 0x7B):
            poison(dev, byte)
            got = [t.clone() for t in fn()]
            torch.cuda.synchronize()
            bad = [i for i, (a, b) in enumerate(zip(want, got)) if not torch.equal(a, b)]
            worst = max(((a.float() - b.float()).abs().nan_to_num(nan=float("inf")).max().item() for a, b in zip(want, got)),
                        default=0.0)
            print(f"{name:40s} poison 0x{byte:02X}: {'same' if not bad else 'DIFFERENT outputs ' + str(bad)}  max abs diff {worst:.3e}",
                  flush=True)
        if name in variants:
            setattr(det, variants[name][0], True)


if __name__ == "__main__":
    with torch.no_grad():
        main()
