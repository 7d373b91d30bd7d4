"""Stage times of the config-3 step (B = 8 sketches x 16 boxes, full depth) on one MI355X: each stage alone on the
stream (HIP events, median of N), then the whole overlapped step.  Development aid, not the bench contract.

    python tools/stage_times.py [--only decoder|encoder|detector] [--iters N]
`--only X` runs just that stage in a loop (for `rocprofv3 --kernel-trace --stats -- python3 tools/stage_times.py --only decoder`).
"""
import argparse
import statistics
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def ev(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts), min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--iters", type=int, default=7)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--boxes", type=int, default=16)
    a = ap.parse_args()
    import bench
    from inklayer_amd import pipeline, synthetic, sam as sm
    dev = torch.device("cuda:0")
    det, seg, _ = bench.build_engines(dev, 0, 1, a.batch)
    pipe = pipeline.InkLayerPipeline(det, seg, overlap=False)
    imgs = [synthetic.synthetic_sketch(i) for i in range(a.batch)]
    raw = pipe.upload(imgs)
    det_in, sam_in, sizes = pipe.preprocess(raw)
    torch.cuda.synchronize()
    dets = det.detect(det_in, top_n=a.boxes)
    emb = seg.encode(sam_in, chan_reverse=True).clone()
    L = seg.cfg.img_size
    all_boxes, iob = [], []
    for b, (bc, sc) in enumerate(dets):
        bx = bc.double().numpy()
        xyxy = np.stack([bx[:, 0] - bx[:, 2] / 2, bx[:, 1] - bx[:, 3] / 2, bx[:, 0] + bx[:, 2] / 2, bx[:, 1] + bx[:, 3] / 2], -1)
        all_boxes.append(pipeline.boxes_to_pixels(xyxy, 1024, 1024))
        iob += [b] * len(bx)
    boxes = torch.cat(all_boxes, 0)
    low, _ = seg.decode_low_res(emb, boxes, iob)
    stages = {
        "preprocess": lambda: pipe.preprocess(raw),
        "detector": lambda: det.forward(det_in, allow_graph=False),
        "encoder": lambda: seg.encode(sam_in, chan_reverse=True),
        "decoder": lambda: seg.decode_low_res(emb, boxes, iob),
        "postprocess": lambda: pipeline.ops_sam_postprocess(low, L, (1024, 1024), (1024, 1024), 0.0),
    }
    if a.only:
        # profiling mode (tools/prof_stage.sh): a marker kernel that nothing else launches separates set-up from the
        # measured passes in the kernel trace (tools/prof_trace.py cuts there)
        stages[a.only]()
        torch.cuda.synchronize()
        torch.flip(torch.arange(7, device=dev), (0,))
        torch.cuda.synchronize()
        for _ in range(a.iters):
            stages[a.only]()
        torch.cuda.synchronize()
        return
    tot = 0.0
    for name, fn in stages.items():
        med, mn = ev(fn, a.iters)
        tot += med
        print(f"{name:12s} median {med:8.3f} ms   min {mn:8.3f} ms")
    print(f"{'sum':12s}        {tot:8.3f} ms")
    med, mn = ev(lambda: pipe.run_uploaded(raw, top_n=a.boxes), a.iters)
    print(f"{'serial step':12s} median {med:8.3f} ms   min {mn:8.3f} ms")
    both = pipeline.InkLayerPipeline(det, seg, overlap=True)
    med, mn = ev(lambda: both.run_uploaded(raw, top_n=a.boxes), a.iters)
    print(f"{'overlap step':12s} median {med:8.3f} ms   min {mn:8.3f} ms")
    both.encoder_first = not both.encoder_first
    med, mn = ev(lambda: both.run_uploaded(raw, top_n=a.boxes), a.iters)
    print(f"{'overlap step, encoder_first=' + str(both.encoder_first):12s} median {med:8.3f} ms   min {mn:8.3f} ms")
    both.encoder_first = not both.encoder_first
    med, mn = ev(lambda: both.run_uploaded(raw, top_n=a.boxes), a.iters)
    print(f"{'overlap step, encoder_first=' + str(both.encoder_first):12s} median {med:8.3f} ms   min {mn:8.3f} ms")
    # interleaved A/B of the GEMM dispatch: -1 = product heuristic (persistent workgroups for the f16-output ViT-H
    # projections), -2 = the same heuristic with one tile per workgroup everywhere
    from inklayer_amd import _lib
    for rnd in range(3):
        for var, tag in ((-2, "one tile per workgroup"), (-1, "persistent qkv / lin1")):
            _lib.lib().ink_gemm_set_variant(var)
            e_med, e_mn = ev(stages["encoder"], a.iters)
            s_med, s_mn = ev(lambda: both.run_uploaded(raw, top_n=a.boxes), a.iters)
            print(f"A/B round {rnd} {tag:24s} encoder median {e_med:7.3f} min {e_mn:7.3f} ms | overlap step median {s_med:7.3f} min {s_mn:7.3f} ms")
    _lib.lib().ink_gemm_set_variant(-1)


if __name__ == "__main__":
    main()
