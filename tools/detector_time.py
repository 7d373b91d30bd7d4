"""Detector alone at batch 8 (800x800): wall per forward, and kernel-level stage split with HIP events (development aid)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from inklayer_amd import ops, pipeline, synthetic
dev = torch.device("cuda:0")
det, seg, _ = bench.build_engines(dev, 0, 1, 8)
pipe = pipeline.InkLayerPipeline(det, seg, overlap=False)
imgs = [synthetic.synthetic_sketch(i) for i in range(8)]
d, s, z = pipe.prepare(imgs)
for _ in range(2):
    det.forward(d, allow_graph=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    det.forward(d, allow_graph=False)
torch.cuda.synchronize()
print(f"detector forward B=8: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms wall")
# stage split
pl = det.plan(800, 800, 8)
ev = lambda: torch.cuda.Event(enable_timing=True)
marks = [ev() for _ in range(5)]
marks[0].record(); feats = det.backbone(d, pl)
marks[1].record(); src = det.neck(feats, pl, 8)
marks[2].record(); memory, text = det.encoder(src, pl, 8)
marks[3].record(); det.decoder(memory, text, pl, 8, None)
marks[4].record(); torch.cuda.synchronize()
for n, a, b in zip(("swin backbone", "input_proj+GN", "encoder x6", "two-stage + decoder x6"), marks[:-1], marks[1:]):
    print(f"  {n:26s} {a.elapsed_time(b):7.2f} ms")
t0 = time.perf_counter()
for _ in range(5):
    seg.encode(s, chan_reverse=True)
torch.cuda.synchronize()
print(f"SAM encoder B=8: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms wall")
