"""Single-sketch numbers (BASELINE.json configs[1]: SAM ViT-H encoder only, batch 1; and the whole pipeline at batch 1),
each call synchronised (latency, not throughput), plus a HIP-graph capture/replay check of the static parts."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from inklayer_amd import pipeline, synthetic

dev = torch.device("cuda:0")
det, seg, _ = bench.build_engines(dev, 0, 1, 1)
pipe = pipeline.InkLayerPipeline(det, seg)
raw = pipe.upload([synthetic.synthetic_sketch(0, 1024, 1024)])
d_in, s_in, sz = pipe.preprocess(raw)


def each(fn, n=10, sync=True):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn()
        if sync: torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    return float(np.median(ts))


tf = each(lambda: pipe.run_uploaded(raw, top_n=16))          # two-stream pipeline first (eager detector)
te, tde = each(lambda: seg.encode(s_in)), each(lambda: det.detect(d_in, top_n=16))
tf2 = each(lambda: pipe.run_uploaded(raw, top_n=16))         # again, now that a detector graph exists in the process
print(f"B=1 latency: SAM ViT-H encoder {te:.2f} ms ({5.65 / te:.2f} PFLOP/s on 5.65 TFLOP) | detector {tde:.2f} ms | "
      f"whole pipeline, 16 boxes {tf:.2f} ms = {1e3 / tf:.1f} sketches/s (again after the graph capture: {tf2:.2f} ms)")
print(f"host issue time per call: encoder {each(lambda: seg.encode(s_in), sync=False):.2f} ms, detector forward "
      f"{each(lambda: det.forward(d_in), sync=False):.2f} ms")


def capture(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()                     # warm-up: lazy attribute calls, allocator pools
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


g_enc, emb = capture(lambda: seg.encode(s_in))
ref = seg.encode(s_in).clone(); g_enc.replay(); torch.cuda.synchronize()
print(f"HIP graph of the encoder: bit-identical {torch.equal(emb, ref)}, replay {each(lambda: g_enc.replay()):.2f} ms")
g_det, (lg, bx) = capture(lambda: det.forward(d_in))
r_lg, r_bx = det.forward(d_in); r_lg, r_bx = r_lg.clone(), r_bx.clone(); g_det.replay(); torch.cuda.synchronize()
print(f"HIP graph of the detector forward: bit-identical {torch.equal(lg, r_lg) and torch.equal(bx, r_bx)}, "
      f"replay {each(lambda: g_det.replay()):.2f} ms")
