"""Per-shape GEMM time inside the real pipeline (development aid)."""
import sys, collections
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
pipe.run_prepared(d, s, z, top_n=16)
trace = []
ops.set_gemm_trace(trace)
pipe.run_prepared(d, s, z, top_n=16)
torch.cuda.synchronize()
ops.set_gemm_trace(None)
agg = collections.OrderedDict()
for fl, e0, e1, key in trace:
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"total gemm ms {tot:.2f}")
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{ms:7.2f} ms {100*ms/tot:5.1f}%  x{n:3d}  {fl/ms/1e9:6.0f} TF  {key}")
