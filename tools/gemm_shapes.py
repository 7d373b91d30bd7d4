"""Development aid: every GEMM of one pipeline step (batch 8, serial stream order) grouped by shape: calls, time, TFLOP/s."""
import sys, collections
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from inklayer_amd import ops, pipeline

dev = torch.device("cuda:0")
B = 8
det, seg, _ = bench.build_engines(dev, 0, 1, B)
pipe = pipeline.InkLayerPipeline(det, seg, overlap=False)
rs = np.random.RandomState(0)
raw = pipe.upload([rs.randint(0, 255, (1024, 1024, 3), dtype=np.uint8) for _ in range(B)])
for _ in range(2):
    pipe.run_uploaded(raw, top_n=16)
torch.cuda.synchronize()
tr = []
ops.set_gemm_trace(tr)
pipe.run_uploaded(raw, top_n=16)
torch.cuda.synchronize()
ops.set_gemm_trace(None)
agg = collections.OrderedDict()
for fl, e0, e1, key in tr:
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"{len(tr)} GEMM launches, {tot:.2f} ms")
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{str(key):58s} x{n:3d}  {ms:7.3f} ms  {ms / n * 1e3:7.1f} us each  {fl / ms / 1e9:6.0f} TFLOP/s")
