"""Development aid: wall time of every host-to-host step of the bench loop (are slow runs uniformly slow or a few stalls?)."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from inklayer_amd import pipeline

dev = torch.device("cuda:0")
B, boxes = 8, 16
det, seg, _ = bench.build_engines(dev, 0, 1, B)
pipe = pipeline.InkLayerPipeline(det, seg)
rs = np.random.RandomState(0)
host = pipe.pinned_like([rs.randint(0, 255, (1024, 1024, 3), dtype=np.uint8) for _ in range(B)])
import gc
gc_log = []
def _gc_cb(phase, info):
    if phase == "start": gc_log.append([len(times), info["generation"], time.perf_counter()])
    else: gc_log[-1][2] = time.perf_counter() - gc_log[-1][2]
gc.callbacks.append(_gc_cb)
if len(sys.argv) > 1 and sys.argv[1] == "nogc":
    gc.disable()
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams", "reserved_bytes.all.current")
snap = lambda: {k: torch.cuda.memory_stats(dev).get(k, 0) for k in keys}
prev = None
times = []
stats = []
for i in range(45):
    stats.append(snap())
    t0 = time.perf_counter()
    t = pipe.submit_host(host, top_n=boxes)
    t1 = time.perf_counter()
    if prev is not None:
        pipe.collect_host(prev)
    t2 = time.perf_counter()
    prev = t
    times.append((t1 - t0, t2 - t1))
pipe.collect_host(prev)
sub = np.array([a for a, b in times[5:]]) * 1e3
col = np.array([b for a, b in times[5:]]) * 1e3
tot = sub + col
print("per step ms: submit mean %.1f max %.1f | collect mean %.1f max %.1f | total mean %.1f median %.1f min %.1f max %.1f"
      % (sub.mean(), sub.max(), col.mean(), col.max(), tot.mean(), np.median(tot), tot.min(), tot.max()))
print("totals:", " ".join("%.0f" % v for v in tot))
print("gc events (step, generation, ms):", [(a, b, round(c * 1e3, 1)) for a, b, c in gc_log if a >= 5])
for i in range(6, len(stats)):
    d = {k: stats[i][k] - stats[i - 1][k] for k in keys}
    if any(v != 0 for v in d.values()):
        print("step", i - 1, "allocator deltas", d)
print("final", stats[-1])
