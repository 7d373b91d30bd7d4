"""Development aid: host time per SAM ViT-H encoder call (eager launches vs graph replay) and per pipeline step."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import sam, weights_init

dev = torch.device("cuda:0")
B = 8
cfg = sam.SamConfig()
eng = sam.SamEngine(weights_init.random_sam_state_dict(cfg, dev, seed=1), cfg, device=dev, max_batch=B)
imgs = [torch.randint(0, 255, (1024, 1024, 3), dtype=torch.uint8, device=dev) for _ in range(B)]
for mode in ("eager", "graph"):
    eng.graph_blocks = mode == "graph"
    for _ in range(3):
        eng.encode(imgs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        eng.encode(imgs)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode}: host returns after {(t1 - t0) / n * 1e3:.2f} ms per encode; wall {(t2 - t0) / n * 1e3:.2f} ms per encode")
