"""Host <-> device copy rates of this box with pinned memory (what the host-to-host bench legs cost)."""
import time
import torch
dev = torch.device("cuda:0")
for mb in (25, 134):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name} {mb} MB: {dt * 1e3:.2f} ms  {n / dt / 1e9:.1f} GB/s")
