"""Yardstick for the dense ViT-H projections: this library's ping-pong kernel against the vendor library GEMM that
torch.matmul dispatches to (hipBLASLt / rocBLAS) on the same box, same shapes, f16 in / f32 accumulate / f16 out, no
epilogue on either side (development aid; the product never calls the vendor library)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops

dev = torch.device("cuda:0")

def ev_time(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

torch.manual_seed(0)
shapes = [(32768, 3840, 1280, "qkv"), (32768, 1280, 1280, "proj"), (32768, 5120, 1280, "lin1"), (32768, 1280, 5120, "lin2"),
          (8192, 8192, 8192, "square 8k")]
for (m, n, k, nm) in shapes:
    a = torch.randn(m, k, device=dev).half(); w = (torch.randn(n, k, device=dev) * 0.05).half()
    wt = w.t().contiguous()                      # [K, N]: the NN form, in case the library prefers it
    o = torch.empty(m, n, device=dev, dtype=torch.float16)
    fl = 2.0 * m * n * k
    t = {}
    for rnd in range(3):
        t.setdefault("ink_gemm_f16 (this library)", []).append(ev_time(lambda: ops.gemm(a, w, None, out=o)))
        t.setdefault("torch.matmul(a, w.T)  [NT]", []).append(ev_time(lambda: torch.matmul(a, w.t(), out=o)))
        t.setdefault("torch.matmul(a, wt)   [NN]", []).append(ev_time(lambda: torch.matmul(a, wt, out=o)))
    ref = torch.matmul(a[:512], w.t()); mine = ops.gemm(a[:512].contiguous(), w, None, out_dtype=torch.float16)
    print(f"{nm} {m}x{n}x{k}   (max |ours - library| on 512 rows: {(ref.float() - mine.float()).abs().max().item():.3e})")
    for name, v in t.items():
        print(f"   {name:34s} min {min(v):8.1f} us  median {sorted(v)[1]:8.1f} us   {fl / min(v) / 1e6:6.0f} TFLOP/s")
