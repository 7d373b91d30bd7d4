"""Cost of the GEMM epilogue options on the SAM ViT-H shapes (development aid)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops

dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 32768
for (N, K, nm) in [(5120, 1280, "lin1"), (1280, 5120, "lin2"), (1280, 1280, "proj"), (3840, 1280, "qkv")]:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half(); b = torch.randn(N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.float16); x = torch.randn(M, N, device=dev)
    r = {}
    r["f16 out"] = t(lambda: ops.gemm(a, w, b, out=o16))
    r["f16 out + gelu"] = t(lambda: ops.gemm(a, w, b, act="gelu", out=o16))
    r["f32 out"] = t(lambda: ops.gemm(a, w, b, out=x))
    r["f32 out + residual (in place)"] = t(lambda: ops.gemm(a, w, b, residual=x, out=x))
    print(nm, f"{M}x{N}x{K}:", "  ".join(f"{k} {v:.0f} us ({2 * M * N * K / v / 1e6:.0f} TF)" for k, v in r.items()))
