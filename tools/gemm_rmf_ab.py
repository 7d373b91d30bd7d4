"""A/B of the two residual paths of the ViT-H in-place f32 projections (proj, lin2) on the 256x320 ping-pong kernel:
variant 45 = residual preloaded into the accumulators, variant 54 = residual through the MFMA pipe during the K loop
(development aid).  Also checks the two against each other and against an f64 product."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops, _lib

dev = torch.device("cuda:0")
L = _lib.lib()

def ev_time(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

torch.manual_seed(0)
for (m, n, k, nm) in [(32768, 1280, 1280, "proj"), (32768, 1280, 5120, "lin2"), (4096 + 77, 1280, 1280, "ragged")]:
    a = torch.randn(m, k, device=dev).half(); w = (torch.randn(n, k, device=dev) * 0.05).half()
    bias = torch.randn(n, device=dev); x = torch.randn(m, n, device=dev) * 3
    outs = {}
    for var in (445, 454):
        L.ink_gemm_set_variant(var)
        outs[var] = ops.gemm(a, w, bias, residual=x, out=torch.empty(m, n, device=dev))
    L.ink_gemm_set_variant(-1)
    ref = (a[:2048].double() @ w.double().T + bias.double() + x[:2048].double())
    e45 = (outs[445][:2048].double() - ref).abs().max().item(); e54 = (outs[454][:2048].double() - ref).abs().max().item()
    d = (outs[445] - outs[454]).abs().max().item()
    print(f"{nm} {m}x{n}x{k}: max |45 - f64| {e45:.3e}   max |54 - f64| {e54:.3e}   max |45 - 54| {d:.3e}")
    if m < 32768: continue
    fl = 2.0 * m * n * k
    t = {}
    for rnd in range(4):
        for var in (445, 454):
            L.ink_gemm_set_variant(var)
            t.setdefault(var, []).append(ev_time(lambda: ops.gemm(a, w, bias, residual=x, out=x), iters=10))
    L.ink_gemm_set_variant(-1)
    for var, v in t.items():
        print(f"   variant {var}: min {min(v):7.1f} us  median {sorted(v)[len(v) // 2]:7.1f} us   {fl / min(v) / 1e6:6.0f} TF")
