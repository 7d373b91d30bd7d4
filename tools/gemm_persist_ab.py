"""A/B of the one-tile-per-workgroup ping-pong kernel (variant 45) and its persistent form (variant 55) on the four
ViT-H projections with the product's epilogues, plus equality checks on ragged shapes (development aid)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops, _lib

dev = torch.device("cuda:0")
L = _lib.lib()

def ev_time(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

def forms(a, w, bias, x):
    return {"f16 out": lambda out: ops.gemm(a, w, bias, out=out) if out is not None else ops.gemm(a, w, bias, out_dtype=torch.float16),
            "gelu f16 out": lambda out: ops.gemm(a, w, bias, act="gelu", out=out) if out is not None else ops.gemm(a, w, bias, act="gelu", out_dtype=torch.float16),
            "f32 + residual": lambda out: ops.gemm(a, w, bias, residual=x, out=out) if out is not None else ops.gemm(a, w, bias, residual=x),
            "f32 plain": lambda out: ops.gemm(a, w, bias, out=out) if out is not None else ops.gemm(a, w, bias)}

torch.manual_seed(0)
print("== equality, variant 55 vs 45 (bit-exact expected: same arithmetic order)")
for (m, n, k) in [(300, 320, 128), (4096 + 77, 640, 1280), (256 * 70, 1280, 256), (256 * 300 + 5, 320, 192), (32768, 3840, 1280)]:
    a = torch.randn(m, k, device=dev).half(); w = (torch.randn(n, k, device=dev) * 0.05).half()
    bias = torch.randn(n, device=dev); x = torch.randn(m, n, device=dev)
    for name, fn in forms(a, w, bias, x).items():
        if m == 32768 and name != "f16 out": continue
        L.ink_gemm_set_variant(445); r45 = fn(None)
        L.ink_gemm_set_variant(455); r55 = fn(None)
        L.ink_gemm_set_variant(-1)
        d = (r45.float() - r55.float()).abs().max().item()
        print(f"   {m}x{n}x{k} {name:16s} max |45 - 55| = {d:.3e}  finite {bool(torch.isfinite(r55).all())}")
print("== timing (interleaved, min / median of 4 rounds of 10)")
for (m, n, k, nm, form) in [(32768, 3840, 1280, "qkv", "f16 out"), (32768, 1280, 1280, "proj", "f32 + residual"),
                            (32768, 5120, 1280, "lin1", "gelu f16 out"), (32768, 1280, 5120, "lin2", "f32 + residual"),
                            (39200, 3840, 1280, "qkv-win", "f16 out"), (39200, 1280, 1280, "proj-win", "f32 + residual")]:
    a = torch.randn(m, k, device=dev).half(); w = (torch.randn(n, k, device=dev) * 0.05).half()
    bias = torch.randn(n, device=dev); x = torch.randn(m, n, device=dev)
    out = x if "f32" in form else torch.empty(m, n, device=dev, dtype=torch.float16)
    fn = forms(a, w, bias, x)[form]
    t = {}
    for rnd in range(4):
        for var in (445, 455):
            L.ink_gemm_set_variant(var)
            t.setdefault(var, []).append(ev_time(lambda: fn(out)))
    L.ink_gemm_set_variant(-1)
    fl = 2.0 * m * n * k
    print(f"{nm} {m}x{n}x{k} ({form})")
    for var, v in t.items():
        print(f"   variant {var}: min {min(v):7.1f} us  median {sorted(v)[2]:7.1f} us   {fl / min(v) / 1e6:6.0f} TFLOP/s")
