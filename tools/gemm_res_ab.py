"""A/B of the residual handling of the ViT-H f32-residual GEMMs (proj, lin2): preload vs late vs none (development aid)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops

dev = torch.device("cuda:0")
def ev_time(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for (m, n, k, nm) in [(32768, 1280, 1280, "proj"), (32768, 1280, 5120, "lin2"), (32768, 3840, 1280, "qkv"), (32768, 5120, 1280, "lin1")]:
    a = torch.randn(m, k, device=dev).half(); w = (torch.randn(n, k, device=dev) * 0.05).half()
    bias = torch.randn(n, device=dev); x = torch.randn(m, n, device=dev); ones = torch.ones(n, device=dev)
    o32 = torch.empty(m, n, device=dev); o16 = torch.empty(m, n, device=dev, dtype=torch.float16)
    fl = 2.0 * m * n * k
    rows = []
    rows.append(("f16 out, no residual", ev_time(lambda: ops.gemm(a, w, bias, out=o16))))
    rows.append(("f16 out, gelu", ev_time(lambda: ops.gemm(a, w, bias, act="gelu", out=o16))))
    rows.append(("f32 out, no residual", ev_time(lambda: ops.gemm(a, w, bias, out=o32))))
    rows.append(("f32 out, residual preloaded (product)", ev_time(lambda: ops.gemm(a, w, bias, residual=x, out=o32))))
    rows.append(("f32 out, in place x += (product)", ev_time(lambda: ops.gemm(a, w, bias, residual=x, out=x))))
    rows.append(("f32 out, late residual (col_scale=1)", ev_time(lambda: ops.gemm(a, w, bias, residual=x, col_scale=ones, out=o32))))
    rows.append(("f16 out, residual preloaded", ev_time(lambda: ops.gemm(a, w, bias, residual=x, out=o16))))
    # the same four product calls with the run-time-dispatch epilogue (variant 10000 + 4*100 + 45), interleaved rounds
    from inklayer_amd import _lib
    calls = {"qkv-type f16": lambda: ops.gemm(a, w, bias, out=o16), "lin1-type gelu f16": lambda: ops.gemm(a, w, bias, act="gelu", out=o16),
             "proj-type in place f32": lambda: ops.gemm(a, w, bias, residual=x, out=x)}
    ab = {}
    for rnd in range(3):
        for tag, var in (("compile-time modes", -1), ("run-time dispatch", 10445)):
            _lib.lib().ink_gemm_set_variant(var)
            for cn, fn in calls.items():
                ab.setdefault((cn, tag), []).append(ev_time(fn, iters=10))
    _lib.lib().ink_gemm_set_variant(-1)
    print(f"{nm} {m}x{n}x{k}")
    for name, us in rows:
        print(f"   {name:42s} {us:8.1f} us  {fl / us / 1e6:6.0f} TF")
    for (cn, tag), v in sorted(ab.items()):
        print(f"   A/B {cn:24s} {tag:20s} min {min(v):8.1f} us  median {sorted(v)[1]:8.1f} us")
