"""Sweep GEMM tile variants on the pipeline's real shapes (development aid)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops, _lib

def ev_time(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

dev = torch.device("cuda:0")
shapes = [(32768, 3840, 1280, "qkv-glob"), (39200, 3840, 1280, "qkv-win"), (39200, 1280, 1280, "proj-win"),
          (32768, 5120, 1280, "lin1"), (32768, 1280, 5120, "lin2"), (4096, 3840, 1280, "qkv B=1"),
          (106352, 2048, 256, "dino ffn1"), (106352, 256, 2048, "dino ffn2"), (106352, 384, 256, "msda proj"),
          (320000, 288, 96, "swin qkv s0"), (7200, 256, 256, "dec"),
          (320000, 384, 96, "swin fc1 s0"), (329672, 96, 96, "swin proj s0")]
variants = [int(v) for v in sys.argv[1:]] or [0, 410, 440, 445]   # gm*100 + variant (see ink_gemm_f16)
print("shape".ljust(34), *[f"v{v}".rjust(8) for v in variants])
for (m, n, k, nm) in shapes:
    a = torch.randn(m, k, device=dev).half(); w = (torch.randn(n, k, device=dev) * 0.05).half()
    bias = torch.randn(n, device=dev)
    out = torch.empty(m, n, device=dev, dtype=torch.float16)
    ref = None
    row = []
    for v in variants:
        _lib.lib().ink_gemm_set_variant(v)
        ops.gemm(a, w, bias, out=out)
        if ref is None:
            idx = torch.cat([torch.arange(0, 512, device=dev), torch.randint(0, m, (2048,), device=dev),
                             torch.arange(max(0, m - 300), m, device=dev)])
            ref = (a[idx].float() @ w.float().t() + bias).half()
        err = (out[idx].float() - ref.float()).abs().max().item()
        out.zero_()
        t = ev_time(lambda: ops.gemm(a, w, bias, out=out))
        row.append(f"{2*m*n*k/t/1e9:6.0f}" + ("!" if err > 0.05 else " "))
    print(f"{nm} {m}x{n}x{k}".ljust(34), *[r.rjust(8) for r in row])
_lib.lib().ink_gemm_set_variant(-1)
