"""Locate GEMM mismatches by row/column pattern (development aid)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops, _lib
dev = torch.device("cuda:0")
M, N, K = [int(x) for x in sys.argv[1:4]]
v = int(sys.argv[4]) if len(sys.argv) > 4 else -1
g = torch.Generator().manual_seed(1)
a = (torch.randn(M, K, generator=g) * 0.5).half().to(dev); w = (torch.randn(N, K, generator=g) * 0.1).half().to(dev)
bias = torch.randn(N, generator=g).to(dev)
_lib.lib().ink_gemm_set_variant(v)
for od in (torch.float32, torch.float16):
    out = ops.gemm(a, w, bias, out_dtype=od)
    ref = a.double() @ w.double().t() + bias.double()
    bad = ~((out.double() - ref).abs() < 1e-2 * ref.abs().max())
    print(od, "bad elements", int(bad.sum()), "of", bad.numel())
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print(" bad rows", rows[:40].tolist(), "... n=", len(rows)); print(" bad cols", cols[:40].tolist(), "... n=", len(cols))
        r0 = int(rows[0]); print(" row", r0, "bad cols:", bad[r0].nonzero().flatten()[:64].tolist())
        print(" out", out[r0, cols[:8]].tolist()); print(" ref", ref[r0, cols[:8]].tolist())
