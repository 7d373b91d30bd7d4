"""One GEMM shape, a few launches (for rocprofv3 --pmc)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops, _lib
m, n, k = (int(v) for v in sys.argv[1:4])
variant = int(sys.argv[4]) if len(sys.argv) > 4 else -1
dev = torch.device("cuda:0")
a = torch.randn(m, k, device=dev).half(); w = (torch.randn(n, k, device=dev) * 0.05).half()
bias = torch.randn(n, device=dev)
out = torch.empty(m, n, device=dev, dtype=torch.float16)
_lib.lib().ink_gemm_set_variant(variant)
for _ in range(5):
    ops.gemm(a, w, bias, out=out)
torch.cuda.synchronize()
