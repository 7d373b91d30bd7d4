"""Same-box A/B of the ViT-H projection forms at batch 8 (M = 32768): ABI-3 (f32 stream, LayerNorm'd f16 operand) against
ABI-4 (split-f16 stream, folded LayerNorm), interleaved rounds in one process, median us per launch."""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops

dev = torch.device("cuda:0")
M, D = 32768, 1280
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
x32 = rn(M, D)
a16 = rn(M, D).half()
h16 = rn(M, 4 * D).half()
chunk = ops.gemm_stats_chunk(M, D, D)
hi, lo = torch.empty(M, D, device=dev, dtype=torch.float16), torch.empty(M, D, device=dev, dtype=torch.float16)
st = torch.empty(M, D // chunk, 2, device=dev)
ops.hilo_split_stats(x32, hi, lo, st, chunk)
W = {n: (rn(n, k) / k ** 0.5).half() for n, k in ((3 * D, D), (D, D), (4 * D, D))}
W2 = (rn(D, 4 * D) / (4 * D) ** 0.5).half()
b = {n: 0.1 * rn(n) for n in (D, 3 * D, 4 * D)}
cs = {n: W[n].float().sum(1).contiguous() for n in (3 * D, 4 * D)}
o_qkv = torch.empty(M, 3 * D, device=dev, dtype=torch.float16)
o_hid = torch.empty(M, 4 * D, device=dev, dtype=torch.float16)
forms = {
    "qkv  ABI-3 (plain f16 operand)": lambda: ops.gemm(a16, W[3 * D], b[3 * D], out=o_qkv),
    "qkv  ABI-4 (LayerNorm folded)": lambda: ops.gemm(hi, W[3 * D], b[3 * D], out=o_qkv, ln=(st, D, 1e-6, cs[3 * D])),
    "lin1 ABI-3 (+GELU)": lambda: ops.gemm(a16, W[4 * D], b[4 * D], act="gelu", out=o_hid),
    "lin1 ABI-4 (LayerNorm folded, +GELU)": lambda: ops.gemm(hi, W[4 * D], b[4 * D], act="gelu", out=o_hid, ln=(st, D, 1e-6, cs[4 * D])),
    "proj ABI-3 (f32 residual in place)": lambda: ops.gemm(a16, W[D], b[D], residual=x32, out=x32),
    "proj ABI-4 (split residual in place + stats)": lambda: ops.gemm(a16, W[D], b[D], residual_hilo=(hi, lo), out_hilo=(hi, lo), stats_out=st),
    "lin2 ABI-3": lambda: ops.gemm(h16, W2, b[D], residual=x32, out=x32),
    "lin2 ABI-4": lambda: ops.gemm(h16, W2, b[D], residual_hilo=(hi, lo), out_hilo=(hi, lo), stats_out=st),
    "layernorm_rows (what ABI-4 removes, x2 per block)": lambda: ops.layernorm_rows(x32, b[D], b[D], 1e-6, out=a16),
}
T = {k: [] for k in forms}
for k, f in forms.items():
    f()
torch.cuda.synchronize()
for rnd in range(7):
    for k, f in forms.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            f()
        e1.record()
        torch.cuda.synchronize()
        T[k].append(e0.elapsed_time(e1) / 4 * 1e3)
for k, v in T.items():
    print(f"{k:52s} median {statistics.median(v):7.1f} us   min {min(v):7.1f} us")
