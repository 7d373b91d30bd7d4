"""Time the fused FFN kernel variants built by tools/ffn_variants.sh at the detector's shape (B = 8: 106352 rows,
d_ffn 2048): HIP events around 20 launches each.  Development aid."""
import ctypes as C
import glob
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
M, HID = 8 * 13294, 2048
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 256, generator=g).to(dev)
x16 = x.half()
w1 = (torch.randn(HID, 256, generator=g) / 16).half().to(dev)
w2 = (torch.randn(256, HID, generator=g) / 45).half().to(dev)
b1, b2 = torch.randn(HID, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
lg, lb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
out = torch.empty_like(x)
libs = {}
for path in sorted(glob.glob(str(ROOT / "tools" / "micro" / "ffn_*.so"))):
    L = C.CDLL(path)
    need = C.c_int64(0)
    L.ink_ffn256_pack_bytes(HID, C.byref(need))
    blob = torch.empty(need.value // 2, device=dev, dtype=torch.float16)
    vp = C.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    assert L.ink_ffn256_pack(vp(w1.data_ptr()), vp(b1.data_ptr()), vp(w2.data_ptr()), HID, vp(blob.data_ptr()), st) == 0
    libs[Path(path).stem] = (L, blob)
vp = C.c_void_p
st = vp(torch.cuda.current_stream().cuda_stream)
best = {}
for rep in range(3):                       # interleaved passes: the first kernels of a process see a cold clock
    for name, (L, blob) in libs.items():
        call = lambda: L.ink_ffn256_fused(vp(x16.data_ptr()), C.c_int64(256), vp(x.data_ptr()), vp(blob.data_ptr()),
                                          vp(b2.data_ptr()), vp(lg.data_ptr()), vp(lb.data_ptr()), C.c_float(1e-5), M, HID,
                                          vp(out.data_ptr()), st)
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        best.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3 / 20)
for name, ts in best.items():
    us = min(ts)
    print(f"{name:24s} {us:8.1f} us per launch (min of {[round(t, 1) for t in ts]})   {2 * 2 * M * 256 * HID / us / 1e6:6.0f} TFLOP/s", flush=True)
