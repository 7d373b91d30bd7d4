"""Development aid: s_memtime stamps inside the global attention kernel (ablation build)."""
import ctypes, os, sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops

dev = torch.device("cuda:0")
B, H, hd, g = 8, 16, 80, 64
D, T = H * hd, g * g
qkv = (torch.randn(B * T, 3 * D, device=dev) * 0.5).half()
rh64 = torch.randn(2 * g - 1, hd, device=dev) * 0.2; rw64 = torch.randn(2 * g - 1, hd, device=dev) * 0.2
kg = dict(n_batch=B, n_heads=H, head_dim=hd, scale=hd ** -0.5)
rh, rw = ops.relpos_bias(qkv[:, :D], rh64, rw64, S=g, **kg)
out = torch.empty(B * T, D, device=dev, dtype=torch.float16)
for _ in range(3):
    ops.flash_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], rel_h=rh, rel_w=rw, grid_w=g, out=out, **kg)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["INKLAYER_HIP_LIB"])
buf = np.zeros(4 * 16 * 8, dtype=np.uint64)
assert lib.ink_glob4_read_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(4, 16, 8).astype(np.int64)
names = ["unit 1 (PV_B || smA)", "unit 2 (S_B || smA)", "barrier", "unit 3 (PV_A || smB + K reads + LDS writes)", "unit 4 (S_A || smB)"]
for w in (0, 3):
    for t in range(2, 10):
        d = np.diff(st[w, t, :6])
        print(f"w{w} tile {t + 8}: " + "  ".join(f"{n}:{int(v)}" for n, v in zip(names, d)) + f"   total {int(st[w, t + 1, 0] - st[w, t, 0])}")
