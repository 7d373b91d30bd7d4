"""Development aid: s_memtime stamps inside the window attention kernel (ablation build: python -m inklayer_amd.build
--ablation; run with INKLAYER_HIP_LIB=inklayer_amd/lib/libinklayer_hip_ablation.so)."""
import ctypes, os, sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops, _lib

dev = torch.device("cuda:0")
B, H, hd, S, g = 8, 16, 80, 14, 64
D, T, nwin = H * hd, g * g, 5
Mw = nwin * nwin * S * S
r = torch.arange(B * Mw); b, rr = r // Mw, r % Mw
win, pos = rr // (S * S), rr % (S * S)
y, x = (win // nwin) * S + pos // S, (win % nwin) * S + pos % S
wm = torch.where((y < g) & (x < g), b * T + y * g + x, torch.full_like(r, -1)).to(torch.int32).to(dev)
qkv = (torch.randn(B * T, 3 * D, device=dev) * 0.5).half()
pad_k = torch.randn(D, device=dev).half(); pad_v = torch.randn(D, device=dev).half()
rel_h = torch.randn(2 * S - 1, hd, device=dev) * 0.2; rel_w = torch.randn(2 * S - 1, hd, device=dev) * 0.2
out = torch.empty(B * T, D, device=dev, dtype=torch.float16)
nb = B * nwin * nwin
kw = dict(n_batch=nb, n_heads=H, head_dim=hd, scale=hd ** -0.5)
aug = ops.relpos_bias(qkv[:, :D], rel_h, rel_w, S=S, tok_rows=wm, **kw)
for _ in range(3):
    ops.flash_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], n_q=S * S, n_k=S * S, rel_aug=aug, grid_w=S,
                   tok_rows=wm, pad_k=pad_k, pad_v=pad_v, out=out, **kw)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["INKLAYER_HIP_LIB"])
buf = np.zeros(4 * 16 * 16, dtype=np.uint64)
assert lib.ink_win4_read_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(4, 16, 16).astype(np.int64)
names = ["top", "handoff written", "barrier 2", "prefetch issued", "S_A(0)", "tile 0", "tile 1", "tile 2 + tail S",
         "Q issued", "tail done", "stores issued"]
print("s_memtime ticks (100 MHz constant clock -> x10 ns); wave 0 and wave 3, blocks 1..6")
for w in (0, 3):
    for blk in range(1, 7):
        row = st[w, blk, :11]
        d = np.diff(row)
        nxt = st[w, blk + 1, 0] - row[10]
        print(f"w{w} b{blk}: " + "  ".join(f"{n}:{int(v)}" for n, v in zip(names[1:], d)) + f"  ->next top:{int(nxt)}   block total {int(st[w, blk + 1, 0] - row[0])}")
