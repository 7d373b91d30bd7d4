"""Time the SAM ViT-H attention calls at the bench shapes (development aid)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops
from inklayer_amd.sam import SamConfig

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

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

tw = t(lambda: ops.flash_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], n_q=S * S, n_k=S * S, rel_aug=aug, grid_w=S,
                              tok_rows=wm, pad_k=pad_k, pad_v=pad_v, out=out, **kw))
tr = t(lambda: ops.relpos_bias(qkv[:, :D], rel_h, rel_w, S=S, tok_rows=wm, out=aug, **kw))
byts = B * T * 4 * D * 2
print(f"window attention: {tw:7.1f} us  ({byts / tw / 1e6:.2f} TB/s of q,k,v,o = {byts / tw / 1e6 / 8 * 100:.0f} % of 8 TB/s)   rel-pos: {tr:6.1f} us")
rh64 = torch.randn(2 * g - 1, hd, device=dev) * 0.2; rw64 = torch.randn(2 * g - 1, hd, device=dev) * 0.2
kg = dict(n_batch=B, n_heads=H, head_dim=hd, scale=hd ** -0.5)
rh, rw = ops.relpos_bias(qkv[:, :D], rh64, rw64, S=g, **kg)
tg = t(lambda: ops.flash_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], rel_h=rh, rel_w=rw, grid_w=g, out=out, **kg), n=5)
fl = B * H * T * T * hd * 4
print(f"global attention: {tg:7.1f} us  ({fl / tg / 1e6:.0f} TFLOP/s;  {byts / tg / 1e6:.2f} TB/s of q,k,v,o)")


