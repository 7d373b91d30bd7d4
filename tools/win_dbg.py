"""Development aid: where does the window attention kernel disagree with float64?"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import ops

dev = torch.device("cuda:0")
S, B, H, hd = 14, 5, 2, 80
g = torch.Generator(device="cpu").manual_seed(S + B)
qkv = (torch.randn(B * S * S, 3 * H * hd, generator=g) * 1.5).half().to(dev)
rph = (torch.randn(2 * S - 1, hd, generator=g) * 0.3).to(dev)
rpw = (torch.randn(2 * S - 1, hd, generator=g) * 0.3).to(dev)
scale = hd ** -0.5
q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
r = ops.relpos_bias(q, rph, rpw, S=S, n_batch=B, n_heads=H, head_dim=hd, scale=scale)
out = ops.flash_attn(q, k, v, n_batch=B, n_heads=H, head_dim=hd, scale=scale, rel_aug=r, grid_w=S)
N = S * S
x = qkv.double().view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
qq, kk, vv = x[0], x[1], x[2]
attn = (qq * scale) @ kk.transpose(-2, -1)
idx = (torch.arange(S)[:, None] - torch.arange(S)[None, :] + (S - 1)).to(dev)
Rh, Rw = rph.double()[idx], rpw.double()[idx]
rq = qq.reshape(B, H, S, S, hd)
rel_h = torch.einsum("bnhwc,hkc->bnhwk", rq, Rh)
rel_w = torch.einsum("bnhwc,wkc->bnhwk", rq, Rw)
attn = (attn.view(B, H, S, S, S, S) + rel_h[..., :, None] + rel_w[..., None, :]).view(B, H, N, N)
ref = (attn.softmax(-1) @ vv)                      # B,H,N,hd
o = out.double().view(B, N, H, hd).permute(0, 2, 1, 3)
bad = ~torch.isfinite(o)
print("non-finite:", int(bad.sum()), "of", o.numel())
if bad.any():
    ii = bad.nonzero()
    print("  windows", ii[:, 0].unique().tolist(), "heads", ii[:, 1].unique().tolist())
    print("  queries", ii[:, 2].unique().tolist()[:60])
    print("  d", ii[:, 3].unique().tolist())
err = (o - ref).abs()
err[bad] = 0
print("max err (finite)", err.max().item())
eq = err.amax(dim=(0, 1, 3))
print("err by query block of 32:", [round(eq[i:i + 32].max().item(), 4) for i in range(0, N, 32)])
ed = err.amax(dim=(0, 1, 2))
print("err by d block of 8:", [round(ed[i:i + 8].max().item(), 4) for i in range(0, hd, 8)])
print("err by (window, head):", err.amax(dim=(2, 3)).cpu().numpy().round(4).tolist())
