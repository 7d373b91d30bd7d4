"""Per-CU wall-clock timeline of the ping-pong GEMM (development aid; variants 48/49 are ablation builds that
write (HW_ID, XCC_ID, t_entry, t_filled, t_loop_end, t_stores_issued) per workgroup through `residual`)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import os
from inklayer_amd import build as _b
os.environ.setdefault("INKLAYER_HIP_LIB", str(_b.build(ablation=True, verbose=False)))   # ablation kernels live in their own .so
from inklayer_amd import ops, _lib

dev = torch.device("cuda:0")
m, n, k = 32768, 3840, 1280
a = torch.randn(m, k, device=dev).half(); w = (torch.randn(n, k, device=dev) * 0.05).half()
out = torch.empty(m, n, device=dev, dtype=torch.float16)
for v in [int(x) for x in sys.argv[1:] if x.isdigit()] or [448]:
    ntiles = (m // 256) * (n // (320 if v % 100 in (49, 50, 51, 52, 54) else 256))
    dbg = torch.zeros(m, n, device=dev, dtype=torch.float32)       # stands in for `residual`
    _lib.lib().ink_gemm_set_variant(v)
    for _ in range(3):
        ops.gemm(a, w, None, residual=dbg, out=out)
    torch.cuda.synchronize()
    t = dbg.view(torch.int32).flatten()[: ntiles * 8].cpu().view(ntiles, 8).long() & 0xffffffff
    hw, xcc = t[:, 0], t[:, 1] & 0xf
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    key = xcc * 1000 + se * 100 + sh * 50 + cu
    t0 = t[:, 2].min()
    ent, fil, lo, st = [(t[:, i] - t0) * 10 for i in (2, 3, 4, 5)]      # ns
    print(f"variant {v}: {ntiles} tiles on {key.unique().numel()} CUs; kernel span {(st.max()) / 1000:.1f} us")
    print(f"  fill     mean {float((fil - ent).float().mean()) / 1000:6.2f} us   (p10 {float((fil - ent).float().quantile(0.1)) / 1000:.2f}, p90 {float((fil - ent).float().quantile(0.9)) / 1000:.2f})")
    print(f"  loop     mean {float((lo - fil).float().mean()) / 1000:6.2f} us   (p10 {float((lo - fil).float().quantile(0.1)) / 1000:.2f}, p90 {float((lo - fil).float().quantile(0.9)) / 1000:.2f})")
    print(f"  epilogue mean {float((st - lo).float().mean()) / 1000:6.2f} us   (p10 {float((st - lo).float().quantile(0.1)) / 1000:.2f}, p90 {float((st - lo).float().quantile(0.9)) / 1000:.2f})")
    ghz_loop = (t[:, 6].float() / (lo - fil).float().clamp(min=1)).mean()
    ghz_epi = (t[:, 7].float() / (st - lo).float().clamp(min=1)).mean()
    print(f"  shader clock: {float(ghz_loop):.2f} GHz during the main loop, {float(ghz_epi):.2f} GHz during the epilogue")
    gaps = []
    for kk in key.unique().tolist():
        idx = (key == kk).nonzero().flatten()
        order = idx[ent[idx].argsort()]
        for i in range(1, len(order)):
            gaps.append(int(ent[order[i]] - st[order[i - 1]]))
    g = torch.tensor(gaps, dtype=torch.float32)
    print(f"  gap between a workgroup's last store issue and the next workgroup's entry on the same CU: mean {float(g.mean()) / 1000:.2f} us (p10 {float(g.quantile(0.1)) / 1000:.2f}, p90 {float(g.quantile(0.9)) / 1000:.2f}), n={len(gaps)}")
    kk = key.unique().tolist()[0]
    idx = (key == kk).nonzero().flatten(); order = idx[ent[idx].argsort()]
    print("  one CU:", " | ".join(f"{int(ent[i])/1000:.1f}+{int(fil[i]-ent[i])/1000:.1f}+{int(lo[i]-fil[i])/1000:.1f}+{int(st[i]-lo[i])/1000:.1f}" for i in order))
_lib.lib().ink_gemm_set_variant(-1)
