"""Per-workgroup timeline (fill / main loop / epilogue, in us) of the ping-pong GEMM in its ABI-3 and ABI-4 forms at the
ViT-H shapes, batch 8.  Ablation library only: the stamped kernels write (HW_ID, XCC_ID, t_entry, t_filled, t_loop_end,
t_stores_issued, cycles...) per workgroup through a spare pointer (`residual` for the ABI-3 kernel, `col_scale` for the
ABI-4 ones)."""
import ctypes as C
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import build as _b
os.environ.setdefault("INKLAYER_HIP_LIB", str(_b.build(ablation=True, verbose=False)))
from inklayer_amd import ops, _lib

dev = torch.device("cuda:0")
M, D = 32768, 1280
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
a16 = rn(M, D).half()
hi, lo = torch.empty(M, D, device=dev, dtype=torch.float16), torch.empty(M, D, device=dev, dtype=torch.float16)
st = torch.empty(M, D // 80, 2, device=dev)
ops.hilo_split_stats(rn(M, D), hi, lo, st, 80)
L = _lib.lib()


def launch(variant, N, K, A, W, bias, out, c_f16, act=0, out_lo=None, res_hilo=None, stats=None, ln=None):
    ntiles = (M // 256) * (N // 320)
    dbg = torch.zeros(max(ntiles * 8, M * N if variant == 49 else 0), device=dev, dtype=torch.float32)
    p = _lib.InkGemm()
    p.A, p.W, p.C, p.bias = A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr()
    p.M, p.N, p.K, p.lda, p.ldw, p.ldc, p.act, p.c_f16 = M, N, K, A.stride(0), W.stride(0), out.stride(0), act, c_f16
    if variant == 49:
        p.residual, p.ldr = dbg.data_ptr(), N
    else:
        p.col_scale = dbg.data_ptr()
    if out_lo is not None:
        p.C_lo = out_lo.data_ptr()
    if res_hilo is not None:
        p.res_hi, p.res_lo, p.ldr = res_hilo[0].data_ptr(), res_hilo[1].data_ptr(), res_hilo[0].stride(0)
    if stats is not None:
        p.stats_out, p.stats_parts = stats.data_ptr(), N // 80
    if ln is not None:
        p.ln_stats, p.ln_parts, p.ln_dim, p.ln_eps, p.ln_colsum = ln[0].data_ptr(), ln[0].shape[1], D, 1e-6, ln[1].data_ptr()
    L.ink_gemm_set_variant(variant)
    for _ in range(3):
        _lib.check(L.ink_gemm_f16(C.byref(p), ops._stream()), "gemm")
    torch.cuda.synchronize()
    L.ink_gemm_set_variant(-1)
    t = dbg.view(torch.int32)[: ntiles * 8].cpu().view(ntiles, 8).long() & 0xffffffff
    t0 = t[:, 2].min()
    ent, fil, lp, stt = [(t[:, i] - t0).float() * 0.01 for i in (2, 3, 4, 5)]      # us
    return ent, fil, lp, stt


def report(name, *a, **kw):
    ent, fil, lp, stt = launch(*a, **kw)
    f = lambda v: f"{float(v.mean()):6.2f} (p10 {float(v.quantile(0.1)):5.2f}, p90 {float(v.quantile(0.9)):5.2f})"
    print(f"{name:40s} span {float(stt.max()):7.1f} us | fill {f(fil - ent)} | loop {f(lp - fil)} | epilogue {f(stt - lp)}", flush=True)


Wq = (rn(3 * D, D) / D ** 0.5).half()
bq = 0.1 * rn(3 * D)
csq = Wq.float().sum(1).contiguous()
oq = torch.empty(M, 3 * D, device=dev, dtype=torch.float16)
report("qkv ABI-3 (f16 out; +residual stamps)", 49, 3 * D, D, a16, Wq, bq, oq, 1)
report("qkv ABI-4 (LayerNorm folded)", 61, 3 * D, D, hi, Wq, bq, oq, 1, ln=(st, csq))
Wp = (rn(D, D) / D ** 0.5).half()
bp = 0.1 * rn(D)
x32 = rn(M, D)
report("proj ABI-3 (f32 out; stamps replace res.)", 49, D, D, a16, Wp, bp, x32, 0)
report("proj ABI-4 (split residual, split out, stats)", 63, D, D, a16, Wp, bp, hi, 2, out_lo=lo, res_hilo=(hi, lo), stats=st)
report("proj ABI-4 without stats", 63, D, D, a16, Wp, bp, hi, 2, out_lo=lo, res_hilo=(hi, lo))
