"""Victim / aggressor matrix for the co-residency corruption found in round 3: small kernels
(few registers: they fit next to a big wave on a SIMD) run on one stream and are compared bit for bit with their
quiet-GPU results, while ONE heavy kernel loops on another stream.  Development aid."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    from inklayer_amd import ops
    dev = torch.device("cuda:0")
    F16, F32 = torch.float16, torch.float32
    g = torch.Generator().manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    # ---- victims
    x256, x1280 = rn(106352, 256), rn(32768, 1280)
    g256, b256, g1280, b1280 = rn(256), rn(256), rn(1280), rn(1280)
    y_gn = rn(8 * 10000, 256)
    pos = rn(13294, 256)
    def _gn():
        o = torch.empty(8 * 10000, 256, device=dev)
        ops.groupnorm_nhwc(y_gn, 8, 10000, 32, g256, b256, 1e-5, o, 10000 * 256)
        return o
    victims = {
        "layernorm_rows C=256": lambda: ops.layernorm_rows(x256, g256, b256, 1e-5),
        "layernorm_rows C=1280": lambda: ops.layernorm_rows(x1280, g1280, b1280, 1e-6),
        "groupnorm_nhwc": lambda: _gn(),
        "add_cvt_f16 (+pos)": lambda: ops.add_cvt_f16(x256, pos),
        "add_split_f16": lambda: ops.add_split_f16(x256),
    }
    refs = {k: fn().clone() for k, fn in victims.items()}
    torch.cuda.synchronize()
    # ---- aggressors
    B, H, hd, S, gr = 8, 16, 80, 14, 64
    D, T, nwin = H * hd, gr * gr, 5
    Mw = nwin * nwin * S * S
    r = torch.arange(B * Mw)
    b, rr = r // Mw, r % Mw
    win, p_ = rr // (S * S), rr % (S * S)
    yy, xx = (win // nwin) * S + p_ // S, (win % nwin) * S + p_ % S
    wm = torch.where((yy < gr) & (xx < gr), b * T + yy * gr + xx, torch.full_like(r, -1)).to(torch.int32).to(dev)
    qkv = (rn(B * T, 3 * D) * 0.5).half()
    pad_k, pad_v = rn(D).half(), rn(D).half()
    rel_h, rel_w = rn(2 * S - 1, hd) * 0.2, rn(2 * S - 1, hd) * 0.2
    out = torch.empty(B * T, D, device=dev, dtype=F16)
    kw = dict(n_batch=B * nwin * nwin, n_heads=H, head_dim=hd, scale=hd ** -0.5)
    aug = ops.relpos_bias(qkv[:, :D], rel_h, rel_w, S=S, tok_rows=wm, **kw)
    kg = dict(n_batch=B, n_heads=H, head_dim=hd, scale=hd ** -0.5)
    rh, rw = ops.relpos_bias(qkv[:, :D], rn(2 * gr - 1, hd) * 0.2, rn(2 * gr - 1, hd) * 0.2, S=gr, **kg)
    A = rn(B * T, D).half()
    Wq = (rn(3 * D, D) / 36).half()
    w1, w2, bb1, bb2 = (rn(2048, 256) / 16).half(), (rn(256, 2048) / 45).half(), rn(2048), rn(256)
    blob = ops.ffn256_pack(w1, bb1, w2)
    xs16 = x256.half()
    A2, W2 = rn(106352, 256).half(), (rn(2048, 256) / 16).half()
    # round-3 kernels of the SAM decoder / detector encoder
    n_box, T = 32, 4096
    a_att = rn(n_box * T, 128)
    keys_r = rn(n_box * T, 256)
    pl_blob = ops.proj256_ln_pack(ops.split_weight(rn(256, 128) / 11))
    u0 = rn(n_box * T, 256)
    up_blob = ops.sam_upscale_pack(ops.split_weight(rn(128, 64) / 8))
    hyper = rn(n_box, 32)
    b3 = rn(128)
    g64, b64 = rn(64), rn(64)
    q7 = rn(n_box * 7, 128)
    kv = rn(n_box * T, 256)
    qi = rn(n_box * T, 128)
    k7, v7 = rn(n_box * 7, 128), rn(n_box * 7, 128)
    blob_pre = ops.ffn256_pack(w1, bb1, w2, (rn(256, 256) / 16).half())
    aggressors = {
        "nothing": (lambda: None, 0),
        "proj256_ln": (lambda: ops.proj256_ln(a_att, pl_blob, b256, keys_r, g256, b256, 1e-5), 4),
        "upscale_tail": (lambda: ops.sam_upscale_tail(u0, n_box, 64, g64, b64, 1e-6, up_blob, b3, hyper), 4),
        "attn_fewq16 (2 wave groups)": (lambda: ops.attn_fewq(q7, kv[:, :128], kv[:, 128:], n_batch=n_box, n_heads=8, head_dim=16,
                                                             scale=0.25, n_q=7, n_k=T), 8),
        "attn_fewkeys16": (lambda: ops.attn_fewkeys(qi, k7, v7, B=n_box, n_heads=8, head_dim=16, scale=0.25, n_q=T), 8),
        "ffn256 fused with pre-phase": (lambda: ops.ffn256_fused(xs16, x256, blob_pre, 2048, bb2, g256, b256, 1e-5,
                                                                 pre=(b256, g256, b256)), 5),
        "win4 window attention": (lambda: ops.flash_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], n_q=S * S, n_k=S * S, rel_aug=aug,
                                                         grid_w=S, tok_rows=wm, pad_k=pad_k, pad_v=pad_v, out=out, **kw), 12),
        "glob4 global attention": (lambda: ops.flash_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], rel_h=rh, rel_w=rw, grid_w=gr,
                                                          out=out, **kg), 2),
        "ffn256 fused": (lambda: ops.ffn256_fused(xs16, x256, blob, 2048, bb2, g256, b256, 1e-5), 6),
        "ping-pong GEMM": (lambda: ops.gemm(A, Wq, None, out_dtype=F16), 5),
        "16-wave 256x256 GEMM": (lambda: ops.gemm(A2, W2, None, out_dtype=F16), 8),
    }
    s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
    print(f"{'aggressor':28s} " + " | ".join(f"{k:22s}" for k in victims), flush=True)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for an, (afn, n) in aggressors.items():
        if only and only not in an and an != 'nothing':
            continue
        bad = {k: 0 for k in victims}
        runs = 0
        for rnd in range(8):
            outs = []
            for k in range(6):
                with torch.cuda.stream(s_b):
                    for _ in range(n):
                        afn()
                with torch.cuda.stream(s_a):
                    for vn, vfn in victims.items():
                        outs.append((vn, vfn()))
            torch.cuda.synchronize()
            runs += 6
            for vn, o in outs:
                if not torch.equal(o, refs[vn]):
                    bad[vn] += 1
                    if bad[vn] <= 2:
                        d = (o != refs[vn]).nonzero()
                        print(f"      {vn}: {d.shape[0]} elements differ; first {d[:5].tolist()}; got {o[tuple(d[0])].item():.5f} want {refs[vn][tuple(d[0])].item():.5f}", flush=True)
        print(f"{an:28s} " + " | ".join(f"{str(bad[k]) + '/' + str(runs):22s}" for k in victims), flush=True)


if __name__ == "__main__":
    with torch.no_grad():
        main()
