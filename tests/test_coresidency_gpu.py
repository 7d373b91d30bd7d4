"""Regression test for the co-residency corruption found in round 3: with the SAM window-attention
kernel on one stream and small kernels of another stream sharing its SIMDs, the small kernels read zeros in one register
of a quarter-wave (groupnorm_apply: 48 of 48 runs).  The window kernel now claims the whole register file; kernels that
run next to it must give the bytes they give on a quiet GPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@torch.no_grad()
def test_small_kernels_next_to_the_window_attention_kernel_are_bit_reproducible(dev):
    from inklayer_amd import ops
    F16 = torch.float16
    g = torch.Generator().manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    x256, y_gn = rn(106352, 256), rn(8 * 10000, 256)
    g256, b256 = rn(256), rn(256)

    def groupnorm():
        o = torch.empty(8 * 10000, 256, device=dev)
        ops.groupnorm_nhwc(y_gn, 8, 10000, 32, g256, b256, 1e-5, o, 10000 * 256)
        return o
    victims = {"groupnorm_nhwc": groupnorm,
               "layernorm_rows": lambda: ops.layernorm_rows(x256, g256, b256, 1e-5),
               "add_cvt_f16": lambda: ops.add_cvt_f16(x256)}
    refs = {k: fn().clone() for k, fn in victims.items()}
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
    out = torch.empty(B * T, D, device=dev, dtype=F16)
    kw = dict(n_batch=B * nwin * nwin, n_heads=H, head_dim=hd, scale=hd ** -0.5)
    aug = ops.relpos_bias(qkv[:, :D], rn(2 * S - 1, hd) * 0.2, rn(2 * S - 1, hd) * 0.2, S=S, tok_rows=wm, **kw)
    window = lambda: ops.flash_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], n_q=S * S, n_k=S * S, rel_aug=aug, grid_w=S,
                                    tok_rows=wm, pad_k=pad_k, pad_v=pad_v, out=out, **kw)
    window()
    torch.cuda.synchronize()
    ref_out = out.clone()
    s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
    bad = {k: 0 for k in victims}
    bad_window = 0
    for rnd in range(4):
        outs = []
        for k in range(6):
            with torch.cuda.stream(s_b):
                for _ in range(12):
                    window()
            with torch.cuda.stream(s_a):
                for vn, vfn in victims.items():
                    outs.append((vn, vfn()))
        torch.cuda.synchronize()
        bad_window += int(not torch.equal(out, ref_out))
        for vn, o in outs:
            bad[vn] += int(not torch.equal(o, refs[vn]))
    print("different from the quiet-GPU result:", bad, "window kernel:", bad_window)
    assert not any(bad.values()) and bad_window == 0
