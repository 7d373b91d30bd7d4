"""Which kernel of the SAM stream disturbs the detector when both run at once?  The detector forward is queued on one
stream several times while ONE kind of SAM-encoder kernel loops on another; every detector output is compared with the
quiet-GPU result bit for bit.  Development aid (tools/race_check3.py found the two-compute-stream step wrong in a third
of back-to-back submissions while every stage alone, and one compute stream, are bit-reproducible)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import bench
    from inklayer_amd import ops, pipeline, synthetic
    dev = torch.device("cuda:0")
    det, seg, _ = bench.build_engines(dev, 0, 1, 8)
    pipe = pipeline.InkLayerPipeline(det, seg, overlap=False)
    imgs = [synthetic.synthetic_sketch(i) for i in range(8)]
    det_in, sam_in, sizes = pipe.preprocess(pipe.upload(imgs))
    ref = [t.clone() for t in det.forward(det_in, allow_graph=False)]
    torch.cuda.synchronize()

    F16, F32 = torch.float16, torch.float32
    B, H, hd, S, g = 8, 16, 80, 14, 64
    D, T, nwin = H * hd, g * g, 5
    Mw = nwin * nwin * S * S
    r = torch.arange(B * Mw)
    b, rr = r // Mw, r % Mw
    win, pos = rr // (S * S), rr % (S * S)
    y, x = (win // nwin) * S + pos // S, (win % nwin) * S + pos % S
    wm = torch.where((y < g) & (x < g), b * T + y * g + x, torch.full_like(r, -1)).to(torch.int32).to(dev)
    qkv = (torch.randn(B * T, 3 * D, device=dev) * 0.5).half()
    pad_k, pad_v = torch.randn(D, device=dev).half(), torch.randn(D, device=dev).half()
    rel_h, rel_w = torch.randn(2 * S - 1, hd, device=dev) * 0.2, torch.randn(2 * S - 1, hd, device=dev) * 0.2
    out = torch.empty(B * T, D, device=dev, dtype=F16)
    kw = dict(n_batch=B * nwin * nwin, n_heads=H, head_dim=hd, scale=hd ** -0.5)
    aug = ops.relpos_bias(qkv[:, :D], rel_h, rel_w, S=S, tok_rows=wm, **kw)
    rh64, rw64 = torch.randn(2 * g - 1, hd, device=dev) * 0.2, torch.randn(2 * g - 1, hd, device=dev) * 0.2
    kg = dict(n_batch=B, n_heads=H, head_dim=hd, scale=hd ** -0.5)
    rh, rw = ops.relpos_bias(qkv[:, :D], rh64, rw64, S=g, **kg)
    A = torch.randn(B * T, D, device=dev).half()
    Wq = (torch.randn(3 * D, D, device=dev) / 36).half()
    W1 = (torch.randn(4 * D, D, device=dev) / 36).half()
    x32 = torch.randn(B * T, D, device=dev)
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    hid = torch.empty(B * T, 4 * D, device=dev, dtype=F16)
    xres = torch.randn(B * T, D, device=dev)
    W2 = (torch.randn(D, 4 * D, device=dev) / 72).half()
    bg = {
        "nothing": (lambda: None, 0),
        "ping-pong GEMM qkv (f16 out)": (lambda: ops.gemm(A, Wq, None, out_dtype=F16), 30),
        "ping-pong GEMM lin1 + GELU": (lambda: ops.gemm(A, W1, None, act="gelu", out=hid), 20),
        "ping-pong GEMM lin2 + f32 residual in place": (lambda: ops.gemm(hid, W2, None, residual=xres, out=xres), 25),
        "LayerNorm rows (1280)": (lambda: ops.layernorm_rows(x32, gam, bet, 1e-6), 200),
        "window attention (win4)": (lambda: ops.flash_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], n_q=S * S, n_k=S * S,
                                                           rel_aug=aug, grid_w=S, tok_rows=wm, pad_k=pad_k, pad_v=pad_v, out=out, **kw), 80),
        "window rel-pos": (lambda: ops.relpos_bias(qkv[:, :D], rel_h, rel_w, S=S, tok_rows=wm, out=aug, **kw), 250),
        "global attention (glob4)": (lambda: ops.flash_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], rel_h=rh, rel_w=rw, grid_w=g,
                                                            out=out, **kg), 10),
        "whole SAM encoder": (lambda: seg.encode(sam_in, chan_reverse=True), 1),
    }
    s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
    n_det = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    only = sys.argv[2] if len(sys.argv) > 2 else None
    for name, (fn, n) in bg.items():
        if only and only not in name:
            continue
        bad = 0
        runs = 0
        for rnd in range(4):
            outs = []
            for k in range(n_det):                 # interleave the host-side issue so that both queues stay full
                with torch.cuda.stream(s_b):
                    for _ in range(n):
                        fn()
                with torch.cuda.stream(s_a):
                    outs.append(det.forward(det_in, allow_graph=False))
            torch.cuda.synchronize()
            for lg, bx in outs:
                runs += 1
                bad += int(not (torch.equal(lg, ref[0]) and torch.equal(bx, ref[1])))
        print(f"background: {name:46s} detector outputs different from the quiet run: {bad}/{runs}", flush=True)


if __name__ == "__main__":
    with torch.no_grad():
        main()
