"""Per-stage timing of the SAM path on one MI355X (development aid, not the bench contract)."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import sam, ops, weights_init

def ev_time(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cfg = sam.SamConfig()
    sd = weights_init.random_sam_state_dict(cfg, dev, seed=0)
    eng = sam.SamEngine(sd, cfg, dev, max_batch=B)
    del sd
    imgs = [torch.randint(0, 256, (1024, 1024, 3), dtype=torch.uint8, device=dev) for _ in range(B)]
    t = ev_time(lambda: eng.encode(imgs))
    print(f"encode B={B}: {t:.2f} ms  ({t/B:.2f} ms/img, {5.65e12*B/t/1e9:.0f} TFLOP/s eff)")
    emb = eng.encode(imgs)
    boxes = np.array([[100 + 10*i, 50 + 20*i, 600 + 10*i, 700 + 5*i] for i in range(16)], dtype=np.float32)
    t = ev_time(lambda: eng.decode(emb[0], boxes, (1024, 1024), (1024, 1024)))
    print(f"decode n=16: {t:.2f} ms")
    # individual kernels at ViT-H shapes
    D = 1280
    M = B * 4096
    a = torch.randn(M, D, device=dev).half(); w = torch.randn(3*D, D, device=dev).half()
    for (m, n, k, nm) in [(M, 3*D, D, "qkv"), (M, D, D, "proj"), (M, 4*D, D, "lin1"), (M, D, 4*D, "lin2"), (B*4900, 3*D, D, "qkv-win")]:
        a = torch.randn(m, k, device=dev).half(); w = (torch.randn(n, k, device=dev)*0.05).half()
        bias = torch.randn(n, device=dev)
        o16 = torch.empty(m, n, device=dev, dtype=torch.float16)
        t = ev_time(lambda: ops.gemm(a, w, bias, out=o16), iters=20)
        print(f"gemm {nm} {m}x{n}x{k}: {t*1e3:.1f} us  {2*m*n*k/t/1e9:.0f} TFLOP/s")
        if nm == "lin1":
            t = ev_time(lambda: ops.gemm(a, w, bias, act="gelu", out=o16), iters=20)
            print(f"   +gelu: {t*1e3:.1f} us  {2*m*n*k/t/1e9:.0f} TFLOP/s")
    qkv = torch.randn(B*4096, 3*D, device=dev).half()
    rph = torch.randn(127, 80, device=dev)*0.1
    sc = 80**-0.5
    q, k, v = qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:]
    rh, rw = ops.relpos_bias(q, rph, rph, S=64, n_batch=B, n_heads=16, head_dim=80, scale=sc)
    t = ev_time(lambda: ops.relpos_bias(q, rph, rph, S=64, n_batch=B, n_heads=16, head_dim=80, scale=sc, out=(rh, rw)), iters=10)
    print(f"relpos global: {t*1e3:.1f} us")
    t = ev_time(lambda: ops.flash_attn(q, k, v, n_batch=B, n_heads=16, head_dim=80, scale=sc, rel_h=rh, rel_w=rw, grid_w=64), iters=10)
    print(f"attn global: {t*1e3:.1f} us  {B*85.9e9/t/1e9:.0f} TFLOP/s")
    qkv = torch.randn(B*4900, 3*D, device=dev).half()
    q, k, v = qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:]
    rp = torch.randn(27, 80, device=dev)*0.1
    aug = ops.relpos_bias(q, rp, rp, S=14, n_batch=B*25, n_heads=16, head_dim=80, scale=sc)
    t = ev_time(lambda: ops.relpos_bias(q, rp, rp, S=14, n_batch=B*25, n_heads=16, head_dim=80, scale=sc, out=aug), iters=10)
    print(f"relpos window: {t*1e3:.1f} us")
    t = ev_time(lambda: ops.flash_attn(q, k, v, n_batch=B*25, n_heads=16, head_dim=80, scale=sc, rel_aug=aug, grid_w=14), iters=10)
    print(f"attn window: {t*1e3:.1f} us  {B*25*16*196*196*80*4/t/1e9:.0f} TFLOP/s")
    x = torch.randn(M, D, device=dev); g = torch.ones(D, device=dev)
    y = torch.empty(M, D, device=dev, dtype=torch.float16)
    t = ev_time(lambda: ops.layernorm_rows(x, g, g, 1e-6, out=y), iters=20)
    print(f"layernorm {M}x{D}: {t*1e3:.1f} us  {M*D*6/t/1e6:.0f} GB/s")

main()
