"""Does a GEMM launch give different bytes when other streams are busy?  Foreground: detector-shaped GEMMs on one
stream, each compared with its own quiet-GPU result.  Background (other streams): the ping-pong GEMM of SAM ViT-H, a
window-attention-like mix, and a device-to-host copy.  Development aid for the in-flight mismatch of two host-to-host
steps (tools/race_check*.py)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    from inklayer_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    F16, F32 = torch.float16, torch.float32
    shapes = [(80000, 256, 192, F32), (20000, 256, 384, F32), (5000, 256, 768, F32), (106352, 256, 256, F32),
              (106352, 384, 256, F32), (106352, 256, 256, F16), (7200, 256, 256, F32), (32, 256, 1024, F32)]
    cases = []
    for M, N, K, od in shapes:
        a = torch.randn(M, K, generator=g).half().to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
        b = torch.randn(N, generator=g).to(dev)
        ref = ops.gemm(a, w, b, out_dtype=od)
        cases.append((M, N, K, od, a, w, b, ref))
    torch.cuda.synchronize()
    A = torch.randn(32768, 1280, generator=g).half().to(dev)
    W = (torch.randn(3840, 1280, generator=g) / 36).half().to(dev)
    big = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
    hostbuf = torch.empty(128 << 20, dtype=torch.uint8, pin_memory=True)
    s_bg, s_cp, s_fg = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    for mode in ("quiet", "busy: ping-pong GEMM on a second stream", "busy: GEMM + D2H copy"):
        bad = {}
        total = 0
        for rnd in range(6):
            if mode != "quiet":
                with torch.cuda.stream(s_bg):
                    for _ in range(40):
                        ops.gemm(A, W, None, out_dtype=F16)
                if "D2H" in mode:
                    with torch.cuda.stream(s_cp):
                        for _ in range(8):
                            hostbuf.copy_(big, non_blocking=True)
            with torch.cuda.stream(s_fg):
                outs = []
                for rep in range(6):
                    for ci, (M, N, K, od, a, w, b, ref) in enumerate(cases):
                        outs.append((ci, ops.gemm(a, w, b, out_dtype=od)))
            torch.cuda.synchronize()
            for ci, o in outs:
                total += 1
                ref = cases[ci][7]
                if not torch.equal(o, ref):
                    d = (o != ref).nonzero()
                    bad.setdefault(ci, []).append((int(d.shape[0]), d[:6].tolist(), o[d[0, 0], d[0, 1]].item(), ref[d[0, 0], d[0, 1]].item(),
                                                   cases[ci][6][d[0, 1]].item()))
        print(f"[{mode}] {total} launches, mismatching: { {cases[k][:3]: len(v) for k, v in bad.items()} }", flush=True)
        for k, v in bad.items():
            for n, where, got, want, bias in v[:4]:
                print(f"      shape {cases[k][:3]}: {n} elements differ, first at {where}; got {got:.6f} want {want:.6f} (bias of that column {bias:.6f})", flush=True)


if __name__ == "__main__":
    with torch.no_grad():
        main()
