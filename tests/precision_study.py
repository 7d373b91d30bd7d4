"""CPU study: which f16 rounding points of the SAM path cost mask IoU?  (tool, not product)

Runs the fp32 oracle (oracle/sam_ref.py) with emulated f16 rounding of chosen GEMM operands — a proxy for
`torch.nn.functional` inside the oracle rounds the activation / weight / output of every linear or conv whose
weight name matches a policy — and reports the relative error of the mask logits and the mask IoU against the
un-rounded run.  Split-f16 (hi + lo) operands are emulated as "no rounding" (their error is ~2^-22).

    python tests/precision_study.py --depth 4            # quick
    python tests/precision_study.py --depth 32           # full ViT-H, ~30 s per policy on 8 cores
"""
import argparse
import re
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as RealF

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import sam_ref  # noqa: E402
from inklayer_amd import synthetic  # noqa: E402


def r16(x):
    return x.half().float()


class QuantF:
    """Stand-in for torch.nn.functional inside sam_ref: rounds operands of the layers a policy names."""

    def __init__(self, names, policy):
        self.names, self.policy = names, policy      # id(weight) -> key ; list of (regex, "awo" flags)

    def __getattr__(self, k):
        return getattr(RealF, k)

    def _flags(self, w):
        key = self.names.get(id(w), "?")
        for rx, fl in self.policy:
            if re.search(rx, key):
                return fl
        return ""

    def _wrap(self, fn, x, w, *a, **kw):
        fl = self._flags(w)
        if ("c" in fl or "z" in fl) and fn is RealF.linear:
            # channel-centred operands (round-3 study): x = cm + s with cm the per-channel mean over the tokens; only s is
            # rounded ("c"), and the mean's product with the weights is exact ("z": cm W^T in fp32 instead of cm f16(W)^T)
            cm = x.reshape(-1, x.shape[-1]).mean(0)
            s_ = r16(x - cm) if ("a" in fl and "c" in fl) else (r16(x) - cm if "a" in fl else x - cm)
            w16 = r16(w) if "w" in fl else w
            y = fn(s_, w16, *a, **kw) + RealF.linear(cm, w if "z" in fl else w16)
            return r16(y) if "o" in fl else y
        if "a" in fl:
            x = r16(x)
        if "w" in fl:
            w = r16(w)
        y = fn(x, w, *a, **kw)
        return r16(y) if "o" in fl else y

    def linear(self, x, w, b=None):
        return self._wrap(RealF.linear, x, w, b)

    def conv2d(self, x, w, b=None, **kw):
        return self._wrap(RealF.conv2d, x, w, b, **kw)

    def conv_transpose2d(self, x, w, b=None, **kw):
        return self._wrap(RealF.conv_transpose2d, x, w, b, **kw)


POLICIES = {
    "all-f16 (round-1 design)": [(r"qkv|lin1", "awo"), (r".", "aw")],
    "encoder blocks only": [(r"blocks.*(qkv|lin1)", "awo"), (r"blocks|patch_embed", "aw")],
    "blocks (no patch embed)": [(r"blocks.*(qkv|lin1)", "awo"), (r"blocks", "aw")],
    "blocks, centred activations": [(r"blocks.*(qkv|lin1)", "awoc"), (r"blocks", "awc")],
    "blocks, centred + exact mean product": [(r"blocks.*(qkv|lin1)", "awocz"), (r"blocks", "awcz")],
    "blocks, exact mean product only": [(r"blocks.*(qkv|lin1)", "awoz"), (r"blocks", "awz")],
    "encoder: weights only": [(r"blocks|patch_embed", "w")],
    "encoder: activations only": [(r"blocks.*(qkv|lin1)", "ao"), (r"blocks|patch_embed", "a")],
    "encoder: qkv": [(r"blocks.*qkv", "awo")],
    "encoder: proj": [(r"blocks.*proj", "aw")],
    "encoder: lin1": [(r"blocks.*lin1", "awo")],
    "encoder: lin2": [(r"blocks.*lin2", "aw")],
    "neck only": [(r"neck", "aw")],
    "decoder transformer only": [(r"mask_decoder.transformer", "aw")],
    "upscaler + hyper only": [(r"output_upscaling|hypernetworks", "aw")],
    "neck + decoder + upscaler": [(r"neck|mask_decoder", "aw")],
}


@torch.no_grad()
def run(sd, cfg, img, boxes, policy):
    names = {id(v): k for k, v in sd.items()}
    sam_ref.F = QuantF(names, policy) if policy is not None else RealF
    try:
        logits, low, _ = sam_ref.run_sam(sd, cfg, img, boxes, return_logits=True)
    finally:
        sam_ref.F = RealF
    return logits[:, 0], low[:, 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    glob = tuple(i for i in (7, 15, 23, 31) if i < args.depth) if args.depth >= 8 else (1, 3)[: max(1, args.depth // 2)]
    cfg = sam_ref.SamConfig(depth=args.depth, global_attn_indexes=glob)
    sd = sam_ref.seeded_state_dict(sam_ref.sam_param_shapes(cfg), args.seed)
    img = synthetic.synthetic_sketch(0)
    boxes = torch.tensor([[100.0, 80.0, 700.0, 600.0], [300.0, 300.0, 900.0, 760.0], [20.0, 500.0, 400.0, 1000.0],
                          [600.0, 50.0, 1000.0, 400.0]])
    t0 = time.time()
    ref, ref_low = run(sd, cfg, img, boxes, None)
    print(f"depth {args.depth} global {glob}: fp32 run {time.time() - t0:.1f} s; logit std {ref.std():.3f}, "
          f"mask occupancy {[(m > 0).float().mean().item() for m in ref]}", flush=True)
    for name, pol in POLICIES.items():
        if args.only and args.only not in name:
            continue
        got, low = run(sd, cfg, img, boxes, pol)
        l2 = ((low - ref_low).norm() / ref_low.norm()).item()
        a, b = got > 0, ref > 0
        iou = ((a & b).flatten(1).sum(1).double() / (a | b).flatten(1).sum(1).double()).tolist()
        print(f"{name:34s} low-res logits l2-rel {l2:.2e}   IoU min {min(iou):.5f}  {['%.5f' % i for i in iou]}", flush=True)


if __name__ == "__main__":
    main()
