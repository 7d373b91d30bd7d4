"""Pillow-exact bilinear resize on the GPU: host side (coefficient tables, plan cache).

The reference resizes with PIL (torchvision `F.resize` on a PIL image -> `Image.resize(BILINEAR)`), once for the
detector (GD/datasets/transforms.py:87-117) and once for SAM (SA/utils/transforms.py:26-31).  Pillow's 8-bit
resampler is integer arithmetic on 22-bit fixed-point weights; the weights and sample bounds are computed here
exactly as `precompute_coeffs` / `normalize_coeffs_8bpc` (Pillow 12.2 src/libImaging/Resample.c) do, in the same
double-precision operation order, and the two passes run in `ink_resize_bilinear_u8`.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

from ._lru import LRU

PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """(bounds int32 [out, 2] = (xmin, count), coef int32 [out, ksize]) of one axis."""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale                      # bilinear: support 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)           # (int): truncation toward zero, as in C
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ws = []
        ww = 0.0
        for x in range(xmax):
            a = ((x + xmin) - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w = 1.0 - a if a < 1.0 else 0.0
            ws.append(w)
            ww += w
        for x in range(xmax):
            k = ws[x] / ww if ww != 0.0 else ws[x]
            coef[xx, x] = int(-0.5 + k * one) if k < 0 else int(0.5 + k * one)
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
    return bounds, coef


class ResizePlan:
    """Device-resident tables for one (h, w) -> (oh, ow) resize."""

    def __init__(self, h: int, w: int, oh: int, ow: int, device):
        self.h, self.w, self.oh, self.ow = h, w, oh, ow
        dev = torch.device(device)
        self.xb = self.xk = self.yb = self.yk = None
        self.kx = self.ky = 0
        if ow != w:
            b, k = pil_bilinear_coeffs(w, ow)
            self.xb, self.xk, self.kx = torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), k.shape[1]
        if oh != h:
            b, k = pil_bilinear_coeffs(h, oh)
            self.yb, self.yk, self.ky = torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), k.shape[1]
        self.tmp = torch.empty((h, ow, 3), device=dev, dtype=torch.uint8) if (ow != w and oh != h) else None


_PLANS = LRU(32)      # (source size, target size) pairs seen recently; a plan is a few hundred KB of tables


def plan_for(h: int, w: int, oh: int, ow: int, device) -> ResizePlan:
    key = (h, w, oh, ow, str(device))
    return _PLANS.get_or_make(key, lambda: ResizePlan(h, w, oh, ow, device))
