"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy emulation of Pillow's 8-bit antialiased resize passes.

Third-party algorithm, not part of /root/reference: Pillow (here 12.2.0), src/libImaging/Resample.c
`ImagingResampleHorizontal_8bpc` / `ImagingResampleVertical_8bpc` - the resampler behind the reference's
`F.resize` / `Image.resize(BILINEAR)` calls (GD/datasets/transforms.py:87-117, SA/utils/transforms.py:26-31).
Pinned: `tests/test_resize.py` checks this emulation (fed with `inklayer_amd.resize.pil_bilinear_coeffs`) against
`PIL.Image.resize` itself, bit for bit, on random images at the sizes the pipeline uses.
"""
import numpy as np


def resample_pass(img: np.ndarray, bounds: np.ndarray, coef: np.ndarray, axis: int) -> np.ndarray:
    """One pass along x (axis=0) or y (axis=1) of an HWC uint8 image with 22-bit fixed-point weights."""
    src = img.astype(np.int64)
    if axis == 1:
        src = src.transpose(1, 0, 2)
    n_out = bounds.shape[0]
    out = np.empty((src.shape[0], n_out, src.shape[2]), dtype=np.uint8)
    for o in range(n_out):
        lo, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = (1 << 21) + (src[:, lo:lo + n, :] * coef[o, :n].astype(np.int64)[None, :, None]).sum(axis=1)
        out[:, o, :] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
    return out.transpose(1, 0, 2) if axis == 1 else out


def resize_bilinear_u8(img: np.ndarray, oh: int, ow: int, coeffs) -> np.ndarray:
    """coeffs(in_size, out_size) -> (bounds, coef); horizontal pass first, u8 intermediate, then vertical."""
    h, w = img.shape[:2]
    cur = img
    if ow != w:
        cur = resample_pass(cur, *coeffs(w, ow), axis=0)
    if oh != h:
        cur = resample_pass(cur, *coeffs(h, oh), axis=1)
    return cur
