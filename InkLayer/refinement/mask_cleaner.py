"""Mask cleanup plugin (reference surface: InkLayer/refinement/mask_cleaner.py), MI355X kernels underneath.

  calculate_kernel_size(image_shape, factor) -> (k, k)
  clean_up_mask(binary_mask) -> uint8 0/255
  run_clean_masks_on_sketch_dir(sketch_dir) -> <sketch_dir>/masks_cleaned   (8-bit grayscale PNGs, as the reference)
All masks of a sketch are cleaned in ONE batched GPU call (`ink_mask_cleanup`); `clean_masks_in_memory` is the hand-off
the runner uses, so the masks are never re-read from disk."""
import glob
import os

import numpy as np
from PIL import Image


def calculate_kernel_size(image_shape, factor=0.025):
    from inklayer_amd import refine
    k = refine.calculate_kernel_size(image_shape, factor)
    return (k, k)


def clean_masks_on_device(masks):
    """A sequence of equally sized HxW masks (bool, or uint8 where > 127 is foreground) -> cleaned uint8 [n, H, W]
    (0 / 255) as a CUDA tensor that STAYS in HBM (the sketch NMS and the refinement stage consume it there); None for
    an empty sequence.  The components pass' overflow flag is read once here and raises."""
    import torch
    from inklayer_amd import _lib, refine
    if len(masks) == 0:
        return None
    stack = np.stack([np.asarray(m) for m in masks])
    stack = stack.astype(np.uint8) * 255 if stack.dtype == np.bool_ else stack.astype(np.uint8)
    out = refine.clean_masks(torch.from_numpy(np.ascontiguousarray(stack)).to("cuda"))
    if int(out._ink_overflow_flag.item()) != 0:
        raise _lib.InkLayerHipError("ink_mask_cleanup: a row has more runs than the closing bound allows (workspace overflow)")
    return out


def clean_device_masks(masks01_u8):
    """The segmentor's 0/1 uint8 masks [n, H, W], still on the GPU -> cleaned 0 / 255 CUDA tensor (overflow checked)."""
    from inklayer_amd import _lib, refine
    out = refine.clean_segmentor_masks(masks01_u8.contiguous())
    if int(out._ink_overflow_flag.item()) != 0:
        raise _lib.InkLayerHipError("ink_mask_cleanup: a row has more runs than the closing bound allows (workspace overflow)")
    return out


def clean_masks_in_memory(masks):
    """-> cleaned uint8 [n, H, W] (0 / 255) numpy, computed on the GPU."""
    dev = clean_masks_on_device(masks)
    return np.zeros((0, 0, 0), np.uint8) if dev is None else dev.cpu().numpy()


def clean_up_mask(binary_mask):
    return clean_masks_in_memory([binary_mask])[0]


def run_clean_masks_on_sketch_dir(sketch_dir, masks=None, cleaned=None):
    """`masks` / `cleaned` (this build's extension): raw or already cleaned masks held in memory; without them
    masks/mask_i.png are read."""
    src = os.path.join(sketch_dir, "masks")
    if not os.path.isdir(src):
        print(f"{src} is missing: nothing to clean")
        return None
    dst = os.path.join(sketch_dir, "masks_cleaned")
    os.makedirs(dst, exist_ok=True)
    if cleaned is None:
        if masks is None:
            count = len(glob.glob(os.path.join(src, "mask_*.png")))
            masks = [np.asarray(Image.open(os.path.join(src, f"mask_{i}.png")).convert("L")) for i in range(count)]
        cleaned = clean_masks_in_memory(masks)
    if hasattr(cleaned, "cpu"):
        cleaned = cleaned.cpu().numpy()
    from InkLayer.utils.io import save_all
    save_all(((np.ascontiguousarray(m), os.path.join(dst, f"mask_{i}.png")) for i, m in enumerate(cleaned)), wait=None)
    print(f"cleaned {len(cleaned)} masks -> {dst}")
    return dst
