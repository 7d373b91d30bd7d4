"""Mask cleanup plugin (reference: InkLayer/refinement/mask_cleaner.py), MI355X kernels underneath.

Same entry points: `calculate_kernel_size`, `clean_up_mask(binary_mask) -> uint8 0/255`,
`run_clean_masks_on_sketch_dir(sketch_dir) -> <sketch_dir>/masks_cleaned`, same files written (8-bit grayscale PNGs).
All masks of a sketch are cleaned in ONE batched GPU call; `clean_masks_in_memory` is the hand-off the runner uses so
that the masks never have to be re-read from disk."""
import glob
import os

import numpy as np
from PIL import Image


def calculate_kernel_size(image_shape, factor=0.025):
    kernel_size = int(min(image_shape) * factor)
    kernel_size = kernel_size if kernel_size % 2 != 0 else kernel_size + 1
    return (kernel_size, kernel_size)


def clean_masks_in_memory(masks):
    """masks: sequence of HxW arrays (bool, or uint8 with > 127 = foreground) of ONE size -> uint8 [n, H, W] 0/255
    numpy array (cleaned), computed on the GPU."""
    import torch
    from inklayer_amd import refine
    if len(masks) == 0:
        return np.zeros((0, 0, 0), np.uint8)
    arr = np.stack([np.asarray(m) for m in masks])
    arr = arr.astype(np.uint8) * 255 if arr.dtype == np.bool_ else arr.astype(np.uint8)
    dev = torch.from_numpy(np.ascontiguousarray(arr)).to("cuda")
    return refine.clean_masks(dev).cpu().numpy()


def clean_up_mask(binary_mask):
    return clean_masks_in_memory([binary_mask])[0]


def run_clean_masks_on_sketch_dir(sketch_dir, masks=None, cleaned=None):
    """`masks` / `cleaned` (optional, this build's extension): the raw or already cleaned masks in memory; otherwise
    masks/mask_i.png are read."""
    sam_masks_dir = f"{sketch_dir}/masks"
    if not os.path.exists(sam_masks_dir):
        print(f"Skipping {sam_masks_dir}")
        return
    num_masks = len(glob.glob(f"{sam_masks_dir}/mask_*.png"))
    out_dir = f"{sketch_dir}/masks_cleaned"
    os.makedirs(out_dir, exist_ok=True)
    if cleaned is None:
        if masks is None:
            masks = [np.asarray(Image.open(f"{sam_masks_dir}/mask_{i}.png").convert("L")) for i in range(num_masks)]
        cleaned = clean_masks_in_memory(masks)
    for i in range(len(cleaned)):
        Image.fromarray(cleaned[i], "L").save(f"{out_dir}/mask_{i}.png")
    print(f"Processed {num_masks} masks in {sam_masks_dir}")
    return out_dir
