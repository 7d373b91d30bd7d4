"""Refinement plugin (reference: InkLayer/refinement/refiner.py), this build's host implementation.

`run_refinement_on_sketch_dir(sketch_dir, bboxes_path, out_base_dir=None)` with the reference's outputs:
masks_disjoint/mask_i.png, masks_final/mask_i.png, depth_map.png, segmented_sketch_final.png.  The depth map comes
from the MI355X Depth-Anything-V2 engine (InkLayer.refinement.depth_sort.get_depth_map); the mask logic is sequential
by construction and runs on the host like the reference's (inklayer_amd/refine_host.py explains why, and is pinned bit
for bit by the reference's own committed outputs)."""
import json
import os
import shutil

import numpy as np
from PIL import Image

from InkLayer.refinement.depth_sort import get_depth_map
from inklayer_amd import refine_host as _R

SKETCH_THRESHOLD = _R.SKETCH_THRESHOLD
clean_delicate_mask = _R.clean_delicate_mask
composite_and_parse_masks = _R.composite_and_parse_masks
match_masks_to_boxes = _R.match_masks_to_boxes


def _rgb(sketch_path):
    return np.asarray(Image.open(sketch_path).convert("RGB"))


def parse_masks_to_disjoint_masks(masks_np, bboxes, sketch_path, depth_map=None):
    rgb = _rgb(sketch_path)
    if depth_map is None:
        depth_map = get_depth_map(sketch_path)
    return _R.parse_masks_to_disjoint_masks(masks_np, bboxes, rgb, depth_map)


def refine_masks_with_watershed(sketch_image, original_masks, debug=False):
    return _R.refine_masks_with_watershed(np.asarray(sketch_image), original_masks)


def refine_masks_with_boxes(sketch_image_path, original_masks, boxes):
    return _R.refine_masks_with_boxes(_R.pil_luma(_rgb(sketch_image_path)), original_masks, boxes)


def create_unlabeled_mask(sketch_path, masks):
    return _R.create_unlabeled_mask(_R.png_gray(_rgb(sketch_path)), masks)


def _colour(rgb, masks):
    base = rgb.astype("float32")
    out = base.copy()
    for i, m in enumerate(masks):
        hue = (i * 0.61803398875) % 1.0
        col = 255.0 * np.array([0.6 + 0.4 * abs(((hue * 6 + k) % 6) / 3 - 1) for k in (0, 4, 2)], dtype="float32")
        sel = np.asarray(m) > 0
        out[sel] = 0.5 * base[sel] + 0.5 * col
    return Image.fromarray(out.clip(0, 255).astype("uint8"))


def improve_sam_masks(sketch_image_path, masks_np, bboxes):
    rgb = _rgb(sketch_image_path)
    final_masks = _R.improve_sam_masks(rgb, masks_np, bboxes)
    return {"initial_seg_sketch": _colour(rgb, masks_np), "final_seg_sketch": _colour(rgb, final_masks),
            "final_masks": final_masks}


def run_refinement_on_sketch_dir(sketch_dir, bboxes_path, out_base_dir=None, cleaned_masks=None):
    """`cleaned_masks` (optional, this build's extension): the cleaned masks [n, H, W] uint8 already in memory, indexed
    like masks_cleaned/mask_i.png; otherwise the files are read."""
    if not os.path.exists(sketch_dir):
        print(f"Sketch directory {sketch_dir} does not exist.")
        return
    masks_dir = f"{sketch_dir}/masks_cleaned"
    sketch_path = f"{sketch_dir}/input.png"
    rgb = _rgb(sketch_path)
    h, w = rgb.shape[:2]
    with open(bboxes_path, "r") as f:
        bboxes_data = json.load(f)
    assert len(bboxes_data["bboxes"]) == len(bboxes_data["kept_indices"])
    bboxes = _R.unnormalize_bboxes(bboxes_data["bboxes"], h, w)
    kept = bboxes_data["kept_indices"]
    if cleaned_masks is not None:
        cleaned = [np.asarray(cleaned_masks[i]) for i in kept]
    else:
        cleaned = [np.asarray(Image.open(f"{masks_dir}/mask_{i}.png").convert("L")) for i in kept]
    depth_map = get_depth_map(sketch_path)
    disjoint, sorted_bboxes, _info = _R.parse_masks_to_disjoint_masks(cleaned, bboxes, rgb, depth_map)
    out_base_dir = out_base_dir or sketch_dir
    dis_dir = f"{out_base_dir}/masks_disjoint"
    shutil.rmtree(dis_dir, ignore_errors=True)
    os.makedirs(dis_dir, exist_ok=True)
    for i, m in enumerate(disjoint):
        Image.fromarray(m.astype(np.uint8) * 255, "L").save(f"{dis_dir}/mask_{i}.png")
    res = improve_sam_masks(sketch_path, disjoint, sorted_bboxes)
    out_dir = f"{out_base_dir}/masks_final"
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir, exist_ok=True)
    for i, m in enumerate(res["final_masks"]):
        Image.fromarray((np.asarray(m) > 0).astype(np.uint8) * 255, "L").save(f"{out_dir}/mask_{i}.png")
    lo, hi = float(depth_map.min()), float(depth_map.max())             # cv2.normalize(NORM_MINMAX, 0..255)
    norm = (depth_map - lo) * (255.0 / (hi - lo)) if hi > lo else np.zeros_like(depth_map)
    Image.fromarray(np.clip(norm, 0, 255).astype(np.uint8)).convert("RGB").save(f"{out_base_dir}/depth_map.png")
    res["final_seg_sketch"].save(f"{out_base_dir}/segmented_sketch_final.png")
    print(f"Results saved to {out_dir}")
    return out_dir
