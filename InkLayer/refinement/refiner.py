"""Refinement plugin (reference: InkLayer/refinement/refiner.py), MI355X kernels underneath.

`run_refinement_on_sketch_dir(sketch_dir, bboxes_path, out_base_dir=None)` with the reference's outputs:
masks_disjoint/mask_i.png, masks_final/mask_i.png, depth_map.png, segmented_sketch_final.png.  The cleaned masks, the
sketch and the Depth-Anything map stay in HBM; disjoint parsing, growth and the unlabeled mask are kernels over bit
planes / one label image (inklayer_amd/refine_stage.py), the two pixel-sequential algorithms of the stage (stroke
thinning, raster-order box assignment) are C++ on the host."""
import json
import os
import shutil

import numpy as np
from PIL import Image

from InkLayer.refinement.depth_sort import get_depth_map, get_depth_map_device

SKETCH_THRESHOLD = 250


def _rgb(sketch_path):
    return np.asarray(Image.open(sketch_path).convert("RGB"))


def _stack_on_gpu(masks, shape):
    import torch
    if len(masks) == 0:
        return torch.zeros((0,) + tuple(shape), dtype=torch.uint8, device="cuda")
    a = np.stack([(np.asarray(m) > 0) for m in masks]).astype(np.uint8)
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def parse_masks_to_disjoint_masks(masks_np, bboxes, sketch_path, depth_map=None):
    """refiner.py:91-126 -> (disjoint bool masks in depth order, boxes in depth order, mask info)."""
    import torch
    from inklayer_amd import refine_stage
    rgb = _rgb(sketch_path)
    depth = get_depth_map_device(sketch_path) if depth_map is None else \
        torch.from_numpy(np.ascontiguousarray(depth_map, np.float32)).to("cuda")
    res = refine_stage.refine_masks(_stack_on_gpu(masks_np, rgb.shape[:2]), [list(b) for b in bboxes], rgb,
                                    depth.contiguous(), stop_after_disjoint=True)
    return res.disjoint_masks(), res.sorted_boxes, res.info


def match_masks_to_boxes(masks, boxes):
    from inklayer_amd import refine_stage

    def bb(m):
        ys, xs = np.nonzero(np.asarray(m).any(1))[0], np.nonzero(np.asarray(m).any(0))[0]
        return None if len(ys) == 0 else [int(xs[0]), int(ys[0]), int(xs[-1]), int(ys[-1])]
    return refine_stage.match_boxes_to_masks(boxes, [bb(m) for m in masks])


def _colour(rgb, masks):
    from InkLayer.runner import colour_by_masks
    return Image.fromarray(colour_by_masks(rgb, masks))


def improve_sam_masks(sketch_image_path, masks_np, bboxes):
    """refiner.py:340-372 for DISJOINT masks (what parse_masks_to_disjoint_masks returns), without the intermediate
    visualisations."""
    import torch
    from inklayer_amd import _lib, ops, refine_stage
    rgb = _rgb(sketch_image_path)
    H, W = rgb.shape[:2]
    label = np.zeros((H, W), np.uint8)
    for i, m in enumerate(masks_np):
        label[np.asarray(m) > 0] = i + 1
    dev = torch.device("cuda")
    sk = torch.empty((4, H, (W + 63) // 64), device=dev, dtype=torch.int64)
    ws = torch.zeros(1, device=dev, dtype=torch.int32)
    _lib.check(_lib.lib().ink_refine_sketch_planes(torch.from_numpy(np.ascontiguousarray(rgb)).to(dev).data_ptr(), H, W,
                                                   sk.data_ptr(), ws.data_ptr(), ops._stream()), "ink_refine_sketch_planes")
    final, extra = refine_stage.grow_and_assign(torch.from_numpy(label).to(dev), len(masks_np), [list(b) for b in bboxes], sk)
    final_masks = [final == l for l in range(1, len(masks_np) + 1)]
    if extra is not None:
        final_masks.append(extra.astype(np.uint8))
    return {"initial_seg_sketch": _colour(rgb, masks_np), "final_seg_sketch": _colour(rgb, final_masks),
            "final_masks": final_masks}


def run_refinement_on_sketch_dir(sketch_dir, bboxes_path, out_base_dir=None, cleaned_masks=None, sketch_rgb=None):
    """`cleaned_masks` (optional, this build's extension): the cleaned masks [n, H, W] uint8 already in memory (numpy or
    a CUDA tensor), indexed like masks_cleaned/mask_i.png; otherwise the files are read.  `sketch_rgb` (optional): the
    decoded input.png when the caller still holds it; otherwise the file is read (twice, as in the reference)."""
    import torch
    from inklayer_amd import refine_stage
    if not os.path.exists(sketch_dir):
        print(f"Sketch directory {sketch_dir} does not exist.")
        return
    masks_dir = f"{sketch_dir}/masks_cleaned"
    sketch_path = f"{sketch_dir}/input.png"
    rgb = _rgb(sketch_path) if sketch_rgb is None else sketch_rgb
    h, w = rgb.shape[:2]
    with open(bboxes_path, "r") as f:
        bboxes_data = json.load(f)
    assert len(bboxes_data["bboxes"]) == len(bboxes_data["kept_indices"])
    bboxes = refine_stage.to_pixel_boxes(bboxes_data["bboxes"], h, w)
    kept = bboxes_data["kept_indices"]
    if cleaned_masks is None:
        on_gpu = _stack_on_gpu([np.asarray(Image.open(f"{masks_dir}/mask_{i}.png").convert("L")) for i in kept], (h, w))
    elif torch.is_tensor(cleaned_masks):
        on_gpu = cleaned_masks[torch.as_tensor(kept, dtype=torch.long, device=cleaned_masks.device)].contiguous() \
            if len(kept) else cleaned_masks[:0]
    else:
        on_gpu = _stack_on_gpu([np.asarray(cleaned_masks[i]) for i in kept], (h, w))
    depth_dev = get_depth_map_device(sketch_path, sketch_rgb=sketch_rgb).contiguous()
    res = refine_stage.refine_masks(on_gpu, bboxes, rgb, depth_dev)
    out_base_dir = out_base_dir or sketch_dir
    dis_dir = f"{out_base_dir}/masks_disjoint"
    shutil.rmtree(dis_dir, ignore_errors=True)
    os.makedirs(dis_dir, exist_ok=True)
    from InkLayer.utils.io import save_all
    save_all(((m.astype(np.uint8) * 255, f"{dis_dir}/mask_{i}.png") for i, m in enumerate(res.disjoint_masks())), wait=None)
    out_dir = f"{out_base_dir}/masks_final"
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir, exist_ok=True)
    final_masks = res.final_masks()
    save_all((((np.asarray(m) > 0).astype(np.uint8) * 255, f"{out_dir}/mask_{i}.png") for i, m in enumerate(final_masks)), wait=None)
    depth_map = depth_dev.cpu().numpy()
    lo, hi = float(depth_map.min()), float(depth_map.max())             # cv2.normalize(NORM_MINMAX, 0..255)
    norm = (depth_map - lo) * (255.0 / (hi - lo)) if hi > lo else np.zeros_like(depth_map)
    from InkLayer.runner import colour_by_masks
    save_all([(lambda: np.repeat(np.clip(norm, 0, 255).astype(np.uint8)[..., None], 3, axis=2), f"{out_base_dir}/depth_map.png"),
              (lambda: colour_by_masks(rgb, final_masks), f"{out_base_dir}/segmented_sketch_final.png")], wait=None)
    print(f"Results saved to {out_dir}")
    return out_dir
