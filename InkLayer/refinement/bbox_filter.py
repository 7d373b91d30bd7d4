"""Sketch-NMS plugin (reference surface: InkLayer/refinement/bbox_filter.py), pair table on the MI355X.

Entry points with the reference's names and argument meaning:
  process_json_with_sketch_NMS(sketch_path, masks_dir, input_data, iou_threshold) -> {"bboxes", "scores", "kept_indices", "threshold"}
  run_postprocess_boxes_on_sketch_dir(sketch_dir, sketch_iou_thresh) -> path of <sketch_dir>/bboxes_final.json
The cleaned masks are loaded once (or handed over in memory through `cleaned_masks`) and all pairwise sketch IoUs come
from one GPU launch, where the reference re-opens the sketch and two mask PNGs for every pair of boxes."""
import glob
import json
import os
from typing import Dict, Optional

import numpy as np
from PIL import Image, ImageDraw


def _cleaned_stack(masks_dir: str, count: int, shape) -> np.ndarray:
    if count == 0:
        return np.zeros((0,) + tuple(shape), np.uint8)
    return np.stack([np.asarray(Image.open(os.path.join(masks_dir, f"mask_{i}.png")).convert("L")) for i in range(count)])


def process_json_with_sketch_NMS(sketch_path: str, masks_dir: str, input_data: Dict, iou_threshold: float = 0.2,
                                 cleaned_masks: Optional[np.ndarray] = None, sketch_rgb: Optional[np.ndarray] = None) -> Dict:
    """sketch_rgb (optional): the decoded sketch when the caller still holds it (the runner does) - the reference re-opens
    sketch_path, which is what happens when it is None."""
    import torch
    from inklayer_amd import refine
    sketch = np.asarray(Image.open(sketch_path).convert("RGB")) if sketch_rgb is None else sketch_rgb
    if cleaned_masks is None:
        cleaned_masks = _cleaned_stack(masks_dir, len(input_data["bboxes"]), sketch.shape[:2])
    on_gpu = cleaned_masks if torch.is_tensor(cleaned_masks) else torch.from_numpy(np.ascontiguousarray(cleaned_masks)).to("cuda")
    return refine.process_json_with_sketch_nms(sketch, input_data, on_gpu, iou_threshold)


def _boxes_source(sketch_dir: str) -> str:
    """An mmdet result takes precedence over bboxes.json, as in the reference."""
    alt = glob.glob(os.path.join(sketch_dir, "mmdet_out", "*.json"))
    return alt[0] if alt else os.path.join(sketch_dir, "bboxes.json")


def run_postprocess_boxes_on_sketch_dir(sketch_dir, sketch_iou_thresh=0.5, cleaned_masks=None, sketch_rgb=None):
    if not os.path.isdir(sketch_dir):
        print(f"{sketch_dir}: no such sketch directory")
        return None
    with open(_boxes_source(sketch_dir)) as fh:
        detections = json.load(fh)
    sketch_png = os.path.join(sketch_dir, "input.png")
    kept = process_json_with_sketch_NMS(sketch_png, os.path.join(sketch_dir, "masks_cleaned"), detections,
                                        iou_threshold=sketch_iou_thresh, cleaned_masks=cleaned_masks, sketch_rgb=sketch_rgb)
    target = os.path.join(sketch_dir, "bboxes_final.json")
    with open(target, "w") as fh:
        json.dump(kept, fh, indent=4)
    # bboxes_final.png: a plain visualisation (InkLayer/utils/visualization.py is outside the hot path)
    def _visual():
        canvas = Image.open(sketch_png).convert("RGB") if sketch_rgb is None else Image.fromarray(sketch_rgb)
        pen, (W, H) = ImageDraw.Draw(canvas), canvas.size
        for x1, y1, x2, y2 in kept["bboxes"]:
            pen.rectangle([x1 * W, y1 * H, x2 * W, y2 * H], outline=(220, 40, 40), width=2)
        return canvas
    from InkLayer.utils.io import save_all
    save_all([(_visual, os.path.join(sketch_dir, "bboxes_final.png"))], wait=None)
    print(f"sketch NMS kept {len(kept['bboxes'])} of {len(detections['bboxes'])} boxes -> {target}")
    return target
