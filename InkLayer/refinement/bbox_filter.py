"""Sketch-NMS plugin (reference: InkLayer/refinement/bbox_filter.py), pair table on the MI355X.

`process_json_with_sketch_NMS(sketch_path, masks_dir, input_data, iou_threshold)` and
`run_postprocess_boxes_on_sketch_dir(sketch_dir, sketch_iou_thresh)` as in the reference; the cleaned masks are read
ONCE (or handed over in memory) instead of twice per pair of boxes (nms_sketch.py:214-224)."""
import glob
import json
import os
from typing import Dict

import numpy as np
from PIL import Image, ImageDraw


def _load_cleaned(masks_dir, n):
    return np.stack([np.asarray(Image.open(f"{masks_dir}/mask_{i}.png").convert("L")) for i in range(n)])


def process_json_with_sketch_NMS(sketch_path: str, masks_dir: str, input_data: Dict, iou_threshold: float = 0.2,
                                 cleaned_masks=None) -> Dict:
    import torch
    from inklayer_amd import refine
    rgb = np.asarray(Image.open(sketch_path).convert("RGB"))
    n = len(input_data["bboxes"])
    if cleaned_masks is None:
        cleaned_masks = _load_cleaned(masks_dir, n) if n else np.zeros((0,) + rgb.shape[:2], np.uint8)
    dev = torch.from_numpy(np.ascontiguousarray(cleaned_masks)).to("cuda")
    return refine.process_json_with_sketch_nms(rgb, input_data, dev, iou_threshold)


def run_postprocess_boxes_on_sketch_dir(sketch_dir, sketch_iou_thresh=0.5, cleaned_masks=None):
    if not os.path.exists(sketch_dir):
        print("no sketch dir")
        return
    mmdet_json = glob.glob(f"{sketch_dir}/mmdet_out/*.json")
    json_path = mmdet_json[0] if len(mmdet_json) > 0 else f"{sketch_dir}/bboxes.json"
    with open(json_path, "r") as f:
        input_data = json.load(f)
    filtered_data = process_json_with_sketch_NMS(sketch_path=f"{sketch_dir}/input.png",
                                                 masks_dir=f"{sketch_dir}/masks_cleaned", input_data=input_data,
                                                 iou_threshold=sketch_iou_thresh, cleaned_masks=cleaned_masks)
    print(f"Got filtered data with {len(filtered_data['bboxes'])} boxes")
    out_path = f"{sketch_dir}/bboxes_final.json"
    with open(out_path, "w") as f:
        json.dump(filtered_data, f, indent=4)
    print(f"Output saved to {out_path}")
    # bboxes_final.png (visualisation only; InkLayer/utils/visualization.py is outside the hot path)
    im = Image.open(f"{sketch_dir}/input.png").convert("RGB")
    d = ImageDraw.Draw(im)
    W, H = im.size
    for b in filtered_data["bboxes"]:
        d.rectangle([b[0] * W, b[1] * H, b[2] * W, b[3] * H], outline=(220, 40, 40), width=2)
    im.save(f"{sketch_dir}/bboxes_final.png")
    return out_path
