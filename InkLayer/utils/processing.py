"""Box-format glue between the detector and the segmentor (reference: InkLayer/utils/processing.py).

Same functions, same numerics (the float64 -> float32 round trip and the in-place per-row updates of
process_boxes_ours matter for the exact pixel boxes SAM is prompted with)."""
import json

import numpy as np
import torch


def cxcywh_to_xyxy(boxes):
    """processing.py:56-63.  Deviation: an empty list returns an empty (0, 4) array instead of the
    reference's IndexError (documented reference bug)."""
    b = np.array(boxes, dtype=np.float64).reshape(-1, 4)
    half_w, half_h = b[:, 2] / 2, b[:, 3] / 2
    return np.stack([b[:, 0] - half_w, b[:, 1] - half_h, b[:, 0] + half_w, b[:, 1] + half_h], axis=-1)


def process_boxes_ours(out_dict, input_pil):
    """processing.py:6-28: normalised xyxy -> cxcywh (python floats) -> float32 -> * (W,H,W,H) -> xyxy."""
    norm_boxes = out_dict["bboxes"] if "bboxes" in out_dict else out_dict
    rows = []
    for x1, y1, x2, y2 in norm_boxes:
        w, h = x2 - x1, y2 - y1
        rows.append([x1 + w / 2, y1 + h / 2, w, h])
    W, H = input_pil.size
    boxes = torch.tensor(rows).float().reshape(-1, 4)
    scale = torch.Tensor([W, H, W, H])
    for i in range(boxes.size(0)):
        boxes[i] = boxes[i] * scale
        boxes[i][:2] -= boxes[i][2:] / 2
        boxes[i][2:] += boxes[i][:2]
    return boxes


def process_dino_output(out_dict, input_pil):
    """processing.py:30-32."""
    return process_boxes_ours(out_dict, input_pil), out_dict["labels"]


def save_norm_bboxes(bboxes_list, scores_list, input_pil, out_path, labels=None):
    """processing.py:35-53: pixel boxes -> normalised, JSON with indent 4."""
    W, H = input_pil.size
    obj = {"bboxes": [[b[0] / W, b[1] / H, b[2] / W, b[3] / H] for b in bboxes_list], "scores": scores_list}
    if labels is not None:
        obj["labels"] = labels
    with open(out_path, "w") as fh:
        json.dump(obj, fh, indent=4)
