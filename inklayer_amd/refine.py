"""Mask hand-off stage of InkLayer's refinement on the MI355X: mask cleanup + sketch NMS with the
masks staying resident in HBM.

Reference: runner.py:57-60,69-73 writes every SAM mask to masks/mask_i.png, mask_cleaner.py:39-54 re-reads them one by
one, cleans them with cv2 and writes masks_cleaned/, and nms_sketch.py:186-234 then re-opens the sketch and TWO mask
PNGs for every PAIR of boxes (O(n^2) file reads - the "non-optimized sketch NMS" the README warns about).  Here the
uint8 masks the segmentor produced go straight into `ink_mask_cleanup` (one launch sequence for all masks) and
`ink_mask_sketch_iou_counts` (one pair table); what stays on the host is the data-dependent greedy loop over at most a
few dozen boxes and its box-geometry tests (nms_sketch.py:270-351), restated here with the same float64 arithmetic.
Results are bit-identical to the reference's (tests/test_refine_gpu.py against oracle/refine_ref.py, which is pinned by
the reference's own committed outputs).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import ops


def calculate_kernel_size(image_shape: Sequence[int], factor: float = 0.025) -> int:
    """mask_cleaner.py:6-9."""
    k = int(min(image_shape) * factor)
    return k if k % 2 != 0 else k + 1


def clean_masks(masks_u8: torch.Tensor) -> torch.Tensor:
    """clean_up_mask (mask_cleaner.py:11-36) on [n, H, W] uint8 GPU masks (0/1, 0/255 or anything thresholded at
    > 127 ... note: the segmentor's 0/1 masks must be scaled first, see `clean_segmentor_masks`)."""
    return ops.mask_cleanup(masks_u8, calculate_kernel_size(masks_u8.shape[1:]))


def clean_segmentor_masks(masks01_u8: torch.Tensor) -> torch.Tensor:
    """The segmentor hands over 0/1 bytes; the reference cleans the 0/255 PNG pixels (cv2.threshold at 127).  The
    scaling is a byte multiply on the GPU (plumbing, like the PNG round trip it replaces)."""
    return clean_masks(masks01_u8 * 255)


# ---------------------------------------------------------------------------------------------------------------
# sketch NMS: host logic (float64, as the reference's numpy), pair table from the GPU
# ---------------------------------------------------------------------------------------------------------------
def _png_gray(rgb: np.ndarray) -> np.ndarray:
    """cv2.imread(IMREAD_GRAYSCALE) of an 8-bit RGB PNG: libpng rgb_to_gray with OpenCV's (0.299, 0.587)."""
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))      # 255 * 32768 + 16384 < 2^32
    return ((r * np.uint32(9798) + g * np.uint32(19235) + b * np.uint32(3735) + np.uint32(16384)) >> np.uint32(15)).astype(np.uint8)


def _bbox_iou(box1, box2) -> float:
    """refinement/utils.py:11-32."""
    x1, y1 = max(box1[0], box2[0]), max(box1[1], box2[1])
    x2, y2 = min(box1[2], box2[2]), min(box1[3], box2[3])
    if x2 < x1 or y2 < y1:
        return 0.0
    a1 = (box1[2] - box1[0]) * (box1[3] - box1[1])
    a2 = (box2[2] - box2[0]) * (box2[3] - box2[1])
    ai = (x2 - x1) * (y2 - y1)
    return ai / (a1 + a2 - ai)


def _share_corner(b1, b2, eps: float) -> bool:
    """nms_sketch.py:23-59."""
    for cx, cy in ((b1[0], b1[1]), (b1[0], b1[3]), (b1[2], b1[1]), (b1[2], b1[3])):
        for dx, dy in ((b2[0], b2[1]), (b2[0], b2[3]), (b2[2], b2[1]), (b2[2], b2[3])):
            if ((cx - dx) ** 2 + (cy - dy) ** 2) ** 0.5 <= eps:
                return True
    return False


def _filter_full_or_empty(gray: np.ndarray, bboxes: np.ndarray, size_threshold=0.9, max_contained=5) -> np.ndarray:
    """filter_full_or_empty_bbox (nms_sketch.py:126-174)."""
    h, w = gray.shape
    if np.max(bboxes) <= 1.0:
        bboxes = (bboxes * np.array([w, h, w, h])).astype(int)
    kept = []
    for i, box in enumerate(bboxes):
        ok_area = (box[2] - box[0]) * (box[3] - box[1]) / (h * w) < size_threshold
        x0, y0, x1, y1 = (int(max(0, min(v, lim - 1))) for v, lim in zip(box, (w, h, w, h)))
        ok_content = np.count_nonzero(gray[y0:y1 + 1, x0:x1 + 1]) > 0
        contained = sum(1 for o in bboxes if not np.array_equal(box, o)
                        and box[0] <= o[0] and box[1] <= o[1] and box[2] >= o[2] and box[3] >= o[3])
        if ok_area and ok_content and contained <= max_contained:
            kept.append(i)
    return np.array(kept)


def sketch_nms(sketch_rgb: np.ndarray, bboxes: np.ndarray, scores: np.ndarray, cleaned_masks: torch.Tensor,
               sketch_iou_threshold: float, bbox_iou_threshold: float = 0.7) -> np.ndarray:
    """sketch_nms (nms_sketch.py:270-351) with the pair table from `ink_mask_sketch_iou_counts`.
    sketch_rgb: HxWx3 uint8 (host); cleaned_masks: [n, H, W] uint8 on the GPU (output of clean_masks).
    Returns the kept ORIGINAL box indices in the reference's order."""
    bboxes, scores = np.asarray(bboxes, dtype=np.float64).reshape(-1, 4), np.asarray(scores, dtype=np.float64)
    if len(bboxes) == 0:
        return np.array([])
    rgb_dev = torch.from_numpy(np.ascontiguousarray(sketch_rgb)).to(cleaned_masks.device)
    counts = ops.mask_sketch_iou_counts(cleaned_masks, rgb_dev).cpu().numpy().astype(np.int64)   # one D2H: n*n*2 ints
    gray = _png_gray(sketch_rgb)
    h, w = gray.shape
    kept_idx = _filter_full_or_empty(gray, bboxes)
    if len(kept_idx) == 0:
        return np.array([])
    fb, fs = bboxes[kept_idx], scores[kept_idx]
    order = np.argsort(-fs)
    original = kept_idx[order]
    n = len(fb)
    keep = np.ones(n, dtype=bool)
    eps = 8.0 * (np.sqrt(w ** 2 + h ** 2) / 1000)                     # get_dynamic_threshold (nms_sketch.py:7-20)

    def pair(i: int, j: int) -> Tuple[float, float, int]:
        """content_iou (nms_sketch.py:186-251); i, j are FILTERED indices, and - as in the reference - they also
        select the mask files (mask_{filtered index}.png)."""
        b1, b2 = fb[i].astype(float), fb[j].astype(float)
        if np.all(b1 <= 1.0) and np.all(b2 <= 1.0):
            b1, b2 = b1 * np.array([w, h, w, h]), b2 * np.array([w, h, w, h])
        a1, a2 = (b1[2] - b1[0]) * (b1[3] - b1[1]), (b2[2] - b2[0]) * (b2[3] - b2[1])
        if a1 > a2:
            big, small, bi, si, bs, ss = b1, b2, i, j, fs[i], fs[j]
        else:
            big, small, bi, si, bs, ss = b2, b1, j, i, fs[j], fs[i]
        inter, union = counts[bi, si]
        s_iou = inter / union if union > 0 else 0.0
        contained = (small[0] >= big[0] - eps and small[1] >= big[1] - eps
                     and small[2] <= big[2] + eps and small[3] <= big[3] + eps)
        if not contained or not _share_corner(small, big, eps):
            return 0.0, 0.0, bi
        return s_iou, _bbox_iou(small, big), (bi if bs > ss else si)

    for i in range(n):
        if not keep[i]:
            continue
        remaining = order[i + 1:]
        if len(remaining) == 0:
            continue
        terms = [pair(order[i], r) for r in remaining]
        s_iou = np.array([t[0] for t in terms])
        b_iou = np.array([t[1] for t in terms])
        larger = np.array([t[2] for t in terms])
        for ov in np.where(np.logical_or(s_iou > sketch_iou_threshold, b_iou > bbox_iou_threshold))[0]:
            compared = remaining[ov]
            if larger[ov] == compared:
                keep[i] = False
                break
            keep[np.where(order == compared)[0][0]] = False
    return original[keep]


def process_json_with_sketch_nms(sketch_rgb: np.ndarray, input_data: Dict, cleaned_masks: torch.Tensor,
                                 iou_threshold: float = 0.2) -> Dict:
    """process_json_with_sketch_NMS (bbox_filter.py:12-36)."""
    keep = sketch_nms(sketch_rgb, np.array(input_data["bboxes"]), np.array(input_data["scores"]), cleaned_masks,
                      iou_threshold)
    return {"bboxes": [input_data["bboxes"][i] for i in keep], "scores": [input_data["scores"][i] for i in keep],
            "kept_indices": [int(i) for i in keep], "threshold": iou_threshold}
