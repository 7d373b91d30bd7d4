"""CPU restatement of the mask hand-off stage of InkLayer's refinement (TEST INFRASTRUCTURE ONLY): mask cleanup and
sketch NMS.

The reference does this with cv2 (not importable here or on the GPU box): InkLayer/refinement/mask_cleaner.py:6-36,
InkLayer/refinement/nms_sketch.py:7-351, InkLayer/refinement/bbox_filter.py:12-36, refinement/utils.py:11-32.
This file restates it with numpy / scipy.ndimage / PIL and is PINNED bit-exactly by the reference's own committed
outputs (tests/golden/refine_*.npz = its masks/ -> masks_cleaned/ -> bboxes_final.json sets, tests/test_oracle_refine.py).

Third-party algorithms restated (OpenCV 4.x, absent from /root/reference):
  * cv2.threshold(m, 127, 255, THRESH_BINARY): m > 127.
  * cv2.morphologyEx(MORPH_CLOSE, rect k x k, anchor = centre, default border): dilate then erode, pixels outside the
    image never win (borderValue = -inf for the max, +inf for the min).
  * cv2.connectedComponentsWithStats(connectivity=8): 8-connected components with area and bounding-box width/height.
  * cv2.imread(path, IMREAD_GRAYSCALE) of an 8-bit RGB PNG: libpng's rgb_to_gray with OpenCV's coefficients
    (0.299, 0.587): gray = (r*9798 + g*19235 + b*3735 + 16384) >> 15.  Only its SHAPE and "is any pixel of this
    region non-zero" are used on this path.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


def calculate_kernel_size(image_shape: Sequence[int], factor: float = 0.025) -> int:
    """mask_cleaner.py:6-9 (the kernel is square)."""
    k = int(min(image_shape) * factor)
    return k if k % 2 != 0 else k + 1


def morph_close_rect(b: np.ndarray, k: int) -> np.ndarray:
    """cv2.morphologyEx(b, MORPH_CLOSE, ones(k, k)) on a bool image (k odd, anchor at the centre)."""
    from scipy import ndimage
    st = np.ones((k, k), bool)
    dil = ndimage.binary_dilation(b, structure=st, border_value=0)
    return ndimage.binary_erosion(dil, structure=st, border_value=1)


def clean_up_mask(mask_u8: np.ndarray, size_threshold: int = 500, aspect_ratio_threshold: float = 1.1) -> np.ndarray:
    """mask_cleaner.py:11-36: threshold, close, keep components with area > 500 OR bbox aspect ratio > 1.1.
    uint8 0/255 in, uint8 0/255 out."""
    from scipy import ndimage
    b = mask_u8 > 127
    closed = morph_close_rect(b, calculate_kernel_size(b.shape))
    labels, n = ndimage.label(closed, structure=np.ones((3, 3), bool))
    out = np.zeros(b.shape, np.uint8)
    if n == 0:
        return out
    areas = np.bincount(labels.ravel(), minlength=n + 1)
    for i, sl in enumerate(ndimage.find_objects(labels), start=1):
        height, width = sl[0].stop - sl[0].start, sl[1].stop - sl[1].start
        aspect = max(width, height) / (min(width, height) + 1e-5)
        if areas[i] > size_threshold or aspect > aspect_ratio_threshold:
            out[labels == i] = 255
    return out


def png_gray(rgb: np.ndarray) -> np.ndarray:
    """cv2.imread(IMREAD_GRAYSCALE) of an 8-bit RGB PNG (see the header)."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 9798 + g * 19235 + b * 3735 + 16384) >> 15).astype(np.uint8)


def pil_luma(rgb: np.ndarray) -> np.ndarray:
    """PIL Image.convert("L") of an RGB image (ITU-R 601-2, Pillow's fixed point)."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def sketch_pixels(rgb: np.ndarray) -> np.ndarray:
    """refine_mask_to_sketch_regions (nms_sketch.py:62-78): the stroke pixels, luma < 250 (input and masks have the
    same size on this path, so the BILINEAR resize to the mask size is the identity)."""
    return pil_luma(rgb) < 250


def compute_bbox_iou(box1, box2) -> float:
    """refinement/utils.py:11-32."""
    x1_i, y1_i = max(box1[0], box2[0]), max(box1[1], box2[1])
    x2_i, y2_i = min(box1[2], box2[2]), min(box1[3], box2[3])
    area_1 = (box1[2] - box1[0]) * (box1[3] - box1[1])
    area_2 = (box2[2] - box2[0]) * (box2[3] - box2[1])
    if x2_i < x1_i or y2_i < y1_i:
        return 0.0
    area_i = (x2_i - x1_i) * (y2_i - y1_i)
    return area_i / (area_1 + area_2 - area_i)


def share_corner(box1, box2, epsilon: float) -> bool:
    """nms_sketch.py:23-59."""
    c1 = [(box1[0], box1[1]), (box1[0], box1[3]), (box1[2], box1[1]), (box1[2], box1[3])]
    c2 = [(box2[0], box2[1]), (box2[0], box2[3]), (box2[2], box2[1]), (box2[2], box2[3])]
    return any(((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2) ** 0.5 <= epsilon for a in c1 for b in c2)


def is_contained_bbox(small, big, epsilon: float) -> bool:
    """nms_sketch.py:177-183."""
    return (small[0] >= big[0] - epsilon and small[1] >= big[1] - epsilon
            and small[2] <= big[2] + epsilon and small[3] <= big[3] + epsilon)


def filter_full_or_empty_bbox(gray: np.ndarray, bboxes: np.ndarray, size_threshold: float = 0.9,
                              max_contained_boxes: int = 5) -> np.ndarray:
    """nms_sketch.py:126-174."""
    h, w = gray.shape
    img_area = h * w
    if np.max(bboxes) <= 1.0:
        bboxes = (bboxes * np.array([w, h, w, h])).astype(int)
    kept = []
    for i, box in enumerate(bboxes):
        valid_area = (box[2] - box[0]) * (box[3] - box[1]) / img_area < size_threshold
        x_min, y_min, x_max, y_max = (int(max(0, min(v, lim - 1))) for v, lim in zip(box, (w, h, w, h)))
        valid_content = np.count_nonzero(gray[y_min:y_max + 1, x_min:x_max + 1]) > 0
        contained = 0
        for other in bboxes:
            if np.array_equal(box, other):
                continue
            if box[0] <= other[0] and box[1] <= other[1] and box[2] >= other[2] and box[3] >= other[3]:
                contained += 1
        if valid_area and valid_content and contained <= max_contained_boxes:
            kept.append(i)
    return np.array(kept)


def pair_terms(hw: Tuple[int, int], bboxes: np.ndarray, scores: np.ndarray, i: int, j: int,
               sketch_iou_of) -> Tuple[float, float, int]:
    """content_iou (nms_sketch.py:186-251) for the pair (i, j); `sketch_iou_of(a, b)` returns
    |A & B & S| / |(A | B) & S| of the cleaned masks a, b restricted to the stroke pixels S."""
    h, w = hw
    box1, box2 = bboxes[i].astype(float), bboxes[j].astype(float)
    if np.all(box1 <= 1.0) and np.all(box2 <= 1.0):
        box1 = box1 * np.array([w, h, w, h])
        box2 = box2 * np.array([w, h, w, h])
    area1 = (box1[2] - box1[0]) * (box1[3] - box1[1])
    area2 = (box2[2] - box2[0]) * (box2[3] - box2[1])
    if area1 > area2:
        larger_box, smaller_box, larger_index, smaller_index = box1, box2, i, j
        larger_score, smaller_score = scores[i], scores[j]
    else:
        larger_box, smaller_box, larger_index, smaller_index = box2, box1, j, i
        larger_score, smaller_score = scores[j], scores[i]
    sketch_iou = sketch_iou_of(larger_index, smaller_index)
    eps = 8.0 * (np.sqrt(w ** 2 + h ** 2) / 1000)            # get_dynamic_threshold (nms_sketch.py:7-20)
    contained = is_contained_bbox(smaller_box, larger_box, eps)
    corner = share_corner(smaller_box, larger_box, eps)
    bbox_iou = compute_bbox_iou(smaller_box, larger_box)
    if not contained or not corner:
        return 0.0, 0.0, larger_index
    better = larger_index if larger_score > smaller_score else smaller_index
    return sketch_iou, bbox_iou, better


def sketch_nms(rgb: np.ndarray, bboxes: np.ndarray, scores: np.ndarray, cleaned_masks: Sequence[np.ndarray],
               sketch_iou_threshold: float, bbox_iou_threshold: float = 0.7, sketch_iou_of=None) -> np.ndarray:
    """sketch_nms (nms_sketch.py:270-351), masks handed over in memory instead of being re-read from PNG files per
    pair.  Returns the kept ORIGINAL indices in the reference's order."""
    if len(bboxes) == 0:
        return np.array([])
    gray = png_gray(rgb)
    hw = gray.shape
    if sketch_iou_of is None:
        S = sketch_pixels(rgb)
        refined = [np.logical_and(m > 0, S) for m in cleaned_masks]

        def sketch_iou_of(a, b):
            union = np.sum(refined[a] | refined[b])
            return np.sum(refined[a] & refined[b]) / union if union > 0 else 0.0
    kept_box_idx = filter_full_or_empty_bbox(gray, bboxes)
    if len(kept_box_idx) == 0:
        return np.array([])
    fb, fs = bboxes[kept_box_idx], scores[kept_box_idx]
    order = np.argsort(-fs)
    original = kept_box_idx[order]
    n = len(fb)
    keep = np.ones(n, dtype=bool)
    for i in range(n):
        if not keep[i]:
            continue
        remaining = order[i + 1:]
        if len(remaining) == 0:
            continue
        # NB: the reference maps the pair's masks through the FILTERED index (mask_{filtered index}.png,
        # nms_sketch.py:214-224), not the original one; kept as is.
        terms = [pair_terms(hw, fb, fs, order[i], r, sketch_iou_of) for r in remaining]
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


def process_json_with_sketch_nms(rgb, bboxes_json: Dict, cleaned_masks, iou_threshold: float = 0.2) -> Dict:
    """bbox_filter.py:12-36 (the runner calls it with sketch_iou_thresh=0.2, runner.py:71)."""
    keep = sketch_nms(rgb, np.array(bboxes_json["bboxes"]), np.array(bboxes_json["scores"]), cleaned_masks, iou_threshold)
    return {"bboxes": [bboxes_json["bboxes"][i] for i in keep], "scores": [bboxes_json["scores"][i] for i in keep],
            "kept_indices": [int(i) for i in keep], "threshold": iou_threshold}
