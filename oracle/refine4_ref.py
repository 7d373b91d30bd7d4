"""oracle/refine4_ref.py - TEST INFRASTRUCTURE ONLY (the checker of SURVEY §8(f)-4; the product is
inklayer_amd/refine_stage.py + csrc/refine_stage.hip and never imports this file).

A numpy / scipy restatement, function by function, of InkLayer's refinement stage: depth ordering of the masks, disjoint
parsing, growth of the masks over unlabeled stroke pixels, per-pixel box assignment, the "unlabeled" extra mask.

Reference: InkLayer/refinement/depth_sort.py:49-270 (sparse_sketch_sample, get_mask_depth_score,
build_containment_graph_fast, compute_major_overlap_matrix, sort_sketch_masks) and InkLayer/refinement/refiner.py:21-337
(clean_delicate_mask, composite_and_parse_masks, parse_masks_to_disjoint_masks, refine_masks_with_watershed,
match_masks_to_boxes, refine_masks_with_boxes, create_unlabeled_mask), refinement/utils.py.  It follows the reference's
structure closely on purpose: one function per reference function, the same loops over masks, so that a disagreement with
the HIP stage can be localised.  (Rounds 1-2 shipped this file as the product's host stage; it was retired to a checker
when the stage moved to the GPU.)

cv2 / skimage are not importable here; their functions are restated from their published algorithms:
  * skimage.morphology.disk / binary_dilation / binary_closing -> scipy.ndimage with skimage's border rule (dilation
    ignores the outside, erosion treats it as foreground);
  * skimage.segmentation.watershed(distance, markers, mask=sketch, compactness=0.01) as the reference CALLS it: the
    marker image is initialised to -1 and watershed() treats every non-zero label as a seed, so every pixel inside the
    mask is already labelled and the flood never moves a label; the call reduces to `markers * mask`.  The distance /
    gradient images the reference computes for it (refiner.py:171-189) do not influence the result and are not computed;
  * cv2.morphologyEx(MORPH_OPEN, ones(3,3)), cv2.dilate(ones(2,2)) (anchor (1,1): the window reaches up / left),
    cv2.getStructuringElement(MORPH_ELLIPSE, (3,3)) = the 3x3 cross, cv2.imread(GRAYSCALE) = libpng's rgb_to_gray.
PINNED by the reference's own committed outputs (masks_cleaned/ -> masks_disjoint/ -> masks_final/ of its 7 output
sets, tests/golden/refine_*.npz): tests/test_oracle_refine4.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
from scipy import ndimage

SKETCH_THRESHOLD = 250


# ---------------------------------------------------------------------------------------------------------------
# small restated library pieces
# ---------------------------------------------------------------------------------------------------------------
def disk(r: int) -> np.ndarray:
    """skimage.morphology.disk(r) (strict radius)."""
    L = np.arange(-r, r + 1)
    X, Y = np.meshgrid(L, L)
    return (X ** 2 + Y ** 2) <= r ** 2


def _window(b: np.ndarray, st: np.ndarray):
    """Bounding box of the mask grown by the structuring element's radius, clipped to the image (None if empty)."""
    ys, xs = np.nonzero(b.any(1))[0], np.nonzero(b.any(0))[0]
    if len(ys) == 0:
        return None
    ry, rx = st.shape[0] // 2, st.shape[1] // 2
    return (max(int(ys[0]) - ry, 0), min(int(ys[-1]) + ry + 1, b.shape[0]),
            max(int(xs[0]) - rx, 0), min(int(xs[-1]) + rx + 1, b.shape[1]), ry, rx)


def sk_dilate(b: np.ndarray, st: np.ndarray) -> np.ndarray:
    """skimage binary_dilation (outside = background).  A dilation cannot leave the mask's bounding box grown by the
    structuring element's radius, so only that window is processed, as the OR of the window shifted by every offset
    of the (small, symmetric) structuring element - identical result (tests/test_refiner_cpu.py), ~20x less work
    than scipy's generic routine on the full image."""
    out = np.zeros(b.shape, dtype=bool)
    w = _window(b, st)
    if w is None:
        return out
    y0, y1, x0, x1, ry, rx = w
    p = np.pad(b[y0:y1, x0:x1], ((ry, ry), (rx, rx)), constant_values=False)
    h, wd = y1 - y0, x1 - x0
    acc = np.zeros((h, wd), dtype=bool)
    for dy, dx in zip(*np.nonzero(st)):
        acc |= p[dy:dy + h, dx:dx + wd]
    out[y0:y1, x0:x1] = acc
    return out


def sk_erode(b: np.ndarray, st: np.ndarray) -> np.ndarray:
    """skimage binary_erosion (outside the image = foreground).  An erosion stays inside the mask's bounding box and
    only looks one structuring-element radius beyond it: that window is processed, as the AND of its shifts (where the
    window ends inside the image its margin is background, so the foreground padding only acts where the window
    touches the image edge)."""
    out = np.zeros(b.shape, dtype=bool)
    w = _window(b, st)
    if w is None:
        return out
    y0, y1, x0, x1, ry, rx = w
    p = np.pad(b[y0:y1, x0:x1], ((ry, ry), (rx, rx)), constant_values=True)
    h, wd = y1 - y0, x1 - x0
    acc = np.ones((h, wd), dtype=bool)
    for dy, dx in zip(*np.nonzero(st)):
        acc &= p[dy:dy + h, dx:dx + wd]
    out[y0:y1, x0:x1] = acc
    return out


def sk_closing(b: np.ndarray, st: np.ndarray) -> np.ndarray:
    return sk_erode(sk_dilate(b, st), st)


def png_gray(rgb: np.ndarray) -> np.ndarray:
    """cv2.imread(path, IMREAD_GRAYSCALE) of an 8-bit RGB PNG."""
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return ((r * 9798 + g * 19235 + b * 3735 + 16384) >> 15).astype(np.uint8)


def pil_luma(rgb: np.ndarray) -> np.ndarray:
    """PIL Image.convert("L")."""
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def compute_bbox_iou(box1, box2) -> float:
    """refinement/utils.py:11-32."""
    xi1, yi1 = max(box1[0], box2[0]), max(box1[1], box2[1])
    xi2, yi2 = min(box1[2], box2[2]), min(box1[3], box2[3])
    a1 = (box1[2] - box1[0]) * (box1[3] - box1[1])
    a2 = (box2[2] - box2[0]) * (box2[3] - box2[1])
    if xi2 < xi1 or yi2 < yi1:
        return 0.0
    ai = (xi2 - xi1) * (yi2 - yi1)
    return ai / (a1 + a2 - ai)


def compute_mask_bbox(mask: np.ndarray):
    """refinement/utils.py:34-39."""
    m = np.asarray(mask) != 0
    ys, xs = np.nonzero(m.any(1))[0], np.nonzero(m.any(0))[0]      # (row / column projections instead of np.where)
    if len(ys) == 0:
        return None
    return [xs[0], ys[0], xs[-1], ys[-1]]


def unnormalize_bboxes(bboxes, h: int, w: int) -> List[List[int]]:
    """refinement/utils.py:41-51."""
    return [[int(b[0] * w), int(b[1] * h), int(b[2] * w), int(b[3] * h)] for b in bboxes]


# ---------------------------------------------------------------------------------------------------------------
# depth ordering  (depth_sort.py)
# ---------------------------------------------------------------------------------------------------------------
def sketch_to_01binary(sketch_bgr: np.ndarray) -> np.ndarray:
    """refinement/utils.py:3-9 on a cv2.imread colour image: channel 0 (blue) <= max/2 -> 1.0."""
    s = sketch_bgr if sketch_bgr.ndim == 3 else sketch_bgr[:, :, None]
    return 1.0 * ~(s[:, :, 0] > s.max() / 2)              # (the threshold is the maximum over ALL channels)


def sparse_sketch_sample(binary_edge_map: np.ndarray) -> List[Tuple[int, int]]:
    """depth_sort.py:49-68: greedy thinning of the stroke pixels - take the first remaining pixel (row-major), drop every
    stroke pixel within 1 % of the height of it, repeat.  The reference walks a Python set of point indices
    (`next(iter(remaining))`: for a set built from range(n) that is the smallest remaining index) and asks a KD-tree for
    the ball; here the points sit in an index image and a ball is a disk-masked window of it - the same samples in the
    same order (pinned against the literal form on the reference's sketches, tests/test_refiner_cpu.py), ~6x faster."""
    H, W = binary_edge_map.shape
    radius = H * 0.01
    ys, xs = np.where(binary_edge_map > 0)
    n = len(ys)
    if n == 0:
        return []
    r = int(np.floor(radius))
    idx = np.full((H + 2 * r, W + 2 * r), -1, dtype=np.int64)
    idx[ys + r, xs + r] = np.arange(n)
    dy, dx = np.mgrid[-r:r + 1, -r:r + 1]
    disk = (dy * dy + dx * dx) <= radius * radius          # (query_ball_point is inclusive)
    alive = np.ones(n, dtype=bool)
    sampled, cur = [], 0
    while True:
        rest = alive[cur:]
        k = int(rest.argmax()) if len(rest) else 0
        if len(rest) == 0 or not rest[k]:
            break
        cur += k
        y, x = ys[cur], xs[cur]
        sampled.append((y, x))
        win = idx[y:y + 2 * r + 1, x:x + 2 * r + 1]
        alive[win[disk & (win >= 0)]] = False
    return sampled


def get_binned_frequent(values, bin_width: float = 0.1):
    """refinement/utils.py:53-59."""
    binned = np.round(np.array(values) / bin_width) * bin_width
    vals, counts = np.unique(binned, return_counts=True)
    return vals[np.argmax(counts)]


def get_mask_depth_score(mask: np.ndarray, points, depth_map: np.ndarray):
    """depth_sort.py:72-89 (the per-point Python loop as one gather; the binned mode does not depend on the order)."""
    if len(points) == 0:
        return float("inf")
    p = np.asarray(points)
    inside = np.asarray(mask)[p[:, 0], p[:, 1]] > 0
    vals = depth_map[p[inside, 0], p[inside, 1]]
    return get_binned_frequent(vals) if len(vals) else float("inf")


def build_containment_graph(bboxes, image_size) -> np.ndarray:
    """build_containment_graph_fast (depth_sort.py:114-175): graph[i, j] = box i strictly contains box j."""
    if bboxes is None or len(bboxes) == 0:
        return np.zeros((0, 0), dtype=bool)
    H, W = int(image_size[0]), int(image_size[1])
    b = np.asarray(bboxes, dtype=float)
    if np.max(b) <= 1.0 + 1e-6:
        b[:, [0, 2]] *= W
        b[:, [1, 3]] *= H
    b = np.stack([np.minimum(b[:, 0], b[:, 2]), np.minimum(b[:, 1], b[:, 3]),
                  np.maximum(b[:, 0], b[:, 2]), np.maximum(b[:, 1], b[:, 3])], axis=1)
    eps = float(max(1.0, 0.002 * max(H, W)))
    areas = np.clip(b[:, 2] - b[:, 0], 0, None) * np.clip(b[:, 3] - b[:, 1], 0, None)
    cx, cy = (b[:, 0] + b[:, 2]) * 0.5, (b[:, 1] + b[:, 3]) * 0.5
    b1, b2 = b[:, None, :], b[None, :, :]
    c = ((b1[..., 0] - eps <= b2[..., 0]) & (b1[..., 1] - eps <= b2[..., 1])
         & (b1[..., 2] + eps >= b2[..., 2]) & (b1[..., 3] + eps >= b2[..., 3]))
    c &= (areas[:, None] * (1.0 - 0.02)) > areas[None, :]
    c &= (b1[..., 0] - eps <= cx[None, :]) & (cx[None, :] <= b1[..., 2] + eps)
    c &= (b1[..., 1] - eps <= cy[None, :]) & (cy[None, :] <= b1[..., 3] + eps)
    np.fill_diagonal(c, False)
    return c.astype(bool)


def compute_major_overlap_matrix(masks, bboxes=None, thr: float = 0.6, dilate_px: int = 1) -> np.ndarray:
    """depth_sort.py:177-240: |m_i & m_j| / min(|m_i|, |m_j|) >= thr on masks dilated by the 3x3 ellipse (= cross),
    intersections counted inside the intersection of the two boxes."""
    M = [np.asarray(m).astype(np.uint8) for m in masks]
    if dilate_px and dilate_px > 0:
        cross = ndimage.generate_binary_structure(2, 1) if dilate_px == 1 else disk(dilate_px)
        M = [sk_dilate(m > 0, cross).astype(np.uint8) for m in M]
    areas = np.array([int(m.sum()) for m in M], dtype=np.int64)
    if bboxes is None:
        bboxes = []
        for m in M:
            ys, xs = np.where(m > 0)
            bboxes.append((0, 0, 0, 0) if len(ys) == 0 else (int(xs.min()), int(ys.min()), int(xs.max() + 1), int(ys.max() + 1)))
    b = np.asarray(bboxes, dtype=int)
    N = len(M)
    major = np.zeros((N, N), dtype=bool)
    for i in range(N):
        x1i, y1i, x2i, y2i = b[i]
        if areas[i] == 0 or x2i <= x1i or y2i <= y1i:
            continue
        for j in range(i + 1, N):
            x1j, y1j, x2j, y2j = b[j]
            if areas[j] == 0 or x2j <= x1j or y2j <= y1j:
                continue
            xi1, yi1, xi2, yi2 = max(x1i, x1j), max(y1i, y1j), min(x2i, x2j), min(y2i, y2j)
            if xi2 <= xi1 or yi2 <= yi1:
                continue
            inter = int(np.count_nonzero(M[i][yi1:yi2, xi1:xi2] & M[j][yi1:yi2, xi1:xi2]))
            if inter and inter / float(min(areas[i], areas[j])) >= thr:
                major[i, j] = major[j, i] = True
    return major


def sort_sketch_masks(masks, bboxes, sketch_rgb: np.ndarray, depth_map: np.ndarray):
    """depth_sort.py:244-295: deepest first by the binned depth mode over sparse stroke samples, then containers are
    moved in front of the boxes they contain when the masks overlap.  -> (order, depth_scores, containment)."""
    bgr = sketch_rgb[..., ::-1]
    h, w = bgr.shape[:2]
    binary = sketch_to_01binary(bgr)
    points = sparse_sketch_sample(binary)
    if np.all(np.array(bboxes) <= 1.0):
        bboxes = [np.asarray(box) * np.array([w, h, w, h]) for box in bboxes]
    scores = [get_mask_depth_score(m, points, depth_map) for m in masks]
    containment = build_containment_graph(bboxes, (h, w))
    overlap = compute_major_overlap_matrix([m & binary.astype(bool) for m in masks], bboxes=bboxes, dilate_px=1)
    order = list(np.argsort(scores)[::-1])
    for _ in range(3):
        for i in range(len(order)):
            for j in range(i + 1, len(order)):
                a, b = order[i], order[j]
                if overlap[a, b] and containment[a, b]:
                    order[i], order[j] = order[j], order[i]
    return order, scores, containment


# ---------------------------------------------------------------------------------------------------------------
# disjoint parsing  (refiner.py:21-129)
# ---------------------------------------------------------------------------------------------------------------
def clean_delicate_mask(mask: np.ndarray, isolation_threshold: int = 1) -> np.ndarray:
    """refiner.py:21-33: drop pixels with at most one 8-neighbour (the reference convolves with a 3x3 ring; the
    neighbour count is the sum of the eight shifted images)."""
    p = np.pad((np.asarray(mask) > 0).astype(np.uint8), 1)
    cnt = (p[:-2, :-2] + p[:-2, 1:-1] + p[:-2, 2:] + p[1:-1, :-2] + p[1:-1, 2:] + p[2:, :-2] + p[2:, 1:-1] + p[2:, 2:])
    out = mask.copy()
    out[cnt <= isolation_threshold] = False
    return out


def composite_and_parse_masks(masks, bboxes, empty_threshold: float = 0.05):
    """refiner.py:35-88: earlier masks win overlaps; a mask left with < 5 % of its area is merged into the earlier
    mask it overlaps most."""
    if not masks:
        return [], []
    comp = np.zeros(masks[0].shape, dtype=np.uint8)
    orig_areas = [np.sum(m > 0) for m in masks]
    for i in range(len(masks) - 1, -1, -1):
        comp[masks[i] > 0] = i + 1
    labels = np.unique(comp)[1:]
    parsed = [(comp == lab) for lab in labels]
    info = [{"bbox": bboxes[lab - 1], "original_indices": [lab - 1]} for lab in labels]
    out_masks, out_info = [], []
    for pm, inf in zip(parsed, info):
        oi = inf["original_indices"][0]
        if np.sum(pm) < empty_threshold * orig_areas[oi]:
            best, best_ov = None, 0
            for j in range(oi):
                ov = np.sum(np.logical_and(masks[oi], masks[j]))
                if ov > best_ov:
                    best_ov, best = ov, j
            if best is not None:
                comp[np.logical_or(comp == best + 1, masks[oi])] = best + 1
                continue
        out_masks.append(pm)
        out_info.append(inf)
    return out_masks, out_info


def parse_masks_to_disjoint_masks(masks_np, bboxes, sketch_rgb: np.ndarray, depth_map: Optional[np.ndarray],
                                  order: Optional[Sequence[int]] = None):
    """refiner.py:91-126.  `order` (tests): a given depth order instead of sort_sketch_masks."""
    if order is None:
        order, _, _ = sort_sketch_masks(masks_np, bboxes, sketch_rgb, depth_map)
    order = [int(i) for i in order]
    smasks = [masks_np[i] for i in order]
    sboxes = [bboxes[i] for i in order]
    luma = pil_luma(sketch_rgb)
    sketch_area = np.sum(luma < SKETCH_THRESHOLD)
    n = len(smasks)
    for i, m in enumerate(smasks):
        if n > 1 and np.sum(np.logical_and(m > 0, luma < SKETCH_THRESHOLD)) > 0.9 * sketch_area:
            smasks[i] = np.zeros_like(m)
            n -= 1
    disjoint, info = composite_and_parse_masks(smasks, sboxes)
    cleaned = [clean_delicate_mask(m) for m in disjoint]
    final_info = [{"bbox": inf["bbox"], "original_indices": [order[i] for i in inf["original_indices"]]} for inf in info]
    return cleaned, sboxes, final_info


# ---------------------------------------------------------------------------------------------------------------
# mask growth  (refiner.py:129-337)
# ---------------------------------------------------------------------------------------------------------------
def refine_masks_with_watershed(sketch_luma: np.ndarray, original_masks: Sequence[np.ndarray]) -> List[np.ndarray]:
    """refiner.py:129-196 (see the module header for why no flood is run): every mask takes the unlabeled stroke pixels
    within a disk of radius 3 (when it comes within 3 px of a large unlabeled region) or 2; where two masks reach the
    same pixel the LATER one keeps it; results are restricted to stroke pixels."""
    sketch = ~(sketch_luma > SKETCH_THRESHOLD)
    combined = np.zeros_like(sketch, dtype=bool)
    for m in original_masks:
        combined |= m
    unlabeled = sketch & ~combined
    lab, _ = ndimage.label(sk_closing(unlabeled, disk(3)))
    sizes = np.bincount(lab.ravel())
    big = np.zeros(len(sizes), dtype=bool)
    big[1:] = sizes[1:] > 50
    large = big[lab]
    markers = np.full(sketch.shape, -1, dtype=int)
    d3 = disk(3)
    for i, m in enumerate(original_masks, start=1):
        near_large = np.any(sk_dilate(m, d3) & large)
        grown = sk_dilate(m, disk(3 if near_large else 2))
        markers[grown & unlabeled] = i
        markers[m] = i
    labels = markers * sketch
    return [labels == i for i in range(1, len(original_masks) + 1)]


def match_masks_to_boxes(masks, boxes) -> Optional[Dict[int, int]]:
    """refiner.py:199-225: greedy one-to-one matching of boxes to mask bounding boxes by IoU."""
    mboxes = [bb for bb in (compute_mask_bbox(m) for m in masks) if bb is not None]
    iou = np.zeros((len(boxes), len(mboxes)))
    for i, box in enumerate(boxes):
        for j, mb in enumerate(mboxes):
            iou[i, j] = compute_bbox_iou(box, mb)
    if iou.size == 0:
        return None
    out: Dict[int, int] = {}
    while np.max(iou) != 0:
        bi, mi = np.unravel_index(np.argmax(iou), iou.shape)
        out[int(bi)] = int(mi)
        iou[bi, :] = 0
        iou[:, mi] = 0
    return out


def refine_masks_with_boxes(sketch_luma: np.ndarray, original_masks: Sequence[np.ndarray], boxes) -> List[np.ndarray]:
    """refiner.py:228-297: unlabeled stroke pixels, in raster order, go to the mask of the box that contains them; with
    several boxes, to the one whose mask has the nearest filled pixel - INCLUDING pixels assigned earlier in this very
    loop (sequential by construction).  The reference recomputes all distances per pixel; here the distance to a mask
    is min(exact EDT of the mask as it was, distance to the pixels added since): the same value."""
    sketch = ~(sketch_luma > SKETCH_THRESHOLD)
    b2m = match_masks_to_boxes(original_masks, boxes)
    if b2m is None:
        return list(original_masks)
    combined = np.zeros_like(sketch, dtype=bool)
    for m in original_masks:
        combined |= m
    unlabeled = sketch & ~combined
    refined = [m.copy() for m in original_masks]
    nonempty = [bool(m.any()) for m in refined]
    edt = {}
    added: Dict[int, List[Tuple[int, int]]] = {i: [] for i in range(len(refined))}

    def dist_to(mi: int, y: int, x: int) -> float:
        if mi not in edt:
            edt[mi] = ndimage.distance_transform_edt(~original_masks[mi]) if original_masks[mi].any() else None
        d = float("inf") if edt[mi] is None else float(edt[mi][y, x])
        if added[mi]:
            a = np.asarray(added[mi])
            d = min(d, float(np.sqrt(((a[:, 0] - y) ** 2 + (a[:, 1] - x) ** 2).min())))
        return d

    ys, xs = np.where(unlabeled)
    boxes_arr = np.asarray(boxes)
    for y, x in zip(ys, xs):
        inside = np.nonzero((boxes_arr[:, 0] <= x) & (x <= boxes_arr[:, 2]) & (boxes_arr[:, 1] <= y) & (y <= boxes_arr[:, 3]))[0]
        if len(inside) == 0:
            continue
        if len(inside) > 1:
            best, best_d = None, float("inf")
            for bi in inside:
                bi = int(bi)
                if bi not in b2m:
                    continue
                mi = b2m[bi]
                if nonempty[mi]:
                    d = dist_to(mi, int(y), int(x))
                    if d < best_d:
                        best_d, best = d, bi
            if best is not None and best in b2m:
                mi = b2m[best]
                refined[mi][y, x] = True
                added[mi].append((int(y), int(x)))
                nonempty[mi] = True
        else:
            bi = int(inside[0])
            if bi in b2m:
                mi = b2m[bi]
                refined[mi][y, x] = True
                added[mi].append((int(y), int(x)))
                nonempty[mi] = True
    return refined


def create_unlabeled_mask(sketch_gray: np.ndarray, masks: Sequence[np.ndarray]) -> List[np.ndarray]:
    """refiner.py:301-337: stroke pixels no mask claims, opened with a 3x3 square and dilated with a 2x2 square
    (cv2 anchor (1,1): each pixel also takes its upper / left neighbours); appended as one more mask if non-empty."""
    sketch = sketch_gray < SKETCH_THRESHOLD
    labeled = np.zeros_like(sketch, dtype=bool)
    for m in masks:
        labeled |= m.astype(bool)
    un = sketch & ~labeled
    sq = np.ones((3, 3), bool)
    un = ndimage.binary_dilation(ndimage.binary_erosion(un, structure=sq, border_value=1), structure=sq, border_value=0)
    d = un.copy()
    d[1:, :] |= un[:-1, :]
    d[:, 1:] |= un[:, :-1]
    d[1:, 1:] |= un[:-1, :-1]
    if d.sum() == 0:
        return list(masks)
    return list(masks) + [d.astype(np.uint8)]


def improve_sam_masks(sketch_rgb: np.ndarray, masks_np: Sequence[np.ndarray], bboxes) -> List[np.ndarray]:
    """improve_sam_masks (refiner.py:340-372) without the visualisations: growth -> box assignment -> unlabeled mask."""
    luma = pil_luma(sketch_rgb)
    grown = refine_masks_with_watershed(luma, [m.astype(bool) for m in masks_np])
    boxed = refine_masks_with_boxes(luma, grown, bboxes)
    return create_unlabeled_mask(png_gray(sketch_rgb), boxed)
