"""Refinement stage of InkLayer on the MI355X: depth ordering of the masks, disjoint parsing, growth of
the masks over unlabeled stroke pixels, per-pixel box assignment, the "unlabeled" extra mask.

Reference: InkLayer/refinement/depth_sort.py:49-295 (sort_sketch_masks and helpers) and
InkLayer/refinement/refiner.py:21-372 (parse_masks_to_disjoint_masks, improve_sam_masks), which run numpy / cv2 / skimage
/ scipy over n separate HxW masks on the CPU.

Design (MI355X-first, csrc/refine_stage.hip + csrc/bitplane.h): the cleaned masks and the sketch become row-aligned
BIT PLANES once; everything that is a function of pixels - stroke / mask intersections, dilated overlaps, popcount
tables, the layered composite, the neighbour-count cleaning, closing + connected components of the unlabeled strokes,
the disk growth, bounding boxes, nearest-pixel distances, the opening of the leftover strokes - is a kernel over planes
or over ONE uint8 label image (after the disjoint parsing a pixel belongs to at most one mask).  The host keeps what is
a decision over a few dozen NUMBERS (scores -> order, containment of boxes, which ranks survive, box <-> mask matching)
and the two algorithms that are sequential over pixels by definition: the greedy stroke thinning and the raster-order
pixel assignment, both plain C++ inside libinklayer_hip.so (ink_host_*).  Seven small device<->host hand-overs of
tables; no mask ever crosses PCIe except the two label images the runner writes to disk.
Bit-exact with the reference's own committed outputs (tests/test_refine_stage_gpu.py).
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check
from .ops import _stream

I32, U8, I64, F32 = torch.int32, torch.uint8, torch.int64, torch.float32
INT_MAX = 0x7FFFFFFF


def to_pixel_boxes(norm_boxes, h: int, w: int) -> List[List[int]]:
    """refinement/utils.py:41-51: normalised xyxy -> int()-truncated pixel boxes."""
    return [[int(b[0] * w), int(b[1] * h), int(b[2] * w), int(b[3] * h)] for b in norm_boxes]


# ---------------------------------------------------------------------------------------------------------------
# host natives
# ---------------------------------------------------------------------------------------------------------------
def sparse_sketch_sample(sketch_rgb: np.ndarray) -> np.ndarray:
    """depth_sort.py:49-68 on the sketch as cv2.imread would hand it over: int32 [P, 2] (y, x) samples."""
    rgb = np.ascontiguousarray(sketch_rgb, dtype=np.uint8)
    H, W = rgb.shape[:2]
    cap = max(1024, (H * W) // 16)
    while True:
        out = np.empty((cap, 2), np.int32)
        cnt = C.c_int32(0)
        check(_lib.lib().ink_host_sparse_sample(rgb.ctypes.data, H, W, out.ctypes.data, cap, C.byref(cnt)),
              "ink_host_sparse_sample")
        if cnt.value <= cap:
            return out[:cnt.value]
        cap = cnt.value


def assign_unlabeled(q_yx: np.ndarray, boxes: np.ndarray, box2mask: np.ndarray, d2: np.ndarray, nonempty: np.ndarray,
                     n_masks: int) -> np.ndarray:
    """The raster-order loop of refine_masks_with_boxes (refiner.py:262-295): label per unlabeled pixel (0 = none)."""
    Q = len(q_yx)
    out = np.zeros(Q, np.int32)
    if Q == 0:
        return out
    q = np.ascontiguousarray(q_yx, np.int32)
    b = np.ascontiguousarray(boxes, np.int32).reshape(-1, 4)
    m = np.ascontiguousarray(box2mask, np.int32)
    d = np.ascontiguousarray(d2, np.int32)
    ne = np.ascontiguousarray(nonempty, np.uint8)
    check(_lib.lib().ink_host_assign_unlabeled(q.ctypes.data, Q, b.ctypes.data if len(b) else None, len(b),
                                               m.ctypes.data if len(b) else None, d.ctypes.data if len(b) else None,
                                               ne.ctypes.data if len(b) else None, n_masks, out.ctypes.data),
          "ink_host_assign_unlabeled")
    return out


# ---------------------------------------------------------------------------------------------------------------
# decisions over numbers (host)
# ---------------------------------------------------------------------------------------------------------------
def depth_score(vals: np.ndarray) -> float:
    """get_mask_depth_score's reduction (depth_sort.py:85-89, utils.py:53-59): the most frequent 0.1-wide depth bin of
    the mask's samples (ties -> the smallest bin), +inf without samples.  float32 arithmetic like the reference's."""
    if len(vals) == 0:
        return float("inf")
    binned = np.round(np.asarray(vals) / 0.1) * 0.1
    bins, counts = np.unique(binned, return_counts=True)
    return bins[np.argmax(counts)]


def box_containment(boxes, H: int, W: int) -> np.ndarray:
    """build_containment_graph_fast (depth_sort.py:114-175): g[i, j] = box i contains box j - every side of j inside i
    up to a slack of max(1, 0.2 % of the larger image side), j at least 2 % smaller in area, j's centre inside i."""
    n = len(boxes)
    g = np.zeros((n, n), dtype=bool)
    if n == 0:
        return g
    b = np.array(boxes, dtype=float).reshape(n, 4)
    if b.max() <= 1.0 + 1e-6:                                   # boxes that look normalised are scaled to pixels
        b = b * np.array([W, H, W, H], dtype=float)
    lo = np.minimum(b[:, :2], b[:, 2:])
    hi = np.maximum(b[:, :2], b[:, 2:])
    slack = float(max(1.0, 0.002 * max(H, W)))
    area = np.clip(hi[:, 0] - lo[:, 0], 0, None) * np.clip(hi[:, 1] - lo[:, 1], 0, None)
    centre = (lo + hi) * 0.5
    for i in range(n):
        inner_ok = ((lo[i] - slack <= lo).all(1) & (hi[i] + slack >= hi).all(1)
                    & (area[i] * (1.0 - 0.02) > area)
                    & (lo[i] - slack <= centre).all(1) & (centre <= hi[i] + slack).all(1))
        inner_ok[i] = False
        g[i] = inner_ok
    return g


def _slice_bounds(a: int, b: int, size: int):
    """numpy's resolution of the slice [a:b] on an axis of `size` (negative indices count from the end)."""
    s, e, _ = slice(a, b).indices(size)
    return s, max(s, e)


def overlap_rects(int_boxes: np.ndarray, H: int, W: int) -> np.ndarray:
    """rect[i, j] = rows / columns numpy would slice for the box intersection of (i, j) in compute_major_overlap_matrix
    (depth_sort.py:206-226), (0, 0, 0, 0) where the reference skips the pair."""
    n = len(int_boxes)
    rect = np.zeros((n, n, 4), np.int32)
    for i in range(n):
        x1i, y1i, x2i, y2i = (int(v) for v in int_boxes[i])
        if x2i <= x1i or y2i <= y1i:
            continue
        for j in range(i + 1, n):
            x1j, y1j, x2j, y2j = (int(v) for v in int_boxes[j])
            if x2j <= x1j or y2j <= y1j:
                continue
            xa, ya, xb, yb = max(x1i, x1j), max(y1i, y1j), min(x2i, x2j), min(y2i, y2j)
            if xb <= xa or yb <= ya:
                continue
            ys, ye = _slice_bounds(ya, yb, H)
            xs, xe = _slice_bounds(xa, xb, W)
            rect[i, j] = rect[j, i] = (ys, ye, xs, xe)
    return rect


def depth_order(scores: Sequence[float], overlap: np.ndarray, contains: np.ndarray) -> List[int]:
    """sort_sketch_masks' ordering (depth_sort.py:270-288): deepest score first, then three bubble passes that move a
    containing box in front of a box it contains when their stroke masks overlap."""
    order = list(np.argsort(scores)[::-1])
    for _ in range(3):
        for i in range(len(order)):
            for j in range(i + 1, len(order)):
                a, b = order[i], order[j]
                if overlap[a, b] and contains[a, b]:
                    order[i], order[j] = b, a
    return [int(v) for v in order]


def match_boxes_to_masks(boxes: Sequence[Sequence[int]], mask_boxes: Sequence[Optional[Sequence[int]]]):
    """match_masks_to_boxes (refiner.py:199-225): greedy one-to-one matching by IoU between the boxes and the bounding
    boxes of the NON-EMPTY masks; the returned mask numbers index that compacted list (the reference then uses them on
    the full list - kept, it is the reference's behaviour).  None when there is nothing to match."""
    from .refine import _bbox_iou
    mb = [m for m in mask_boxes if m is not None]
    iou = np.zeros((len(boxes), len(mb)))
    for i, b in enumerate(boxes):
        for j, m in enumerate(mb):
            iou[i, j] = _bbox_iou(b, m)
    if iou.size == 0:
        return None
    pairs = {}
    while iou.max() != 0:
        bi, mi = np.unravel_index(np.argmax(iou), iou.shape)
        pairs[int(bi)] = int(mi)
        iou[bi, :] = 0
        iou[:, mi] = 0
    return pairs


# ---------------------------------------------------------------------------------------------------------------
# the stage
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class RefineResult:
    order: List[int]                       # depth order of the input masks (deepest first)
    scores: List[float]
    sorted_boxes: List[List[int]]          # input boxes in depth order
    disjoint: np.ndarray                   # uint8 [H, W] label image of masks_disjoint/ (label l = file mask_{l-1}.png)
    n_disjoint: int
    info: List[dict] = field(default_factory=list)
    final: Optional[np.ndarray] = None     # uint8 [H, W] label image of masks_final/ (same labels as `disjoint`)
    extra: Optional[np.ndarray] = None     # bool [H, W]: the appended "unlabeled" mask, None if empty
    timings: dict = field(default_factory=dict)

    def disjoint_masks(self) -> List[np.ndarray]:
        return [self.disjoint == l for l in range(1, self.n_disjoint + 1)]

    def final_masks(self) -> List[np.ndarray]:
        out = [self.final == l for l in range(1, self.n_disjoint + 1)]
        if self.extra is not None:
            out.append(self.extra)
        return out


def _i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.int32)).to(dev)


def _unpack_plane(words: np.ndarray, W: int) -> np.ndarray:
    """[H, Wp] uint64 (little endian bit order) -> bool [H, W]."""
    return np.unpackbits(words.view(np.uint8), axis=1, bitorder="little")[:, :W].astype(bool)


@torch.no_grad()
def refine_masks(cleaned_masks: torch.Tensor, boxes_px: Sequence[Sequence[int]], sketch_rgb: np.ndarray,
                 depth: Optional[torch.Tensor], order: Optional[Sequence[int]] = None,
                 stop_after_disjoint: bool = False, only_order: bool = False):
    """parse_masks_to_disjoint_masks + improve_sam_masks (refiner.py:91-126, 340-372) for one sketch.
    cleaned_masks: [n, H, W] uint8 on the GPU (> 0 = inside), one per box; boxes_px: n pixel boxes (ints);
    sketch_rgb: HxWx3 uint8 (host); depth: [H, W] float32 on the GPU (the Depth-Anything map).  `order` (tests): a given
    depth order instead of the one derived from `depth`.  only_order: stop after the ordering and return
    sort_sketch_masks' triple (order, depth scores, containment graph) (depth_sort.py:244-295)."""
    L = _lib.lib()
    dev = cleaned_masks.device
    assert cleaned_masks.dtype == U8 and cleaned_masks.dim() == 3 and cleaned_masks.is_contiguous()
    n, H, W = (int(v) for v in cleaned_masks.shape)
    assert tuple(sketch_rgb.shape) == (H, W, 3) and len(boxes_px) == n and n <= 254
    Wp = (W + 63) // 64
    st = _stream()
    rgb_dev = torch.from_numpy(np.ascontiguousarray(sketch_rgb)).to(dev)
    sk = torch.empty((4, H, Wp), device=dev, dtype=I64)
    scratch = torch.zeros(8, device=dev, dtype=I32)                       # [0] max, [1] sketch area, [2] unl count, [3] extra
    check(L.ink_refine_sketch_planes(rgb_dev.data_ptr(), H, W, sk.data_ptr(), scratch.data_ptr(), st), "ink_refine_sketch_planes")
    planes = torch.empty((max(n, 1), H, Wp), device=dev, dtype=I64)
    if n:
        check(L.ink_bitplane_pack(cleaned_masks.data_ptr(), n, H, W, 0, planes.data_ptr(), st), "ink_bitplane_pack")

    # ---- pair / per-mask tables (+ depth samples) in one hand-over
    boxes_list = [list(b) for b in boxes_px]
    sort_boxes = boxes_list
    if n and np.all(np.array(boxes_list) <= 1.0):                         # depth_sort.py:262-263
        sort_boxes = [np.asarray(b) * np.array([W, H, W, H]) for b in boxes_list]
    int_boxes = np.asarray(sort_boxes, dtype=int).reshape(n, 4)
    pair = torch.zeros((max(n, 1), max(n, 1), 2), device=dev, dtype=I32)
    per = torch.zeros((max(n, 1), 3), device=dev, dtype=I32)
    rect_dev = _i32(overlap_rects(int_boxes, H, W), dev) if n else None
    dil = torch.empty_like(planes)
    check(L.ink_refine_pair_tables(planes.data_ptr() if n else None, sk.data_ptr(), n, H, W,
                                   rect_dev.data_ptr() if n else None, dil.data_ptr() if n else None,
                                   pair.data_ptr() if n else None, per.data_ptr() if n else None,
                                   scratch[1:].data_ptr(), st), "ink_refine_pair_tables")
    scores: List[float] = []
    host_s = {}
    if order is None and n:
        t0 = time.perf_counter()
        pts = sparse_sketch_sample(sketch_rgb)
        host_s["stroke thinning (C++)"] = time.perf_counter() - t0
        if len(pts):
            assert depth is not None and depth.dtype == F32 and tuple(depth.shape) == (H, W) and depth.is_contiguous()
            pts_dev = torch.from_numpy(pts).to(dev)
            vals = torch.empty(len(pts), device=dev, dtype=F32)
            inside = torch.empty((n, len(pts)), device=dev, dtype=U8)
            check(L.ink_refine_depth_samples(planes.data_ptr(), n, depth.data_ptr(), pts_dev.data_ptr(), len(pts), H, W,
                                             vals.data_ptr(), inside.data_ptr(), st), "ink_refine_depth_samples")
            vals_h, inside_h = vals.cpu().numpy(), inside.cpu().numpy().astype(bool)
            scores = [depth_score(vals_h[inside_h[m]]) for m in range(n)]
        else:
            scores = [float("inf")] * n
    pair_h, per_h, sketch_area = pair.cpu().numpy().astype(np.int64), per.cpu().numpy().astype(np.int64), int(scratch[1].item())
    if order is None:
        area_d = per_h[:n, 1]
        overlap = np.zeros((n, n), dtype=bool)
        for i in range(n):
            for j in range(i + 1, n):
                inter = int(pair_h[i, j, 1])
                if inter and area_d[i] and area_d[j] and inter / float(min(area_d[i], area_d[j])) >= 0.6:
                    overlap[i, j] = overlap[j, i] = True
        contains = box_containment(sort_boxes, H, W)
        order = depth_order(scores, overlap, contains) if n else []
        if only_order:
            return order, scores, contains
    order = [int(i) for i in order]
    sorted_boxes = [boxes_list[i] for i in order]

    # ---- disjoint parsing (refiner.py:100-126, 35-88)
    ranks = list(order)                       # mask index per rank; -1 once a rank is emptied
    left = len(ranks)
    for r, m in enumerate(order):             # a mask that covers > 90 % of the strokes is dropped while others remain
        if left > 1 and per_h[m, 2] > 0.9 * sketch_area:
            ranks[r] = -1
            left -= 1
    label = torch.empty((H, W), device=dev, dtype=U8)
    hist = torch.empty(256, device=dev, dtype=I32)
    ranks_dev = _i32(ranks, dev) if n else None
    check(L.ink_refine_composite(planes.data_ptr() if n else None, ranks_dev.data_ptr() if n else None, len(ranks), H, W,
                                 label.data_ptr(), hist.data_ptr(), st), "ink_refine_composite")
    hist_h = hist.cpu().numpy().astype(np.int64)
    hist_h[0] = H * W - int(hist_h[1:].sum())
    # `np.unique(composite)[1:]` (refiner.py:49): the reference drops the SMALLEST value present, which is the
    # background 0 - unless the masks cover every pixel, in which case the first mask label goes instead (kept as is)
    present = [l for l in range(len(ranks) + 1) if hist_h[l] > 0][1:]
    lut = np.zeros(256, np.uint8)
    info, n_dis = [], 0
    for r, m in enumerate(ranks):
        if m < 0 or (r + 1) not in present:
            continue                          # no pixel carries this label
        if hist_h[r + 1] < 0.05 * per_h[m, 0] and any(ranks[q] >= 0 and pair_h[m, ranks[q], 0] > 0 for q in range(r)):
            continue                          # nearly hidden and overlapping an earlier mask: merged away
        n_dis += 1
        lut[r + 1] = n_dis
        info.append({"bbox": sorted_boxes[r], "original_indices": [m]})
    lut_dev = torch.from_numpy(lut).to(dev)
    disjoint = torch.empty_like(label)
    check(L.ink_refine_relabel_clean(label.data_ptr(), lut_dev.data_ptr(), H, W, disjoint.data_ptr(), st),
          "ink_refine_relabel_clean")
    res = RefineResult(order, [float(s) for s in scores], sorted_boxes, disjoint.cpu().numpy(), n_dis, info, timings=host_s)
    if stop_after_disjoint:
        return res
    res.final, res.extra = grow_and_assign(disjoint, n_dis, sorted_boxes, sk, host_s)
    return res


@torch.no_grad()
def grow_and_assign(disjoint: torch.Tensor, n_masks: int, boxes: Sequence[Sequence[int]], sk: torch.Tensor,
                    host_s: Optional[dict] = None):
    """improve_sam_masks (refiner.py:340-372) on a label image: growth, raster-order box assignment, the unlabeled extra
    mask.  disjoint: uint8 [H, W] on the GPU; sk: the 4 sketch planes.  -> (final label image np uint8, extra or None)."""
    L = _lib.lib()
    dev = disjoint.device
    H, W = (int(v) for v in disjoint.shape)
    Wp = (W + 63) // 64
    st = _stream()
    pw, ci = C.c_int64(0), C.c_int64(0)
    check(L.ink_refine_grow_workspace(H, W, C.byref(pw), C.byref(ci)), "ink_refine_grow_workspace")
    pws = torch.empty(pw.value, device=dev, dtype=I64)
    ccw = torch.empty(ci.value, device=dev, dtype=I32)
    flags = torch.empty(256, device=dev, dtype=I32)
    bbox = torch.empty((256, 4), device=dev, dtype=I32)
    cnt = torch.zeros(2, device=dev, dtype=I32)
    grown = torch.empty_like(disjoint)
    cap = 1 << 16
    while True:
        unl = torch.empty((cap, 2), device=dev, dtype=I32)
        check(L.ink_refine_grow(disjoint.data_ptr(), sk.data_ptr(), H, W, pws.data_ptr(), ccw.data_ptr(), flags.data_ptr(),
                                grown.data_ptr(), bbox.data_ptr(), unl.data_ptr(), cap, cnt.data_ptr(), st), "ink_refine_grow")
        Q = int(cnt[0].item())
        if int(ccw[0].item()) != 0:
            raise _lib.InkLayerHipError("ink_refine_grow: run workspace overflow")
        if Q <= cap:
            break
        cap = Q
    assign = np.zeros((0, 3), np.int32)
    if Q and n_masks and len(boxes):
        bb = bbox.cpu().numpy()[1:n_masks + 1]
        mask_boxes = [None if b[2] < 0 else [int(b[0]), int(b[1]), int(b[2]), int(b[3])] for b in bb]
        b2m = match_boxes_to_masks(boxes, mask_boxes)
        if b2m is not None:
            q = unl[:Q].cpu().numpy()
            barr = np.asarray(boxes, dtype=np.int64).reshape(-1, 4)
            box2mask = np.full(len(barr), -1, np.int32)
            for bi, mi in b2m.items():
                box2mask[bi] = mi
            # candidate labels per pixel: the masks matched to the boxes that contain it (only needed with >= 2 boxes)
            ins = ((barr[None, :, 0] <= q[:, None, 1]) & (q[:, None, 1] <= barr[None, :, 2])
                   & (barr[None, :, 1] <= q[:, None, 0]) & (q[:, None, 0] <= barr[None, :, 3]))            # [Q, nb]
            multi = ins.sum(1) > 1
            cand = np.zeros((Q, 4), np.uint64)
            for bi in np.nonzero(box2mask >= 0)[0]:
                lab = int(box2mask[bi]) + 1
                if lab <= 255:
                    sel = ins[:, bi] & multi
                    cand[sel, lab >> 6] |= np.uint64(1 << (lab & 63))
            d2 = torch.empty((Q, 256), device=dev, dtype=I32)
            cand_dev = torch.from_numpy(cand.view(np.int64)).to(dev)
            check(L.ink_refine_query_dists(grown.data_ptr(), unl.data_ptr(), cand_dev.data_ptr(), Q, H, W, d2.data_ptr(), st),
                  "ink_refine_query_dists")
            nonempty = np.array([mb is not None for mb in mask_boxes], np.uint8)
            d2_h = d2.cpu().numpy()
            t0 = time.perf_counter()
            lab = assign_unlabeled(q, barr, box2mask, d2_h, nonempty, n_masks)
            if host_s is not None:
                host_s["raster-order assignment (C++)"] = time.perf_counter() - t0
                host_s["unlabeled pixels"] = Q
            keep = lab > 0
            assign = np.concatenate([q[keep], lab[keep, None]], 1).astype(np.int32)
    extra = torch.empty((H, Wp), device=dev, dtype=I64)
    assign_dev = _i32(assign, dev) if len(assign) else None
    check(L.ink_refine_finalize(grown.data_ptr(), assign_dev.data_ptr() if len(assign) else None, len(assign), sk.data_ptr(),
                                H, W, pws.data_ptr(), extra.data_ptr(), cnt[1:].data_ptr(), st), "ink_refine_finalize")
    final = grown.cpu().numpy()
    extra_np = _unpack_plane(extra.cpu().numpy().view(np.uint64), W) if int(cnt[1].item()) > 0 else None
    return final, extra_np
