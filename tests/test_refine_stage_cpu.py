"""Host half of the GPU refinement stage (inklayer_amd/refine_stage.py): the two pixel-sequential C++ routines of
libinklayer_hip.so (no GPU involved: they take host pointers) and the decisions over numbers, against the checker
oracle/refine4_ref.py on the reference's own sketches / committed outputs (tests/golden/refine_*.npz)."""
import glob
from pathlib import Path

import numpy as np
import pytest

GOLD = sorted(glob.glob(str(Path(__file__).resolve().parent / "golden" / "refine_*.npz")))


def _load(path):
    g = np.load(path)
    h, w = (int(v) for v in g["hw"])
    unpack = lambda a: np.unpackbits(a, axis=-1)[..., :w].astype(bool)
    return g, h, w, unpack


@pytest.mark.parametrize("path", GOLD, ids=lambda p: Path(p).stem)
def test_native_stroke_thinning_equals_the_oracle(path):
    from inklayer_amd import refine_stage as S
    from oracle import refine4_ref as R
    g, h, w, _ = _load(path)
    rgb = g["input"]
    want = R.sparse_sketch_sample(R.sketch_to_01binary(rgb[..., ::-1]))
    got = S.sparse_sketch_sample(rgb)
    assert got.dtype == np.int32 and [tuple(p) for p in got.tolist()] == [(int(y), int(x)) for y, x in want]
    assert len(got) > 50


def test_native_thinning_edge_cases():
    from inklayer_amd import refine_stage as S
    from oracle import refine4_ref as R
    rs = np.random.RandomState(0)
    for h, w in ((37, 91), (100, 100), (200, 64), (300, 333)):          # radius 0.37 .. 3.0, incl. an integer one
        rgb = np.full((h, w, 3), 255, np.uint8)
        rgb[rs.rand(h, w) < 0.3] = 0
        rgb[0, :] = 0
        rgb[:, -1] = 10
        want = R.sparse_sketch_sample(R.sketch_to_01binary(rgb[..., ::-1]))
        got = S.sparse_sketch_sample(rgb)
        assert [tuple(p) for p in got.tolist()] == [(int(y), int(x)) for y, x in want]
    assert len(S.sparse_sketch_sample(np.full((40, 40, 3), 255, np.uint8))) == 0     # a blank page has no strokes
    assert len(S.sparse_sketch_sample(np.zeros((40, 40, 3), np.uint8))) == 1600        # max/2 rule: an all-black page is all stroke (radius 0.4)
    blank = np.full((40, 40, 3), 255, np.uint8)
    blank[0, 0] = 0
    assert S.sparse_sketch_sample(blank).tolist() == [[0, 0]]


@pytest.mark.parametrize("path", GOLD, ids=lambda p: Path(p).stem)
def test_native_raster_assignment_equals_the_oracle(path):
    """ink_host_assign_unlabeled fed with exact squared distances (scipy EDT of the grown masks instead of the GPU's
    distance kernel) against refine_masks_with_boxes of the checker, on the reference's masks_disjoint sets."""
    from scipy import ndimage
    from inklayer_amd import refine_stage as S
    from oracle import refine4_ref as R
    g, h, w, unpack = _load(path)
    dis = unpack(g["masks_disjoint"])[g["masks_disjoint_present"]]
    boxes = R.unnormalize_bboxes(g["final_bboxes"].tolist(), h, w)
    luma = R.pil_luma(g["input"])
    grown = R.refine_masks_with_watershed(luma, [m.astype(bool) for m in dis])
    want = R.refine_masks_with_boxes(luma, grown, boxes)
    sketch = ~(luma > 250)
    unl = sketch & ~np.any(grown, axis=0)
    q = np.argwhere(unl).astype(np.int32)
    mask_boxes = [R.compute_mask_bbox(m) for m in grown]
    b2m = S.match_boxes_to_masks(boxes, [None if b is None else [int(v) for v in b] for b in mask_boxes])
    assert b2m == R.match_masks_to_boxes(grown, boxes)
    box2mask = np.full(len(boxes), -1, np.int32)
    for bi, mi in (b2m or {}).items():
        box2mask[bi] = mi
    d2 = np.full((len(q), 256), 0x7FFFFFFF, np.int32)
    for mi, m in enumerate(grown):
        if m.any():
            e = ndimage.distance_transform_edt(~m)
            d2[:, mi + 1] = np.rint(e[q[:, 0], q[:, 1]] ** 2).astype(np.int32)
    lab = S.assign_unlabeled(q, np.asarray(boxes), box2mask, d2, np.array([m.any() for m in grown], np.uint8), len(grown))
    got = [m.copy() for m in grown]
    for (y, x), l in zip(q, lab):
        if l:
            got[l - 1][y, x] = True
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    print(Path(path).stem, "unlabeled", len(q), "assigned", int((lab > 0).sum()))


def test_decisions_over_numbers_equal_the_oracle():
    from inklayer_amd import refine_stage as S
    from oracle import refine4_ref as R
    rs = np.random.RandomState(3)
    for trial in range(40):
        H, W = int(rs.randint(60, 900)), int(rs.randint(60, 900))
        n = int(rs.randint(1, 14))
        x1 = rs.randint(-5, W - 10, n)
        y1 = rs.randint(-5, H - 10, n)
        boxes = np.stack([x1, y1, x1 + rs.randint(1, W // 2, n), y1 + rs.randint(1, H // 2, n)], 1)
        if trial % 5 == 0:                       # nested boxes: containment must fire
            boxes[0] = [10, 10, W - 10, H - 10]
            boxes[-1] = [20, 20, 40, 40]
        assert np.array_equal(S.box_containment(boxes.tolist(), H, W), R.build_containment_graph(boxes.tolist(), (H, W)))
        # overlap rectangles = what numpy slicing does with the int boxes (negative starts wrap around)
        rect = S.overlap_rects(boxes, H, W)
        probe = rs.rand(H, W)
        for i in range(n):
            for j in range(i + 1, n):
                xa, ya = max(boxes[i, 0], boxes[j, 0]), max(boxes[i, 1], boxes[j, 1])
                xb, yb = min(boxes[i, 2], boxes[j, 2]), min(boxes[i, 3], boxes[j, 3])
                y0, y1_, x0, x1_ = rect[i, j]
                if xb <= xa or yb <= ya:
                    assert (rect[i, j] == 0).all()
                else:
                    assert np.array_equal(probe[ya:yb, xa:xb], probe[y0:y1_, x0:x1_])
    assert S.box_containment([[0.1, 0.1, 0.9, 0.9], [0.2, 0.2, 0.4, 0.4]], 100, 200)[0, 1]      # normalised boxes are scaled
    # depth score: most frequent 0.1 bin, ties -> the smaller bin, float32 like the reference
    v = np.array([0.31, 0.29, 0.52, 0.49, 0.1], np.float32)
    assert S.depth_score(v) == R.get_binned_frequent(v) and S.depth_score(np.zeros(0, np.float32)) == float("inf")
    for _ in range(20):
        v = (rs.rand(rs.randint(1, 300)) * 4).astype(np.float32)
        assert S.depth_score(v) == R.get_binned_frequent(v)
    assert S.to_pixel_boxes([[0.1, 0.2, 0.5, 0.999]], 750, 640) == R.unnormalize_bboxes([[0.1, 0.2, 0.5, 0.999]], 750, 640)
