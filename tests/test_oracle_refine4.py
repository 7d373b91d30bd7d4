"""Row (f)-4, the CHECKER: oracle/refine4_ref.py (the numpy / scipy restatement of the depth-order based disjoint parsing,
mask growth, per-pixel box assignment and the unlabeled extra mask that tests/test_refine_stage_gpu.py compares the HIP
stage with) against the reference's OWN committed outputs (tests/golden/refine_*.npz:
masks_cleaned/ + bboxes_final.json -> masks_disjoint/ -> masks_final/ of its 7 output sets), bit for bit.

Stage B (masks_disjoint -> masks_final: refiner.py:129-372) needs nothing but the fixtures.  Stage A (masks_cleaned ->
masks_disjoint: refiner.py:35-126) depends on the depth ORDER of the masks, which the reference derived from
Depth-Anything-V2 with the real checkpoint (not available): the order is recovered from the fixtures themselves (every
disjoint mask is a subset of exactly one cleaned mask, files are numbered in depth order) and the test then checks that
the compositing / merging / cleaning reproduce masks_disjoint exactly under that order.  The depth scoring itself
(sort_sketch_masks) is checked on synthetic depth maps."""
import glob
import itertools
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
def test_stage_b_reproduces_reference_masks_final(path):
    from oracle import refine4_ref as R
    g, h, w, unpack = _load(path)
    dis = unpack(g["masks_disjoint"])[g["masks_disjoint_present"]]
    fin = unpack(g["masks_final"])[g["masks_final_present"]]
    boxes = R.unnormalize_bboxes(g["final_bboxes"].tolist(), h, w)
    # sorted_bboxes: the kept boxes in depth order = the order of the disjoint files (boxes whose mask vanished last)
    used, order = set(), []
    for bb in (R.compute_mask_bbox(m) for m in dis):
        j = int(np.argmax([(R.compute_bbox_iou(boxes[j], bb) if j not in used else -1) for j in range(len(boxes))]))
        used.add(j)
        order.append(j)
    order += [j for j in range(len(boxes)) if j not in used]
    out = R.improve_sam_masks(g["input"], list(dis), [boxes[j] for j in order])
    assert len(out) == len(fin)
    for o, f in zip(out, fin):
        assert np.array_equal(np.asarray(o) > 0, f)
    changed = sum(int(((np.asarray(o) > 0) != d).sum()) for o, d in zip(out, dis))
    assert changed > 0                               # the stage is not the identity on any set


@pytest.mark.parametrize("path", GOLD, ids=lambda p: Path(p).stem)
def test_stage_a_reproduces_reference_masks_disjoint_under_the_recovered_depth_order(path):
    from oracle import refine4_ref as R
    g, h, w, unpack = _load(path)
    cleaned = unpack(g["masks_cleaned"])
    dis = unpack(g["masks_disjoint"])[g["masks_disjoint_present"]]
    kept = g["final_kept"].tolist()
    boxes = R.unnormalize_bboxes(g["final_bboxes"].tolist(), h, w)
    masks = [cleaned[k].astype(np.uint8) * 255 for k in kept]
    used, order = set(), []
    for d in dis:                                    # the tightest unused cleaned mask that contains the disjoint mask
        cand = [j for j in range(len(masks)) if j not in used and np.logical_and(d, masks[j] > 0).sum() == d.sum()]
        j = min(cand, key=lambda j: int((masks[j] > 0).sum()))
        used.add(j)
        order.append(j)
    rest = [j for j in range(len(masks)) if j not in used]      # masks that vanished / were merged: position unknown
    assert len(rest) <= 2
    ok = False
    for pos in itertools.product(range(len(order) + 1), repeat=len(rest)):
        o = list(order)
        for r, p in sorted(zip(rest, pos), key=lambda t: -t[1]):
            o.insert(p, r)
        out, sboxes, info = R.parse_masks_to_disjoint_masks(masks, boxes, g["input"], None, order=o)
        if len(out) == len(dis) and all(np.array_equal(a, b) for a, b in zip(out, dis)):
            ok = True
            assert sboxes == [boxes[j] for j in o]
            break
    assert ok


def test_depth_ordering_on_synthetic_depth():
    """sort_sketch_masks: deepest (largest binned depth mode over the sparse stroke samples) first; a containing box is
    moved in front of a box it contains when their stroke masks overlap."""
    from oracle import refine4_ref as R
    H = W = 200
    rgb = np.full((H, W, 3), 255, np.uint8)
    rgb[20:180, 20:24] = 0; rgb[20:180, 176:180] = 0; rgb[20:24, 20:180] = 0; rgb[176:180, 20:180] = 0    # big frame
    rgb[60:120, 60:64] = 0; rgb[60:120, 116:120] = 0; rgb[60:64, 60:120] = 0; rgb[116:120, 60:120] = 0     # small frame inside
    rgb[140:160, 130:134] = 0
    m_big = np.zeros((H, W), bool); m_big[18:182, 18:182] = True
    m_small = np.zeros((H, W), bool); m_small[58:122, 58:122] = True
    m_bar = np.zeros((H, W), bool); m_bar[138:162, 128:136] = True
    boxes = [[58, 58, 122, 122], [18, 18, 182, 182], [128, 138, 136, 162]]
    depth = np.zeros((H, W), np.float32)
    depth[58:122, 58:122] = 5.0          # the small frame is the DEEPEST by score ...
    depth[138:162, 128:136] = 3.0
    pts = R.sparse_sketch_sample(R.sketch_to_01binary(rgb[..., ::-1]))
    assert 10 < len(pts) < int((R.pil_luma(rgb) < 250).sum())
    ys, xs = np.array(pts).T
    d = np.sqrt((ys[:, None] - ys[None]) ** 2 + (xs[:, None] - xs[None]) ** 2)
    assert d[~np.eye(len(pts), dtype=bool)].min() > H * 0.01            # greedy thinning: no two samples within the radius
    order, scores, cont = R.sort_sketch_masks([m_small, m_big, m_bar], boxes, rgb, depth)
    assert scores[0] == 5.0 and scores[2] == 3.0 and scores[1] == 0.0
    assert cont[1, 0] and cont[1, 2] and not cont[0, 1]
    # by score the order would be [small, bar, big]; the big frame's strokes do not overlap the small frame's by >= 60 %,
    # so containment alone does not reorder them
    assert order == [0, 2, 1]
    # clean_delicate_mask: isolated pixels and line ends with a single neighbour go
    m = np.zeros((9, 9), bool); m[4, 1:8] = True; m[0, 0] = True
    c = R.clean_delicate_mask(m)
    assert not c[0, 0] and not c[4, 1] and not c[4, 7] and c[4, 2:7].all()



def _sparse_sketch_sample_literal(binary_edge_map):
    """depth_sort.py:49-68 as written there (a Python set of point indices + a KD-tree ball query per sample)."""
    from scipy.spatial import KDTree
    radius = binary_edge_map.shape[0] * 0.01
    pts = np.column_stack(np.where(binary_edge_map > 0))
    tree = KDTree(pts)
    sampled, remaining = [], set(range(len(pts)))
    while remaining:
        cur = next(iter(remaining))
        sampled.append(tuple(pts[cur]))
        remaining.difference_update(tree.query_ball_point(pts[cur], radius))
    return sampled


@pytest.mark.parametrize("name", ["bunny_cook_sketch", "animal_hike_sketch", "office_sketch", "clock_lamp_plant",
                                  "fscoco_animals", "mario_bunny", "Clipasso_brushpen_0249"])
def test_fast_host_forms_equal_the_literal_ones(name):
    """the oracle's index-image thinning and shifted-sum neighbour count against the reference's literal forms (set +
    KD-tree; scipy convolve with the 3x3 ring) on the reference's own sketches and cleaned masks: identical output."""
    from scipy import ndimage
    from oracle import refine4_ref as R
    z = np.load(str(Path(__file__).resolve().parent / "golden" / f"refine_{name}.npz"))
    rgb = z["input"]
    binary = R.sketch_to_01binary(rgb[..., ::-1])
    assert R.sparse_sketch_sample(binary) == _sparse_sketch_sample_literal(binary)
    H, W = [int(v) for v in z["hw"]]
    cleaned = np.unpackbits(z["masks_cleaned"], axis=-1)[..., :W].astype(bool)
    k = np.ones((3, 3), dtype=int)
    k[1, 1] = 0
    for m in cleaned[z["masks_cleaned_present"]][:6]:
        want = m.copy()
        want[ndimage.convolve(m.astype(int), k, mode="constant", cval=0) <= 1] = False
        assert np.array_equal(R.clean_delicate_mask(m), want)


def test_window_morphology_equals_scipy():
    """sk_dilate / sk_erode (shifts of the mask's window) against scipy's generic routines with skimage's border
    conventions, on random masks that touch the image border and on every structuring element the stage uses."""
    from scipy import ndimage
    from oracle import refine4_ref as R
    rs = np.random.RandomState(1)
    sts = (R.disk(1), R.disk(2), R.disk(3), ndimage.generate_binary_structure(2, 1), np.ones((3, 3), bool))
    for t in range(150):
        H, W = rs.randint(12, 90), rs.randint(12, 90)
        m = np.zeros((H, W), bool)
        for _ in range(rs.randint(0, 4)):
            y0, x0 = rs.randint(0, H), rs.randint(0, W)
            y1, x1 = min(H, y0 + rs.randint(1, 40)), min(W, x0 + rs.randint(1, 40))
            m[y0:y1, x0:x1] = rs.rand(y1 - y0, x1 - x0) > 0.2
        for st in sts:
            assert np.array_equal(R.sk_erode(m, st), ndimage.binary_erosion(m, structure=st, border_value=1) & m.any())
            assert np.array_equal(R.sk_dilate(m, st), ndimage.binary_dilation(m, structure=st, border_value=0))
