"""§8(f)-1 on the GPU: ink_mask_cleanup and ink_mask_sketch_iou_counts + the NMS host loop, bit-exact against
(a) the reference's own committed outputs (tests/golden/refine_*.npz) and (b) oracle/refine_ref.py on seeded random
masks incl. the edge cases (empty / full masks, single pixels, widths that are no multiple of 64, tiny images)."""
import glob
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(str(Path(__file__).resolve().parent / "golden" / "refine_*.npz")))


def _load(path):
    g = np.load(path)
    h, w = (int(v) for v in g["hw"])
    unpack = lambda a: np.unpackbits(a, axis=-1)[..., :w].astype(bool)
    return g, unpack(g["masks"]), unpack(g["masks_cleaned"])


@pytest.mark.parametrize("path", GOLD, ids=lambda p: Path(p).stem)
def test_cleanup_and_nms_reproduce_the_reference_outputs(dev, path):
    from inklayer_amd import refine
    g, masks, cleaned = _load(path)
    m01 = torch.from_numpy(masks.astype(np.uint8)).to(dev)
    got = refine.clean_segmentor_masks(m01)                 # the segmentor's 0/1 bytes, all masks in one call
    assert int(got._ink_overflow_flag.item()) == 0
    out = got.cpu().numpy()
    assert set(np.unique(out)) <= {0, 255}
    assert np.array_equal(out > 0, cleaned), [int(((out[i] > 0) != cleaned[i]).sum()) for i in range(len(out))]
    res = refine.process_json_with_sketch_nms(g["input"], {"bboxes": g["bboxes"].tolist(), "scores": g["scores"].tolist()},
                                              got, 0.2)
    assert res["kept_indices"] == g["final_kept"].tolist()
    assert np.array_equal(np.asarray(res["bboxes"]).reshape(-1, 4), g["final_bboxes"])


def _random_masks(rs, n, H, W):
    """blobs + salt noise + lines: many small components, some below / above the area threshold, thin long ones"""
    from scipy import ndimage
    ms = []
    for i in range(n):
        f = ndimage.gaussian_filter(rs.standard_normal((H, W)), sigma=rs.uniform(2, 12))
        m = f > np.quantile(f, rs.uniform(0.5, 0.97))
        m |= rs.rand(H, W) < rs.choice([0.0, 0.0005, 0.003])
        if i % 3 == 0:
            y = rs.randint(0, H)
            m[y:y + 1, rs.randint(0, W // 2):] = True               # a 1-px line reaching the right border
        ms.append(m)
    ms[0][:] = False                                                # empty
    if n > 1:
        ms[1][:] = True                                             # full
    if n > 2:
        ms[2][:] = False
        ms[2][0, 0] = ms[2][H - 1, W - 1] = ms[2][H // 2, W // 2] = True    # isolated pixels incl. corners
    return np.stack(ms)


@pytest.mark.parametrize("H,W", [(750, 750), (1024, 1024), (333, 517), (64, 64), (40, 200), (129, 63), (512, 1000)])
def test_cleanup_matches_oracle_on_random_masks(dev, H, W):
    from oracle import refine_ref
    from inklayer_amd import refine
    rs = np.random.RandomState(H * 7 + W)
    masks = _random_masks(rs, 7, H, W)
    got = refine.clean_masks(torch.from_numpy(masks.astype(np.uint8) * 255).to(dev))
    assert int(got._ink_overflow_flag.item()) == 0
    out = got.cpu().numpy()
    for i in range(len(masks)):
        want = refine_ref.clean_up_mask(masks[i].astype(np.uint8) * 255)
        assert np.array_equal(out[i], want), (i, int((out[i] != want).sum()))
    # gray levels: cv2.threshold at 127 (128 is foreground, 127 is not)
    gray = (masks[3].astype(np.uint8) * 128) + (~masks[3]).astype(np.uint8) * 127
    g2 = refine.clean_masks(torch.from_numpy(gray[None]).to(dev)).cpu().numpy()[0]
    assert np.array_equal(g2, out[3])


def test_sketch_iou_counts_and_nms_match_oracle(dev):
    from oracle import refine_ref
    from inklayer_amd import ops, refine, synthetic
    H, W, n = 600, 750, 9
    rs = np.random.RandomState(5)
    rgb = synthetic.synthetic_sketch(3, H, W)
    masks = np.stack([refine_ref.clean_up_mask(m.astype(np.uint8) * 255) for m in _random_masks(rs, n, H, W)])
    dm = torch.from_numpy(masks).to(dev)
    counts = ops.mask_sketch_iou_counts(dm, torch.from_numpy(rgb).to(dev)).cpu().numpy()
    S = refine_ref.sketch_pixels(rgb)
    for i in range(n):
        for j in range(n):
            a, b = (masks[i] > 0) & S, (masks[j] > 0) & S
            assert counts[i, j, 0] == int((a & b).sum()) and counts[i, j, 1] == int((a | b).sum())
    # nested / corner-sharing boxes so that the NMS has decisions to make
    boxes = np.array([[0.1, 0.1, 0.6, 0.6], [0.1, 0.1, 0.35, 0.4], [0.5, 0.5, 0.9, 0.9], [0.1, 0.1, 0.6, 0.59],
                      [0.0, 0.0, 1.0, 1.0], [0.7, 0.1, 0.9, 0.3], [0.7, 0.1, 0.9, 0.31], [0.3, 0.35, 0.6, 0.6],
                      [0.52, 0.5, 0.9, 0.88]])
    scores = rs.uniform(0.2, 0.9, size=n)
    for thr in (0.05, 0.2, 0.5):
        want = refine_ref.sketch_nms(rgb, boxes, scores, list(masks), thr)
        got = refine.sketch_nms(rgb, boxes, scores, dm, thr)
        assert np.array_equal(got, want), (thr, got, want)
    assert len(refine.sketch_nms(rgb, np.zeros((0, 4)), np.zeros((0,)), dm[:0], 0.2)) == 0
