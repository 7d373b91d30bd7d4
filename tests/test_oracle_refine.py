"""Pins oracle/refine_ref.py (mask cleanup + sketch NMS) to the reference's OWN committed outputs:
the 7 sets masks/ -> masks_cleaned/ -> bboxes_final.json that /root/reference holds (tests/golden/refine_*.npz,
copied as data by tests/golden/make_refine_golden.py).  Bit-exact: every cleaned mask, every kept index."""
import glob
from pathlib import Path

import numpy as np
import pytest

GOLD = sorted(glob.glob(str(Path(__file__).resolve().parent / "golden" / "refine_*.npz")))


def _load(path):
    g = np.load(path)
    h, w = (int(v) for v in g["hw"])
    unpack = lambda a: np.unpackbits(a, axis=-1)[..., :w].astype(bool)
    return g, unpack(g["masks"]), unpack(g["masks_cleaned"])


def test_all_seven_reference_sets_are_present():
    assert len(GOLD) == 7


@pytest.mark.parametrize("path", GOLD, ids=lambda p: Path(p).stem)
def test_clean_up_mask_reproduces_reference_masks_cleaned(path):
    from oracle import refine_ref
    g, masks, cleaned = _load(path)
    changed = 0
    for m, c in zip(masks, cleaned):
        out = refine_ref.clean_up_mask(m.astype(np.uint8) * 255)
        assert out.dtype == np.uint8 and set(np.unique(out)) <= {0, 255}
        assert np.array_equal(out > 0, c)
        changed += int((out > 0).sum() != m.sum())
    assert changed > 0                     # the cleanup is not the identity on these sets


@pytest.mark.parametrize("path", GOLD, ids=lambda p: Path(p).stem)
def test_sketch_nms_reproduces_reference_bboxes_final(path):
    from oracle import refine_ref
    g, masks, cleaned = _load(path)
    assert float(g["final_threshold"]) == 0.2
    out = refine_ref.process_json_with_sketch_nms(
        g["input"], {"bboxes": g["bboxes"].tolist(), "scores": g["scores"].tolist()},
        [c.astype(np.uint8) * 255 for c in cleaned], 0.2)
    assert out["kept_indices"] == g["final_kept"].tolist()
    assert np.array_equal(np.asarray(out["bboxes"]).reshape(-1, 4), g["final_bboxes"])
    assert np.array_equal(np.asarray(out["scores"]), g["final_scores"])
    assert len(out["kept_indices"]) < int(g["n"])      # NMS removed something in every set


def test_kernel_size_rule():
    from oracle import refine_ref
    assert refine_ref.calculate_kernel_size((750, 750)) == 19      # int(18.75) = 18 -> even -> 19
    assert refine_ref.calculate_kernel_size((1024, 1024)) == 25
    assert refine_ref.calculate_kernel_size((512, 800)) == 13      # int(12.8) = 12 -> 13


def test_luma_formulas_match_pillow():
    """PIL Image.convert("L") is the definition of the stroke pixels of the sketch NMS (luma < 250): the oracle's, the
    host stage's and (through tests/test_refine_gpu.py) the GPU kernel's fixed-point formula must reproduce Pillow."""
    from PIL import Image
    from oracle import refine_ref
    from oracle import refine4_ref as refine_host
    rs = np.random.RandomState(1)
    rgb = rs.randint(0, 256, size=(257, 301, 3)).astype(np.uint8)
    rgb[:8, :8] = [[250, 250, 250]]; rgb[8:16, :8] = [[249, 250, 251]]       # values around the 250 threshold
    want = np.asarray(Image.fromarray(rgb).convert("L"))
    assert np.array_equal(refine_ref.pil_luma(rgb), want) and np.array_equal(refine_host.pil_luma(rgb), want)
    assert np.array_equal(refine_ref.png_gray(rgb), refine_host.png_gray(rgb))
