"""Packs the reference's OWN committed pipeline outputs into small fixtures for the refinement rows.

The reference holds 7 complete output sets (output/bunny_cook_sketch, custom_interface/static/outputs/*): for each,
input.png, bboxes.json, masks/ -> masks_cleaned/ -> bboxes_final.json (-> masks_disjoint/ -> masks_final/).  They were
produced by the reference itself with cv2 / skimage (neither importable here), so they are input -> output pairs that
pin an oracle for mask cleanup (mask_cleaner.py:11-36) and sketch NMS (nms_sketch.py:186-351) bit-exactly.
This script only COPIES DATA (pixels and JSON numbers), bit-packed, into tests/golden/refine_<set>.npz.

    python tests/golden/make_refine_golden.py          # build container only (/root/reference)
"""
import glob
import json
import os
from pathlib import Path

import numpy as np
from PIL import Image

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
SETS = [REF / "output" / "bunny_cook_sketch"] + sorted(
    Path(p) for p in glob.glob(str(REF / "custom_interface/static/outputs/*/")) if os.path.isdir(p))


def _stack(dirpath: Path, n: int):
    """masks of one stage as packed bits [n, H, ceil(W/8)] + a flag array telling whether each file exists."""
    arrs, present = [], []
    shape = None
    for i in range(n):
        f = dirpath / f"mask_{i}.png"
        if f.exists():
            a = np.asarray(Image.open(f).convert("L")) > 127
            shape = a.shape
            arrs.append(a)
            present.append(True)
        else:
            arrs.append(None)
            present.append(False)
    if shape is None:
        return np.zeros((0,), np.uint8), np.zeros((0,), bool)
    arrs = [a if a is not None else np.zeros(shape, bool) for a in arrs]
    return np.packbits(np.stack(arrs), axis=-1), np.asarray(present)


def main():
    for d in SETS:
        name = d.name
        inp = np.asarray(Image.open(d / "input.png").convert("RGB"))
        bj = json.loads((d / "bboxes.json").read_text())
        fj = json.loads((d / "bboxes_final.json").read_text())
        n = len(glob.glob(str(d / "masks" / "mask_*.png")))
        assert n == len(bj["bboxes"]), (name, n, len(bj["bboxes"]))
        raw_modes = sorted({Image.open(f).mode for f in glob.glob(str(d / "masks" / "mask_*.png"))})
        out = {"input": inp, "n": np.int64(n), "hw": np.asarray(inp.shape[:2]),
               "bboxes": np.asarray(bj["bboxes"], np.float64), "scores": np.asarray(bj["scores"], np.float64),
               "final_bboxes": np.asarray(fj["bboxes"], np.float64).reshape(-1, 4),
               "final_scores": np.asarray(fj["scores"], np.float64),
               "final_kept": np.asarray(fj["kept_indices"], np.int64), "final_threshold": np.float64(fj["threshold"]),
               "mask_modes": np.asarray(raw_modes)}
        for stage in ("masks", "masks_cleaned", "masks_disjoint", "masks_final"):
            if (d / stage).is_dir():
                packed, present = _stack(d / stage, n)
                out[stage], out[stage + "_present"] = packed, present
        np.savez_compressed(OUT / f"refine_{name}.npz", **out)
        print(name, inp.shape, n, "masks ->", (OUT / f"refine_{name}.npz").stat().st_size >> 10, "KiB",
              "final kept", fj["kept_indices"])


if __name__ == "__main__":
    main()
