"""BASELINE config 5 plumbing on the CPU: the batched directory runner (inklayer_amd/batch_runner.py) and its launcher
(tools/run_dir.py) with the GPU plugins replaced by deterministic stand-ins - chunking into batches, mixed sizes, the
output trees, and a 2-rank image-parallel run (static round-robin shard, fresh child processes on 127.0.0.1)."""
import json
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent

FAKES = textwrap.dedent('''
    import numpy as np, torch
    from PIL import Image
    import InkLayer.runner as R
    import InkLayer.refinement.mask_cleaner as MC
    import InkLayer.refinement.bbox_filter as BF
    import InkLayer.refinement.refiner as RF
    from inklayer_amd import batch_runner, refine_stage, pipeline
    from oracle import refine4_ref

    class FakePipe:
        class seg: max_batch = 64
        calls = []
        def run_batch(self, images, top_n=None):
            FakePipe.calls.append([im.shape[:2] for im in images])
            out = []
            for im in images:
                H, W = im.shape[:2]
                xyxy = np.array([[0.1, 0.2, 0.5, 0.6], [0.3333, 0.25, 0.9, 0.8]])
                pix = pipeline.boxes_to_pixels(xyxy, W, H)
                m = torch.zeros((2, H, W), dtype=torch.uint8)
                for k, b in enumerate(pix.tolist()):
                    m[k, int(b[1]):int(b[3]), int(b[0]):int(b[2])] = 1
                out.append(pipeline.SketchResult(xyxy, np.array([0.9, 0.4]), pix, m))
            return out

    def install():
        batch_runner._pipeline = lambda: FakePipe()
        MC.clean_masks_on_device = lambda masks: np.stack([np.asarray(m, dtype=np.uint8) * 255 for m in masks])
        MC.clean_device_masks = lambda m: m.numpy() * 255
        BF.process_json_with_sketch_NMS = lambda sp, md, d, iou_threshold=0.2, cleaned_masks=None, sketch_rgb=None: {
            "bboxes": d["bboxes"][:1], "scores": d["scores"][:1], "kept_indices": [0], "threshold": iou_threshold}
        RF.get_depth_map_device = lambda path, sketch_rgb=None: torch.from_numpy(np.tile(
            np.linspace(0, 3, Image.open(path).size[0], dtype=np.float32), (Image.open(path).size[1], 1)))
        RF._stack_on_gpu = lambda masks, shape: torch.from_numpy(np.stack([(np.asarray(m) > 0) for m in masks]).astype(np.uint8))

        def fake_stage(masks, boxes, rgb, depth, **kw):
            ms = [m.numpy() * 255 for m in masks]
            dis, sboxes, info = refine4_ref.parse_masks_to_disjoint_masks(ms, boxes, rgb, depth.numpy())
            fin = refine4_ref.improve_sam_masks(rgb, dis, sboxes)
            lab = lambda lst: sum(((np.asarray(m) > 0).astype(np.uint8) * (i + 1) for i, m in enumerate(lst)),
                                  np.zeros(rgb.shape[:2], np.uint8))
            extra = np.asarray(fin[-1]) > 0 if len(fin) > len(dis) else None
            return refine_stage.RefineResult([], [], sboxes, lab(dis), len(dis), info, lab(fin[:len(dis)]), extra)
        refine_stage.refine_masks = fake_stage
''')

TREE = ["bboxes.json", "bboxes.png", "bboxes_final.json", "bboxes_final.png", "depth_map.png", "input.png", "masks",
        "masks_cleaned", "masks_disjoint", "masks_final", "segmented_sketch.png", "segmented_sketch_final.png"]


def _make_dir(d, sizes):
    d.mkdir()
    for i, (w, h) in enumerate(sizes):
        a = np.full((h, w, 3), 255, np.uint8)
        a[h // 3: h // 3 + 3, 5: w - 5] = 0
        a[5: h - 5, w // 2: w // 2 + 2] = 0
        Image.fromarray(a).save(d / f"sk{i}.v1.png")


@pytest.fixture
def restore_plugins():
    """The stand-ins are installed by plain assignment (the same text runs in the child processes of the launcher test):
    put the real attributes back afterwards."""
    import InkLayer.runner as R
    import InkLayer.refinement.mask_cleaner as MC
    import InkLayer.refinement.bbox_filter as BF
    import InkLayer.refinement.refiner as RF
    from inklayer_amd import batch_runner, refine_stage
    saved = [(m, k, getattr(m, k)) for m, k in ((batch_runner, "_pipeline"), (MC, "clean_masks_on_device"), (MC, "clean_device_masks"),
                                                (BF, "process_json_with_sketch_NMS"), (RF, "get_depth_map_device"),
                                                (RF, "_stack_on_gpu"), (refine_stage, "refine_masks"),
                                                (R, "run_ft_dino_on_sketch"), (R, "run_SAM"))]
    yield
    for m, k, v in saved:
        setattr(m, k, v)


def test_batch_runner_chunks_and_writes_the_runner_tree(tmp_path, restore_plugins):
    ns = {}
    exec(FAKES, ns)
    ns["install"]()
    from inklayer_amd import batch_runner
    sizes = [(80, 60), (64, 64), (80, 60), (100, 50), (64, 64)]
    _make_dir(tmp_path / "in", sizes)
    files = sorted(str(p) for p in (tmp_path / "in").glob("*.png"))
    stages = {}
    outs = batch_runner.run_files(files, str(tmp_path / "out"), batch=2, stage_s=stages)
    assert [Path(o).name for o in outs] == [f"sk{i}" for i in range(5)]
    assert ns["FakePipe"].calls == [[(60, 80), (64, 64)], [(60, 80), (50, 100)], [(64, 64)]]      # batches of 2, in order
    for o, (w, h) in zip(outs, sizes):
        assert sorted(p.name for p in Path(o).iterdir()) == TREE
        d = json.loads((Path(o) / "bboxes.json").read_text())
        assert len(d["bboxes"]) == 2 and d["scores"] == [0.9, 0.4]
        m0 = Image.open(Path(o) / "masks" / "mask_0.png")
        assert m0.mode == "1" and m0.size == (w, h)
        assert len(list((Path(o) / "masks_final").iterdir())) >= 1
    assert {"decode + input.png", "detector + segmentor (batched hot path)", "tree + refinement (per file)"} <= set(stages)
    assert any(k.startswith("  of which ") for k in stages)
    # same tree as the per-file entry point on the same stand-ins
    import InkLayer.runner as R
    R.run_ft_dino_on_sketch = lambda sketch_path: {"bboxes": [[0.1, 0.2, 0.5, 0.6], [0.3333, 0.25, 0.9, 0.8]],
                                                   "scores": [0.9, 0.4], "labels": ["object"] * 2}

    def fake_sam(image_pil, boxes_filt):
        W, H = image_pil.size
        ms = []
        for b in boxes_filt.tolist():
            m = np.zeros((H, W), dtype=bool)
            m[int(b[1]):int(b[3]), int(b[0]):int(b[2])] = True
            ms.append(m)
        return ms
    R.run_SAM = fake_sam
    solo = R.run_inklayer_pipeline(files[3], str(tmp_path / "solo"))
    for sub in ("masks", "masks_cleaned", "masks_disjoint", "masks_final"):
        a, b = sorted((Path(outs[3]) / sub).iterdir()), sorted((Path(solo) / sub).iterdir())
        assert [p.name for p in a] == [p.name for p in b]
        assert all(np.array_equal(np.asarray(Image.open(x)), np.asarray(Image.open(y))) for x, y in zip(a, b))
    assert json.loads((Path(outs[3]) / "bboxes.json").read_text()) == json.loads((Path(solo) / "bboxes.json").read_text())


def test_run_dir_launcher_two_ranks(tmp_path):
    """tools/run_dir.py --gpus 2 semantics: two fresh ranks (RANK / WORLD_SIZE in the environment), files i % 2 == rank."""
    from inklayer_amd import dist as idist
    sizes = [(80, 60), (64, 64), (70, 50), (100, 50), (64, 72)]
    _make_dir(tmp_path / "in", sizes)
    child = tmp_path / "child.py"
    child.write_text(f"import sys\nsys.path.insert(0, {str(ROOT)!r})\n" + FAKES + textwrap.dedent(f'''
        install()
        sys.path.insert(0, {str(ROOT / "tools")!r})
        import run_dir
        sys.argv = ["run_dir.py", "--dir", {str(tmp_path / "in")!r}, "--out_dir", {str(tmp_path / "out")!r}, "--batch", "2"]
        run_dir.main()
    '''))
    rc, out0 = idist.launch_ranks([sys.executable, str(child)], 2, timeout_s=300)
    assert rc == 0, out0
    assert "[rank 0/2] 3 of 5 sketches" in out0
    done = sorted(p.name for p in (tmp_path / "out").iterdir())
    assert done == [f"sk{i}" for i in range(5)]                    # both ranks' shares are there
    for i in range(5):
        assert sorted(p.name for p in (tmp_path / "out" / f"sk{i}").iterdir()) == TREE
