"""Depth model plugin (reference: InkLayer/refinement/depth_sort.py:35-46), MI355X engine underneath.

`get_depth_map(sketch_path) -> HxW float32 numpy` = DepthAnythingV2("vitb").infer_image(cv2.imread(sketch_path)).
The model is a lazily created singleton kept resident in HBM (the reference builds it at import).  The mask-ordering
logic (sort_sketch_masks and helpers, depth_sort.py:49-295) is host code, as in the reference: inklayer_amd/refine_host.py."""
import os

import numpy as np
from PIL import Image

from InkLayer.utils.paths import get_model_path

encoder = "vitb"
_engine = None


def _get_engine():
    global _engine
    if _engine is None:
        from inklayer_amd import depth
        if os.environ.get("INKLAYER_RANDOM_WEIGHTS") == "1":          # no checkpoints exist offline
            import torch
            cfg = depth.DepthConfig()
            from inklayer_amd import weights_init
            _engine = depth.DepthEngine(weights_init.random_depth_state_dict(cfg, "cuda"), cfg, "cuda")
        else:
            path = get_model_path(f"depth_anything_v2_{encoder}.pth")
            if not os.path.exists(path):
                raise FileNotFoundError(f"Checkpoint not found at {path}")
            _engine = depth.build_depth(path)
    return _engine


def get_depth_map(sketch_path):
    rgb = np.asarray(Image.open(sketch_path).convert("RGB"))
    bgr = np.ascontiguousarray(rgb[..., ::-1])                        # cv2.imread returns BGR
    return _get_engine().infer_image(bgr).cpu().numpy()


def sort_sketch_masks(masks, bboxes, sketch_path, depth_sketch=None):
    """depth_sort.py:244-295: -> (order deepest first, depth scores, containment graph)."""
    from inklayer_amd import refine_host
    assert os.path.exists(sketch_path), f"Sketch path {sketch_path} does not exist."
    if depth_sketch is None:
        depth_sketch = get_depth_map(sketch_path)
    rgb = np.asarray(Image.open(sketch_path).convert("RGB"))
    return refine_host.sort_sketch_masks([np.asarray(m) > 0 for m in masks], bboxes, rgb, depth_sketch)
