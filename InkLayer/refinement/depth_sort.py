"""Depth model plugin + mask ordering (reference: InkLayer/refinement/depth_sort.py:35-46, 244-295), MI355X underneath.

`get_depth_map(sketch_path) -> HxW float32 numpy` = DepthAnythingV2("vitb").infer_image(cv2.imread(sketch_path)); the
model is a lazily created singleton kept resident in HBM (the reference builds it at import).
`sort_sketch_masks(masks, bboxes, sketch_path, depth_sketch=None) -> (order deepest first, depth scores, containment)`:
the pixel work (stroke / mask intersections, dilated overlaps, depth samples) runs on the GPU
(inklayer_amd/refine_stage.py), the ordering of a few dozen numbers on the host."""
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
            cfg = depth.DepthConfig()
            from inklayer_amd import weights_init
            _engine = depth.DepthEngine(weights_init.random_depth_state_dict(cfg, "cuda"), cfg, "cuda")
        else:
            path = get_model_path(f"depth_anything_v2_{encoder}.pth")
            if not os.path.exists(path):
                raise FileNotFoundError(f"Checkpoint not found at {path}")
            _engine = depth.build_depth(path)
    return _engine


def get_depth_map_device(sketch_path, sketch_rgb=None):
    """The depth map as a float32 [H, W] tensor that stays on the GPU (what the refinement stage consumes).
    sketch_rgb (optional): the decoded sketch when the caller still holds it; None: sketch_path is read."""
    rgb = np.asarray(Image.open(sketch_path).convert("RGB")) if sketch_rgb is None else sketch_rgb
    bgr = np.ascontiguousarray(rgb[..., ::-1])                        # cv2.imread returns BGR
    return _get_engine().infer_image(bgr)


def get_depth_map(sketch_path):
    return get_depth_map_device(sketch_path).cpu().numpy()


def sort_sketch_masks(masks, bboxes, sketch_path, depth_sketch=None):
    import torch
    from inklayer_amd import refine_stage
    assert os.path.exists(sketch_path), f"Sketch path {sketch_path} does not exist."
    rgb = np.asarray(Image.open(sketch_path).convert("RGB"))
    depth = get_depth_map_device(sketch_path) if depth_sketch is None else \
        torch.from_numpy(np.ascontiguousarray(depth_sketch, np.float32)).to("cuda")
    stack = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(m) > 0 for m in masks]).astype(np.uint8))).to("cuda") \
        if len(masks) else torch.zeros((0,) + rgb.shape[:2], dtype=torch.uint8, device="cuda")
    return refine_stage.refine_masks(stack, [list(b) for b in bboxes], rgb, depth.contiguous(), only_order=True)
