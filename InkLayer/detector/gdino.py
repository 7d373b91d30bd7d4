"""Detector plugin (reference: InkLayer/detector/gdino.py:12-30), MI355X engine underneath.

`run_ft_dino_on_sketch(path) -> {"bboxes": normalised xyxy lists, "scores": [...], "labels": [...]}`
with caption "object", box_threshold 0.2, text_threshold 0 exactly as the reference.  The model is a
module-level singleton like the reference's `model`, but created on first use and kept resident in
HBM (the reference re-`.to(device)`s it every call, GD/util/inference.py:64)."""
import os

import numpy as np
from PIL import Image

from InkLayer.utils.paths import get_model_path
from InkLayer.utils.processing import cxcywh_to_xyxy

gdino_config_path = get_model_path("GroundingDINO_SwinT_OGC.py")
weights_path = get_model_path("inklayer_gdino.pth")
model = None


def load_model(config_path=None, checkpoint_path=None, device="cuda"):
    """groundingdino.util.inference.load_model (GD/util/inference.py:29-36) for the HIP engine.  The defaults are the
    module attributes `gdino_config_path` / `weights_path`, read at CALL time."""
    import torch
    checkpoint_path = weights_path if checkpoint_path is None else checkpoint_path
    from inklayer_amd import gdino, text_branch, weights_init
    cfg = gdino.GDinoConfig()
    if os.environ.get("INKLAYER_RANDOM_WEIGHTS") == "1":      # no checkpoints exist offline
        sd = weights_init.random_gdino_state_dict(cfg, device)
        text = weights_init.random_text_features(cfg, device)
    else:
        if not os.path.exists(checkpoint_path):
            raise FileNotFoundError(f"Checkpoint not found at {checkpoint_path}")
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        sd = gdino.clean_state_dict(ckpt["model"] if "model" in ckpt else ckpt)   # GD/util/inference.py:33-34
        text = text_branch.encode_caption_from_checkpoint(sd, gdino.DEFAULT_TOKEN_IDS)
    return gdino.GDinoEngine(sd, cfg, device, encoded_text=text)


def get_model():
    global model
    if model is None:
        model = load_model()
    return model


def run_ft_dino_on_sketch(sketch_path):
    import torch
    from inklayer_amd import gdino
    eng = get_model()
    image_source = np.asarray(Image.open(sketch_path).convert("RGB"))
    from inklayer_amd import ops
    raw = torch.from_numpy(np.ascontiguousarray(image_source)).to(eng.dev)
    oh, ow = gdino.resize_shape(image_source.shape[1], image_source.shape[0])   # load_image's RandomResize([800], 1333)
    boxes, scores = eng.detect([ops.resize_bilinear_u8(raw, oh, ow)])[0]
    normalized = cxcywh_to_xyxy(boxes.tolist()).tolist()        # cxcywh -> xyxy in float64
    return {"bboxes": normalized, "scores": scores.tolist(), "labels": ["object"] * len(normalized)}
