"""Segmentor plugin (reference: InkLayer/segmentor/sam.py:16-43), MI355X engine underneath.

`run_SAM(image_pil, boxes_filt, sam_checkpoint=default_ckpt) -> list of HxW bool arrays`."""
import os

import torch

from InkLayer.utils.paths import get_model_path

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
default_ckpt = get_model_path("sam_vit_h_4b8939.pth")
_engine = None


def _get_engine(sam_checkpoint):
    global _engine
    from inklayer_amd import sam, weights_init
    if os.environ.get("INKLAYER_RANDOM_WEIGHTS") == "1":
        if _engine is None:
            cfg = sam.SamConfig()
            _engine = sam.SamEngine(weights_init.random_sam_state_dict(cfg, "cuda"), cfg, "cuda")
        return _engine
    if not os.path.exists(sam_checkpoint):
        raise FileNotFoundError(f"Checkpoint not found at {sam_checkpoint}")   # reference: print + breakpoint()
    return sam.build_sam(sam_checkpoint)


def run_SAM(image_pil, boxes_filt, sam_checkpoint=default_ckpt):
    from inklayer_amd import sam
    if len(boxes_filt) == 0:
        return []                                               # reference returns a 2-tuple of empty arrays
    return sam.run_SAM(image_pil, boxes_filt, engine=_get_engine(sam_checkpoint))
