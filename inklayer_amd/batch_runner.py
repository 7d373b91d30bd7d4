"""BASELINE config 5 as a batched job: the whole runner (detector, segmentor, mask cleanup, sketch NMS,
Depth-Anything-V2, refinement) over a list of sketch files, B files per pass of the hot path.

Reference: main.py:27-32 loops `run_inklayer_pipeline` over the files one by one (and InkLayer/runner.py:21-103 reloads
SAM per file).  Here the detector -> segmentor part of B files is ONE pass of InkLayerPipeline (the unit bench.py times;
files of different sizes run as size groups), the engines are the plugin modules' own resident singletons, and every
file then goes through `InkLayer.runner.finish_sketch` - the same code that writes the tree for the per-file entry
point - so the outputs are those of `main.py --dir`.  Image-parallel across ranks by static round-robin, no collectives
(tools/run_dir.py is the launcher)."""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


def _pipeline():
    import InkLayer.detector.gdino as DET
    import InkLayer.segmentor.sam as SEG
    from . import pipeline
    det = DET.get_model()
    seg = SEG._get_engine(SEG.default_ckpt)
    return pipeline.InkLayerPipeline(det, seg)


@torch.no_grad()
def run_files(files: Sequence[str], out_base_dir: str, batch: int = 8, no_intermediate: bool = False,
              stage_s: Optional[Dict[str, float]] = None, pipe=None, top_n: Optional[int] = None) -> List[str]:
    """-> the output directories, one per file, in order.  stage_s (optional) accumulates wall seconds per stage.
    pipe: an InkLayerPipeline to use instead of the plugin singletons' engines; top_n: keep the n best boxes per
    sketch instead of the 0.2 threshold (measurement runs on random weights, which put every score above it)."""
    from PIL import Image
    import InkLayer.runner as R
    import InkLayer.segmentor.sam as SEG
    pipe = pipe or _pipeline()
    R.STAGE_S = {} if stage_s is not None else None
    if pipe.seg.max_batch < batch:
        pipe.seg._alloc(batch)
    outs: List[str] = []

    def tick(name, t0):
        if stage_s is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            stage_s[name] = stage_s.get(name, 0.0) + time.perf_counter() - t0

    for i in range(0, len(files), batch):
        chunk = list(files[i:i + batch])
        t0 = time.perf_counter()
        prepared = [R._prepare_out_dir(f, out_base_dir, wait=False) for f in chunk]      # (out_dir, PIL RGB), input.png queued
        images = [np.asarray(pil) for _, pil in prepared]
        tick("decode + input.png", t0)
        t0 = time.perf_counter()
        results = pipe.run_batch(images, top_n=top_n)                           # threshold path by default, as the plugin
        masks_h = [r.masks.cpu().numpy().view(np.bool_) for r in results]      # 0 / 1 bytes: a zero-copy bool view
        tick("detector + segmentor (batched hot path)", t0)
        t0 = time.perf_counter()
        for (out_dir, pil), r, m in zip(prepared, results, masks_h):
            dino_out = {"bboxes": r.boxes_xyxy_norm.tolist(), "scores": r.scores.tolist(),
                        "labels": ["object"] * len(r.scores)}
            outs.append(R.finish_sketch(out_dir, pil, dino_out, r.boxes_pixel, [m[k] for k in range(m.shape[0])],
                                        no_intermediate=no_intermediate, masks_dev=r.masks, flush_files=False))
        tick("tree + refinement (per file)", t0)
    t0 = time.perf_counter()
    from InkLayer.utils.io import flush
    flush()                                         # every file of every sketch is on disk when the call returns
    tick("waiting for the PNG encoders", t0)
    if stage_s is not None:
        for k, v in R.STAGE_S.items():
            stage_s["  of which " + k] = v
        R.STAGE_S = None
    return outs
