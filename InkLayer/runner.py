"""Pipeline orchestration (reference: InkLayer/runner.py:21-103): same signature, same output tree
for the detector -> segmentor part:  <out_base_dir>/<name>/{input.png, bboxes.json, masks/mask_i.png,
segmented_sketch.png, bboxes.png, masks_cleaned/, bboxes_final.json, bboxes_final.png, masks_disjoint/, masks_final/,
depth_map.png, segmented_sketch_final.png}.  Detector, segmentor, mask cleanup, sketch-NMS pair table and the depth model
run on the GPU; inpainting (diffusers) is outside this build's scope and is skipped with a message."""
import os
import shutil

from PIL import Image, ImageDraw

from InkLayer.detector.gdino import run_ft_dino_on_sketch
from InkLayer.segmentor.sam import run_SAM
from InkLayer.utils.io import flush, save_all
from InkLayer.utils.processing import process_dino_output, save_norm_bboxes


def _draw_boxes(input_pil, boxes):
    im = input_pil.copy()
    d = ImageDraw.Draw(im)
    for b in boxes:
        d.rectangle([b[0], b[1], b[2], b[3]], outline=(220, 40, 40), width=2)
    return im


def colour_by_masks(rgb, masks):
    """A plain visualisation (InkLayer/utils/visualization.py is outside the hot path): every mask tints its pixels
    half / half with its own colour, later masks on top.  One label image + one table look-up in uint8 instead of a
    float pass over the image per mask.  rgb: uint8 [H, W, 3]; masks: sequence of [H, W] arrays (non-zero = inside)."""
    import numpy as np
    base = np.asarray(rgb)
    label = np.zeros(base.shape[:2], np.uint16)
    half = np.zeros((len(masks) + 1, 3), np.uint8)
    for i, m in enumerate(masks):
        hue = (i * 0.61803398875) % 1.0
        half[i + 1] = [int(127.5 * (0.6 + 0.4 * abs(((hue * 6 + k) % 6) / 3 - 1))) for k in (0, 4, 2)]
        label[np.asarray(m) != 0] = i + 1
    out = base.copy()
    sel = label > 0
    out[sel] = (base[sel] >> 1) + half[label[sel]]
    return out


def _prepare_out_dir(input_path, out_base_dir, wait=True):
    """runner.py:22-29: <out_base_dir>/<basename before the first dot>, wiped if it has content, input.png saved.
    wait=False: input.png is encoded on the I/O threads (flush() waits for it) - for callers that hand the decoded
    sketch to the later stages instead of letting them re-open the file, as finish_sketch does."""
    input_name = os.path.basename(input_path).split(".")[0]
    input_pil = Image.open(input_path).convert("RGB")
    out_dir = os.path.join(out_base_dir, input_name)
    if os.path.exists(out_dir) and len(os.listdir(out_dir)) > 0:
        shutil.rmtree(out_dir)                                   # reference: `rm -r`
    os.makedirs(out_dir, exist_ok=True)
    save_all([(input_pil, os.path.join(out_dir, "input.png"))], wait=wait)
    return out_dir, input_pil


STAGE_S = None      # measurement hook: a dict here accumulates wall seconds per stage of finish_sketch


def _tick(name, t0):
    if STAGE_S is not None:
        import time
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except ImportError:
            pass
        STAGE_S[name] = STAGE_S.get(name, 0.0) + time.perf_counter() - t0


def finish_sketch(out_dir, input_pil, dino_out, boxes_tensor, masks_np, no_intermediate=False, inpaint=False,
                  masks_dev=None, flush_files=True):
    """Everything of run_inklayer_pipeline after the detector and the segmentor have answered (runner.py:35-101): the
    output tree of the detection stage, then the refinement stage.  Shared by the per-file entry point below and by the
    batched directory runner (inklayer_amd/batch_runner.py), so both write the same tree.  masks_dev (optional): the same
    masks as a uint8 0/1 [n, H, W] CUDA tensor when the caller still has them in HBM (saves the re-upload).
    flush_files=False: the PNG encoding of this sketch may still be running on the I/O threads when the call returns
    (InkLayer.utils.io.flush() waits for it) - the batched runner overlaps it with the next sketch's GPU stages."""
    import time
    import InkLayer.utils.io as _io
    _io.DEFER = True            # the plugin functions below hand their PNGs to the I/O threads and return
    try:
        t0 = time.perf_counter()
        boxes_int = [[int(v) for v in box] for box in boxes_tensor.tolist()]
        save_norm_bboxes(bboxes_list=boxes_int, scores_list=dino_out["scores"], input_pil=input_pil,
                         out_path=os.path.join(out_dir, "bboxes.json"))
        import numpy as np
        masks_dir = os.path.join(out_dir, "masks")
        os.makedirs(masks_dir, exist_ok=True)
        rgb = np.asarray(input_pil)
        save_all([(np.asarray(m, dtype=bool), os.path.join(masks_dir, f"mask_{i}.png")) for i, m in enumerate(masks_np)]  # 1-bit, PIL mode "1"
                 + [(lambda: colour_by_masks(rgb, masks_np), os.path.join(out_dir, "segmented_sketch.png")),
                    (lambda: _draw_boxes(input_pil, boxes_int), os.path.join(out_dir, "bboxes.png"))], wait=False)
        _tick("masks/ + detection visualisations (files)", t0)
        t0 = time.perf_counter()

        # Refinement (runner.py:69-73).  Mask cleanup, sketch NMS, Depth-Anything-V2 and the refinement stage (depth order,
        # disjoint parsing, growth, unlabeled mask) run on the GPU with the cleaned masks staying IN HBM from stage to stage
        # (the files masks_cleaned/, bboxes_final.json, masks_disjoint/, masks_final/ are still written: they are part of the
        # output tree); only the stroke thinning and the raster-order box assignment are host code (inklayer_amd/refine_stage.py).
        from InkLayer.refinement.mask_cleaner import run_clean_masks_on_sketch_dir, clean_masks_on_device
        from InkLayer.refinement.bbox_filter import run_postprocess_boxes_on_sketch_dir
        if masks_dev is not None and len(masks_np):
            from InkLayer.refinement.mask_cleaner import clean_device_masks
            cleaned = clean_device_masks(masks_dev)
        else:
            cleaned = clean_masks_on_device(masks_np)
        if cleaned is None:
            import numpy as _np
            cleaned = _np.zeros((0,) + input_pil.size[::-1], _np.uint8)
        _tick("mask cleanup (GPU)", t0)
        t0 = time.perf_counter()
        run_clean_masks_on_sketch_dir(out_dir, cleaned=cleaned)
        _tick("masks_cleaned/ (D2H + files)", t0)
        t0 = time.perf_counter()
        bbox_out_path = run_postprocess_boxes_on_sketch_dir(out_dir, sketch_iou_thresh=0.2, cleaned_masks=cleaned, sketch_rgb=rgb)
        _tick("sketch NMS (GPU pair table + host loop + files)", t0)
        t0 = time.perf_counter()
        from InkLayer.refinement.refiner import run_refinement_on_sketch_dir
        run_refinement_on_sketch_dir(out_dir, bbox_out_path, cleaned_masks=cleaned, sketch_rgb=rgb)
        _tick("depth + refinement stage + masks_disjoint/ masks_final/ (files)", t0)
        if inpaint:
            print("Inpainting (diffusers) is not part of this build: skipped.")
        else:
            print("Skipping inpainting step as 'inpaint' is set to False.")
    finally:
        _io.DEFER = False
    if flush_files:
        flush()
    if no_intermediate:
        flush()
        keep = {"masks_final", "complete_layers", "complete_layers_rgba", "bboxes_final.json",
                "bboxes_final.png", "segmented_sketch_final.png", "depth_map.png", "input.png"}
        for item in os.listdir(out_dir):
            if item in keep:
                continue
            path = os.path.join(out_dir, item)
            shutil.rmtree(path) if os.path.isdir(path) else os.remove(path)
    return out_dir


def run_inklayer_pipeline(input_path, out_base_dir, no_intermediate=False, inpaint=False):
    out_dir, input_pil = _prepare_out_dir(input_path, out_base_dir, wait=False)      # (finish_sketch flushes)
    # detector -> boxes (runner.py:34-44): JSON gets the int()-truncated pixel boxes, SAM the float ones
    dino_out = run_ft_dino_on_sketch(sketch_path=input_path)
    boxes_tensor, phrases = process_dino_output(dino_out, input_pil)
    # segmentor -> masks (runner.py:49-64)
    input_pil = Image.open(input_path).convert("RGB")
    masks_np = run_SAM(image_pil=input_pil, boxes_filt=boxes_tensor)
    return finish_sketch(out_dir, input_pil, dino_out, boxes_tensor, masks_np, no_intermediate, inpaint)


def run_inpaint_single_layer(request_data, cur_dir, out_dir):
    raise NotImplementedError("layer inpainting (InkLayer/inpainting, diffusers) is outside this build's scope")
