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
from InkLayer.utils.io import save_all
from InkLayer.utils.processing import process_dino_output, save_norm_bboxes


def _draw_boxes(input_pil, boxes):
    im = input_pil.copy()
    d = ImageDraw.Draw(im)
    for b in boxes:
        d.rectangle([b[0], b[1], b[2], b[3]], outline=(220, 40, 40), width=2)
    return im


def _colour_by_masks(input_pil, masks_pils):
    import numpy as np
    base = np.asarray(input_pil).astype("float32")
    out = base.copy()
    for i, m in enumerate(masks_pils):
        hue = (i * 0.61803398875) % 1.0
        col = 255.0 * np.array([0.6 + 0.4 * abs(((hue * 6 + k) % 6) / 3 - 1) for k in (0, 4, 2)], dtype="float32")
        sel = np.asarray(m, dtype=bool)
        out[sel] = 0.5 * base[sel] + 0.5 * col
    return Image.fromarray(out.clip(0, 255).astype("uint8"))


def _prepare_out_dir(input_path, out_base_dir):
    """runner.py:22-29: <out_base_dir>/<basename before the first dot>, wiped if it has content, input.png saved."""
    input_name = os.path.basename(input_path).split(".")[0]
    input_pil = Image.open(input_path).convert("RGB")
    out_dir = os.path.join(out_base_dir, input_name)
    if os.path.exists(out_dir) and len(os.listdir(out_dir)) > 0:
        shutil.rmtree(out_dir)                                   # reference: `rm -r`
    os.makedirs(out_dir, exist_ok=True)
    input_pil.save(os.path.join(out_dir, "input.png"))
    return out_dir, input_pil


def finish_sketch(out_dir, input_pil, dino_out, boxes_tensor, masks_np, no_intermediate=False, inpaint=False):
    """Everything of run_inklayer_pipeline after the detector and the segmentor have answered (runner.py:35-101): the
    output tree of the detection stage, then the refinement stage.  Shared by the per-file entry point below and by the
    batched directory runner (inklayer_amd/batch_runner.py), so both write the same tree."""
    boxes_int = [[int(v) for v in box] for box in boxes_tensor.tolist()]
    save_norm_bboxes(bboxes_list=boxes_int, scores_list=dino_out["scores"], input_pil=input_pil,
                     out_path=os.path.join(out_dir, "bboxes.json"))
    masks_pils = [Image.fromarray(m) for m in masks_np]
    masks_dir = os.path.join(out_dir, "masks")
    os.makedirs(masks_dir, exist_ok=True)
    save_all([(m, os.path.join(masks_dir, f"mask_{i}.png")) for i, m in enumerate(masks_pils)]       # PIL mode "1"
              + [(_colour_by_masks(input_pil, masks_pils), os.path.join(out_dir, "segmented_sketch.png")),
                 (_draw_boxes(input_pil, boxes_int), os.path.join(out_dir, "bboxes.png")),
                 (input_pil, os.path.join(out_dir, "input.png"))])

    # Refinement (runner.py:69-73).  Mask cleanup, sketch NMS, Depth-Anything-V2 and the refinement stage (depth order,
    # disjoint parsing, growth, unlabeled mask) run on the GPU with the cleaned masks staying IN HBM from stage to stage
    # (the files masks_cleaned/, bboxes_final.json, masks_disjoint/, masks_final/ are still written: they are part of the
    # output tree); only the stroke thinning and the raster-order box assignment are host code (inklayer_amd/refine_stage.py).
    from InkLayer.refinement.mask_cleaner import run_clean_masks_on_sketch_dir, clean_masks_on_device
    from InkLayer.refinement.bbox_filter import run_postprocess_boxes_on_sketch_dir
    cleaned = clean_masks_on_device(masks_np)
    if cleaned is None:
        import numpy as _np
        cleaned = _np.zeros((0,) + input_pil.size[::-1], _np.uint8)
    run_clean_masks_on_sketch_dir(out_dir, cleaned=cleaned)
    bbox_out_path = run_postprocess_boxes_on_sketch_dir(out_dir, sketch_iou_thresh=0.2, cleaned_masks=cleaned)
    from InkLayer.refinement.refiner import run_refinement_on_sketch_dir
    run_refinement_on_sketch_dir(out_dir, bbox_out_path, cleaned_masks=cleaned)
    if inpaint:
        print("Inpainting (diffusers) is not part of this build: skipped.")
    else:
        print("Skipping inpainting step as 'inpaint' is set to False.")
    if no_intermediate:
        keep = {"masks_final", "complete_layers", "complete_layers_rgba", "bboxes_final.json",
                "bboxes_final.png", "segmented_sketch_final.png", "depth_map.png", "input.png"}
        for item in os.listdir(out_dir):
            if item in keep:
                continue
            path = os.path.join(out_dir, item)
            shutil.rmtree(path) if os.path.isdir(path) else os.remove(path)
    return out_dir


def run_inklayer_pipeline(input_path, out_base_dir, no_intermediate=False, inpaint=False):
    out_dir, input_pil = _prepare_out_dir(input_path, out_base_dir)
    # detector -> boxes (runner.py:34-44): JSON gets the int()-truncated pixel boxes, SAM the float ones
    dino_out = run_ft_dino_on_sketch(sketch_path=input_path)
    boxes_tensor, phrases = process_dino_output(dino_out, input_pil)
    # segmentor -> masks (runner.py:49-64)
    input_pil = Image.open(input_path).convert("RGB")
    masks_np = run_SAM(image_pil=input_pil, boxes_filt=boxes_tensor)
    return finish_sketch(out_dir, input_pil, dino_out, boxes_tensor, masks_np, no_intermediate, inpaint)


def run_inpaint_single_layer(request_data, cur_dir, out_dir):
    raise NotImplementedError("layer inpainting (InkLayer/inpainting, diffusers) is outside this build's scope")
