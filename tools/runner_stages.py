"""Per-stage wall time of the whole runner (BASELINE config 5's stages) on generated sketches, random weights.
usage: INKLAYER_RANDOM_WEIGHTS=1 python tools/runner_stages.py [n_sketches]"""
import os, sys, time, tempfile
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ.setdefault("INKLAYER_RANDOM_WEIGHTS", "1")
from PIL import Image
from inklayer_amd import synthetic
import InkLayer.detector.gdino as DET
import InkLayer.segmentor.sam as SEG
import InkLayer.refinement.mask_cleaner as MC
import InkLayer.refinement.bbox_filter as BF
import InkLayer.refinement.refiner as RF
import InkLayer.refinement.depth_sort as DS
from InkLayer.utils import processing as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tmp = Path(tempfile.mkdtemp())
eng = DET.get_model()
eng.w["dec.norm.w"].mul_(0.05); eng.w["dec.norm.b"].mul_(0.05)     # un-saturate the random-weight scores
T = {}
def tick(name, t0):
    torch.cuda.synchronize()
    T.setdefault(name, []).append(time.perf_counter() - t0)
for i in range(n + 1):
    png = tmp / f"s{i}.png"
    Image.fromarray(synthetic.synthetic_sketch(i, 750, 750)).save(png)
    pil = Image.open(png).convert("RGB")
    t0 = time.perf_counter(); d = DET.run_ft_dino_on_sketch(str(png)); tick("detector", t0)
    keep = np.argsort(-np.asarray(d["scores"]))[:16]                  # random weights keep ~900 boxes: take 16
    d = {k: [v[j] for j in keep] for k, v in d.items()}
    boxes, _ = P.process_dino_output(d, pil)
    t0 = time.perf_counter(); masks = SEG.run_SAM(pil, boxes); tick("sam (16 boxes)", t0)
    t0 = time.perf_counter(); cleaned = MC.clean_masks_in_memory(masks); tick("mask cleanup (GPU, in memory)", t0)
    from inklayer_amd import refine as _rf
    noisy = torch.from_numpy((np.random.RandomState(i).rand(16, 750, 750) < 0.5).astype(np.uint8) * 255).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); _rf.clean_masks(noisy); tick("mask cleanup of 16 salt-and-pepper masks (device resident)", t0)
    blob = torch.from_numpy(np.stack(masks).astype(np.uint8) * 255).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); _rf.clean_masks(blob); tick("mask cleanup of the 16 SAM masks (device resident)", t0)
    out = tmp / f"o{i}"; (out / "masks_cleaned").mkdir(parents=True); pil.save(out / "input.png")
    W, H = pil.size
    inp = {"bboxes": [[int(v[0]) / W, int(v[1]) / H, int(v[2]) / W, int(v[3]) / H] for v in boxes.tolist()], "scores": d["scores"]}
    t0 = time.perf_counter(); fin = BF.process_json_with_sketch_NMS(str(out / "input.png"), "", inp, 0.2, cleaned_masks=cleaned); tick("sketch NMS (GPU pair table)", t0)
    t0 = time.perf_counter(); depth = DS.get_depth_map(str(out / "input.png")); tick("depth (GPU)", t0)
    from inklayer_amd import refine_stage
    rgb = np.asarray(pil)
    cleaned_dev = torch.from_numpy(cleaned).cuda()
    depth_dev = torch.from_numpy(depth).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bx = refine_stage.to_pixel_boxes(fin["bboxes"], H, W)
    sel = cleaned_dev[torch.as_tensor(fin["kept_indices"], dtype=torch.long, device="cuda")].contiguous()
    res = refine_stage.refine_masks(sel, bx, rgb, depth_dev)
    tick("refinement stage (GPU + host natives)", t0)
    for k, v in res.timings.items():
        T.setdefault("   of which " + k, []).append(v if "pixels" not in k else v * 1e-3)
print(f"{n} sketches of 750x750 (first one discarded as warm-up); seconds per sketch:")
for k, v in T.items():
    print(f"  {k:40s} {np.median(v[1:]) * 1e3:9.1f} ms")
