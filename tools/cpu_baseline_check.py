"""Validates bench.py's COMPOSED cpu_baseline (1 windowed + 1 global ViT-H block timed, encoder = 28 + 4 of them) once
against a full run of the same oracle on the same box: GroundingDINO in full + the 32-block ViT-H + 16 boxes through the
mask decoder and postprocess, one synthetic 1024x1024 sketch, weights resident.  Prints both and their ratio.
    python tools/cpu_baseline_check.py > profiles/rNN_cpu_baseline_check.txt"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from inklayer_amd import gdino as pgd, sam as psam, synthetic, weights_init
from oracle import gdino_ref, sam_ref

composed = bench.cpu_baseline(16)
ncpu = bench._cpu_share()
torch.set_num_threads(ncpu)
img = synthetic.synthetic_sketch(0)
ssd = weights_init.random_sam_state_dict(psam.SamConfig(), "cpu", 0)
gsd = weights_init.random_gdino_state_dict(pgd.GDinoConfig(), "cpu", 1)
text = weights_init.random_text_features(pgd.GDinoConfig(), "cpu")
sm, pid = gdino_ref.text_masks_and_position_ids([101, 4874, 1012, 102])
with torch.no_grad():
    t0 = time.time()
    logits, boxes = gdino_ref.detector_forward(gsd, gdino_ref.GDinoConfig(), gdino_ref.load_image(img)[None], text, sm, pid)
    t_det = time.time() - t0
    order = torch.sort(logits[0].sigmoid().max(-1)[0], descending=True, stable=True)[1][:16]
    b = boxes[0][order].double().numpy()
    pix = torch.tensor(np.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2], -1) * 1024.0).float()
    t0 = time.time()
    masks = sam_ref.run_sam(ssd, sam_ref.SamConfig(), img, pix)
    t_sam = time.time() - t0
full = t_det + t_sam
print(f"cores {ncpu}")
print(f"composed (bench.py cpu_baseline): {composed['per_sketch_s']:.2f} s per sketch  [{composed['sample']}]")
print(f"full oracle run, same box, no warm-up: detector {t_det:.2f} s + SAM (32 blocks, 16 boxes) {t_sam:.2f} s = {full:.2f} s per sketch")
print(f"ratio full / composed = {full / composed['per_sketch_s']:.3f}")
