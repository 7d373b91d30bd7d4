"""bench.py — sketches/sec end-to-end (GroundingDINO Swin-T + SAM ViT-H) at 1024x1024 on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one batch of B synthetic 1024x1024 sketches per GPU
(BASELINE.json config "Full GroundingDINO Swin-T + SAM ViT-H pipeline, batch=8, 1 MI355X"): detector
forward (B images) -> host threshold/box glue (top-16 boxes per image so the work does not depend on the
random weights) -> SAM encoder (B images) -> prompt encoder + mask decoder + postprocess
(16 boxes per image) -> B x 16 bool masks at 1024x1024.  The resized uint8 images are resident in HBM
when the timed region starts; masks stay on the GPU, boxes/scores cross to the host (they steer the
control flow).  Image-parallel over ranks (weak scaling), one RCCL weight broadcast at start-up, no
per-batch collectives.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_F16_TFLOPS = 2500.0          # dense f16/bf16 MFMA peak of MI355X (MI355X_MICROARCH.md)
FLOP_PER_SKETCH = 6.21e12         # 5.65 T (SAM enc) + 0.50 T (GroundingDINO) + 0.058 T (16 boxes)


def build_engines(dev, rank, world, batch):
    from inklayer_amd import dist as idist, gdino, sam, weights_init
    scfg, gcfg = sam.SamConfig(), gdino.GDinoConfig()
    spec = {("sam." + k): (v, torch.float32) for k, v in weights_init.sam_param_shapes(scfg).items()}
    spec.update({("det." + k): (v, torch.float32) for k, v in weights_init.gdino_param_shapes(gcfg).items()})
    spec["txt"] = ((4, gcfg.hidden_dim), torch.float32)
    sd = None
    if rank == 0:   # rank 0 owns the (random-init) weights; everyone else receives them over RCCL
        sd = {("sam." + k): v for k, v in weights_init.random_sam_state_dict(scfg, dev, 0).items()}
        sd.update({("det." + k): v for k, v in weights_init.random_gdino_state_dict(gcfg, dev, 1).items()})
        sd["txt"] = weights_init.random_text_features(gcfg, dev)
    t0 = time.time()
    sd = idist.broadcast_state_dict(spec, sd, dev)
    torch.cuda.synchronize()
    bcast_s = time.time() - t0
    seg = sam.SamEngine({k[4:]: v for k, v in sd.items() if k.startswith("sam.")}, scfg, dev, max_batch=batch)
    det = gdino.GDinoEngine({k[4:]: v for k, v in sd.items() if k.startswith("det.")}, gcfg, dev,
                            encoded_text=sd["txt"])
    del sd
    torch.cuda.empty_cache()
    return det, seg, bcast_s


def cpu_baseline(n_boxes):
    """The CPU oracle (oracle/, a port of the reference's PyTorch modules pinned by tests/golden) timed on
    this box's host cores on ONE sketch of the same workload."""
    from oracle import gdino_ref, sam_ref
    from inklayer_amd import synthetic, weights_init, sam as psam, gdino as pgd
    torch.manual_seed(0)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, ncpu))       # the threads this process may actually run on
    scfg, gcfg = sam_ref.SamConfig(), gdino_ref.GDinoConfig()
    ssd = weights_init.random_sam_state_dict(psam.SamConfig(), "cpu", 0)
    gsd = weights_init.random_gdino_state_dict(pgd.GDinoConfig(), "cpu", 1)
    text = weights_init.random_text_features(pgd.GDinoConfig(), "cpu")
    sm, pid = gdino_ref.text_masks_and_position_ids([101, 4874, 1012, 102])
    img = synthetic.synthetic_sketch(0)
    t0 = time.time()
    with torch.no_grad():
        x = gdino_ref.load_image(img)
        logits, boxes = gdino_ref.detector_forward(gsd, gcfg, x[None], text, sm, pid)
        score = logits[0].sigmoid().max(-1)[0]
        order = torch.sort(score, descending=True, stable=True)[1][:n_boxes]
        b = boxes[0][order].double().numpy()
        xyxy = np.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2], -1)
        pix = torch.tensor(xyxy * 1024.0).float()
        masks = sam_ref.run_sam(ssd, scfg, img, pix)
    dt = time.time() - t0
    assert len(masks) == n_boxes
    return {"value": 1.0 / dt, "unit": "sketches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 synthetic 1024x1024 sketch, full GroundingDINO Swin-T + SAM ViT-H, {n_boxes} boxes, "
                      f"fp32 torch CPU oracle, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="sketches per GPU per step")
    ap.add_argument("--boxes", type=int, default=16, help="boxes per sketch (top-n by score)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from inklayer_amd import dist as idist, ops, pipeline, synthetic
    rank, world, local = idist.init_process_group()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    det, seg, bcast_s = build_engines(dev, rank, world, args.batch)
    pipe = pipeline.InkLayerPipeline(det, seg)
    B = args.batch
    # rank r owns global images r, r+world, ... (static round-robin shard); synthetic, seeded per image
    imgs = [synthetic.synthetic_sketch(i) for i in idist.shard_indices(B * world, rank, world)]
    det_in, sam_in, sizes = pipe.prepare(imgs)
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.run_prepared(det_in, sam_in, sizes, top_n=args.boxes)
    trace = []
    ops.set_gemm_trace(trace)
    torch.cuda.synchronize()
    idist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = pipe.run_prepared(det_in, sam_in, sizes, top_n=args.boxes)
    torch.cuda.synchronize()
    idist.barrier()
    dt = time.perf_counter() - t0
    ops.set_gemm_trace(None)
    dt = idist.max_over_ranks(dt, dev)
    assert len(res) == B and res[0].masks.shape == (args.boxes, 1024, 1024)

    if rank == 0:
        gemm_ms = sum(t[1].elapsed_time(t[2]) for t in trace)
        gemm_flops = sum(t[0] for t in trace)
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        sketches = B * world * args.steps
        out = {
            "metric": "sketches/sec end-to-end (GroundingDINO+SAM) at 1024x1024",
            "value": sketches / dt, "unit": "sketches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 (MFMA, f32 accumulate; f32 residual/norm/softmax)",
            "data": "synthetic",
            "config": {"workload": "full GroundingDINO Swin-T + SAM ViT-H pipeline, batch=8 per GPU, "
                                   "1024x1024 synthetic sketches, 16 boxes/sketch, random-init weights",
                       "global_batch": B * world, "boxes_per_sketch": args.boxes,
                       "parallelism": f"image-parallel x{world}", "weight_broadcast_s": round(bcast_s, 3)},
            "roofline": {"bound": "mfma", "kernel": "gemm_f16_nt_128 (all dense projections)",
                         "achieved": achieved, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F16_TFLOPS, "traffic": None,
                         "launches_per_step": len(trace) // max(1, args.steps),
                         "gemm_ms_per_step": gemm_ms / max(1, args.steps),
                         "gemm_tflop_per_sketch": gemm_flops / max(1, args.steps) / B / 1e12,
                         "end_to_end_tflops": FLOP_PER_SKETCH * sketches / dt / 1e12 / world},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.boxes)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
