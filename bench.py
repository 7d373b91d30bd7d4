"""bench.py — sketches/sec end-to-end (GroundingDINO Swin-T + SAM ViT-H) at 1024x1024 on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1: either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N)
    or directly - bench.py then starts its N ranks itself (fresh child processes, before anything touches the GPU).

A step = one pass of the hot path over one batch of B synthetic 1024x1024 sketches per GPU (BASELINE.json config
"Full GroundingDINO Swin-T + SAM ViT-H pipeline, batch=8, 1 MI355X"), HOST MEMORY TO HOST MEMORY as SURVEY §8(d) defines
the metric: decoded RGB u8 sketches in (pinned) host memory -> upload -> the two Pillow-exact resizes on the GPU ->
detector forward (B images) -> host threshold/box glue (top-16 boxes per image so the work does not depend on the
random weights) -> SAM encoder (B images) -> prompt encoder + mask decoder + postprocess (16 boxes per image) ->
B x 16 uint8 0/1 masks of 1024x1024 downloaded into pinned host memory (1 byte per pixel, the reference's numpy bool
format; 134 MB per step), boxes/scores on the host.  Uploads and downloads ride a copy stream and overlap with the
neighbouring step's compute (two slots); every one of the K timed steps is complete - masks in host memory - when the
clock stops.  `device_resident` repeats the measurement without the PCIe legs (round 1's number).
Image-parallel over ranks (weak scaling), one RCCL weight broadcast at start-up, no per-batch collectives.
Prints ONE JSON line on rank 0.
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


def _cpu_share() -> int:
    """Host cores this process can really use: min(affinity, cgroup cpu.max quota, 16).  The GPU box exposes
    all of the host's hardware threads to sched_getaffinity but gives one-GPU jobs a ~16-CPU share; running
    torch with hundreds of threads on that share oversubscribes it by 10x and never finishes."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(n_boxes):
    """The CPU oracle (oracle/, a port of the reference's PyTorch modules pinned by tests/golden) timed on this box's
    host cores on a BOUNDED sample of ONE sketch of the same workload (B = 1, weights resident, SURVEY §8d protocol
    scaled to ~25 s of CPU work): every component gets one warm-up run and is then timed `reps` times, the MEDIAN is
    used; GroundingDINO is run in full; of SAM's 32 ViT-H blocks one windowed and one global block are run and the
    encoder time is composed as 28 x windowed + 4 x global + patch-embed + neck; the mask decoder runs on 4 boxes and
    is scaled to n_boxes.  The JSON says so (`extrapolated`, `measured_s`)."""
    import statistics
    import torch.nn.functional as F
    from oracle import gdino_ref, sam_ref
    from inklayer_amd import synthetic, weights_init, sam as psam, gdino as pgd
    ncpu = _cpu_share()
    torch.set_num_threads(ncpu)
    say = lambda m: print(f"[cpu_baseline] {m}", file=sys.stderr, flush=True)
    scfg = sam_ref.SamConfig(depth=8, global_attn_indexes=(7,))          # blocks 0 (windowed) and 7 (global) are timed
    gcfg = gdino_ref.GDinoConfig()
    say(f"{ncpu} threads; generating random weights")
    ssd = weights_init.random_sam_state_dict(psam.SamConfig(depth=8, global_attn_indexes=(7,)), "cpu", 0)
    gsd = weights_init.random_gdino_state_dict(pgd.GDinoConfig(), "cpu", 1)
    text = weights_init.random_text_features(pgd.GDinoConfig(), "cpu")
    sm, pid = gdino_ref.text_masks_and_position_ids([101, 4874, 1012, 102])
    img = synthetic.synthetic_sketch(0)
    t_all = time.time()

    def timed(fn, reps, warm=1):
        for _ in range(warm):
            out = fn()
        ts = []
        for _ in range(reps):
            t0 = time.time()
            out = fn()
            ts.append(time.time() - t0)
        return statistics.median(ts), out

    T = {}
    with torch.no_grad():
        def detector():
            x = gdino_ref.load_image(img)
            return gdino_ref.detector_forward(gsd, gcfg, x[None], text, sm, pid)
        T["detector"], (logits, boxes) = timed(detector, reps=2)
        say(f"detector {T['detector']:.2f} s (median of 2 after 1 warm-up)")
        score = logits[0].sigmoid().max(-1)[0]
        order = torch.sort(score, descending=True, stable=True)[1][:4]
        b = boxes[0][order].double().numpy()
        xyxy = np.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2], -1)
        pix = torch.tensor(xyxy * 1024.0).float()
        xin = sam_ref.preprocess(scfg, torch.from_numpy(img[..., ::-1].copy()).permute(2, 0, 1))[None]
        T["patch_embed"], tok = timed(lambda: sam_ref.image_encoder(ssd, scfg, xin, upto=0), reps=3)
        T["win_block"], tok1 = timed(lambda: sam_ref.vit_block(ssd, scfg, 0, tok), reps=3)
        T["glob_block"], tok2 = timed(lambda: sam_ref.vit_block(ssd, scfg, 7, tok1), reps=3)
        say(f"ViT-H blocks: windowed {T['win_block']:.2f} s, global {T['glob_block']:.2f} s (medians of 3)")

        def neck():
            e = tok2.permute(0, 3, 1, 2)
            e = F.conv2d(e, ssd["image_encoder.neck.0.weight"])
            e = sam_ref._ln2d(e, ssd["image_encoder.neck.1.weight"], ssd["image_encoder.neck.1.bias"])
            e = F.conv2d(e, ssd["image_encoder.neck.2.weight"], padding=1)
            return sam_ref._ln2d(e, ssd["image_encoder.neck.3.weight"], ssd["image_encoder.neck.3.bias"])
        T["neck"], emb = timed(neck, reps=3)

        def decoder():
            low, _ = sam_ref.mask_decoder(ssd, scfg, emb, sam_ref.dense_pe(ssd, scfg), sam_ref.embed_boxes(ssd, scfg, pix))
            return sam_ref.postprocess_masks(scfg, low, (1024, 1024), (1024, 1024)) > 0
        T["decoder4"], _ = timed(decoder, reps=3)
    total = (T["detector"] + T["patch_embed"] + 28 * T["win_block"] + 4 * T["glob_block"] + T["neck"]
             + T["decoder4"] * n_boxes / 4.0)
    work = time.time() - t_all
    say(f"{work:.1f} s of CPU work -> {total:.1f} s per sketch composed")
    return {"value": 1.0 / total, "unit": "sketches/s", "cores": ncpu, "kind": "port", "extrapolated": True,
            "measured_s": round(work, 1), "per_sketch_s": round(total, 2),
            "sample": f"1 synthetic 1024x1024 sketch, B=1, fp32 torch CPU oracle, weights resident, every component 1 "
                      f"warm-up then median of 2-3 runs: GroundingDINO Swin-T in full ({T['detector']:.2f} s); SAM ViT-H "
                      f"patch-embed ({T['patch_embed']:.2f} s) + 1 windowed block ({T['win_block']:.2f} s) + 1 global "
                      f"block ({T['glob_block']:.2f} s) + neck ({T['neck']:.2f} s), encoder composed as 28 x windowed + "
                      f"4 x global; mask decoder + postprocess on 4 boxes ({T['decoder4']:.2f} s) scaled to {n_boxes}"}


def runner_leg(pipe, batch, n_boxes):
    """Secondary figure (BASELINE config 5, not the metric): the WHOLE runner - the batched hot path plus mask cleanup,
    sketch NMS, Depth-Anything-V2 ViT-B, the refinement stage and the reference's complete output tree of PNG / JSON
    files per sketch - over a directory of `batch` synthetic 1024x1024 sketch PNGs (inklayer_amd/batch_runner.py, what
    tools/run_dir.py runs per rank).  One warm-up pass, one timed pass; random weights, so the n best boxes per sketch
    are kept instead of the 0.2 threshold."""
    import shutil
    import tempfile
    from PIL import Image
    from inklayer_amd import batch_runner, synthetic
    os.environ["INKLAYER_RANDOM_WEIGHTS"] = "1"            # the depth plugin's singleton: no checkpoints exist offline
    tmp = Path(tempfile.mkdtemp(prefix="ink_runner_"))
    try:
        (tmp / "in").mkdir()
        for i in range(batch):
            Image.fromarray(synthetic.synthetic_sketch(100 + i)).save(tmp / "in" / f"s{i}.png")
        files = sorted(str(p) for p in (tmp / "in").glob("*.png"))
        batch_runner.run_files(files[:2], str(tmp / "warm"), batch=batch, pipe=pipe, top_n=n_boxes)
        stages = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = batch_runner.run_files(files, str(tmp / "out"), batch=batch, pipe=pipe, top_n=n_boxes, stage_s=stages)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_files = sum(len(list(Path(o).rglob("*.*"))) for o in outs)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"value": len(files) / dt, "unit": "sketches/s", "sketches": len(files), "seconds": dt,
            "files_written": n_files, "stage_ms_per_sketch": {k: v / len(files) * 1e3 for k, v in stages.items()},
            "note": "whole runner incl. Depth-Anything-V2, refinement and the reference's output tree on disk "
                    "(PNG encoding on 8 host threads); one GPU, not the metric"}


ATTN_ALGO_BYTES = lambda rows, width: 4.0 * rows * width * 2      # read q, k, v + write o once, f16
PEAK_HBM_TBS = 8.0


def attention_summary(trace, steps):
    """North-star figure: ViT-H attention cores against the HBM roofline (algorithmic q,k,v,o bytes / kernel time)
    and as MFMA throughput.  trace rows: ops.set_attn_trace; only head_dim-80 launches (SAM ViT-H) are counted."""
    out = {}
    for name, mode in (("windowed", 2), ("global", 1)):
        rows = [t for t in trace if t[4] == 80 and t[5] == mode]
        if not rows:
            continue
        us = [t[7].elapsed_time(t[8]) * 1e3 for t in rows]
        nb, nh, nq, nk, hd = rows[0][:5]
        width = nh * hd
        abytes = ATTN_ALGO_BYTES(rows[0][6], width)
        flop = 4.0 * nb * nh * nq * nk * hd
        avg = sum(us) / len(us)
        out[name] = {"launches_per_step": len(rows) // max(1, steps), "avg_us": avg,
                     "algorithmic_bytes": abytes, "GBps": abytes / avg / 1e3,
                     "frac_hbm": abytes / (avg * 1e-6) / (PEAK_HBM_TBS * 1e12),
                     "TFLOPs": flop / avg / 1e6, "frac_mfma": flop / (avg * 1e-6) / (PEAK_F16_TFLOPS * 1e12)}
    if out:
        tot_b = sum(v["algorithmic_bytes"] * v["launches_per_step"] for v in out.values())
        tot_t = sum(v["avg_us"] * v["launches_per_step"] for v in out.values())
        out["all_32_blocks"] = {"us_per_step": tot_t, "GBps": tot_b / tot_t / 1e3,
                                "frac_hbm": tot_b / (tot_t * 1e-6) / (PEAK_HBM_TBS * 1e12)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="sketches per GPU per step")
    ap.add_argument("--boxes", type=int, default=16, help="boxes per sketch (top-n by score)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-runner", action="store_true", help="skip the secondary whole-runner (config 5) measurement")
    args = ap.parse_args()

    from inklayer_amd import dist as idist
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: become the launcher.  Nothing in this process has touched the GPU yet (importing
        # torch does not); the N ranks are fresh child processes running this same script.
        rc, out0 = idist.launch_ranks([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], args.gpus)
        for line in out0.splitlines():       # rank 0's JSON line goes to stdout, any library chatter to stderr
            print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
        sys.exit(rc)

    from inklayer_amd import ops, pipeline, synthetic
    rank, world, local = idist.init_process_group()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    det, seg, bcast_s = build_engines(dev, rank, world, args.batch)
    pipe = pipeline.InkLayerPipeline(det, seg)
    B = args.batch
    # rank r owns global images r, r+world, ... (static round-robin shard); synthetic, seeded per image
    imgs = [synthetic.synthetic_sketch(i) for i in idist.shard_indices(B * world, rank, world)]
    host = pipe.pinned_like(imgs)      # decoded RGB u8 sketches in pinned host memory: where the timed region starts
    torch.cuda.synchronize()

    def run_steps(n):
        """n complete host-to-host steps, software-pipelined two deep (download of step i under step i+1)."""
        prev, last = None, None
        for _ in range(n):
            t = pipe.submit_host(host, top_n=args.boxes)
            if prev is not None:
                last = pipe.collect_host(prev)
            prev = t
        if prev is not None:
            last = pipe.collect_host(prev)
        return last

    run_steps(args.warmup)
    torch.cuda.synchronize()
    idist.barrier()
    t0 = time.perf_counter()
    res = run_steps(args.steps)
    torch.cuda.synchronize()
    idist.barrier()
    dt = time.perf_counter() - t0
    dt = idist.max_over_ranks(dt, dev)
    assert len(res) == B and res[0][3].shape == (args.boxes, 1024, 1024) and res[0][3].dtype == np.uint8
    # the LAST timed step (submitted with its predecessor still in flight) against one synchronous step: same bytes?
    # (round 3: kernels co-resident with the window-attention kernel returned wrong results in a third of in-flight steps
    # until that kernel claimed its SIMDs' register file - DESIGN.md section 7; the bench now says so if it ever recurs)
    timed_boxes = [np.array(r[0], copy=True) for r in res]
    timed_sum = [int(r[3].sum(dtype=np.int64)) for r in res]
    torch.cuda.synchronize()
    chk = pipe.collect_host(pipe.submit_host(host, top_n=args.boxes))
    verified = all(np.array_equal(a, r[0]) and sa == int(r[3].sum(dtype=np.int64))
                   for a, sa, r in zip(timed_boxes, timed_sum, chk))

    # secondary figure: the same steps with inputs and outputs resident in HBM (no PCIe legs)
    raw = pipe.upload(imgs)
    torch.cuda.synchronize()
    idist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.run_uploaded(raw, top_n=args.boxes)
    torch.cuda.synchronize()
    idist.barrier()
    dt_dev = idist.max_over_ranks(time.perf_counter() - t0, dev)

    # Roofline instrumentation: the same steps once more with every GEMM / attention launch bracketed by HIP events on
    # its launch stream, in SERIAL stream order (detector, then SAM) so that the events time the kernel itself and not
    # the co-scheduling delay of the two-stream overlap (which the timed region above uses).
    trace, atrace = [], []
    roof_steps = min(2, args.steps)
    if rank == 0:
        serial = pipeline.InkLayerPipeline(det, seg, overlap=False)
        ops.set_gemm_trace(trace)
        ops.set_attn_trace(atrace)
        for _ in range(roof_steps):
            serial.run_uploaded(raw, top_n=args.boxes)
        torch.cuda.synchronize()
        ops.set_gemm_trace(None)
        ops.set_attn_trace(None)
        # ... and once in the two-stream arrangement of the timed region: there a launch's bracket also contains the time
        # its workgroups wait for CUs held by the other stream's kernels - what a rocprofv3 --stats average over this
        # whole command mostly consists of (reported as avg_launch_us_two_streams, not used for `frac`)
        trace2 = []
        ops.set_gemm_trace(trace2)
        for _ in range(roof_steps):
            pipe.run_uploaded(raw, top_n=args.boxes)
        torch.cuda.synchronize()
        ops.set_gemm_trace(None)

    if rank == 0:
        from inklayer_amd import _lib
        gemm_ms = sum(t[1].elapsed_time(t[2]) for t in trace)
        gemm_flops = sum(t[0] for t in trace)
        # the dominant kernel = the tile variant with the largest total time: its launches, flops, bytes, durations
        names = {45: "gemm_f16_nt_pp<4,5> (ping-pong 256x320x32 tile, 8 waves: dense projections of SAM ViT-H)",
                 10: "gemm_f16_nt<256,256,64,4,4,2> (16-wave 256x256 tile: DINO FFN and other large projections)",
                 0: "gemm_f16_nt<128,128,64,2,2,2>", 32: "gemm_f16_nt<128,128,32,2,2,2>"}
        by_var = {}
        for t in trace:
            by_var.setdefault(_lib.lib().ink_gemm_query_variant(t[3][0], t[3][1], t[3][2]), []).append(t)
        dom_var = max(by_var, key=lambda v: sum(t[1].elapsed_time(t[2]) for t in by_var[v]))
        dom = by_var[dom_var]
        traffic = None
        pmcs = sorted((ROOT / "profiles").glob("r*_gemm_pmc.json"))     # the newest round's counters
        pmc = pmcs[-1] if pmcs else ROOT / "profiles" / "none"
        if pmc.exists():      # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
            pj = json.loads(pmc.read_text())
            if pj.get("variant") == dom_var:
                traffic = pj.get("traffic_bytes_per_launch")
        dom_ms = sum(t[1].elapsed_time(t[2]) for t in dom)
        dom_flops = sum(t[0] for t in dom)
        abytes = lambda k: (2.0 * (k[0] * k[2] + k[1] * k[2])                       # A + W in f16
                            + k[0] * k[1] * ((2 if k[6] == "f16" else 4) + (4 if k[4] == "res" else 0)))  # C (+ residual)
        dom_bytes = sum(abytes(t[3]) for t in dom)
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        sketches = B * world * args.steps
        out = {
            "metric": "sketches/sec end-to-end (GroundingDINO+SAM) at 1024x1024",
            "value": sketches / dt, "unit": "sketches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 (MFMA, f32 accumulate; f32 residual/norm/softmax; split-f16 = fp32-grade operands for "
                     "patch-embed, neck, mask decoder)",
            "data": "synthetic",
            "timed_step_equals_synchronous_step": bool(verified),
            "config": {"workload": "full GroundingDINO Swin-T + SAM ViT-H pipeline, batch=8 per GPU, "
                                   "1024x1024 synthetic sketches, 16 boxes/sketch, random-init weights",
                       "global_batch": B * world, "boxes_per_sketch": args.boxes,
                       "parallelism": f"image-parallel x{world}", "weight_broadcast_s": round(bcast_s, 3),
                       "timed_region": "host to host: pinned u8 sketches -> H2D -> pipeline -> D2H of u8 masks (1 B/pixel, "
                                       "134 MB/step at batch 8) + boxes; copies on a copy stream, 2 slots"},
            "device_resident": {"value": sketches / dt_dev, "unit": "sketches/s", "ms_per_step": dt_dev / args.steps * 1e3,
                                "note": "same steps with sketches and masks left in HBM (no PCIe legs); not the metric"},
            "roofline": {"bound": "mfma", "kernel": names.get(dom_var, str(dom_var)),
                         "achieved": achieved, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F16_TFLOPS, "traffic": traffic,
                         "traffic_unit": "HBM-side bytes per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc)",
                         "algorithmic_bytes_per_launch": dom_bytes / max(1, len(dom)),
                         "algorithmic_flop_per_launch": dom_flops / max(1, len(dom)),
                         "avg_launch_us": dom_ms * 1e3 / max(1, len(dom)),
                         "avg_launch_us_two_streams": (lambda d2: sum(t[1].elapsed_time(t[2]) for t in d2) * 1e3 / max(1, len(d2)))(
                             [t for t in trace2 if _lib.lib().ink_gemm_query_variant(t[3][0], t[3][1], t[3][2]) == dom_var]),
                         "launches_per_step": len(dom) // max(1, roof_steps),
                         "ms_per_step": dom_ms / max(1, roof_steps),
                         "all_gemm_launches_per_step": len(trace) // max(1, roof_steps),
                         "all_gemm_ms_per_step": gemm_ms / max(1, roof_steps),
                         "gemm_tflop_per_sketch": gemm_flops / max(1, roof_steps) / B / 1e12,
                         "note": "HIP-event brackets per launch over %d extra steps run in serial stream order right after "
                                 "the timed region (the timed region overlaps detector and SAM encoder on two streams)" % roof_steps,
                         "end_to_end_tflops": FLOP_PER_SKETCH * sketches / dt / 1e12 / world},
            "attention": attention_summary(atrace, roof_steps),
        }
        if world == 1 and not args.no_runner:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):     # the runner prints the reference's progress lines: the
                out["runner"] = runner_leg(pipe, args.batch, args.boxes)   # bench's stdout is ONE JSON line
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.boxes)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
