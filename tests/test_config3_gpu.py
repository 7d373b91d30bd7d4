"""BASELINE config 3 - the workload bench.py's headline is quoted on - checked against the CPU oracle.

The 8 bench sketches (synthetic seeds 0-7), the bench's weight seeds (SAM 0, detector 1, text 2), full depth (6+6 DINO
layers, 900 queries, 32 ViT-H blocks), through the same entry points the bench times:
InkLayerPipeline.submit_host / collect_host with top_n=16.

  * 4 of the 8 images: every one of the 16 masks against oracle.sam_ref.run_sam prompted with the boxes the HIP
    detector produced (north-star tolerance: IoU >= 0.999 PER INSTANCE; every flipped pixel within 1 % of the logit
    scale of the threshold), and the detector's 900 boxes / logits against oracle.gdino_ref.detector_forward pinned to
    the HIP path's two-stage selection (every quantile incl. the maximum <= 2x the fp32 oracle's own movement under
    the stated f16 operand rounding; kept boxes: per-query sensitivity bound).
  * the other 4 images: batch == solo (the image's results do not depend on what else is in the batch).
GPU box only: ~3 minutes of oracle work on the box's host cores."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ORACLE_IMAGES = (0, 3, 5, 6)          # checked against the oracle; the rest by batch == solo
TOP_N = 16


def _iou(g, r):
    return float((g & r).sum() / max(1, (g | r).sum()))


@pytest.fixture(scope="module")
def bench_pipeline(dev):
    """Engines exactly as bench.build_engines makes them (rank 0, world 1)."""
    from inklayer_amd import gdino, pipeline, sam, weights_init
    scfg, gcfg = sam.SamConfig(), gdino.GDinoConfig()
    ssd = weights_init.random_sam_state_dict(scfg, dev, 0)
    gsd = weights_init.random_gdino_state_dict(gcfg, dev, 1)
    text = weights_init.random_text_features(gcfg, dev)
    seg = sam.SamEngine(ssd, scfg, dev, max_batch=8)
    det = gdino.GDinoEngine(gsd, gcfg, dev, encoded_text=text)
    cpu = ({k: v.cpu() for k, v in ssd.items()}, {k: v.cpu() for k, v in gsd.items()}, text.cpu())
    del ssd, gsd
    torch.cuda.empty_cache()
    yield pipeline.InkLayerPipeline(det, seg), cpu
    del det, seg
    torch.cuda.empty_cache()


@torch.no_grad()
def test_config3_batch8_x16_boxes_against_oracle(dev, bench_pipeline):
    from oracle import gdino_ref, sam_ref
    from inklayer_amd import gdino, synthetic
    pipe, (ssd, gsd, text) = bench_pipeline
    torch.set_num_threads(min(16, os.cpu_count() or 16))
    imgs = [synthetic.synthetic_sketch(i) for i in range(8)]
    host = pipe.pinned_like(imgs)
    # two steps in flight over the two slots, as bench.run_steps does; both must give the same bytes
    t1 = pipe.submit_host(host, top_n=TOP_N)
    t2 = pipe.submit_host(host, top_n=TOP_N)
    first = [(a.copy(), b.copy(), c.clone(), d.copy()) for a, b, c, d in pipe.collect_host(t1)]
    res = pipe.collect_host(t2)
    assert len(res) == 8
    for (xyxy, sc, pix, m), (xyxy1, sc1, pix1, m1) in zip(res, first):
        assert m.shape == (TOP_N, 1024, 1024) and m.dtype == np.uint8 and set(np.unique(m)) <= {0, 1}
        assert np.array_equal(m, m1) and np.array_equal(xyxy, xyxy1)
    # ... and so must further pairs of in-flight steps (round 3: a third of the SECOND steps of such pairs came out with
    # different boxes until the window-attention kernel claimed its SIMDs' whole register file)
    for rep in range(4):
        ta, tb = pipe.submit_host(host, top_n=TOP_N), pipe.submit_host(host, top_n=TOP_N)
        for t in (ta, tb):
            for (xyxy, sc, pix, m), (xyxy1, sc1, pix1, m1) in zip(pipe.collect_host(t), first):
                assert np.array_equal(xyxy, xyxy1) and np.array_equal(m, m1), f"in-flight pair {rep}"

    # ---------------- SAM: 16 masks per image vs the oracle prompted with the HIP detector's boxes
    all_iou = []
    for b in ORACLE_IMAGES:
        xyxy, sc, pix, m = res[b]
        ref_logits, _, _ = sam_ref.run_sam(ssd, sam_ref.SamConfig(), imgs[b], pix, return_logits=True)
        ref = (ref_logits[:, 0] > 0).numpy()
        got = m.astype(bool)
        ious = [_iou(g, r) for g, r in zip(got, ref)]
        all_iou += ious
        tol = 1e-2 * ref_logits.std().item()
        far = np.abs(ref_logits[:, 0].numpy()[got != ref]).max(initial=0.0)
        print(f"image {b}: mask IoU min {min(ious):.5f} median {np.median(ious):.5f}; farthest flipped logit "
              f"{far:.3e} (1 % of the logit scale = {tol:.3e})")
        assert min(ious) >= 0.999, (b, ious)
        assert far < tol
    print(f"config 3: {len(all_iou)} instances, IoU min {min(all_iou):.5f} median {np.median(all_iou):.5f}")

    # ---------------- detector at the bench's size (800x800 -> 13294 tokens, 6+6 layers, 900 queries)
    import torch.nn.functional as RealF

    class _F16Operands:
        def __getattr__(self, k):
            return getattr(RealF, k)

        def linear(self, a, w, b=None):
            return RealF.linear(a.half().float(), w.half().float(), b)

        def conv2d(self, a, w, b=None, **kw):
            return RealF.conv2d(a.half().float(), w.half().float(), b, **kw)

    det = pipe.det
    det_in, _, _ = pipe.preprocess(pipe.upload(imgs))
    st = {}
    logits, boxes = det._forward_eager(det_in, stages=st)
    both = torch.cat([logits, boxes], -1).cpu()
    kept = det.postprocess(both, top_n=TOP_N)
    sm, pid = gdino_ref.text_masks_and_position_ids(list(gdino.DEFAULT_TOKEN_IDS))
    ocfg = gdino_ref.GDinoConfig()
    T = det.T
    for b in ORACLE_IMAGES:
        # the eager forward with stage taps is the forward the pipeline ran: same kept boxes
        kb = kept[b][0].double().numpy()
        kx = np.stack([kb[:, 0] - kb[:, 2] / 2, kb[:, 1] - kb[:, 3] / 2, kb[:, 0] + kb[:, 2] / 2, kb[:, 1] + kb[:, 3] / 2], -1)
        assert np.array_equal(kx, res[b][0])
        x = gdino_ref.load_image(imgs[b])
        assert tuple(x.shape[-2:]) == tuple(det_in[b].shape[:2]) == (800, 800)
        pin = {"force_topk": st["topk"][b:b + 1].cpu()}
        ost = dict(pin)
        ref_logits, ref_boxes = gdino_ref.detector_forward(gsd, ocfg, x[None], text, sm, pid, stages=ost)
        for name, mine, want in (("src", st["src"].view(8, -1, 256)[b], ost["src"][0]),
                                 ("memory", st["memory"].view(8, -1, 256)[b], ost["memory"][0])):
            assert mine.shape[0] == 13294
            l2 = ((mine.cpu().double() - want.double()).norm() / want.double().norm()).item()
            print(f"image {b}: {name} [13294 x 256] l2-rel {l2:.2e}")
            assert l2 < 5e-3
        d = (both[b, :, T:] - ref_boxes[0]).abs().max(-1)[0]
        dl = (both[b, :, :T] - ref_logits[0]).abs().max(-1)[0] / ref_logits.abs().max()
        gdino_ref.F = _F16Operands()
        try:
            el, eb = gdino_ref.detector_forward(gsd, ocfg, x[None], text, sm, pid, stages=dict(pin))
        finally:
            gdino_ref.F = RealF
        sb = (eb[0] - ref_boxes[0]).abs().max(-1)[0]
        sl = (el[0] - ref_logits[0]).abs().max(-1)[0] / ref_logits.abs().max()
        for name, mine, emul in (("box", d, sb), ("logit", dl, sl)):
            for qt in (0.5, 0.9, 0.99, 1.0):
                hq, eq = mine.quantile(qt).item(), emul.quantile(qt).item()
                print(f"image {b}: {name} err q{qt}: HIP {hq:.2e}  emulated-f16 oracle {eq:.2e}")
                assert hq <= 2.0 * eq + 1e-3, (b, name, qt, hq, eq)
        # the 16 kept queries (what SAM is prompted with)
        score = both[b, :, :T].sigmoid().max(1)[0]
        order = torch.sort(score, descending=True, stable=True)[1][:TOP_N]
        ke, ks = d[order], sb[order]
        print(f"image {b}: kept boxes err median {ke.median().item():.2e} max {ke.max().item():.2e} "
              f"(own f16 sensitivity max {ks.max().item():.2e}); scores {res[b][1].min():.4f}..{res[b][1].max():.4f}")
        assert ke.median().item() < 3e-3
        assert bool((ke <= 2e-3 + 20 * ks).all())
        rs = ref_logits[0].sigmoid().max(1)[0][order]
        assert np.abs(res[b][1] - rs.numpy()).max() < 3e-3

    # ---------------- the other images: batch == solo
    for b in [i for i in range(8) if i not in ORACLE_IMAGES]:
        solo = pipe.run_batch([imgs[b]], top_n=TOP_N)[0]
        torch.cuda.synchronize()
        xyxy, sc, pix, m = res[b]
        assert np.allclose(solo.boxes_xyxy_norm, xyxy, atol=1e-5, rtol=0)
        sm_ = solo.masks.cpu().numpy()
        if np.array_equal(solo.boxes_xyxy_norm, xyxy):
            assert np.array_equal(sm_, m), f"image {b}: same boxes, different masks in the batch"
        else:      # the detector's GEMM tile variant depends on the batch's row count: boxes may move in the last bits
            ious = [_iou(g.astype(bool), r.astype(bool)) for g, r in zip(sm_, m)]
            print(f"image {b}: boxes differ by {np.abs(solo.boxes_xyxy_norm - xyxy).max():.1e}; IoU min {min(ious):.6f}")
            assert min(ious) >= 0.9999
