"""Batched detector -> segmentor pipeline on one MI355X (the unit that is sharded image-parallel).

One call = B sketches: GroundingDINO proposes boxes for all B images in one batched forward, the
boxes cross to the host once (900 x 8 floats per image) for the threshold + box glue that the
reference also does on the CPU (GD/util/inference.py:70-75, InkLayer/utils/processing.py:6-28), SAM
encodes all B images in one batched forward and decodes each image's boxes.  Both engines stay
resident in HBM (the reference reloads SAM's 2.4 GB checkpoint per image, InkLayer/segmentor/sam.py:23).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import gdino as gd
from . import ops
from . import sam as sm
from .ops import sam_postprocess as ops_sam_postprocess


@dataclass
class SketchResult:
    boxes_xyxy_norm: np.ndarray       # [n,4] float64, normalised (the detector plugin's "bboxes")
    scores: np.ndarray                # [n]
    boxes_pixel: torch.Tensor         # [n,4] float32 pixel xyxy (what SAM is prompted with)
    masks: torch.Tensor               # [n,H,W] uint8 on the GPU (0/1)


def boxes_to_pixels(norm_xyxy: np.ndarray, W: int, H: int) -> torch.Tensor:
    """process_boxes_ours (InkLayer/utils/processing.py:6-28), vectorised with the same f32 steps."""
    if len(norm_xyxy) == 0:
        return torch.zeros((0, 4), dtype=torch.float32)
    x1, y1, x2, y2 = (norm_xyxy[:, i] for i in range(4))
    w, h = x2 - x1, y2 - y1
    b = torch.tensor(np.stack([x1 + w / 2, y1 + h / 2, w, h], -1)).float()
    b = b * torch.tensor([W, H, W, H], dtype=torch.float32)
    b[:, :2] -= b[:, 2:] / 2
    b[:, 2:] += b[:, :2]
    return b


def gpu_preprocess(img_u8: torch.Tensor, L: int):
    """The reference's two PIL resizes of one decoded RGB sketch (HWC uint8, on the GPU), bit for bit:
    detector input (GD/util/inference.py:39-50: shorter side 800, longer <= 1333) and SAM input
    (SA/utils/transforms.py:26-31: longest side L).  Returns (det_u8, sam_u8), both RGB order."""
    h, w = int(img_u8.shape[0]), int(img_u8.shape[1])
    dh, dw = gd.resize_shape(w, h)
    sh, sw = sm.preprocess_shape(h, w, L)
    return ops.resize_bilinear_u8(img_u8, dh, dw), ops.resize_bilinear_u8(img_u8, sh, sw)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class InkLayerPipeline:
    def __init__(self, detector: gd.GDinoEngine, segmentor: sm.SamEngine, overlap: bool = True):
        self.det, self.seg = detector, segmentor
        self.dev = detector.dev
        # The detector and the SAM image encoder are independent until the mask decoder needs the boxes: they run
        # on two HIP streams.  The encoder's GEMM workgroups fill a CU (144 KB LDS, 8 waves x 232 VGPRs), so the two
        # streams mostly time-slice rather than co-reside; what the second stream buys is the gaps (kernel tails,
        # the host round trip for the boxes): ~8 ms of the detector's ~20 ms per batch of 8.
        self.overlap = overlap
        self.s_det = torch.cuda.Stream(device=self.dev) if overlap else None
        self.s_seg = torch.cuda.Stream(device=self.dev) if overlap else None

    def upload(self, images_rgb: Sequence[np.ndarray]) -> List[torch.Tensor]:
        """Decoded RGB sketches (HWC uint8, host) -> device, one transfer each.  Everything after this point -
        both resizes, normalisation, patchify - runs on the GPU."""
        return [torch.from_numpy(np.ascontiguousarray(im)).to(self.dev, non_blocking=True) for im in images_rgb]

    def preprocess(self, raw: Sequence[torch.Tensor]):
        """GPU side of load_image / ResizeLongestSide (Pillow-exact bilinear, `ink_resize_bilinear_u8`), plus
        SAM's channel quirk (InkLayer/segmentor/sam.py:24-26 feeds the encoder channel-reversed pixels)."""
        det_in, sam_in, sizes = [], [], []
        L = self.seg.cfg.img_size
        for im in raw:
            d, s_ = gpu_preprocess(im, L)
            det_in.append(d)
            sam_in.append(s_.flip(-1).contiguous())
            sizes.append(((int(im.shape[0]), int(im.shape[1])), (int(s_.shape[0]), int(s_.shape[1]))))
        return det_in, sam_in, sizes

    def prepare(self, images_rgb: Sequence[np.ndarray]):
        """upload + preprocess: (detector inputs, SAM inputs, ((H, W), (h_sam, w_sam)) per image)."""
        return self.preprocess(self.upload(images_rgb))

    @torch.no_grad()
    def run_uploaded(self, raw: Sequence[torch.Tensor], top_n: Optional[int] = None) -> List[SketchResult]:
        """The whole hot path D1 -> S9 from device-resident decoded sketches."""
        return self.run_prepared(*self.preprocess(raw), top_n=top_n)

    @torch.no_grad()
    def run_prepared(self, det_in, sam_in, sizes, top_n: Optional[int] = None,
                     defer_sync: bool = False) -> List[SketchResult]:
        """defer_sync=True (needs overlap): the caller's stream is NOT made to wait for the results; consecutive
        batches then pipeline (batch i+1's detector runs under batch i's encoder/decoder).  The caller must call
        `synchronize()` (or wait on `s_seg`) before touching the returned masks."""
        if self.overlap:
            cur = torch.cuda.current_stream(self.dev)
            self.s_det.wait_stream(cur)
            self.s_seg.wait_stream(cur)
            with torch.cuda.stream(self.s_det):
                logits, boxes = self.det.forward(det_in, allow_graph=False)   # graphs do not overlap across streams
                both = torch.cat([logits, boxes], dim=-1)
                host = torch.empty(both.shape, dtype=both.dtype, pin_memory=True)
                host.copy_(both, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.s_det)
            with torch.cuda.stream(self.s_seg):
                emb = self.seg.encode(sam_in)
            ev.synchronize()                               # host needs the boxes; the encoder keeps running
            dets = self.det.postprocess(host, top_n=top_n)
            stream_ctx = torch.cuda.stream(self.s_seg)
        else:
            dets = self.det.detect(det_in, top_n=top_n)
            emb = self.seg.encode(sam_in)
            stream_ctx = _NullCtx()
        with stream_ctx:
            out = self._decode_all(dets, emb, sizes)
        if self.overlap and not defer_sync:
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_stream(self.s_seg)
            cur.wait_stream(self.s_det)
            for r in out:                                  # results are consumed on the caller's stream
                r.masks.record_stream(cur)
        return out

    def synchronize(self) -> None:
        """Join both pipeline streams into the caller's stream (after a run of defer_sync=True batches)."""
        if self.overlap:
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_stream(self.s_seg)
            cur.wait_stream(self.s_det)

    def _decode_all(self, dets, emb, sizes) -> List[SketchResult]:
        L = self.seg.cfg.img_size
        per_img, all_boxes, img_of_box = [], [], []
        for b, ((boxes_cxcywh, scores), ((oh, ow), (ih, iw))) in enumerate(zip(dets, sizes)):
            bx = boxes_cxcywh.double().numpy().reshape(-1, 4)
            xyxy = np.stack([bx[:, 0] - bx[:, 2] / 2, bx[:, 1] - bx[:, 3] / 2, bx[:, 0] + bx[:, 2] / 2,
                             bx[:, 1] + bx[:, 3] / 2], -1)
            pix = boxes_to_pixels(xyxy, ow, oh)
            per_img.append((xyxy, scores.numpy(), pix))
            if len(pix):
                # ResizeLongestSide.apply_boxes_torch (SA/utils/transforms.py:67-91)
                nh, nw = sm.preprocess_shape(oh, ow, L)
                tb = pix.reshape(-1, 2, 2).clone()
                tb[..., 0] = tb[..., 0] * (nw / ow)
                tb[..., 1] = tb[..., 1] * (nh / oh)
                all_boxes.append(tb.reshape(-1, 4))
                img_of_box += [b] * len(pix)
        # prompt encoder + mask decoder for ALL boxes of the batch in one pass
        low = None
        if img_of_box:
            low, _ = self.seg.decode_low_res(emb, torch.cat(all_boxes, 0), img_of_box)
        out, off = [], 0
        thr = self.seg.cfg.mask_threshold
        # postprocess: one launch per run of images that share (input size, original size)
        groups = []
        for b, (xyxy, sc, pix) in enumerate(per_img):
            n = len(pix)
            key = sizes[b]
            if groups and groups[-1][0] == key:
                groups[-1][2].append((b, n))
                groups[-1][1][1] += n
            else:
                groups.append([key, [off, n], [(b, n)]])
            off += n
        masks_per_img = {}
        for (oh_ow, ih_iw), (start, cnt), members in groups:
            if cnt == 0:
                for b, n in members:
                    masks_per_img[b] = torch.zeros((0, oh_ow[0], oh_ow[1]), dtype=torch.uint8, device=self.dev)
                continue
            m = ops_sam_postprocess(low[start:start + cnt], L, ih_iw, oh_ow, thr)
            o = 0
            for b, n in members:
                masks_per_img[b] = m[o:o + n]
                o += n
        for b, (xyxy, sc, pix) in enumerate(per_img):
            out.append(SketchResult(xyxy, sc, pix, masks_per_img[b]))
        return out

    def run_batch(self, images_rgb: Sequence[np.ndarray], top_n: Optional[int] = None) -> List[SketchResult]:
        return self.run_prepared(*self.prepare(images_rgb), top_n=top_n)
