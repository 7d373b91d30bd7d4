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
class HostTicket:
    """One in-flight host-to-host batch (InkLayerPipeline.submit_host / collect_host)."""
    slot: int
    done: "torch.cuda.Event"
    dets: list                         # per image: (xyxy float64 [n,4], scores [n], pixel boxes [n,4])
    views: list                        # per image: (offset into the slot's pinned mask buffer, n, H, W)


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
    det_priority = 0
    seg_priority = 0
    # which stage's launches the host queues first (the other stage's kernels cannot start before the host gets to
    # them: ~600 detector launches are ~4 ms of host time, ~370 encoder launches ~2.5 ms); A/B'd by tools/stage_times.py
    encoder_first = True

    def __init__(self, detector: gd.GDinoEngine, segmentor: sm.SamEngine, overlap: bool = True):
        self.det, self.seg = detector, segmentor
        self.dev = detector.dev
        # The detector and the SAM image encoder are independent until the mask decoder needs the boxes: they run
        # on two HIP streams.  The encoder's GEMM workgroups fill a CU (144 KB LDS, 8 waves x 232 VGPRs), so the two
        # streams mostly time-slice rather than co-reside; what the second stream buys is the gaps (kernel tails,
        # the host round trip for the boxes): ~8 ms of the detector's ~20 ms per batch of 8.
        self.overlap = overlap
        # stream priorities (class attributes so that tools can A/B them): see DESIGN.md §7
        self.s_det = torch.cuda.Stream(device=self.dev, priority=self.det_priority) if overlap else None
        self.s_seg = torch.cuda.Stream(device=self.dev, priority=self.seg_priority) if overlap else None
        # host <-> device traffic of the host-to-host entry points rides its own streams, one per direction (on ONE
        # stream the upload of batch i+1 would queue behind the mask download of batch i and put both on the
        # critical path): the upload of batch i+1 and the download of batch i run under the neighbouring batch's
        # compute (two pinned slots)
        self.s_h2d = torch.cuda.Stream(device=self.dev)
        self.s_d2h = torch.cuda.Stream(device=self.dev)
        self._host_in: List[Optional[torch.Tensor]] = [None, None]
        self._host_out: List[Optional[torch.Tensor]] = [None, None]
        self._slot_free: List[Optional[torch.cuda.Event]] = [None, None]
        self._next_slot = 0

    def upload(self, images_rgb: Sequence[np.ndarray]) -> List[torch.Tensor]:
        """Decoded RGB sketches (HWC uint8, host) -> device, one transfer each.  Everything after this point -
        both resizes, normalisation, patchify - runs on the GPU."""
        out = []
        for im in images_rgb:
            a = np.ascontiguousarray(im)
            if not a.flags.writeable:
                a = a.copy()
            out.append(torch.from_numpy(a).to(self.dev, non_blocking=True))
        return out

    # ------------------------------------------------------------------ host-to-host (SURVEY §8d metric)
    @staticmethod
    def pinned_like(images_rgb: Sequence[np.ndarray]) -> List[torch.Tensor]:
        """Copies of decoded sketches in page-locked host memory (where an image decoder of a serving process would
        put them): the source of submit_host's asynchronous uploads."""
        return [torch.from_numpy(np.array(im, copy=True)).pin_memory() for im in images_rgb]

    @torch.no_grad()
    def submit_host(self, images_pinned: Sequence[torch.Tensor], top_n: Optional[int] = None) -> HostTicket:
        """The metric's unit of work, host memory to host memory: decoded RGB u8 sketches (pinned host tensors) ->
        boxes + uint8 0/1 masks [n, H, W] in pinned host memory (the reference returns exactly that: one numpy bool
        array per box, InkLayer/segmentor/sam.py:39-41; 1 byte per pixel, no bit-packing).  Returns as soon as the
        work is queued - the only host wait inside is the detector's boxes - so the caller can submit batch i+1
        before collecting batch i: the mask download of i (134 MB for 8 x 16 masks of 1024^2) and the upload of
        i+1 (25 MB) then overlap with compute.  Two slots: collect a ticket before submitting two more."""
        slot = self._next_slot
        self._next_slot ^= 1
        cur = torch.cuda.current_stream(self.dev)
        if self._slot_free[slot] is not None:
            self._slot_free[slot].synchronize()           # the download that last used this slot has finished
        with torch.cuda.stream(self.s_h2d):
            raw = [t.to(self.dev, non_blocking=True) for t in images_pinned]
            up = torch.cuda.Event()
            up.record(self.s_h2d)
        cur.wait_event(up)
        for r in raw:
            r.record_stream(cur)
        res = self.run_uploaded(raw, top_n=top_n)
        total = sum(int(r.masks.numel()) for r in res)
        if self._host_out[slot] is None or self._host_out[slot].numel() < total:
            self._host_out[slot] = torch.empty(max(total, 1), dtype=torch.uint8, pin_memory=True)
        hbuf = self._host_out[slot]
        self.s_d2h.wait_stream(cur)
        views, off = [], 0
        with torch.cuda.stream(self.s_d2h):
            # masks of consecutive images that share one postprocess launch are one contiguous device tensor:
            # copy run by run (usually ONE transfer per batch)
            i = 0
            while i < len(res):
                j, base = i, res[i].masks
                n_el = int(base.numel())
                while (j + 1 < len(res) and res[j + 1].masks.numel() > 0 and n_el > 0
                       and res[j + 1].masks.data_ptr() == base.data_ptr() + n_el
                       and res[j + 1].masks.untyped_storage().data_ptr() == base.untyped_storage().data_ptr()):
                    j += 1
                    n_el += int(res[j].masks.numel())
                if n_el > 0:
                    src = torch.as_strided(base, (n_el,), (1,))
                    src.record_stream(self.s_d2h)
                    hbuf[off:off + n_el].copy_(src, non_blocking=True)
                for k in range(i, j + 1):
                    m = res[k].masks
                    views.append((off, int(m.shape[0]), int(m.shape[1]), int(m.shape[2])))
                    off += int(m.numel())
                i = j + 1
            done = torch.cuda.Event()
            done.record(self.s_d2h)
        self._slot_free[slot] = done
        return HostTicket(slot, done, [(r.boxes_xyxy_norm, r.scores, r.boxes_pixel) for r in res], views)

    def collect_host(self, t: HostTicket):
        """Wait for a ticket's masks to be in host memory.  -> per image (boxes_xyxy_norm, scores, boxes_pixel,
        masks uint8 numpy [n, H, W]); the mask arrays are views of the slot's pinned buffer, valid until the
        slot is reused (two submits later)."""
        t.done.synchronize()
        hb = self._host_out[t.slot].numpy()
        return [(d[0], d[1], d[2], hb[off:off + n * h * w].reshape(n, h, w))
                for d, (off, n, h, w) in zip(t.dets, t.views)]

    def preprocess(self, raw: Sequence[torch.Tensor]):
        """GPU side of load_image / ResizeLongestSide (Pillow-exact bilinear, `ink_resize_bilinear_u8`).  SAM's
        channel quirk (InkLayer/segmentor/sam.py:24-26 feeds the encoder channel-reversed pixels) is applied by the
        patchify kernel (chan_reverse), not by a copy."""
        det_in, sam_in, sizes = [], [], []
        L = self.seg.cfg.img_size
        for im in raw:
            d, s_ = gpu_preprocess(im, L)
            det_in.append(d)
            sam_in.append(s_)              # RGB order; the channel reversal happens inside ink_sam_patchify
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
    def run_prepared(self, det_in, sam_in, sizes, top_n: Optional[int] = None) -> List[SketchResult]:
        """Detector and SAM encoder on two HIP streams, joined before the mask decoder; results are ready on the
        caller's stream.  (Letting consecutive batches overlap as well was measured slower - one batch already
        saturates the GPU - and was removed.)"""
        if self.overlap:
            cur = torch.cuda.current_stream(self.dev)
            self.s_det.wait_stream(cur)
            self.s_seg.wait_stream(cur)
            emb = None
            if self.encoder_first:
                with torch.cuda.stream(self.s_seg):
                    emb = self.seg.encode(sam_in, chan_reverse=True)
            with torch.cuda.stream(self.s_det):
                logits, boxes = self.det.forward(det_in, allow_graph=False)   # graphs do not overlap across streams
                both = torch.cat([logits, boxes], dim=-1)
                host = torch.empty(both.shape, dtype=both.dtype, pin_memory=True)
                host.copy_(both, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.s_det)
            if emb is None:
                with torch.cuda.stream(self.s_seg):
                    emb = self.seg.encode(sam_in, chan_reverse=True)
            ev.synchronize()                               # host needs the boxes; the encoder keeps running
            dets = self.det.postprocess(host, top_n=top_n)
            stream_ctx = torch.cuda.stream(self.s_seg)
        else:
            dets = self.det.detect(det_in, top_n=top_n)
            emb = self.seg.encode(sam_in, chan_reverse=True)
            stream_ctx = _NullCtx()
        with stream_ctx:
            out = self._decode_all(dets, emb, sizes)
        if self.overlap:
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_stream(self.s_seg)
            cur.wait_stream(self.s_det)
            for t in list(det_in) + list(sam_in):          # allocated on the caller's stream, read on s_det / s_seg
                t.record_stream(self.s_det)
                t.record_stream(self.s_seg)
            for r in out:                                  # allocated on s_seg, consumed on the caller's stream
                r.masks.record_stream(cur)
        return out

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
