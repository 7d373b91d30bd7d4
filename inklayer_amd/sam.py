"""SAM ViT-H segmentor on MI355X: host-side composition of the HIP kernels.

Mirrors the reference surfaces for this path
  * segment_anything.build_sam / SamPredictor  (SA/build_sam.py:55-107, SA/predictor.py:17-243)
  * InkLayer.segmentor.sam.run_SAM              (InkLayer/segmentor/sam.py:16-43)
with the same state_dict key names, so `sam_vit_h_4b8939.pth` drops in unchanged.

Design (MI355X-first, not a translation of the nn.Module tree):
  * tokens live as ONE f32 residual stream [B*4096, 1280]; every GEMM input is produced in
    f16 by the kernel that precedes it (LayerNorm, GELU epilogue, attention), every GEMM
    accumulates in f32 and applies bias / GELU / residual / window-unpartition in its epilogue;
  * window partition + zero padding is a row-gather fused into LayerNorm; un-partition + crop
    is a row-scatter fused into the proj GEMM epilogue (same int32 map for both);
  * attention never materialises scores; the decomposed rel-pos bias rides inside the MFMA
    accumulator (attention.hip);
  * ConvTranspose2d k2s2 = a [C -> 4*C'] projection whose pixel shuffle is deferred to the
    final mask-logit kernel; the two bilinear resizes + threshold are one kernel;
  * precision: the 32 ViT-H blocks run on plain f16 operands; the neck, the prompt / mask
    decoder, the upscaler and the hyper-network - 1 % of the FLOPs, but the layers whose f16 rounding dominated
    the mask error - run on SPLIT-f16 operands (hi + lo, three MFMA products per term, fp32-grade) with f32
    activations and f32-I/O attention, because the reference thresholds fp32 logits at exactly 0.
No torch compute ops are used on the hot path — torch supplies memory, streams, copies.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from ._lru import LRU

F16, F32 = torch.float16, torch.float32


@dataclass
class SamConfig:
    """SA/build_sam.py:14-21,55-101 (defaults = ViT-H, the model InkLayer ships)."""
    embed_dim: int = 1280
    depth: int = 32
    num_heads: int = 16
    global_attn_indexes: Tuple[int, ...] = (7, 15, 23, 31)
    window_size: int = 14
    img_size: int = 1024
    patch_size: int = 16
    mlp_ratio: float = 4.0
    prompt_embed_dim: int = 256
    dec_depth: int = 2
    dec_heads: int = 8
    dec_mlp_dim: int = 2048
    num_mask_tokens: int = 4
    pixel_mean: Tuple[float, ...] = (123.675, 116.28, 103.53)
    pixel_std: Tuple[float, ...] = (58.395, 57.12, 57.375)
    mask_threshold: float = 0.0

    @property
    def grid(self) -> int:
        return self.img_size // self.patch_size


def preprocess_shape(h: int, w: int, L: int) -> Tuple[int, int]:
    """ResizeLongestSide.get_preprocess_shape (SA/utils/transforms.py:93-102)."""
    sc = L * 1.0 / max(h, w)
    return int(h * sc + 0.5), int(w * sc + 0.5)


def resize_longest_side(img: np.ndarray, L: int) -> np.ndarray:
    """ResizeLongestSide.apply_image (SA/utils/transforms.py:26-31) with PIL on the host.  Not on the product
    path any more (ops.resize_bilinear_u8 is bit-identical on the GPU); kept as the reference the tests compare to."""
    from PIL import Image
    nh, nw = preprocess_shape(img.shape[0], img.shape[1], L)
    if (nh, nw) == img.shape[:2]:
        return np.ascontiguousarray(img)
    return np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))


def _to_dev_async(t: torch.Tensor, dev) -> torch.Tensor:
    """Small host tensor -> device through pinned memory, non-blocking.  A pageable `.to(dev)` makes the host wait
    until the stream has drained (here: the whole image encoder) before it can queue the decoder's launches."""
    return t.contiguous().pin_memory().to(dev, non_blocking=True)


class SamEngine:
    """Weights packed for the HIP kernels + preallocated activations for up to `max_batch` images."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: Optional[SamConfig] = None,
                 device: str | torch.device = "cuda", max_batch: int = 1, precise_tail: bool = True,
                 ln_fold: bool = False, bias_correction: bool = True):
        """precise_tail=True (the product setting): neck + decoder on split-f16 operands.  False keeps the plain
        f16 tail of round 1 (faster by a few %, mask IoU 0.998 instead of >= 0.999; kept for A/B measurements).
        ln_fold=True (built and parity-tested in round 3, NOT the product setting): the residual stream of the 32 blocks
        lives as two f16 planes (hi + lo), the hi plane is the operand of qkv / lin1 and both LayerNorms are folded into
        those projections - no LayerNorm kernel ever reads the stream.  Measured slower than what it removes (DESIGN.md
        §7: the split residual preload and the fold's prologue / epilogue cost 275 us per block against 90 us of
        LayerNorm kernels), so the product keeps the f32 stream + layernorm_rows (False).
        bias_correction=True (the product setting): the biases of the 4 x 32 block projections absorb the EXPECTED error
        of rounding their weights to f16 (see _calibrate_bias_correction)."""
        cfg = cfg or SamConfig()
        self.cfg, self.dev = cfg, torch.device(device)
        self.precise_tail = bool(precise_tail)
        self.ln_fold = bool(ln_fold)
        assert self.dev.type == "cuda", "the InkLayer segmentor runs on MI355X only"
        D, g = cfg.embed_dim, cfg.grid
        assert D // cfg.num_heads == 80 and g == 64 and cfg.window_size == 14, \
            "HIP attention kernels are specialised for SAM's head_dim 80 / 64x64 grid / 14x14 windows"
        E = cfg.prompt_embed_dim
        assert E // cfg.dec_heads == 32 and (E // 2) // cfg.dec_heads == 16
        self.T = g * g
        self.scale = 80 ** -0.5
        sd = state_dict
        dev = self.dev

        def h(name, shape=None):  # matrix -> f16 [N, K]
            t = sd[name].detach().to(torch.float32)
            if shape is not None:
                t = t.reshape(shape)
            return t.to(dev, F16).contiguous()

        def f(name):  # vector / table -> f32
            return ops.own_f32(sd[name], dev)

        self.w: Dict[str, torch.Tensor] = {}
        w = self.w
        P = cfg.patch_size
        w["pe.w"] = h("image_encoder.patch_embed.proj.weight", (D, 3 * P * P))
        w["pe.b"] = f("image_encoder.patch_embed.proj.bias")
        w["pos"] = f("image_encoder.pos_embed").reshape(self.T, D).contiguous()
        for i in range(cfg.depth):
            p = f"image_encoder.blocks.{i}."
            for n in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "attn.qkv.bias",
                      "attn.proj.bias", "mlp.lin1.bias", "mlp.lin2.bias", "attn.rel_pos_h",
                      "attn.rel_pos_w"):
                w[f"b{i}.{n}"] = f(p + n)
            for n in ("attn.qkv.weight", "attn.proj.weight", "mlp.lin1.weight", "mlp.lin2.weight"):
                w[f"b{i}.{n}"] = h(p + n)
            if self.ln_fold:
                # LayerNorm folded into the projection that consumes it: W' = gamma o W in f16, its row
                # sums (of the ROUNDED weights: what the MFMA really multiplies the mean with) and beta W^T + b
                for lin, nrm in (("attn.qkv", "norm1"), ("mlp.lin1", "norm2")):
                    W32 = sd[p + lin + ".weight"].detach().to(dev, torch.float32)
                    g32, b32 = w[f"b{i}.{nrm}.weight"], w[f"b{i}.{nrm}.bias"]
                    wl = (W32 * g32[None, :]).to(F16).contiguous()
                    w[f"b{i}.{lin}.w_ln"] = wl
                    w[f"b{i}.{lin}.colsum"] = wl.double().sum(1).float().contiguous()
                    w[f"b{i}.{lin}.bias_ln"] = (W32.double() @ b32.double() + w[f"b{i}.{lin}.bias"].double()).float().contiguous()
                    del w[f"b{i}.{lin}.weight"]               # the unfolded f16 copy is not used on this path
            if i not in cfg.global_attn_indexes:
                # window padding (image_encoder.py:256-259 pads AFTER norm1 with zeros): the k / v rows of a
                # padded token are qkv(0) = the bias, rounded to f16 exactly like a projected row would be
                qb = w[f"b{i}.attn.qkv.bias"]
                w[f"b{i}.pad_k"] = qb[D:2 * D].to(F16).contiguous()
                w[f"b{i}.pad_v"] = qb[2 * D:].to(F16).contiguous()
        w["neck0.w"] = h("image_encoder.neck.0.weight", (E, D))
        w["neck1.w"], w["neck1.b"] = f("image_encoder.neck.1.weight"), f("image_encoder.neck.1.bias")
        # 3x3 conv weight re-laid out as [co][(ky,kx,ci)] to match the NHWC im2col
        w["neck2.w"] = (sd["image_encoder.neck.2.weight"].detach().to(torch.float32)
                        .permute(0, 2, 3, 1).reshape(E, 9 * E).to(dev, F16).contiguous())
        w["neck3.w"], w["neck3.b"] = f("image_encoder.neck.3.weight"), f("image_encoder.neck.3.bias")

        # ---- prompt encoder constants
        w["gauss"] = f("prompt_encoder.pe_layer.positional_encoding_gaussian_matrix")
        w["corner"] = torch.cat([f("prompt_encoder.point_embeddings.2.weight"),
                                 f("prompt_encoder.point_embeddings.3.weight")], 0).contiguous()
        w["no_mask"] = f("prompt_encoder.no_mask_embed.weight").reshape(-1).contiguous()
        # dense positional encoding of the 64x64 grid (prompt_encoder.py:195-206): constant
        ar = (torch.arange(g, dtype=torch.float32) + 0.5) / g
        grid_xy = torch.stack([ar[None, :].expand(g, g), ar[:, None].expand(g, g)], -1).reshape(-1, 2)
        self.dense_pe = ops.sam_pe_encode(grid_xy.contiguous().to(dev), w["gauss"])  # [T, E]

        # ---- mask decoder
        t = "mask_decoder.transformer."
        def attn(dst, src, fuse_qkv=False, fuse_qk=False):
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                w[f"{dst}.{n}.w"] = h(f"{src}.{n}.weight")
                w[f"{dst}.{n}.b"] = f(f"{src}.{n}.bias")
            if fuse_qkv:
                w[f"{dst}.qkv.w"] = torch.cat([w[f"{dst}.{n}.w"] for n in ("q_proj", "k_proj", "v_proj")]).contiguous()
                w[f"{dst}.qkv.b"] = torch.cat([w[f"{dst}.{n}.b"] for n in ("q_proj", "k_proj", "v_proj")]).contiguous()
            if fuse_qk:
                w[f"{dst}.qk.w"] = torch.cat([w[f"{dst}.{n}.w"] for n in ("q_proj", "k_proj")]).contiguous()
                w[f"{dst}.qk.b"] = torch.cat([w[f"{dst}.{n}.b"] for n in ("q_proj", "k_proj")]).contiguous()
        for i in range(cfg.dec_depth):
            p = f"{t}layers.{i}."
            attn(f"d{i}.self", p + "self_attn", fuse_qkv=(i == 0), fuse_qk=(i > 0))
            attn(f"d{i}.t2i", p + "cross_attn_token_to_image")
            attn(f"d{i}.i2t", p + "cross_attn_image_to_token")
            for n in ("norm1", "norm2", "norm3", "norm4"):
                w[f"d{i}.{n}.w"], w[f"d{i}.{n}.b"] = f(p + n + ".weight"), f(p + n + ".bias")
            w[f"d{i}.lin1.w"], w[f"d{i}.lin1.b"] = h(p + "mlp.lin1.weight"), f(p + "mlp.lin1.bias")
            w[f"d{i}.lin2.w"], w[f"d{i}.lin2.b"] = h(p + "mlp.lin2.weight"), f(p + "mlp.lin2.bias")
        attn("dfin", t + "final_attn_token_to_image")
        w["dfin.norm.w"], w["dfin.norm.b"] = f(t + "norm_final_attn.weight"), f(t + "norm_final_attn.bias")
        w["out_tok"] = torch.cat([f("mask_decoder.iou_token.weight"),
                                  f("mask_decoder.mask_tokens.weight")], 0).contiguous()  # [5, E]
        # ConvTranspose2d(k2,s2) as a projection to (dy,dx,co): W'[(dy*2+dx)*Co + co][ci]
        u = "mask_decoder.output_upscaling."
        w["up0.w"] = (sd[u + "0.weight"].detach().to(torch.float32).permute(2, 3, 1, 0)
                      .reshape(4 * (E // 4), E).to(dev, F16).contiguous())
        w["up0.b"] = f(u + "0.bias").repeat(4).contiguous()
        w["up1.w"], w["up1.b"] = f(u + "1.weight"), f(u + "1.bias")
        w["up3.w"] = (sd[u + "3.weight"].detach().to(torch.float32).permute(2, 3, 1, 0)
                      .reshape(4 * (E // 8), E // 4).to(dev, F16).contiguous())
        w["up3.b"] = f(u + "3.bias").repeat(4).contiguous()
        for j in range(3):  # hyper-network of mask token 0 (multimask_output=False keeps mask 0 only)
            w[f"hyp{j}.w"] = h(f"mask_decoder.output_hypernetworks_mlps.0.layers.{j}.weight")
            w[f"hyp{j}.b"] = f(f"mask_decoder.output_hypernetworks_mlps.0.layers.{j}.bias")
            w[f"iou{j}.w"] = h(f"mask_decoder.iou_prediction_head.layers.{j}.weight")
            w[f"iou{j}.b"] = f(f"mask_decoder.iou_prediction_head.layers.{j}.bias")
        # the last hyper layer has N = 32 outputs, the iou head N = 4: both fine for the GEMM (N % 4)

        if self.precise_tail:
            self._pack_split(sd)
        self._alloc(max_batch)
        if bias_correction:
            self._calibrate_bias_correction(sd)

    # ------------------------------------------------------------------ load-time bias correction
    @torch.no_grad()
    def _calibrate_bias_correction(self, sd) -> None:
        """More than half of the energy of the ViT-H residual stream - and of the attention / GELU outputs - is a
        per-channel constant (the same for every token, nearly the same for every sketch).  Rounding a weight matrix to
        f16 therefore produces, besides noise, a SYSTEMATIC output error  sum_k (W - f16(W))[n, k] * mean_k  that is the
        same for every token: a bias error.  It is measured once, at load time, on a calibration page (one encoder
        forward of this engine on a blank sketch page; torch reductions, never on the hot path) and folded into the bias:
        b_n += sum_k (W_eff - f16(W_eff))[n, k] * mean_k,  W_eff = the weights as the kernel sees them (gamma o W for the
        LayerNorm-folded projections, whose operand mean is that of the normalised stream).  Zero run-time cost; measured
        with tests/precision_study.py it halves the logit error of the 32 blocks (the known "bias correction" of
        post-training quantisation, applied to f16)."""
        cfg, w, D = self.cfg, self.w, self.cfg.embed_dim
        page = torch.full((cfg.img_size, cfg.img_size, 3), 255, dtype=torch.uint8, device=self.dev)
        means: Dict[str, torch.Tensor] = {}
        self._calib = means
        try:
            self.encode([page])
        finally:
            self._calib = None
        for i in range(cfg.depth):
            p = f"image_encoder.blocks.{i}."
            for lin, nrm in (("attn.qkv", "norm1"), ("mlp.lin1", "norm2"), ("attn.proj", None), ("mlp.lin2", None)):
                W32 = sd[p + lin + ".weight"].detach().to(self.dev, torch.float32)
                if nrm is not None and self.ln_fold:
                    W32 = W32 * w[f"b{i}.{nrm}.weight"][None, :]
                    w16, bkey = w[f"b{i}.{lin}.w_ln"], f"b{i}.{lin}.bias_ln"
                else:
                    w16, bkey = w[f"b{i}.{lin}.weight"], f"b{i}.{lin}.bias"
                # a NEW tensor: w[bkey] may share storage with the caller's state dict, which must stay untouched
                w[bkey] = (w[bkey].double() + (W32 - w16.float()).double() @ means[f"b{i}.{lin}"].double()).float().contiguous()

    def _pack_split(self, sd) -> None:
        """Split-f16 weight copies [N, 3K] (ops.split_weight) of the neck / decoder matrices, keyed '<name>.ws'."""
        cfg, w, dev = self.cfg, self.w, self.dev
        E, D = cfg.prompt_embed_dim, cfg.embed_dim

        def m(name):
            return sd[name].detach().to(dev, torch.float32)

        P = cfg.patch_size
        w["pe.ws"] = ops.split_weight(m("image_encoder.patch_embed.proj.weight").reshape(D, 3 * P * P))
        w["neck0.ws"] = ops.split_weight(m("image_encoder.neck.0.weight").reshape(E, D))
        # 3x3 conv as [co][(ky,kx)][ci]: the im2col of a split activation row is (ky,kx) x [hi | lo | hi/64]
        w["neck2.ws"] = ops.split_weight(m("image_encoder.neck.2.weight").permute(0, 2, 3, 1).reshape(E, 9, E)) \
            .reshape(E, 27 * E).contiguous()
        t = "mask_decoder.transformer."

        def attn(dst, src):
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                w[f"{dst}.{n}.ws"] = ops.split_weight(m(f"{src}.{n}.weight"))

        for i in range(cfg.dec_depth):
            p = f"{t}layers.{i}."
            for a_, b_ in ((f"d{i}.self", p + "self_attn"), (f"d{i}.t2i", p + "cross_attn_token_to_image"),
                           (f"d{i}.i2t", p + "cross_attn_image_to_token")):
                attn(a_, b_)
            d = f"d{i}.self"
            if i == 0:
                w[d + ".qkv.ws"] = torch.cat([w[f"{d}.{n}.ws"] for n in ("q_proj", "k_proj", "v_proj")]).contiguous()
            else:
                w[d + ".qk.ws"] = torch.cat([w[f"{d}.{n}.ws"] for n in ("q_proj", "k_proj")]).contiguous()
            w[f"d{i}.lin1.ws"] = ops.split_weight(m(p + "mlp.lin1.weight"))
            w[f"d{i}.lin2.ws"] = ops.split_weight(m(p + "mlp.lin2.weight"))
        attn("dfin", t + "final_attn_token_to_image")
        # (keys + pe) W = keys W + pe W: ONE split pass of the image keys feeds the token->image k and v projections and
        # the image->token q projection (one GEMM, N = 384); pe W is a per-position constant [T, 128] that the attention
        # kernels add (k_add / q_add).  The constants come from the same split-f16 GEMM (fp32-grade).
        pe_split = ops.add_split_f16(self.dense_pe)
        for i in range(cfg.dec_depth):
            d = f"d{i}"
            w[d + ".kvq.ws"] = torch.cat([w[d + ".t2i.k_proj.ws"], w[d + ".t2i.v_proj.ws"], w[d + ".i2t.q_proj.ws"]]).contiguous()
            w[d + ".kvq.b"] = torch.cat([w[d + ".t2i.k_proj.b"], w[d + ".t2i.v_proj.b"], w[d + ".i2t.q_proj.b"]]).contiguous()
            w[d + ".t2i.k_pe"] = ops.gemm(pe_split, w[d + ".t2i.k_proj.ws"]).contiguous()
            w[d + ".i2t.q_pe"] = ops.gemm(pe_split, w[d + ".i2t.q_proj.ws"]).contiguous()
        w["dfin.kv.ws"] = torch.cat([w["dfin.k_proj.ws"], w["dfin.v_proj.ws"]]).contiguous()
        w["dfin.kv.b"] = torch.cat([w["dfin.k_proj.b"], w["dfin.v_proj.b"]]).contiguous()
        w["dfin.k_pe"] = ops.gemm(pe_split, w["dfin.k_proj.ws"]).contiguous()
        u = "mask_decoder.output_upscaling."
        w["up0.ws"] = ops.split_weight(m(u + "0.weight").permute(2, 3, 1, 0).reshape(4 * (E // 4), E))
        w["up3.ws"] = ops.split_weight(m(u + "3.weight").permute(2, 3, 1, 0).reshape(4 * (E // 8), E // 4))
        # the final token->image k / v projections and the first transposed convolution read the same split operand of
        # the final keys: ONE GEMM [k | v | up0] (N = 128 + 128 + 256)
        w["dfin.kvu.ws"] = torch.cat([w["dfin.kv.ws"], w["up0.ws"]]).contiguous()
        w["dfin.kvu.b"] = torch.cat([w["dfin.kv.b"], w["up0.b"]]).contiguous()
        if self.fuse_proj_ln and E == 256:
            for i in range(cfg.dec_depth):
                w[f"d{i}.i2t.out_proj.blob"] = ops.proj256_ln_pack(w[f"d{i}.i2t.out_proj.ws"].contiguous())   # csrc/proj_ln.hip
        if self.fuse_upscale_tail and tuple(w["up3.ws"].shape) == (128, 192):
            w["up3.blob"] = ops.sam_upscale_pack(w["up3.ws"].contiguous())          # csrc/upscale_tail.hip
        for j in range(3):
            w[f"hyp{j}.ws"] = ops.split_weight(m(f"mask_decoder.output_hypernetworks_mlps.0.layers.{j}.weight"))
            w[f"iou{j}.ws"] = ops.split_weight(m(f"mask_decoder.iou_prediction_head.layers.{j}.weight"))

    # ------------------------------------------------------------------ buffers
    def _alloc(self, B: int) -> None:
        cfg, dev, T, D = self.cfg, self.dev, self.T, self.cfg.embed_dim
        g, ws = cfg.grid, cfg.window_size
        self.max_batch = B
        nwin = -(-g // ws)                       # 5 windows per side (64 -> 70 padded)
        self.nwin = nwin
        Mw = nwin * nwin * ws * ws               # 4900 rows per image after padding
        self.Mw = Mw
        self._enc_graphs = LRU(self.graph_cache_size)
        self._enc_seen: Dict[int, int] = {}
        # window gather/scatter map (window_partition / window_unpartition, image_encoder.py:243-289)
        r = torch.arange(B * Mw)
        b, rr = r // Mw, r % Mw
        win, pos = rr // (ws * ws), rr % (ws * ws)
        y = (win // nwin) * ws + pos // ws
        x = (win % nwin) * ws + pos % ws
        m = torch.where((y < g) & (x < g), b * T + y * g + x, torch.full_like(r, -1))
        self.win_map = m.to(torch.int32).to(dev)
        P = cfg.patch_size
        e = lambda *s, dt=F16: torch.empty(s, device=dev, dtype=dt)
        self.buf_patches = e(B * T, 3 * P * P * (3 if self.precise_tail else 1))
        self.x = e(B * T, D, dt=F32)
        self.pos_rep = self.w["pos"].repeat(B, 1).contiguous()      # position embedding tiled over the batch (residual of the patch projection)
        self.y = e(B * T, D)
        if self.ln_fold:
            self.x_hi, self.x_lo = e(B * T, D), e(B * T, D)
            # row statistics of the stream, per column chunk of the projection kernel that writes them (the chunk
            # width depends on the tile variant the row count selects: sized for the finest one, viewed per call)
            self.x_stats = e(B * T * (D // 64) * 2, dt=F32)
        self.qkv = e(B * T, 3 * D)
        self.att = e(B * T, D)
        self.hid = e(B * T, int(D * cfg.mlp_ratio))
        H = cfg.num_heads
        self.rel_aug = e(B * nwin * nwin * H * ws * ws, 32)
        # decomposed rel-pos terms of the global blocks: f16 tables at SAM's own grid (the one-wave-per-SIMD kernel converts
        # them on the way into LDS / the exponent: half the bytes written by relpos_bias and read back), f32 otherwise
        rdt = F16 if (self.rel_f16_tables and cfg.grid == 64) else F32
        self.rel_h = e(B * H * T, 64, dt=rdt)
        self.rel_w = e(B * H * T, 64, dt=rdt)

    # ------------------------------------------------------------------ encoder
    def encode(self, images_u8: Sequence[torch.Tensor], chan_reverse: bool = False,
               upto: Optional[int] = None) -> torch.Tensor:
        """images_u8: resized HWC uint8 CUDA tensors (long side = img_size).
        Returns the image embeddings as tokens [B, 4096, 256] f32 (NHWC; the reference's NCHW
        `features` is `.permute(0, 2, 1).view(B, 256, 64, 64)`)."""
        cfg, w, T, D = self.cfg, self.w, self.T, self.cfg.embed_dim
        B = len(images_u8)
        assert 1 <= B <= self.max_batch
        H, Mw = cfg.num_heads, self.Mw
        x = self.x[:B * T]
        for b, img in enumerate(images_u8):
            # the patch embedding runs on split-f16 operands too: its rounding error would sit in the residual stream
            # of all 32 blocks (measured: the largest single contribution to the mask error)
            ops.sam_patchify(img, cfg.img_size, cfg.patch_size, cfg.pixel_mean, cfg.pixel_std,
                             chan_reverse, self.buf_patches[b * T:(b + 1) * T], split=self.precise_tail)
        # ONE projection for the batch (at 8 images N = 1280 and 512 tiles make it a ping-pong-kernel launch: ~1000 TFLOP/s
        # instead of eight 128-tile launches at ~500), the position embedding as a residual tiled per image
        ops.gemm(self.buf_patches[:B * T], w["pe.ws" if self.precise_tail else "pe.w"], w["pe.b"],
                 residual=self.pos_rep[:B * T], out=x)
        if upto is None and self.graph_blocks and ops.tracing_off():
            return self._blocks_graphed(B)
        return self._blocks(B, upto)

    # ------------------------------------------------------------------ HIP graph of the 32 blocks + neck
    # The blocks and the neck only touch engine-owned buffers of fixed shape per batch size: ~370 launches that can be
    # replayed as one graph.  Measured at batch 8 (tools/host_time.py, tools/bench_ab.py): the host needs 2.4 ms to
    # issue them eagerly (6.4 us per launch through ctypes) and 0.2 ms to replay the graph, against 53 ms of GPU work -
    # the step is not host-bound, and same-box A/B runs of the whole pipeline agree within the pool's noise (76.8-77.7 ms
    # either way).  So the graph is OFF by default and kept for hosts with slow cores (one attribute to flip).  A batch
    # size is captured the second time it is seen (a capture costs three forwards and pins a private pool for the
    # neck's intermediates); the result is cloned out of that pool.
    graph_blocks = False
    rel_f16_tables = True        # global attention's rel-pos tables in f16 (grid 64 only)
    fuse_proj_ln = True          # image-side out_proj + residual + norm4 (+ split operand) as one kernel (csrc/proj_ln.hip)
    fuse_upscale_tail = True     # LayerNorm2d + GELU + ConvT + GELU + hyper product as one kernel (csrc/upscale_tail.hip)
    graph_cache_size = 2

    def _blocks_graphed(self, B: int) -> torch.Tensor:
        g = self._enc_graphs.get(B)
        if g is None:
            self._enc_seen[B] = self._enc_seen.get(B, 0) + 1
            if self._enc_seen[B] < 2:
                return self._blocks(B, None)
            x0 = self.x[:B * self.T].clone()              # the capture's warm-ups run the blocks in place
            cur = torch.cuda.current_stream(self.dev)
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):                 # warm-up off the capture: lazy attribute calls, plans
                self._blocks(B, None)
                self.x[:B * self.T].copy_(x0)
            cur.wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._blocks(B, None)
            self.x[:B * self.T].copy_(x0)
            g = (graph, out)
            self._enc_graphs.put(B, g)
        g[0].replay()
        return g[1].clone()                               # the static output is overwritten by the next replay

    def _blocks(self, B: int, upto: Optional[int]) -> torch.Tensor:
        cfg, w, T, D = self.cfg, self.w, self.T, self.cfg.embed_dim
        H, Mw = cfg.num_heads, self.Mw
        x = self.x[:B * T]
        nblk = cfg.depth if upto is None else upto
        fold = self.ln_fold
        if fold:
            # the stream as two f16 planes + its row statistics (written by every projection that updates it)
            xh, xl = self.x_hi[:B * T], self.x_lo[:B * T]
            chunk = ops.gemm_stats_chunk(B * T, D, D)
            st = self.x_stats[:B * T * (D // chunk) * 2].view(B * T, D // chunk, 2)
            ops.hilo_split_stats(x, xh, xl, st, chunk)
            hilo = (xh, xl)

        calib = getattr(self, "_calib", None)

        def note(key, lin, nrm):
            """calibration pass only: per-channel mean (over the tokens) of the operand of projection `lin` as the
            kernel's arithmetic sees it - the normalised stream for a folded LayerNorm, else the f16 operand"""
            if calib is not None and fold:
                x32 = ops.hilo_join(xh, xl)
                calib[key + lin] = torch.nn.functional.layer_norm(x32, (D,), None, None, 1e-6).mean(0)

        def qkv_of(k):
            note(k, "attn.qkv", "norm1")
            if fold:
                return ops.gemm(xh, w[k + "attn.qkv.w_ln"], w[k + "attn.qkv.bias_ln"], out=self.qkv[:B * T],
                                ln=(st, D, 1e-6, w[k + "attn.qkv.colsum"]))
            y = ops.layernorm_rows(x, w[k + "norm1.weight"], w[k + "norm1.bias"], 1e-6, out=self.y[:B * T])
            if calib is not None:
                calib[k + "attn.qkv"] = y.float().mean(0)
            return ops.gemm(y, w[k + "attn.qkv.weight"], w[k + "attn.qkv.bias"], out=self.qkv[:B * T])

        def add_proj(k, o):
            if fold:
                ops.gemm(o, w[k + "attn.proj.weight"], w[k + "attn.proj.bias"], residual_hilo=hilo, out_hilo=hilo, stats_out=st)
            else:
                ops.gemm(o, w[k + "attn.proj.weight"], w[k + "attn.proj.bias"], residual=x, out=x)

        for i in range(nblk):
            k = f"b{i}."
            qkv = qkv_of(k)
            q, kk, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
            if i in cfg.global_attn_indexes:
                rh, rw = ops.relpos_bias(q, w[k + "attn.rel_pos_h"], w[k + "attn.rel_pos_w"], S=cfg.grid,
                                         n_batch=B, n_heads=H, head_dim=80, scale=self.scale,
                                         out=(self.rel_h, self.rel_w), f16_tables=self.rel_h.dtype == F16)
                o = ops.flash_attn(q, kk, v, n_batch=B, n_heads=H, head_dim=80, scale=self.scale,
                                   rel_h=rh, rel_w=rw, grid_w=cfg.grid, out=self.att[:B * T])
            else:
                # windowed block: everything stays in token order; window_partition / window_unpartition
                # (image_encoder.py:243-289) is the token-row map the attention gathers and scatters through,
                # so the 19.6 % padding rows (70x70 vs 64x64) are never normalised, projected or written
                wm = self.win_map[:B * Mw]
                nb = B * self.nwin * self.nwin
                aug = ops.relpos_bias(q, w[k + "attn.rel_pos_h"], w[k + "attn.rel_pos_w"],
                                      S=cfg.window_size, n_batch=nb, n_heads=H, head_dim=80,
                                      scale=self.scale, out=self.rel_aug, tok_rows=wm)
                o = ops.flash_attn(q, kk, v, n_batch=nb, n_heads=H, head_dim=80, scale=self.scale,
                                   n_q=cfg.window_size ** 2, n_k=cfg.window_size ** 2,
                                   rel_aug=aug, grid_w=cfg.window_size, tok_rows=wm,
                                   pad_k=w[k + "pad_k"], pad_v=w[k + "pad_v"], out=self.att[:B * T])
            if calib is not None:
                calib[k + "attn.proj"] = o.float().mean(0)
            add_proj(k, o)
            if fold:
                note(k, "mlp.lin1", "norm2")
                hd = ops.gemm(xh, w[k + "mlp.lin1.w_ln"], w[k + "mlp.lin1.bias_ln"], act="gelu", out=self.hid[:B * T],
                              ln=(st, D, 1e-6, w[k + "mlp.lin1.colsum"]))
                if calib is not None:
                    calib[k + "mlp.lin2"] = hd.float().mean(0)
                ops.gemm(hd, w[k + "mlp.lin2.weight"], w[k + "mlp.lin2.bias"], residual_hilo=hilo, out_hilo=hilo, stats_out=st)
            else:
                y = ops.layernorm_rows(x, w[k + "norm2.weight"], w[k + "norm2.bias"], 1e-6, out=self.y[:B * T])
                hd = ops.gemm(y, w[k + "mlp.lin1.weight"], w[k + "mlp.lin1.bias"], act="gelu", out=self.hid[:B * T])
                if calib is not None:
                    calib[k + "mlp.lin1"], calib[k + "mlp.lin2"] = y.float().mean(0), hd.float().mean(0)
                ops.gemm(hd, w[k + "mlp.lin2.weight"], w[k + "mlp.lin2.bias"], residual=x, out=x)
        if fold:
            ops.hilo_join(xh, xl, out=x)                 # f32 view of the stream for the neck / the stage taps
        if upto is not None:
            return x.view(B, T, D)
        # neck (image_encoder.py:88-104): 1x1 conv -> LN2d -> 3x3 conv -> LN2d, all on NHWC tokens
        if self.precise_tail:
            n0 = ops.gemm(ops.add_split_f16(x), w["neck0.ws"])
            n1 = ops.layernorm_rows(n0, w["neck1.w"], w["neck1.b"], 1e-6, split=True)       # [B*T, 3E]
            n2 = ops.gemm(ops.im2col3x3(n1, B, cfg.grid, cfg.grid), w["neck2.ws"])
            emb = ops.layernorm_rows(n2, w["neck3.w"], w["neck3.b"], 1e-6, out_dtype=F32)
            return emb.view(B, T, cfg.prompt_embed_dim)
        xh = ops.add_cvt_f16(x, out=self.y[:B * T])
        n0 = ops.gemm(xh, w["neck0.w"])
        n1 = ops.layernorm_rows(n0, w["neck1.w"], w["neck1.b"], 1e-6)
        col = ops.im2col3x3(n1, B, cfg.grid, cfg.grid)
        n2 = ops.gemm(col, w["neck2.w"])
        emb = ops.layernorm_rows(n2, w["neck3.w"], w["neck3.b"], 1e-6, out_dtype=F32)
        return emb.view(B, T, cfg.prompt_embed_dim)

    # ------------------------------------------------------------------ decoder
    def _dec_attn(self, name: str, q16, k16, v16, n: int, n_q: int, n_k: int, hd: int,
                  residual: Optional[torch.Tensor], kv_rows=None, q_rows=None,
                  pre_q=None, pre_k=None, pre_v=None) -> torch.Tensor:
        """transformer.py Attention.forward (:218-240): projections + fused attention + out_proj.
        pre_* let a caller pass already-projected operands (shared across boxes)."""
        w = self.w
        Hh = self.cfg.dec_heads
        q = pre_q if pre_q is not None else ops.gemm(q16, w[name + ".q_proj.w"], w[name + ".q_proj.b"], out_dtype=F16)
        k = pre_k if pre_k is not None else ops.gemm(k16, w[name + ".k_proj.w"], w[name + ".k_proj.b"], out_dtype=F16)
        v = pre_v if pre_v is not None else ops.gemm(v16, w[name + ".v_proj.w"], w[name + ".v_proj.b"], out_dtype=F16)
        if n_q <= 8 and n_k >= 256:      # tokens -> image: few queries, many keys
            a = ops.attn_fewq(q, k, v, n_batch=n, n_heads=Hh, head_dim=hd, scale=1.0 / math.sqrt(hd), n_q=n_q,
                              n_k=n_k, q_batch_rows=q_rows, kv_batch_rows=kv_rows)
        else:
            a = ops.flash_attn(q, k, v, n_batch=n, n_heads=Hh, head_dim=hd, scale=1.0 / math.sqrt(hd),
                               n_q=n_q, n_k=n_k, q_batch_rows=q_rows, kv_batch_rows=kv_rows)
        return ops.gemm(a, w[name + ".out_proj.w"], w[name + ".out_proj.b"], residual=residual)

    def decode(self, emb: torch.Tensor, boxes_input_frame: np.ndarray | torch.Tensor,
               input_hw: Tuple[int, int], orig_hw: Tuple[int, int], want_logits: bool = False):
        """One image: emb [4096, 256] f32 tokens, boxes [n, 4] xyxy in the resized-input frame
        (host memory).  PromptEncoder + MaskDecoder(multimask_output=False) + postprocess.
        Returns uint8 masks [n, H, W] on the GPU (+ low-res logits, iou, full logits if asked)."""
        boxes = torch.as_tensor(np.asarray(boxes_input_frame), dtype=torch.float32).reshape(-1, 4)
        low, iou = self.decode_low_res(emb.reshape(1, self.T, -1), boxes, [0] * boxes.shape[0])
        res = ops.sam_postprocess(low, self.cfg.img_size, input_hw, orig_hw, self.cfg.mask_threshold, want_logits)
        if want_logits:
            return res[0], low, iou, res[1]
        return res

    def decode_low_res(self, emb: torch.Tensor, boxes: torch.Tensor, img_of_box: Sequence[int]):
        """Prompt encoder + mask decoder for N boxes spread over B images in ONE pass (the reference decodes
        one image at a time; batching only changes which rows share a launch).  emb [B, 4096, 256] f32,
        boxes [N, 4] xyxy in the resized-input frame (host), img_of_box[i] = image of box i.
        -> (low-res logits [N, 256, 256] f32, iou [N, 1] f32)."""
        if self.precise_tail:
            return self._decode_low_res_split(emb, boxes, img_of_box)
        cfg, w, T, dev = self.cfg, self.w, self.T, self.dev
        E, L, g = cfg.prompt_embed_dim, cfg.img_size, cfg.grid
        B = emb.shape[0]
        n = boxes.shape[0]
        assert n > 0 and len(img_of_box) == n
        NT = 5 + 2                                         # iou + 4 mask tokens + 2 box corners
        # --- prompt encoder (_embed_boxes): host-side affine of 4n numbers, Fourier features on GPU
        coords = _to_dev_async((boxes + 0.5).reshape(-1, 2) / float(L), dev)
        sparse = ops.sam_pe_encode(coords, w["gauss"], add=w["corner"])           # [2n, E]
        tokens = torch.empty((n, NT, E), device=dev, dtype=F32)
        tokens[:, :5] = w["out_tok"]                       # plumbing copies (no math)
        tokens[:, 5:] = sparse.view(n, 2, E)
        qpe = tokens.view(n * NT, E)
        iob = torch.as_tensor(list(img_of_box), dtype=torch.int64)
        img_rows = _to_dev_async((iob * T).to(torch.int32), dev)       # first key row of each box's image
        iob_dev = _to_dev_async(iob, dev)

        # --- image side, shared by all boxes of an image: src = emb + no_mask_embed; key_pe = dense PE
        keys0 = ops.add_f32(emb.reshape(B * T, E).contiguous(), w["no_mask"])      # [B*T, E]
        kpe = self.dense_pe
        queries = qpe
        keys = None                                        # per-box keys [n*T, E] after layer 0
        for i in range(cfg.dec_depth):
            d = f"d{i}"
            # (1) token self-attention
            if i == 0:
                t16 = ops.add_cvt_f16(queries)
                qkv = ops.gemm(t16, w[d + ".self.qkv.w"], w[d + ".self.qkv.b"], out_dtype=F16)
                queries = self._dec_attn(d + ".self", None, None, None, n, NT, NT, 32, None,
                                         pre_q=qkv[:, :E], pre_k=qkv[:, E:2 * E], pre_v=qkv[:, 2 * E:])
            else:
                qk16 = ops.add_cvt_f16(queries, qpe)
                qk = ops.gemm(qk16, w[d + ".self.qk.w"], w[d + ".self.qk.b"], out_dtype=F16)
                queries = self._dec_attn(d + ".self", None, None, ops.add_cvt_f16(queries), n, NT, NT, 32,
                                         queries, pre_q=qk[:, :E], pre_k=qk[:, E:])
            queries = ops.layernorm_rows(queries, w[d + ".norm1.w"], w[d + ".norm1.b"], 1e-5, out_dtype=F32)
            # (2) tokens -> image
            q16 = ops.add_cvt_f16(queries, qpe)
            if keys is None:
                k16 = ops.add_cvt_f16(keys0, kpe)
                v16 = ops.add_cvt_f16(keys0)
                att = self._dec_attn(d + ".t2i", q16, k16, v16, n, NT, T, 16, queries, kv_rows=img_rows)
            else:
                k16 = ops.add_cvt_f16(keys, kpe)
                v16 = ops.add_cvt_f16(keys)
                att = self._dec_attn(d + ".t2i", q16, k16, v16, n, NT, T, 16, queries)
            queries = ops.layernorm_rows(att, w[d + ".norm2.w"], w[d + ".norm2.b"], 1e-5, out_dtype=F32)
            # (3) token MLP
            hmid = ops.gemm(ops.add_cvt_f16(queries), w[d + ".lin1.w"], w[d + ".lin1.b"], act="relu",
                            out_dtype=F16)
            queries = ops.layernorm_rows(ops.gemm(hmid, w[d + ".lin2.w"], w[d + ".lin2.b"], residual=queries),
                                         w[d + ".norm3.w"], w[d + ".norm3.b"], 1e-5, out_dtype=F32)
            # (4) image -> tokens (q = keys + key_pe = k16 from step 2)
            tk16 = ops.add_cvt_f16(queries, qpe)
            tv16 = ops.add_cvt_f16(queries)
            if keys is None:
                # per-box copy of the image keys (repeat_interleave of mask_decoder.py:124; a pure memory copy)
                keys = keys0.view(B, T * E).index_select(0, iob_dev).view(n * T, E)
                att = self._dec_attn(d + ".i2t", k16, tk16, tv16, n, T, NT, 16, keys, q_rows=img_rows)
            else:
                att = self._dec_attn(d + ".i2t", k16, tk16, tv16, n, T, NT, 16, keys)
            keys = ops.layernorm_rows(att, w[d + ".norm4.w"], w[d + ".norm4.b"], 1e-5, out_dtype=F32)
        # final tokens -> image attention
        q16 = ops.add_cvt_f16(queries, qpe)
        att = self._dec_attn("dfin", q16, ops.add_cvt_f16(keys, kpe), ops.add_cvt_f16(keys), n, NT, T, 16,
                             queries)
        queries = ops.layernorm_rows(att, w["dfin.norm.w"], w["dfin.norm.b"], 1e-5, out_dtype=F32)
        hs16 = ops.add_cvt_f16(queries).view(n, NT * E)

        def mlp3(prefix: str, x16: torch.Tensor) -> torch.Tensor:
            a = ops.gemm(x16, w[prefix + "0.w"], w[prefix + "0.b"], act="relu", out_dtype=F16)
            a = ops.gemm(a, w[prefix + "1.w"], w[prefix + "1.b"], act="relu", out_dtype=F16)
            return ops.gemm(a, w[prefix + "2.w"], w[prefix + "2.b"])

        hyper = mlp3("hyp", hs16[:, E:2 * E])               # mask token 0 -> [n, 32]
        iou = mlp3("iou", hs16[:, :E])[:, :1]               # iou token -> [n, 4] -> mask 0
        # upscaling: ConvT(256->64) -> LN2d -> GELU -> ConvT(64->32) -> GELU, un-shuffled
        u0 = ops.gemm(ops.add_cvt_f16(keys), w["up0.w"], w["up0.b"])             # [n*T, 4*64]
        u1 = ops.layernorm_rows(u0.view(n * T * 4, E // 4), w["up1.w"], w["up1.b"], 1e-6, act="gelu")
        u2 = ops.gemm(u1, w["up3.w"], w["up3.b"], act="gelu")                     # [n*T*4, 4*32]
        low = ops.sam_mask_logits(u2, hyper.contiguous(), n, g)                   # [n, 256, 256]
        return low, iou

    def _decode_low_res_split(self, emb: torch.Tensor, boxes: torch.Tensor, img_of_box: Sequence[int]):
        """decode_low_res on split-f16 operands: every activation stays f32, every projection multiplies a
        [hi | lo*64 | hi/64] operand (ops.add_split_f16 / layernorm_rows(split=True)) with a '.ws' weight, the three
        attentions read and write f32 rows.  Same structure as the reference (mask_decoder.py:112-149,
        transformer.py:62-106,151-182)."""
        cfg, w, T, dev = self.cfg, self.w, self.T, self.dev
        E, L, g = cfg.prompt_embed_dim, cfg.img_size, cfg.grid
        B = emb.shape[0]
        n = boxes.shape[0]
        assert n > 0 and len(img_of_box) == n
        NT, Hh = 5 + 2, cfg.dec_heads
        SP = ops.add_split_f16

        def lin(x_split, name, bias=True, **kw):
            return ops.gemm(x_split, w[name + ".ws"], w[name + ".b"] if bias else None, **kw)

        coords = _to_dev_async((boxes + 0.5).reshape(-1, 2) / float(L), dev)
        sparse = ops.sam_pe_encode(coords, w["gauss"], add=w["corner"])           # [2n, E]
        tokens = torch.empty((n, NT, E), device=dev, dtype=F32)
        tokens[:, :5] = w["out_tok"]
        tokens[:, 5:] = sparse.view(n, 2, E)
        qpe = tokens.view(n * NT, E)
        iob = torch.as_tensor(list(img_of_box), dtype=torch.int64)
        img_rows = _to_dev_async((iob * T).to(torch.int32), dev)
        keys = ops.add_f32(emb.reshape(B * T, E).contiguous(), w["no_mask"])      # [B*T, E], shared per image
        shared = True                                       # keys still one copy per IMAGE (layer 0)
        ks = None                                           # split-f16 operand of the per-box keys (layers >= 1)
        queries = qpe
        sc32, sc16 = 1.0 / math.sqrt(32), 1.0 / math.sqrt(16)

        Eh = E // 2                                           # internal width of the cross attentions (128)

        def t2i_attend(name, queries, k, v, k_pe, residual):
            q = lin(SP(queries, qpe), name + ".q_proj")
            a = ops.attn_fewq(q, k, v, n_batch=n, n_heads=Hh, head_dim=16, scale=sc16, n_q=NT, n_k=T,
                              kv_batch_rows=img_rows if shared else None, k_add=k_pe)
            return lin(SP(a), name + ".out_proj", residual=residual)

        for i in range(cfg.dec_depth):
            d = f"d{i}"
            # (1) token self-attention (layer 0: no positional add, output replaces the queries)
            if i == 0:
                qkv = ops.gemm(SP(queries), w[d + ".self.qkv.ws"], w[d + ".self.qkv.b"])
                a = ops.attn_fewkeys(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], B=n, n_heads=Hh, head_dim=32,
                                     scale=sc32)
                queries = lin(SP(a), d + ".self.out_proj")
            else:
                qk = ops.gemm(SP(queries, qpe), w[d + ".self.qk.ws"], w[d + ".self.qk.b"])
                v = lin(SP(queries), d + ".self.v_proj")
                a = ops.attn_fewkeys(qk[:, :E], qk[:, E:], v, B=n, n_heads=Hh, head_dim=32, scale=sc32)
                queries = lin(SP(a), d + ".self.out_proj", residual=queries)
            queries = ops.layernorm_rows(queries, w[d + ".norm1.w"], w[d + ".norm1.b"], 1e-5, out_dtype=F32)
            # image-side projections of this layer from ONE split pass of the keys: [k_t2i | v_t2i | q_i2t]
            # (from layer 1 on the split operand was written by the LayerNorm that produced the keys)
            kvq = ops.gemm(SP(keys) if ks is None else ks, w[d + ".kvq.ws"], w[d + ".kvq.b"])   # [B*T or n*T, 384] f32
            # (2) tokens -> image  (k = (keys + pe) Wk = kvq[:, :128] + pe Wk, added inside the attention)
            queries = ops.layernorm_rows(t2i_attend(d + ".t2i", queries, kvq[:, :Eh], kvq[:, Eh:2 * Eh],
                                                    w[d + ".t2i.k_pe"], queries),
                                         w[d + ".norm2.w"], w[d + ".norm2.b"], 1e-5, out_dtype=F32)
            # (3) token MLP
            hmid = lin(SP(queries), d + ".lin1", act="relu")
            queries = ops.layernorm_rows(lin(SP(hmid), d + ".lin2", residual=queries),
                                         w[d + ".norm3.w"], w[d + ".norm3.b"], 1e-5, out_dtype=F32)
            # (4) image -> tokens: q = (keys + pe) Wq = kvq[:, 256:] + pe Wq, k = queries + qpe, v = queries
            ik = lin(SP(queries, qpe), d + ".i2t.k_proj")
            iv = lin(SP(queries), d + ".i2t.v_proj")
            a = ops.attn_fewkeys(kvq[:, 2 * Eh:], ik, iv, B=n, n_heads=Hh, head_dim=16, scale=sc16, n_q=T,
                                 q_batch_rows=img_rows if shared else None, q_add=w[d + ".i2t.q_pe"])   # [n*T, 128]
            # norm4 writes the next consumer's split operand in the same pass (the keys feed only projections from here
            # on); the f32 copy is kept only while a later layer still adds to it
            last = i + 1 == cfg.dec_depth
            if self.fuse_proj_ln and (d + ".i2t.out_proj.blob") in w:
                # out_proj + residual + norm4 (+ the split operand) in one kernel (csrc/proj_ln.hip)
                keys, ks = ops.proj256_ln(a, w[d + ".i2t.out_proj.blob"], w[d + ".i2t.out_proj.b"], keys, w[d + ".norm4.w"],
                                          w[d + ".norm4.b"], 1e-5, res_batch_rows=img_rows if shared else None,
                                          rows_per_batch=T if shared else 0, want_f32=not last, want_split=True)
                shared = False
                continue
            if shared:
                # keys are still one copy per IMAGE: the residual add happens inside the LayerNorm through a per-box
                # row gather - no per-box copy of the image keys (repeat_interleave of mask_decoder.py:124) is made
                y = lin(SP(a), d + ".i2t.out_proj")
                keys_next = None if last else torch.empty_like(y)
                ks = ops.layernorm_rows(y, w[d + ".norm4.w"], w[d + ".norm4.b"], 1e-5, split=True, add=keys,
                                        add_batch_rows=img_rows, rows_per_batch=T, split_f32=keys_next)
                shared = False
            else:
                y = lin(SP(a), d + ".i2t.out_proj", residual=keys)
                keys_next = None if last else y
                ks = ops.layernorm_rows(y, w[d + ".norm4.w"], w[d + ".norm4.b"], 1e-5, split=True, split_f32=keys_next)
            keys = keys_next
        kvu = ops.gemm(ks, w["dfin.kvu.ws"], w["dfin.kvu.b"])                     # [n*T, k 128 | v 128 | up0 4*64]
        kv = kvu[:, :2 * Eh]
        queries = ops.layernorm_rows(t2i_attend("dfin", queries, kv[:, :Eh], kv[:, Eh:], w["dfin.k_pe"], queries),
                                     w["dfin.norm.w"], w["dfin.norm.b"], 1e-5, out_dtype=F32)
        hs = queries.view(n, NT, E)

        def mlp3(prefix: str, x: torch.Tensor) -> torch.Tensor:
            a = lin(SP(x), prefix + "0", act="relu")
            a = lin(SP(a), prefix + "1", act="relu")
            return lin(SP(a), prefix + "2")

        hyper = mlp3("hyp", hs[:, 1].contiguous())          # mask token 0 -> [n, 32]
        iou = mlp3("iou", hs[:, 0].contiguous())[:, :1]     # iou token -> [n, 4] -> mask 0
        u0 = kvu[:, 2 * Eh:]                                                     # [n*T, 4*64], row stride 512
        if "up3.blob" in w and (T * 4) % 32 == 0:
            # LayerNorm2d + GELU + the second transposed convolution + GELU + the hyper-network product in one kernel
            # (csrc/upscale_tail.hip): u0 is read once, 4 floats per row are written
            low = ops.sam_upscale_tail(u0, n, g, w["up1.w"], w["up1.b"], 1e-6, w["up3.blob"], w["up3.b"],
                                       hyper.contiguous())
        else:
            u1 = ops.layernorm_rows(u0.contiguous().view(n * T * 4, E // 4), w["up1.w"], w["up1.b"], 1e-6, act="gelu",
                                    split=True)
            u2 = lin(u1, "up3", act="gelu")                                          # [n*T*4, 4*32]
            low = ops.sam_mask_logits(u2, hyper.contiguous(), n, g)                   # [n, 256, 256]
        return low, iou


# ----------------------------------------------------------------------------------------
# reference-shaped API
# ----------------------------------------------------------------------------------------
class SamPredictor:
    """Same call surface as SA/predictor.py:17-243 for the box-prompt path InkLayer uses."""

    def __init__(self, engine: SamEngine):
        self.engine = engine
        self.cfg = engine.cfg
        self.reset_image()

    def reset_image(self) -> None:
        self.is_image_set = False
        self.features = None
        self.original_size = None
        self.input_size = None

    def set_image(self, image: np.ndarray, image_format: str = "RGB") -> None:
        assert image_format in ("RGB", "BGR")
        if image_format != "RGB":
            image = image[..., ::-1]
        # ResizeLongestSide.apply_image on the GPU (Pillow-exact, ops.resize_bilinear_u8)
        raw = torch.from_numpy(np.ascontiguousarray(image)).to(self.engine.dev)
        nh, nw = preprocess_shape(image.shape[0], image.shape[1], self.cfg.img_size)
        dev_img = ops.resize_bilinear_u8(raw, nh, nw)
        self.original_size = tuple(image.shape[:2])
        self.input_size = (nh, nw)
        self.features = self.engine.encode([dev_img])[0]
        self.is_image_set = True

    def apply_boxes(self, boxes: torch.Tensor) -> torch.Tensor:
        """ResizeLongestSide.apply_boxes_torch (SA/utils/transforms.py:67-91), host side."""
        oh, ow = self.original_size
        nh, nw = preprocess_shape(oh, ow, self.cfg.img_size)
        c = boxes.detach().cpu().reshape(-1, 2, 2).clone().to(torch.float)
        c[..., 0] = c[..., 0] * (nw / ow)
        c[..., 1] = c[..., 1] * (nh / oh)
        return c.reshape(-1, 4)

    def predict_torch(self, point_coords=None, point_labels=None, boxes: torch.Tensor = None,
                      mask_input=None, multimask_output: bool = False, return_logits: bool = False):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        if point_coords is not None or mask_input is not None or multimask_output:
            raise NotImplementedError("InkLayer only uses box prompts with multimask_output=False")
        masks, low, iou, logits = self.engine.decode(self.features, boxes.detach().cpu(), self.input_size,
                                                     self.original_size, want_logits=True)
        n = masks.shape[0]
        out = logits if return_logits else masks.bool()
        return out.view(n, 1, *self.original_size), iou, low.view(n, 1, *low.shape[-2:])


_ENGINES: Dict[str, SamEngine] = {}


def build_sam(checkpoint: Optional[str] = None, state_dict=None, device="cuda",
              max_batch: int = 1) -> SamEngine:
    """SA/build_sam.py:55-107 (ViT-H).  Unlike the reference (which rebuilds the model and reloads
    the 2.4 GB checkpoint on EVERY run_SAM call, InkLayer/segmentor/sam.py:23) the engine is cached
    per checkpoint path and stays resident in HBM."""
    if state_dict is None:
        if checkpoint in _ENGINES:
            return _ENGINES[checkpoint]
        state_dict = torch.load(checkpoint, map_location="cpu", weights_only=True)
    eng = SamEngine(state_dict, SamConfig(), device, max_batch)
    if checkpoint is not None:
        _ENGINES[checkpoint] = eng
    return eng


def run_SAM(image_pil, boxes_filt: torch.Tensor, sam_checkpoint: Optional[str] = None,
            engine: Optional[SamEngine] = None) -> List[np.ndarray]:
    """InkLayer/segmentor/sam.py:16-43.  Returns a list of HxW bool arrays, one per box.
    Reference quirks kept: the RGB array goes through a channel reversal before being declared
    "RGB" (:24-26).  Deviation (documented reference bug): zero boxes return []."""
    if len(boxes_filt) == 0:
        return []
    eng = engine if engine is not None else build_sam(sam_checkpoint)
    pred = SamPredictor(eng)
    image = np.array(image_pil)[..., ::-1]                 # cv2.COLOR_BGR2RGB on an RGB array
    pred.set_image(np.ascontiguousarray(image))
    tb = pred.apply_boxes(boxes_filt)
    masks, _, _ = pred.predict_torch(None, None, boxes=tb, multimask_output=False)
    m = masks[:, 0].cpu().numpy()                          # ONE device->host copy for all boxes
    return [m[i] for i in range(m.shape[0])]
