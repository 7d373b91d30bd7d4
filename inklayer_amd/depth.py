"""Depth-Anything-V2 ViT-B on MI355X: the depth model InkLayer's refinement uses to order masks.

Mirrors InkLayer/refinement/depth_sort.py:35-45 (`depth_model`, `get_depth_map`) and DA/dpt.py DepthAnythingV2
(`forward`, `infer_image`) with the reference's state_dict key names, so depth_anything_v2_vitb.pth drops in.

Same design as the SAM / GroundingDINO engines: tokens and feature maps are [rows, channels] row-major (NHWC), one f32
residual stream, f16 GEMM operands produced by the kernel that precedes the GEMM, everything dense on
`ink_gemm_f16` (LayerScale = its per-column scale, residual in the epilogue), global attention (1370 tokens, 12 heads x
64) on `ink_flash_attn`; convolutions are im2col GEMMs on NHWC maps; ConvTranspose2d (k = stride) is a projection to
(dy, dx, co) followed by a pixel-shuffle copy.  The patch embedding runs on split-f16 operands (its rounding error
stays in the residual stream of all 12 blocks).  No torch compute ops: torch supplies memory, streams, copies.
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
class DepthConfig:
    """depth_model_configs["vitb"] (depth_sort.py:21-22) + DINOv2("vitb") (DA/dinov2.py:397-415)."""
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    patch_size: int = 14
    img_size: int = 518
    mlp_ratio: int = 4
    features: int = 128
    out_channels: Tuple[int, ...] = (96, 192, 384, 768)
    layer_idx: Tuple[int, ...] = (2, 5, 8, 11)
    interpolate_offset: float = 0.1
    input_size: int = 518
    pixel_mean: Tuple[float, ...] = (0.485, 0.456, 0.406)
    pixel_std: Tuple[float, ...] = (0.229, 0.224, 0.225)


def resize_shape(h: int, w: int, target: int = 518, multiple: int = 14) -> Tuple[int, int]:
    """Resize.get_size, keep_aspect_ratio + lower_bound + ensure_multiple_of=14 (DA/util/transform.py:61-108)."""
    sh, sw = target / h, target / w
    if sw > sh:
        sh = sw
    else:
        sw = sh

    def constrain(x):
        y = int(np.round(x / multiple) * multiple)
        if y < target:
            y = int(np.ceil(x / multiple) * multiple)
        return y

    return constrain(sh * h), constrain(sw * w)


class DepthEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: Optional[DepthConfig] = None,
                 device: str | torch.device = "cuda"):
        cfg = cfg or DepthConfig()
        self.cfg, self.dev = cfg, torch.device(device)
        assert self.dev.type == "cuda", "the InkLayer depth model runs on MI355X only"
        D, P = cfg.embed_dim, cfg.patch_size
        assert D // cfg.num_heads == 64, "flash attention instance: head_dim 64"
        sd, dev = state_dict, self.dev
        self.w: Dict[str, torch.Tensor] = {}
        w = self.w

        def m32(name):
            return sd[name].detach().to(dev, torch.float32)

        def h(name, shape=None):
            t = m32(name)
            return (t.reshape(shape) if shape is not None else t).to(F16).contiguous()

        def f(name):
            return ops.own_f32(sd[name], dev)

        # ---- ViT
        self.KP = -(-3 * P * P // 32) * 32                       # 588 -> 608 columns per split segment
        pe = torch.zeros((D, self.KP), device=dev)
        pe[:, :3 * P * P] = m32("pretrained.patch_embed.proj.weight").reshape(D, 3 * P * P)
        w["pe.ws"] = ops.split_weight(pe)
        w["pe.b"] = f("pretrained.patch_embed.proj.bias")
        self.cls = m32("pretrained.cls_token").reshape(1, D)
        self.pos_embed = sd["pretrained.pos_embed"].detach().to(torch.float32).cpu()
        for i in range(cfg.depth):
            p, d = f"pretrained.blocks.{i}.", f"b{i}."
            for n in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "attn.qkv.bias", "attn.proj.bias",
                      "mlp.fc1.bias", "mlp.fc2.bias", "ls1.gamma", "ls2.gamma"):
                w[d + n] = f(p + n)
            for n in ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"):
                w[d + n] = h(p + n)
        w["norm.w"], w["norm.b"] = f("pretrained.norm.weight"), f("pretrained.norm.bias")
        # ---- DPT head
        hd = "depth_head."
        oc, Fe = cfg.out_channels, cfg.features

        def conv3(name):                                         # [co, ci, 3, 3] -> [co, (ky, kx, ci)]
            t = m32(name)
            return t.permute(0, 2, 3, 1).reshape(t.shape[0], -1).to(F16).contiguous()

        for i, c in enumerate(oc):
            w[f"proj{i}.w"], w[f"proj{i}.b"] = h(f"{hd}projects.{i}.weight", (c, D)), f(f"{hd}projects.{i}.bias")
            w[f"rn{i}.w"] = conv3(f"{hd}scratch.layer{i + 1}_rn.weight")
        for i, s_ in ((0, 4), (1, 2)):                           # ConvTranspose2d [ci, co, k, k] -> [(dy, dx, co), ci]
            t = m32(f"{hd}resize_layers.{i}.weight")
            w[f"up{i}.w"] = t.permute(2, 3, 1, 0).reshape(s_ * s_ * t.shape[1], t.shape[0]).to(F16).contiguous()
            w[f"up{i}.b"] = f(f"{hd}resize_layers.{i}.bias").repeat(s_ * s_).contiguous()
        w["down3.w"], w["down3.b"] = conv3(hd + "resize_layers.3.weight"), f(hd + "resize_layers.3.bias")
        for r in (1, 2, 3, 4):
            p = f"{hd}scratch.refinenet{r}."
            w[f"ref{r}.out.w"], w[f"ref{r}.out.b"] = h(p + "out_conv.weight", (Fe, Fe)), f(p + "out_conv.bias")
            for u in (1, 2):
                for c in (1, 2):
                    w[f"ref{r}.u{u}.c{c}.w"] = conv3(f"{p}resConfUnit{u}.conv{c}.weight")
                    w[f"ref{r}.u{u}.c{c}.b"] = f(f"{p}resConfUnit{u}.conv{c}.bias")
        w["oc1.w"], w["oc1.b"] = conv3(hd + "scratch.output_conv1.weight"), f(hd + "scratch.output_conv1.bias")
        w["oc2.w"], w["oc2.b"] = conv3(hd + "scratch.output_conv2.0.weight"), f(hd + "scratch.output_conv2.0.bias")
        last = torch.zeros((4, 32), device=dev)                  # 1x1 conv to ONE channel, N padded to 4 for the GEMM
        last[0] = m32(hd + "scratch.output_conv2.2.weight").reshape(32)
        w["oc3.w"] = last.to(F16).contiguous()
        b3 = torch.zeros(4, device=dev)
        b3[0] = m32(hd + "scratch.output_conv2.2.bias").reshape(())
        w["oc3.b"] = b3
        self._pos = LRU(8)                                       # position embeddings per (patch rows, patch cols)

    # ------------------------------------------------------------------ constants per input size
    def _pos_tokens(self, ph: int, pw: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """interpolate_pos_encoding (DA/dinov2.py:179-209) folded per input size: (cls + pos[0] [1, D], pos[1:] [T, D]).
        The reference names the image HEIGHT `w` and the WIDTH `h`; the bicubic interpolation runs once per size on the
        host (a load-time constant, like the sine embeddings of the detector)."""
        def make():
            cfg = self.cfg
            pos = self.pos_embed
            N = pos.shape[1] - 1
            if ph * pw == N and ph == pw:
                patch = pos[0, 1:]
            else:
                sq = math.sqrt(N)
                w0, h0 = ph + cfg.interpolate_offset, pw + cfg.interpolate_offset
                pp = torch.nn.functional.interpolate(
                    pos[:, 1:].reshape(1, int(sq), int(sq), -1).permute(0, 3, 1, 2),
                    scale_factor=(float(w0) / sq, float(h0) / sq), mode="bicubic", antialias=False)
                assert (pp.shape[-2], pp.shape[-1]) == (ph, pw)
                patch = pp.permute(0, 2, 3, 1).reshape(ph * pw, -1)
            cls = (self.cls + pos[0, :1].to(self.dev)).contiguous()
            return cls, patch.to(self.dev).contiguous()
        return self._pos.get_or_make((ph, pw), make)

    # ------------------------------------------------------------------ ViT-B
    def encode(self, patches: Sequence[torch.Tensor], ph: int, pw: int) -> List[torch.Tensor]:
        """patches: per image the split-f16 im2col rows [T, 3*KP].  -> the 4 normalised patch-token maps
        [B*T, D] f16 (get_intermediate_layers(norm=True), DA/dinov2.py:293-321; the class tokens are unused since
        use_clstoken=False, DA/dpt.py:122-128)."""
        cfg, w = self.cfg, self.w
        B, T, D, H = len(patches), ph * pw, cfg.embed_dim, cfg.num_heads
        N = T + 1
        cls, pos = self._pos_tokens(ph, pw)
        x = torch.empty((B * N, D), device=self.dev, dtype=F32)
        for b, pt in enumerate(patches):
            x[b * N:b * N + 1] = cls                                           # plumbing copy (1 row)
            ops.gemm(pt, w["pe.ws"], w["pe.b"], residual=pos, out=x[b * N + 1:(b + 1) * N])
        outs = []
        scale = 64 ** -0.5
        for i in range(cfg.depth):
            k = f"b{i}."
            y = ops.layernorm_rows(x, w[k + "norm1.weight"], w[k + "norm1.bias"], 1e-6)
            qkv = ops.gemm(y, w[k + "attn.qkv.weight"], w[k + "attn.qkv.bias"], out_dtype=F16)
            o = ops.flash_attn(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], n_batch=B, n_heads=H, head_dim=64,
                               scale=scale, n_q=N, n_k=N)
            # x = x + ls1 * proj(o): LayerScale is the GEMM's per-column scale (needs the late residual form)
            ops.gemm(o, w[k + "attn.proj.weight"], w[k + "attn.proj.bias"], col_scale=w[k + "ls1.gamma"], residual=x, out=x)
            y = ops.layernorm_rows(x, w[k + "norm2.weight"], w[k + "norm2.bias"], 1e-6)
            hmid = ops.gemm(y, w[k + "mlp.fc1.weight"], w[k + "mlp.fc1.bias"], act="gelu", out_dtype=F16)
            ops.gemm(hmid, w[k + "mlp.fc2.weight"], w[k + "mlp.fc2.bias"], col_scale=w[k + "ls2.gamma"], residual=x, out=x)
            if i in cfg.layer_idx:
                n16 = ops.layernorm_rows(x, w["norm.w"], w["norm.b"], 1e-6)   # [B*N, D] f16
                outs.append(n16.view(B, N, D)[:, 1:].reshape(B * T, D) if B > 1 else n16[1:])
        return outs

    # ------------------------------------------------------------------ DPT head (one image)
    def _conv3(self, x16: torch.Tensor, hh: int, ww: int, wname: str, bias: Optional[str] = None, *, relu_in=False,
               stride=1, **kw) -> torch.Tensor:
        col = ops.im2col3x3_ex(x16, 1, hh, ww, stride=stride, relu=relu_in)
        return ops.gemm(col, self.w[wname], self.w[bias] if bias else None, **kw)

    def _rcu(self, x: torch.Tensor, hh: int, ww: int, p: str) -> torch.Tensor:
        """ResidualConvUnit (DA/util/blocks.py:59-84): conv2(relu(conv1(relu(x)))) + x, f32 in / f32 out."""
        a = self._conv3(ops.add_cvt_f16(x), hh, ww, p + ".c1.w", p + ".c1.b", relu_in=True, act="relu", out_dtype=F16)
        return self._conv3(a, hh, ww, p + ".c2.w", p + ".c2.b", residual=x)

    def _fusion(self, r: int, x0: torch.Tensor, x1: Optional[torch.Tensor], hw: Tuple[int, int],
                size: Optional[Tuple[int, int]]) -> Tuple[torch.Tensor, Tuple[int, int]]:
        """FeatureFusionBlock (DA/util/blocks.py:123-148), align_corners=True."""
        hh, ww = hw
        out = x0
        if x1 is not None:
            out = ops.add_f32(out, self._rcu(x1, hh, ww, f"ref{r}.u1"))
        out = self._rcu(out, hh, ww, f"ref{r}.u2")
        H2, W2 = size if size is not None else (2 * hh, 2 * ww)
        up16 = ops.resize_bilinear_ac(out, 1, hh, ww, H2, W2, out_dtype=F16)
        return ops.gemm(up16, self.w[f"ref{r}.out.w"], self.w[f"ref{r}.out.b"]), (H2, W2)

    def head(self, feats16: Sequence[torch.Tensor], ph: int, pw: int, stages: Optional[dict] = None) -> torch.Tensor:
        """DPTHead.forward (DA/dpt.py:118-150) for ONE image: 4 x [T, D] f16 -> depth [14*ph, 14*pw] f32."""
        w, cfg = self.w, self.cfg
        oc = cfg.out_channels

        def shuffle(t: torch.Tensor, s: int, c: int) -> torch.Tensor:     # [(ph, pw), (dy, dx, c)] -> NHWC upscaled
            return t.view(ph, pw, s, s, c).permute(0, 2, 1, 3, 4).reshape(ph * s * pw * s, c).contiguous()

        l0 = shuffle(ops.gemm(ops.gemm(feats16[0], w["proj0.w"], w["proj0.b"], out_dtype=F16), w["up0.w"], w["up0.b"],
                              out_dtype=F16), 4, oc[0])
        l1 = shuffle(ops.gemm(ops.gemm(feats16[1], w["proj1.w"], w["proj1.b"], out_dtype=F16), w["up1.w"], w["up1.b"],
                              out_dtype=F16), 2, oc[1])
        l2 = ops.gemm(feats16[2], w["proj2.w"], w["proj2.b"], out_dtype=F16)
        l3 = self._conv3(ops.gemm(feats16[3], w["proj3.w"], w["proj3.b"], out_dtype=F16), ph, pw, "down3.w", "down3.b",
                         stride=2, out_dtype=F16)
        s0, s1, s2 = (4 * ph, 4 * pw), (2 * ph, 2 * pw), (ph, pw)
        s3 = ((ph - 1) // 2 + 1, (pw - 1) // 2 + 1)
        rn = [self._conv3(l, hw[0], hw[1], f"rn{i}.w") for i, (l, hw) in enumerate(((l0, s0), (l1, s1), (l2, s2), (l3, s3)))]
        p4, hw4 = self._fusion(4, rn[3], None, s3, s2)
        p3, hw3 = self._fusion(3, p4, rn[2], hw4, s1)
        p2, hw2 = self._fusion(2, p3, rn[1], hw3, s0)
        p1, hw1 = self._fusion(1, p2, rn[0], hw2, None)
        if stages is not None:
            stages.update(rn=rn, path=[p1, p2, p3, p4])
        o = self._conv3(ops.add_cvt_f16(p1), hw1[0], hw1[1], "oc1.w", "oc1.b")                 # [4ph*2 x .., 64] f32
        H, W = 14 * ph, 14 * pw
        o16 = ops.resize_bilinear_ac(o, 1, hw1[0], hw1[1], H, W, out_dtype=F16)
        o = self._conv3(o16, H, W, "oc2.w", "oc2.b", act="relu", out_dtype=F16)                # [H*W, 32]
        d = ops.gemm(o, w["oc3.w"], w["oc3.b"], act="relu")                                    # [H*W, 4], column 0
        return d[:, 0].reshape(H, W)

    # ------------------------------------------------------------------ reference-shaped API
    @torch.no_grad()
    def infer_image(self, raw_bgr: np.ndarray | torch.Tensor, stages: Optional[dict] = None) -> torch.Tensor:
        """DepthAnythingV2.infer_image (DA/dpt.py:189-197) for a BGR uint8 image as cv2.imread returns it
        (host array or HWC uint8 CUDA tensor).  -> depth [h, w] f32 on the GPU."""
        cfg = self.cfg
        img = raw_bgr if isinstance(raw_bgr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(raw_bgr)).to(self.dev)
        h, w_ = int(img.shape[0]), int(img.shape[1])
        nh, nw = resize_shape(h, w_, cfg.input_size, cfg.patch_size)
        ph, pw = nh // cfg.patch_size, nw // cfg.patch_size
        patches = ops.depth_patchify(img, nh, nw, cfg.patch_size, self.KP, cfg.pixel_mean, cfg.pixel_std, chan_reverse=True)
        feats = self.encode([patches], ph, pw)
        if stages is not None:
            stages["feats"] = feats
        d = self.head(feats, ph, pw, stages)
        if stages is not None:
            stages["depth_net"] = d
        return ops.resize_bilinear_ac(d.reshape(nh * nw, 1).contiguous(), 1, nh, nw, h, w_).reshape(h, w_)


_ENGINES: Dict[str, DepthEngine] = {}


def build_depth(checkpoint: str, device="cuda") -> DepthEngine:
    """depth_sort.py:35-40: the model is a module-level singleton there; cached per checkpoint path here."""
    if checkpoint not in _ENGINES:
        _ENGINES[checkpoint] = DepthEngine(torch.load(checkpoint, map_location="cpu", weights_only=True), DepthConfig(), device)
    return _ENGINES[checkpoint]
