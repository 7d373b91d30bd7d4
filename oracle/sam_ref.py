"""CPU fp32 restatement of the SAM half of the InkLayer hot path (TEST INFRASTRUCTURE ONLY).

Functional style over a flat ``state_dict`` that uses the reference's own key names
(SA/build_sam.py:66-106), so a real ``sam_vit_h_4b8939.pth`` drops in unchanged.
Pinned against the reference modules by tests/golden/sam_small.npz
(generator: tests/golden/make_sam_golden.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


@dataclass
class SamConfig:
    """Hyper-parameters of SA/build_sam.py:14-21,55-101 (defaults = ViT-H)."""
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
    iou_head_hidden: int = 256
    mask_in_chans: int = 16
    pixel_mean: Tuple[float, ...] = (123.675, 116.28, 103.53)
    pixel_std: Tuple[float, ...] = (58.395, 57.12, 57.375)

    @property
    def grid(self) -> int:
        return self.img_size // self.patch_size


# ----------------------------------------------------------------------------------------
# parameter inventory (names + shapes), used for seeded random weights and for packing
# ----------------------------------------------------------------------------------------
def sam_param_shapes(cfg: SamConfig) -> Dict[str, Tuple[int, ...]]:
    D, P, E = cfg.embed_dim, cfg.patch_size, cfg.prompt_embed_dim
    hd = D // cfg.num_heads
    mlp = int(D * cfg.mlp_ratio)
    s: Dict[str, Tuple[int, ...]] = {}
    s["image_encoder.pos_embed"] = (1, cfg.grid, cfg.grid, D)
    s["image_encoder.patch_embed.proj.weight"] = (D, 3, P, P)
    s["image_encoder.patch_embed.proj.bias"] = (D,)
    for i in range(cfg.depth):
        p = f"image_encoder.blocks.{i}."
        S = cfg.grid if i in cfg.global_attn_indexes else cfg.window_size
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.rel_pos_h"] = (2 * S - 1, hd)
        s[p + "attn.rel_pos_w"] = (2 * S - 1, hd)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.lin1.weight"] = (mlp, D)
        s[p + "mlp.lin1.bias"] = (mlp,)
        s[p + "mlp.lin2.weight"] = (D, mlp)
        s[p + "mlp.lin2.bias"] = (D,)
    s["image_encoder.neck.0.weight"] = (E, D, 1, 1)
    s["image_encoder.neck.1.weight"] = (E,)
    s["image_encoder.neck.1.bias"] = (E,)
    s["image_encoder.neck.2.weight"] = (E, E, 3, 3)
    s["image_encoder.neck.3.weight"] = (E,)
    s["image_encoder.neck.3.bias"] = (E,)
    # prompt encoder (SA/modeling/prompt_encoder.py:16-61)
    s["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"] = (2, E // 2)
    for i in range(4):
        s[f"prompt_encoder.point_embeddings.{i}.weight"] = (1, E)
    s["prompt_encoder.not_a_point_embed.weight"] = (1, E)
    mc = cfg.mask_in_chans
    s["prompt_encoder.mask_downscaling.0.weight"] = (mc // 4, 1, 2, 2)
    s["prompt_encoder.mask_downscaling.0.bias"] = (mc // 4,)
    s["prompt_encoder.mask_downscaling.1.weight"] = (mc // 4,)
    s["prompt_encoder.mask_downscaling.1.bias"] = (mc // 4,)
    s["prompt_encoder.mask_downscaling.3.weight"] = (mc, mc // 4, 2, 2)
    s["prompt_encoder.mask_downscaling.3.bias"] = (mc,)
    s["prompt_encoder.mask_downscaling.4.weight"] = (mc,)
    s["prompt_encoder.mask_downscaling.4.bias"] = (mc,)
    s["prompt_encoder.mask_downscaling.6.weight"] = (E, mc, 1, 1)
    s["prompt_encoder.mask_downscaling.6.bias"] = (E,)
    s["prompt_encoder.no_mask_embed.weight"] = (1, E)
    # mask decoder (SA/modeling/mask_decoder.py:16-69, SA/modeling/transformer.py:16-240)
    def attn(prefix: str, internal: int) -> None:
        for n in ("q_proj", "k_proj", "v_proj"):
            s[f"{prefix}.{n}.weight"] = (internal, E)
            s[f"{prefix}.{n}.bias"] = (internal,)
        s[f"{prefix}.out_proj.weight"] = (E, internal)
        s[f"{prefix}.out_proj.bias"] = (E,)

    t = "mask_decoder.transformer."
    for i in range(cfg.dec_depth):
        p = f"{t}layers.{i}."
        attn(p + "self_attn", E)
        attn(p + "cross_attn_token_to_image", E // 2)
        attn(p + "cross_attn_image_to_token", E // 2)
        for n in ("norm1", "norm2", "norm3", "norm4"):
            s[p + n + ".weight"] = (E,)
            s[p + n + ".bias"] = (E,)
        s[p + "mlp.lin1.weight"] = (cfg.dec_mlp_dim, E)
        s[p + "mlp.lin1.bias"] = (cfg.dec_mlp_dim,)
        s[p + "mlp.lin2.weight"] = (E, cfg.dec_mlp_dim)
        s[p + "mlp.lin2.bias"] = (E,)
    attn(t + "final_attn_token_to_image", E // 2)
    s[t + "norm_final_attn.weight"] = (E,)
    s[t + "norm_final_attn.bias"] = (E,)
    s["mask_decoder.iou_token.weight"] = (1, E)
    s["mask_decoder.mask_tokens.weight"] = (cfg.num_mask_tokens, E)
    s["mask_decoder.output_upscaling.0.weight"] = (E, E // 4, 2, 2)
    s["mask_decoder.output_upscaling.0.bias"] = (E // 4,)
    s["mask_decoder.output_upscaling.1.weight"] = (E // 4,)
    s["mask_decoder.output_upscaling.1.bias"] = (E // 4,)
    s["mask_decoder.output_upscaling.3.weight"] = (E // 4, E // 8, 2, 2)
    s["mask_decoder.output_upscaling.3.bias"] = (E // 8,)
    for i in range(cfg.num_mask_tokens):
        p = f"mask_decoder.output_hypernetworks_mlps.{i}.layers."
        dims = [E, E, E, E // 8]
        for j in range(3):
            s[f"{p}{j}.weight"] = (dims[j + 1], dims[j])
            s[f"{p}{j}.bias"] = (dims[j + 1],)
    dims = [E, cfg.iou_head_hidden, cfg.iou_head_hidden, cfg.num_mask_tokens]
    for j in range(3):
        s[f"mask_decoder.iou_prediction_head.layers.{j}.weight"] = (dims[j + 1], dims[j])
        s[f"mask_decoder.iou_prediction_head.layers.{j}.bias"] = (dims[j + 1],)
    return s


def seeded_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int,
                      gain: float = 1.0) -> SD:
    """Deterministic weights from numpy's frozen MT19937 stream (identical on every machine).

    Every parameter is non-zero — the reference zero-initialises rel_pos_*, pos_embed, … which
    would make parity tests vacuous.  Matrices ~ N(0, gain/fan_in), norm weights
    ~ 1 + 0.1 N, everything else ~ 0.02..0.5 N.
    """
    rs = np.random.RandomState(seed)
    sd: SD = {}
    embed_like = ("point_embeddings", "not_a_point_embed", "no_mask_embed", "iou_token",
                  "mask_tokens", "tgt_embed", "level_embed", "label_enc")
    for name, shape in shapes.items():
        x = rs.standard_normal(shape).astype(np.float32)
        leaf = name.rsplit(".", 1)[-1]
        if any(e in name for e in embed_like):
            x = 0.5 * x
        elif leaf == "weight" and len(shape) == 1:        # norm scales
            x = 1.0 + 0.1 * x
        elif leaf == "bias":
            x = 0.1 * x
        elif leaf == "weight":                            # linear / conv / conv-transpose
            fan_in = shape[0] if "output_upscaling" in name else int(np.prod(shape[1:]))
            x = x * math.sqrt(gain / fan_in)
        elif "rel_pos" in name or "relative_position_bias_table" in name:
            x = 0.2 * x
        elif leaf == "pos_embed":
            x = 0.5 * x
        # everything else (e.g. the gaussian PE matrix) stays N(0, 1)
        sd[name] = torch.from_numpy(np.ascontiguousarray(x))
    return sd


# ----------------------------------------------------------------------------------------
# image encoder  (SA/modeling/image_encoder.py)
# ----------------------------------------------------------------------------------------
def _ln(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _ln2d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """Channel-wise LayerNorm on NCHW (SA/modeling/common.py:31-43)."""
    mu = x.mean(1, keepdim=True)
    var = (x - mu).pow(2).mean(1, keepdim=True)
    return w[None, :, None, None] * ((x - mu) / torch.sqrt(var + eps)) + b[None, :, None, None]


def _rel_table(S: int, rel_pos: torch.Tensor) -> torch.Tensor:
    """get_rel_pos for q_size == k_size == S (SA/modeling/image_encoder.py:292-322): R[q, k] =
    rel_pos[q - k + S - 1].  The interpolation branch is not taken at SAM's sizes."""
    assert rel_pos.shape[0] == 2 * S - 1
    idx = torch.arange(S)[:, None] - torch.arange(S)[None, :] + (S - 1)
    return rel_pos[idx]


def vit_attention(sd: SD, p: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """Attention.forward + add_decomposed_rel_pos (image_encoder.py:224-240, 325-361).
    x: [B, S, S, C] (a batch of windows, or whole images for the global blocks)."""
    B, S, _, C = x.shape
    hd = C // heads
    qkv = F.linear(x.reshape(B, S * S, C), sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    q, k, v = qkv.reshape(B, S * S, 3, heads, hd).permute(2, 0, 3, 1, 4).reshape(3, B * heads, S * S, hd)
    attn = (q * hd ** -0.5) @ k.transpose(1, 2)
    Rh, Rw = _rel_table(S, sd[p + "rel_pos_h"]), _rel_table(S, sd[p + "rel_pos_w"])
    rq = q.reshape(B * heads, S, S, hd)  # NB: un-scaled q, as in the reference
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    attn = (attn.view(-1, S, S, S, S) + rel_h[..., :, None] + rel_w[..., None, :]).view(-1, S * S, S * S)
    o = attn.softmax(-1) @ v
    o = o.view(B, heads, S, S, hd).permute(0, 2, 3, 1, 4).reshape(B, S, S, C)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def vit_block(sd: SD, cfg: SamConfig, i: int, x: torch.Tensor) -> torch.Tensor:
    """Block.forward (image_encoder.py:166-182) incl. window_partition/unpartition (:243-289):
    zero padding happens AFTER norm1 and the padded tokens take part as keys."""
    p = f"image_encoder.blocks.{i}."
    B, H, W, C = x.shape
    y = _ln(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    if i in cfg.global_attn_indexes:
        y = vit_attention(sd, p + "attn.", y, cfg.num_heads)
    else:
        ws = cfg.window_size
        ph, pw = (-H) % ws, (-W) % ws
        y = F.pad(y, (0, 0, 0, pw, 0, ph))
        Hp, Wp = H + ph, W + pw
        y = y.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
        y = vit_attention(sd, p + "attn.", y, cfg.num_heads)
        y = y.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        y = y[:, :H, :W]
    x = x + y
    y = _ln(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    y = F.linear(F.gelu(F.linear(y, sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"])),
                 sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"])
    return x + y


def image_encoder(sd: SD, cfg: SamConfig, x: torch.Tensor, upto: int | None = None,
                  taps: Dict[int, torch.Tensor] | None = None) -> torch.Tensor:
    """ImageEncoderViT.forward (image_encoder.py:106-116): [B,3,L,L] -> [B,E,L/16,L/16].
    `upto` = stop after that many blocks and return the NHWC token map (stage-level checks);
    `taps` = {n_blocks: None} is filled with the NHWC token map after that many blocks (full-depth error growth)."""
    x = F.conv2d(x, sd["image_encoder.patch_embed.proj.weight"],
                 sd["image_encoder.patch_embed.proj.bias"], stride=cfg.patch_size)
    x = x.permute(0, 2, 3, 1) + sd["image_encoder.pos_embed"]
    for i in range(cfg.depth if upto is None else upto):
        x = vit_block(sd, cfg, i, x)
        if taps is not None and (i + 1) in taps:
            taps[i + 1] = x.clone()
    if upto is not None:
        return x
    x = x.permute(0, 3, 1, 2)
    x = F.conv2d(x, sd["image_encoder.neck.0.weight"])
    x = _ln2d(x, sd["image_encoder.neck.1.weight"], sd["image_encoder.neck.1.bias"])
    x = F.conv2d(x, sd["image_encoder.neck.2.weight"], padding=1)
    return _ln2d(x, sd["image_encoder.neck.3.weight"], sd["image_encoder.neck.3.bias"])


# ----------------------------------------------------------------------------------------
# pre/post-processing  (SA/utils/transforms.py, SA/modeling/sam.py, SA/predictor.py)
# ----------------------------------------------------------------------------------------
def preprocess_shape(h: int, w: int, L: int) -> Tuple[int, int]:
    """ResizeLongestSide.get_preprocess_shape (transforms.py:93-102)."""
    sc = L * 1.0 / max(h, w)
    return int(h * sc + 0.5), int(w * sc + 0.5)


def apply_image(img: np.ndarray, L: int) -> np.ndarray:
    """ResizeLongestSide.apply_image (transforms.py:26-31): torchvision resize(to_pil_image(x))
    == PIL bilinear resize (with PIL's built-in antialiasing when shrinking)."""
    from PIL import Image
    nh, nw = preprocess_shape(img.shape[0], img.shape[1], L)
    return np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))


def preprocess(cfg: SamConfig, x: torch.Tensor) -> torch.Tensor:
    """Sam.preprocess (sam.py:164-174): normalise, zero-pad bottom/right to L x L. x: [3,h,w]."""
    mean = torch.tensor(cfg.pixel_mean).view(3, 1, 1)
    std = torch.tensor(cfg.pixel_std).view(3, 1, 1)
    x = (x.float() - mean) / std
    return F.pad(x, (0, cfg.img_size - x.shape[-1], 0, cfg.img_size - x.shape[-2]))


def apply_boxes(boxes: torch.Tensor, orig_hw: Tuple[int, int], L: int) -> torch.Tensor:
    """ResizeLongestSide.apply_boxes_torch (transforms.py:67-91)."""
    oh, ow = orig_hw
    nh, nw = preprocess_shape(oh, ow, L)
    c = boxes.reshape(-1, 2, 2).clone().to(torch.float)
    c[..., 0] = c[..., 0] * (nw / ow)
    c[..., 1] = c[..., 1] * (nh / oh)
    return c.reshape(-1, 4)


def postprocess_masks(cfg: SamConfig, masks: torch.Tensor, input_hw: Tuple[int, int],
                      orig_hw: Tuple[int, int]) -> torch.Tensor:
    """Sam.postprocess_masks (sam.py:133-162)."""
    m = F.interpolate(masks, (cfg.img_size, cfg.img_size), mode="bilinear", align_corners=False)
    m = m[..., : input_hw[0], : input_hw[1]]
    return F.interpolate(m, orig_hw, mode="bilinear", align_corners=False)


# ----------------------------------------------------------------------------------------
# prompt encoder  (SA/modeling/prompt_encoder.py)
# ----------------------------------------------------------------------------------------
def _pe_encoding(sd: SD, coords01: torch.Tensor) -> torch.Tensor:
    """PositionEmbeddingRandom._pe_encoding (prompt_encoder.py:186-193)."""
    c = 2 * coords01 - 1
    c = c @ sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(sd: SD, cfg: SamConfig) -> torch.Tensor:
    """PromptEncoder.get_dense_pe (prompt_encoder.py:62-71, 195-206): [1,E,g,g]."""
    g = cfg.grid
    ones = torch.ones((g, g), dtype=torch.float32)
    y = (ones.cumsum(0) - 0.5) / g
    x = (ones.cumsum(1) - 0.5) / g
    return _pe_encoding(sd, torch.stack([x, y], -1)).permute(2, 0, 1).unsqueeze(0)


def embed_boxes(sd: SD, cfg: SamConfig, boxes: torch.Tensor) -> torch.Tensor:
    """PromptEncoder._embed_boxes (prompt_encoder.py:93-100, 208-214): [n,4] -> [n,2,E]."""
    c = (boxes + 0.5).reshape(-1, 2, 2).clone()
    c[..., 0] = c[..., 0] / cfg.img_size
    c[..., 1] = c[..., 1] / cfg.img_size
    e = _pe_encoding(sd, c.to(torch.float))
    e[:, 0, :] += sd["prompt_encoder.point_embeddings.2.weight"]
    e[:, 1, :] += sd["prompt_encoder.point_embeddings.3.weight"]
    return e


# ----------------------------------------------------------------------------------------
# mask decoder  (SA/modeling/mask_decoder.py, SA/modeling/transformer.py)
# ----------------------------------------------------------------------------------------
def _dec_attn(sd: SD, p: str, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """transformer.py Attention.forward (:218-240)."""
    q = F.linear(q, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"])
    k = F.linear(k, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    v = F.linear(v, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    def split(t):
        b, n, c = t.shape
        return t.reshape(b, n, heads, c // heads).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    a = (q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    o = a.softmax(-1) @ v
    b, h, n, c = o.shape
    o = o.transpose(1, 2).reshape(b, n, h * c)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def two_way_transformer(sd: SD, cfg: SamConfig, src: torch.Tensor, pos: torch.Tensor,
                        tokens: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """TwoWayTransformer.forward + TwoWayAttentionBlock.forward (transformer.py:62-106, 151-182)."""
    t = "mask_decoder.transformer."
    keys = src.flatten(2).permute(0, 2, 1)
    kpe = pos.flatten(2).permute(0, 2, 1)
    queries, qpe, H = tokens, tokens, cfg.dec_heads
    def ln(x, n):
        return _ln(x, sd[n + ".weight"], sd[n + ".bias"], 1e-5)
    for i in range(cfg.dec_depth):
        p = f"{t}layers.{i}."
        if i == 0:
            queries = _dec_attn(sd, p + "self_attn.", queries, queries, queries, H)
        else:
            q = queries + qpe
            queries = queries + _dec_attn(sd, p + "self_attn.", q, q, queries, H)
        queries = ln(queries, p + "norm1")
        queries = ln(queries + _dec_attn(sd, p + "cross_attn_token_to_image.", queries + qpe,
                                         keys + kpe, keys, H), p + "norm2")
        m = F.linear(F.relu(F.linear(queries, sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"])),
                     sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"])
        queries = ln(queries + m, p + "norm3")
        keys = ln(keys + _dec_attn(sd, p + "cross_attn_image_to_token.", keys + kpe, queries + qpe,
                                   queries, H), p + "norm4")
    queries = ln(queries + _dec_attn(sd, t + "final_attn_token_to_image.", queries + qpe, keys + kpe,
                                     keys, H), t + "norm_final_attn")
    return queries, keys


def _mlp3(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """mask_decoder.py MLP (:154-176), 3 layers, ReLU between."""
    for j in range(3):
        x = F.linear(x, sd[f"{p}{j}.weight"], sd[f"{p}{j}.bias"])
        if j < 2:
            x = F.relu(x)
    return x


def mask_decoder(sd: SD, cfg: SamConfig, image_emb: torch.Tensor, image_pe: torch.Tensor,
                 sparse: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """MaskDecoder.forward with multimask_output=False (mask_decoder.py:71-149); the dense prompt
    is the broadcast no_mask_embed (prompt_encoder.py:161-166).  -> ([n,1,4g,4g], [n,1])."""
    n = sparse.shape[0]
    out_tok = torch.cat([sd["mask_decoder.iou_token.weight"], sd["mask_decoder.mask_tokens.weight"]], 0)
    tokens = torch.cat([out_tok.unsqueeze(0).expand(n, -1, -1), sparse], 1)
    dense = sd["prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1)
    src = torch.repeat_interleave(image_emb, n, 0) + dense
    pos = torch.repeat_interleave(image_pe, n, 0)
    b, c, h, w = src.shape
    hs, src2 = two_way_transformer(sd, cfg, src, pos, tokens)
    iou_tok = hs[:, 0]
    mask_toks = hs[:, 1:1 + cfg.num_mask_tokens]
    src2 = src2.transpose(1, 2).reshape(b, c, h, w)
    u = "mask_decoder.output_upscaling."
    x = F.conv_transpose2d(src2, sd[u + "0.weight"], sd[u + "0.bias"], stride=2)
    x = F.gelu(_ln2d(x, sd[u + "1.weight"], sd[u + "1.bias"]))
    x = F.gelu(F.conv_transpose2d(x, sd[u + "3.weight"], sd[u + "3.bias"], stride=2))
    hyper = torch.stack([_mlp3(sd, f"mask_decoder.output_hypernetworks_mlps.{i}.layers.", mask_toks[:, i])
                         for i in range(cfg.num_mask_tokens)], 1)
    bb, cc, hh, ww = x.shape
    masks = (hyper @ x.view(bb, cc, hh * ww)).view(bb, -1, hh, ww)
    iou = _mlp3(sd, "mask_decoder.iou_prediction_head.layers.", iou_tok)
    return masks[:, 0:1], iou[:, 0:1]


# ----------------------------------------------------------------------------------------
# plugin-level entry (InkLayer/segmentor/sam.py:16-43 + SA/predictor.py:34-90,169-243)
# ----------------------------------------------------------------------------------------
@torch.no_grad()
def run_sam(sd: SD, cfg: SamConfig, image_rgb: np.ndarray, boxes_xyxy: torch.Tensor,
            return_logits: bool = False, taps: Dict[int, torch.Tensor] | None = None):
    """image_rgb: HxWx3 uint8 as np.array(PIL RGB); boxes in original pixel coords.
    Reproduces the reference's channel quirk: run_SAM passes the RGB array through
    COLOR_BGR2RGB (a channel reversal) and then declares it "RGB" (sam.py:24-26)."""
    img = np.ascontiguousarray(image_rgb[..., ::-1])
    oh, ow = img.shape[:2]
    rs = apply_image(img, cfg.img_size)
    x = torch.as_tensor(rs).permute(2, 0, 1).contiguous()
    ih, iw = x.shape[-2:]
    emb = image_encoder(sd, cfg, preprocess(cfg, x)[None], taps=taps)
    if taps is not None:
        taps[-1] = emb
    tb = apply_boxes(boxes_xyxy, (oh, ow), cfg.img_size)
    sparse = embed_boxes(sd, cfg, tb)
    low, iou = mask_decoder(sd, cfg, emb, dense_pe(sd, cfg), sparse)
    logits = postprocess_masks(cfg, low, (ih, iw), (oh, ow))
    if return_logits:
        return logits, low, iou
    masks = logits > 0.0
    return [m[0].numpy() for m in masks]
