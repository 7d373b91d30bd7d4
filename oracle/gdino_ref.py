"""CPU fp32 restatement of the GroundingDINO half of the InkLayer hot path (TEST INFRASTRUCTURE ONLY).

Functional style over a flat state_dict with the reference's key names (GD/models/GroundingDINO/
groundingdino.py module tree), so `inklayer_gdino.pth` drops in.  Pinned against the reference's
own modules by tests/golden/gdino_small.npz (generator: tests/golden/make_gdino_golden.py).

Scope notes
  * batches are equal-sized images (NestedTensor masks all False, valid_ratios == 1), which is
    what InkLayer feeds (one image per call, GD/util/inference.py:67);
  * the text branch (tokenizer + BERT + feat_map, groundingdino.py:248-297) is image-independent
    for the hard-coded caption "object." and enters as the tensor `encoded_text` [n_text, 256];
    the BERT encoder itself is PARITY UNPINNED (no bert-base-uncased files offline).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


@dataclass
class GDinoConfig:
    """models/GroundingDINO_SwinT_OGC.py:1-43 + swin_T_224_1k (swin_transformer.py:771-773)."""
    embed_dim: int = 96
    depths: Tuple[int, ...] = (2, 2, 6, 2)
    num_heads: Tuple[int, ...] = (3, 6, 12, 24)
    window_size: int = 7
    out_indices: Tuple[int, ...] = (1, 2, 3)
    hidden_dim: int = 256
    nheads: int = 8
    enc_layers: int = 6
    dec_layers: int = 6
    dim_feedforward: int = 2048
    num_queries: int = 900
    num_feature_levels: int = 4
    n_points: int = 4
    max_text_len: int = 256
    pe_temperature: float = 20.0
    box_threshold: float = 0.2

    @property
    def fusion_dim(self) -> int:  # BiAttentionBlock(embed_dim=dim_feedforward // 2), transformer.py:97
        return self.dim_feedforward // 2


# ----------------------------------------------------------------------------------------
# parameter inventory
# ----------------------------------------------------------------------------------------
def gdino_param_shapes(cfg: GDinoConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    C0, ws, D, Fd = cfg.embed_dim, cfg.window_size, cfg.hidden_dim, cfg.dim_feedforward
    s["backbone.0.patch_embed.proj.weight"] = (C0, 3, 4, 4)
    s["backbone.0.patch_embed.proj.bias"] = (C0,)
    s["backbone.0.patch_embed.norm.weight"] = (C0,)
    s["backbone.0.patch_embed.norm.bias"] = (C0,)
    for i, (dep, nh) in enumerate(zip(cfg.depths, cfg.num_heads)):
        C = C0 * 2 ** i
        for b in range(dep):
            p = f"backbone.0.layers.{i}.blocks.{b}."
            s[p + "norm1.weight"] = (C,)
            s[p + "norm1.bias"] = (C,)
            s[p + "attn.relative_position_bias_table"] = ((2 * ws - 1) ** 2, nh)
            s[p + "attn.qkv.weight"] = (3 * C, C)
            s[p + "attn.qkv.bias"] = (3 * C,)
            s[p + "attn.proj.weight"] = (C, C)
            s[p + "attn.proj.bias"] = (C,)
            s[p + "norm2.weight"] = (C,)
            s[p + "norm2.bias"] = (C,)
            s[p + "mlp.fc1.weight"] = (4 * C, C)
            s[p + "mlp.fc1.bias"] = (4 * C,)
            s[p + "mlp.fc2.weight"] = (C, 4 * C)
            s[p + "mlp.fc2.bias"] = (C,)
        if i < len(cfg.depths) - 1:
            p = f"backbone.0.layers.{i}.downsample."
            s[p + "reduction.weight"] = (2 * C, 4 * C)
            s[p + "norm.weight"] = (4 * C,)
            s[p + "norm.bias"] = (4 * C,)
        if i in cfg.out_indices:
            s[f"backbone.0.norm{i}.weight"] = (C,)
            s[f"backbone.0.norm{i}.bias"] = (C,)
    chans = [C0 * 2 ** i for i in cfg.out_indices]
    for l in range(cfg.num_feature_levels):
        if l < len(chans):
            s[f"input_proj.{l}.0.weight"] = (D, chans[l], 1, 1)
        else:
            s[f"input_proj.{l}.0.weight"] = (D, chans[-1] if l == len(chans) else D, 3, 3)
        s[f"input_proj.{l}.0.bias"] = (D,)
        s[f"input_proj.{l}.1.weight"] = (D,)
        s[f"input_proj.{l}.1.bias"] = (D,)
    t = "transformer."
    s[t + "level_embed"] = (cfg.num_feature_levels, D)
    M, L, P = cfg.nheads, cfg.num_feature_levels, cfg.n_points

    def msda(p: str) -> None:
        s[p + "sampling_offsets.weight"] = (M * L * P * 2, D)
        s[p + "sampling_offsets.bias"] = (M * L * P * 2,)
        s[p + "attention_weights.weight"] = (M * L * P, D)
        s[p + "attention_weights.bias"] = (M * L * P,)
        for n in ("value_proj", "output_proj"):
            s[p + n + ".weight"] = (D, D)
            s[p + n + ".bias"] = (D,)

    def mha(p: str) -> None:
        s[p + "in_proj_weight"] = (3 * D, D)
        s[p + "in_proj_bias"] = (3 * D,)
        s[p + "out_proj.weight"] = (D, D)
        s[p + "out_proj.bias"] = (D,)

    def ln(p: str, n: int = D) -> None:
        s[p + ".weight"] = (n,)
        s[p + ".bias"] = (n,)

    def lin(p: str, o: int, i: int) -> None:
        s[p + ".weight"] = (o, i)
        s[p + ".bias"] = (o,)

    for i in range(cfg.enc_layers):
        p = f"{t}encoder.layers.{i}."
        msda(p + "self_attn.")
        ln(p + "norm1"); lin(p + "linear1", Fd, D); lin(p + "linear2", D, Fd); ln(p + "norm2")
        p = f"{t}encoder.text_layers.{i}."
        mha(p + "self_attn.")
        lin(p + "linear1", Fd // 2, D); lin(p + "linear2", D, Fd // 2); ln(p + "norm1"); ln(p + "norm2")
        p = f"{t}encoder.fusion_layers.{i}."
        ln(p + "layer_norm_v"); ln(p + "layer_norm_l")
        E = cfg.fusion_dim
        for n in ("v_proj", "l_proj", "values_v_proj", "values_l_proj"):
            lin(p + "attn." + n, E, D)
        lin(p + "attn.out_v_proj", D, E); lin(p + "attn.out_l_proj", D, E)
        s[p + "gamma_v"] = (D,)
        s[p + "gamma_l"] = (D,)
    for i in range(cfg.dec_layers):
        p = f"{t}decoder.layers.{i}."
        msda(p + "cross_attn.")
        ln(p + "norm1"); mha(p + "ca_text."); ln(p + "catext_norm"); mha(p + "self_attn."); ln(p + "norm2")
        lin(p + "linear1", Fd, D); lin(p + "linear2", D, Fd); ln(p + "norm3")
    ln(t + "decoder.norm")
    lin(t + "decoder.ref_point_head.layers.0", D, 2 * D)
    lin(t + "decoder.ref_point_head.layers.1", D, D)
    s[t + "tgt_embed.weight"] = (cfg.num_queries, D)
    lin(t + "enc_output", D, D); ln(t + "enc_output_norm")
    for j, (o, i_) in enumerate(((D, D), (D, D), (4, D))):
        lin(f"{t}enc_out_bbox_embed.layers.{j}", o, i_)
        lin(f"bbox_embed.0.layers.{j}", o, i_)   # shared by all decoder layers (groundingdino.py:165-171)
    return s


# ----------------------------------------------------------------------------------------
# pre-processing  (GD/util/inference.py:39-50, GD/datasets/transforms.py:87-117,264-293)
# ----------------------------------------------------------------------------------------
def resize_shape(w: int, h: int, size: int = 800, max_size: int = 1333) -> Tuple[int, int]:
    """get_size_with_aspect_ratio (transforms.py:90-108) -> (oh, ow)."""
    mn, mx = float(min(w, h)), float(max(w, h))
    if mx / mn * size > max_size:
        size = int(round(max_size * mn / mx))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def load_image(image_rgb: np.ndarray) -> torch.Tensor:
    """RandomResize([800], 1333) + ToTensor + Normalize (inference.py:40-49): HWC u8 -> 3xhxw f32."""
    from PIL import Image
    im = Image.fromarray(image_rgb)
    oh, ow = resize_shape(im.size[0], im.size[1])
    if (ow, oh) != im.size:
        im = im.resize((ow, oh), Image.BILINEAR)
    x = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float() / 255.0
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    return (x - mean) / std


# ----------------------------------------------------------------------------------------
# text-side helpers (bertwarper.py:224-273, utils.py:24-53)
# ----------------------------------------------------------------------------------------
def text_masks_and_position_ids(input_ids: Sequence[int], special: Sequence[int] = (101, 102, 1012, 1029)):
    """generate_masks_with_special_tokens_and_transfer_map for one sentence:
    block-diagonal self-attention mask between special tokens and per-block position ids."""
    n = len(input_ids)
    attn = torch.eye(n, dtype=torch.bool)
    pos = torch.zeros(n, dtype=torch.long)
    prev = 0
    for col, tok in enumerate(input_ids):
        if tok not in special:
            continue
        if col == 0 or col == n - 1:
            attn[col, col] = True
            pos[col] = 0
        else:
            attn[prev + 1: col + 1, prev + 1: col + 1] = True
            pos[prev + 1: col + 1] = torch.arange(0, col - prev)
        prev = col
    return attn, pos


def sine_pos_embed_1d(x: torch.Tensor, num_pos_feats: int = 256, temperature: float = 10000.0) -> torch.Tensor:
    """get_sine_pos_embed for a [..., 1] tensor (utils.py:24-53, exchange_xy irrelevant for n=1)."""
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    v = x[..., None].float() * (2 * math.pi) / dim_t
    return torch.stack((v[..., 0::2].sin(), v[..., 1::2].cos()), dim=-1).flatten(-2)


# ----------------------------------------------------------------------------------------
# Swin-T backbone  (GD/models/GroundingDINO/backbone/swin_transformer.py)
# ----------------------------------------------------------------------------------------
def _ln(x, sd, name, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def swin_rel_index(ws: int) -> torch.Tensor:
    """relative_position_index (swin_transformer.py:111-121)."""
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def swin_shift_mask(Hp: int, Wp: int, ws: int) -> torch.Tensor:
    """SW-MSA mask (0 / -100) of BasicLayer.forward (swin_transformer.py:417-441): [nW, ws*ws, ws*ws]."""
    sh = ws // 2
    img = torch.zeros((Hp, Wp))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
        for wsl in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = mw[:, None, :] - mw[:, :, None]
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))


def swin_block(sd: SD, p: str, x: torch.Tensor, H: int, W: int, nh: int, ws: int, shift: int,
               mask: torch.Tensor) -> torch.Tensor:
    """SwinTransformerBlock.forward + WindowAttention.forward (swin_transformer.py:134-174, 238-298)."""
    B, L, C = x.shape
    y = _ln(x, sd, p + "norm1").view(B, H, W, C)
    pr, pb = (-W) % ws, (-H) % ws
    y = F.pad(y, (0, 0, 0, pr, 0, pb))
    Hp, Wp = H + pb, W + pr
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    yw = y.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    Bw, N = yw.shape[0], ws * ws
    hd = C // nh
    qkv = F.linear(yw, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(Bw, N, 3, nh, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = sd[p + "attn.relative_position_bias_table"][swin_rel_index(ws).view(-1)].view(N, N, nh).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift > 0:
        nW = mask.shape[0]
        attn = (attn.view(Bw // nW, nW, nh, N, N) + mask[None, :, None]).view(-1, nh, N, N)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(Bw, N, C)
    o = F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    o = o.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    x = x + o[:, :H, :W].reshape(B, H * W, C)
    y = _ln(x, sd, p + "norm2")
    y = F.linear(F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])),
                 sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + y


def patch_merging(sd: SD, p: str, x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """PatchMerging.forward (swin_transformer.py:314-340)."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.view(B, -1, 4 * C)
    return F.linear(_ln(x, sd, p + "norm"), sd[p + "reduction.weight"])


def swin_forward(sd: SD, cfg: GDinoConfig, img: torch.Tensor) -> List[torch.Tensor]:
    """SwinTransformer.forward (swin_transformer.py:712-754) -> NCHW maps of out_indices."""
    pfx = "backbone.0."
    _, _, H0, W0 = img.shape
    img = F.pad(img, (0, (-W0) % 4, 0, (-H0) % 4))
    x = F.conv2d(img, sd[pfx + "patch_embed.proj.weight"], sd[pfx + "patch_embed.proj.bias"], stride=4)
    B, C, H, W = x.shape
    x = _ln(x.flatten(2).transpose(1, 2), sd, pfx + "patch_embed.norm")
    outs = []
    ws = cfg.window_size
    for i, (dep, nh) in enumerate(zip(cfg.depths, cfg.num_heads)):
        Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
        mask = swin_shift_mask(Hp, Wp, ws)
        for b in range(dep):
            x = swin_block(sd, f"{pfx}layers.{i}.blocks.{b}.", x, H, W, nh, ws, 0 if b % 2 == 0 else ws // 2, mask)
        if i in cfg.out_indices:
            o = _ln(x, sd, f"{pfx}norm{i}")
            outs.append(o.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous())
        if i < len(cfg.depths) - 1:
            x = patch_merging(sd, f"{pfx}layers.{i}.downsample.", x, H, W)
            H, W = (H + 1) // 2, (W + 1) // 2
    return outs


def pos_sine_hw(cfg: GDinoConfig, B: int, h: int, w: int) -> torch.Tensor:
    """PositionEmbeddingSineHW.forward for an all-False mask (position_encoding.py:98-131): [B,256,h,w]."""
    npf = cfg.hidden_dim // 2
    y = torch.arange(1, h + 1, dtype=torch.float32)[:, None].expand(h, w)
    x = torch.arange(1, w + 1, dtype=torch.float32)[None, :].expand(h, w)
    y = y / (float(h) + 1e-6) * (2 * math.pi)
    x = x / (float(w) + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = cfg.pe_temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).permute(2, 0, 1)[None].expand(B, -1, -1, -1)


# ----------------------------------------------------------------------------------------
# multi-scale deformable attention  (ms_deform_attn.py, csrc/MsDeformAttn/ms_deform_im2col_cuda.cuh)
# ----------------------------------------------------------------------------------------
def msda_core(value: torch.Tensor, shapes: Sequence[Tuple[int, int]], loc: torch.Tensor,
              w: torch.Tensor) -> torch.Tensor:
    """ms_deformable_im2col (ms_deform_im2col_cuda.cuh:33-84, 237-299), restated with explicit
    gathers: value [B,S,M,C], loc [B,Q,M,L,P,2] in [0,1] (x,y), w [B,Q,M,L,P] -> [B,Q,M*C].
    h_im = y*H - 0.5, w_im = x*W - 0.5; bilinear; samples outside (-1, H) x (-1, W) contribute 0
    and out-of-range corners contribute 0 (== grid_sample(zeros, align_corners=False))."""
    B, S, M, C = value.shape
    Q, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    out = torch.zeros(B, Q, M, C, dtype=value.dtype)
    start = 0
    for l, (H, W) in enumerate(shapes):
        v = value[:, start:start + H * W].permute(0, 2, 1, 3)          # B,M,HW,C
        x = loc[:, :, :, l, :, 0] * W - 0.5                            # B,Q,M,P
        y = loc[:, :, :, l, :, 1] * H - 0.5
        x0, y0 = torch.floor(x), torch.floor(y)
        lx, ly = x - x0, y - y0
        inside = (y > -1) & (x > -1) & (y < H) & (x < W)
        acc = torch.zeros(B, Q, M, P, C, dtype=value.dtype)
        for dy, dx, wt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx),
                           (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
            yy, xx = (y0 + dy).long(), (x0 + dx).long()
            ok = inside & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).permute(0, 2, 1, 3).reshape(B, M, Q * P)
            g = torch.gather(v, 2, idx[..., None].expand(-1, -1, -1, C)).view(B, M, Q, P, C).permute(0, 2, 1, 3, 4)
            acc = acc + g * (wt * ok)[..., None]
        out = out + (acc * w[:, :, :, l, :, None]).sum(3)
        start += H * W
    return out.reshape(B, Q, M * C)


def msda_module(sd: SD, p: str, cfg: GDinoConfig, query: torch.Tensor, ref: torch.Tensor,
                value_in: torch.Tensor, shapes: Sequence[Tuple[int, int]]) -> torch.Tensor:
    """MultiScaleDeformableAttention.forward, batch_first, no padding mask (ms_deform_attn.py:232-359).
    query [B,Q,D], ref [B,Q,L,2|4], value_in [B,S,D]."""
    B, Q, D = query.shape
    M, L, P = cfg.nheads, cfg.num_feature_levels, cfg.n_points
    value = F.linear(value_in, sd[p + "value_proj.weight"], sd[p + "value_proj.bias"]).view(B, -1, M, D // M)
    off = F.linear(query, sd[p + "sampling_offsets.weight"], sd[p + "sampling_offsets.bias"]).view(B, Q, M, L, P, 2)
    aw = F.linear(query, sd[p + "attention_weights.weight"], sd[p + "attention_weights.bias"]).view(B, Q, M, L * P)
    aw = aw.softmax(-1).view(B, Q, M, L, P)
    if ref.shape[-1] == 2:
        norm = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float32)
        loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    o = msda_core(value, shapes, loc, aw)
    return F.linear(o, sd[p + "output_proj.weight"], sd[p + "output_proj.bias"])


# ----------------------------------------------------------------------------------------
# encoder pieces  (fuse_modules.py:146-295, transformer_vanilla.py:101-123, transformer.py:780-799)
# ----------------------------------------------------------------------------------------
def mha(sd: SD, p: str, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, nhead: int,
        attn_mask: torch.Tensor | None = None) -> torch.Tensor:
    """nn.MultiheadAttention forward (batch-first here): q [B,Nq,D], k/v [B,Nk,D];
    attn_mask bool [Nq,Nk] or [B,Nq,Nk] with True = NOT allowed."""
    D = q.shape[-1]
    W, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    qp = F.linear(q, W[:D], b[:D])
    kp = F.linear(k, W[D:2 * D], b[D:2 * D])
    vp = F.linear(v, W[2 * D:], b[2 * D:])
    B, Nq, _ = qp.shape
    hd = D // nhead
    sp = lambda t: t.view(B, -1, nhead, hd).transpose(1, 2)
    a = (sp(qp) * hd ** -0.5) @ sp(kp).transpose(-1, -2)
    if attn_mask is not None:
        m = attn_mask if attn_mask.dim() == 3 else attn_mask[None]
        a = a.masked_fill(m[:, None], float("-inf"))
    o = (a.softmax(-1) @ sp(vp)).transpose(1, 2).reshape(B, Nq, D)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def fusion_layer(sd: SD, p: str, v: torch.Tensor, l: torch.Tensor, heads: int = 4):
    """BiAttentionBlock.forward + BiMultiHeadAttention.forward, no padding (fuse_modules.py:146-295).
    NB the residual is taken from the NORMALISED v / l."""
    v = _ln(v, sd, p + "layer_norm_v")
    l = _ln(l, sd, p + "layer_norm_l")
    a = p + "attn."
    B, Nv, _ = v.shape
    E = sd[a + "v_proj.weight"].shape[0]
    hd = E // heads
    lin = lambda x, n: F.linear(x, sd[a + n + ".weight"], sd[a + n + ".bias"])
    sp = lambda t: t.view(B, -1, heads, hd).transpose(1, 2).reshape(B * heads, -1, hd)
    q = sp(lin(v, "v_proj") * hd ** -0.5)
    k = sp(lin(l, "l_proj"))
    vv, vl = sp(lin(v, "values_v_proj")), sp(lin(l, "values_l_proj"))
    aw = torch.bmm(q, k.transpose(1, 2))
    aw = aw - aw.max()                                      # stable_softmax_2d: GLOBAL max (:181-182)
    aw = torch.clamp(aw, min=-50000, max=50000)
    awT = aw.transpose(1, 2)
    awl = torch.clamp(awT - awT.max(dim=-1, keepdim=True)[0], min=-50000, max=50000).softmax(-1)
    awv = aw.softmax(-1)
    ov = torch.bmm(awv, vl).view(B, heads, Nv, hd).transpose(1, 2).reshape(B, Nv, E)
    ol = torch.bmm(awl, vv).view(B, heads, -1, hd).transpose(1, 2).reshape(B, -1, E)
    v = v + sd[p + "gamma_v"] * lin(ov, "out_v_proj")
    l = l + sd[p + "gamma_l"] * lin(ol, "out_l_proj")
    return v, l


def text_layer(sd: SD, p: str, src: torch.Tensor, pos: torch.Tensor, self_mask: torch.Tensor, nhead: int = 4):
    """transformer_vanilla.TransformerEncoderLayer.forward (post-norm)."""
    q = src + pos
    src = _ln(src + mha(sd, p + "self_attn.", q, q, src, nhead, attn_mask=~self_mask), sd, p + "norm1")
    f = F.linear(F.relu(F.linear(src, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                 sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return _ln(src + f, sd, p + "norm2")


def enc_reference_points(shapes: Sequence[Tuple[int, int]]) -> torch.Tensor:
    """TransformerEncoder.get_reference_points with valid_ratios == 1 (transformer.py:465-480): [S, 2]."""
    refs = []
    for H, W in shapes:
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
        refs.append(torch.stack((rx.reshape(-1) / W, ry.reshape(-1) / H), -1))
    return torch.cat(refs, 0)


def deform_enc_layer(sd: SD, p: str, cfg: GDinoConfig, src, pos, ref, shapes):
    src = _ln(src + msda_module(sd, p + "self_attn.", cfg, src + pos, ref, src, shapes), sd, p + "norm1")
    f = F.linear(F.relu(F.linear(src, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                 sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return _ln(src + f, sd, p + "norm2")


def mlp(sd: SD, p: str, x: torch.Tensor, n: int) -> torch.Tensor:
    """utils.MLP (utils.py:171-185)."""
    for j in range(n):
        x = F.linear(x, sd[f"{p}layers.{j}.weight"], sd[f"{p}layers.{j}.bias"])
        if j < n - 1:
            x = F.relu(x)
    return x


def inverse_sigmoid(x: torch.Tensor, eps: float = 1e-3) -> torch.Tensor:
    """GD/util/misc.py:704-708."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def sine_embed_4d(p: torch.Tensor) -> torch.Tensor:
    """gen_sineembed_for_position for [..., 4] (utils.py:204-230) -> [..., 512] as (y, x, w, h)."""
    dim_t = torch.arange(128, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / 128)
    def one(c):
        v = (c * (2 * math.pi))[..., None] / dim_t
        return torch.stack((v[..., 0::2].sin(), v[..., 1::2].cos()), dim=-1).flatten(-2)
    return torch.cat((one(p[..., 1]), one(p[..., 0]), one(p[..., 2]), one(p[..., 3])), dim=-1)


def encoder_proposals(shapes: Sequence[Tuple[int, int]]):
    """gen_encoder_output_proposals without padding (utils.py:56-116): ([S,4] unsigmoid / inf, [S] valid)."""
    props = []
    for lvl, (H, W) in enumerate(shapes):
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
        grid = (torch.stack((gx, gy), -1) + 0.5) / torch.tensor([float(W), float(H)])
        wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
        props.append(torch.cat((grid, wh), -1).view(-1, 4))
    pr = torch.cat(props, 0)
    valid = ((pr > 0.01) & (pr < 0.99)).all(-1)
    pr = torch.log(pr / (1 - pr))
    pr = pr.masked_fill(~valid[:, None], float("inf"))
    return pr, valid


# ----------------------------------------------------------------------------------------
# whole detector
# ----------------------------------------------------------------------------------------
@torch.no_grad()
def detector_forward(sd: SD, cfg: GDinoConfig, img: torch.Tensor, encoded_text: torch.Tensor,
                     text_self_mask: torch.Tensor, position_ids: torch.Tensor, stages: dict | None = None):
    """GroundingDINO.forward (groundingdino.py:227-365) from the normalised image [B,3,h,w] and the
    constant text features [n_text, 256].  Returns (pred_logits [B,nq,n_text], pred_boxes [B,nq,4])."""
    B = img.shape[0]
    D = cfg.hidden_dim
    feats = swin_forward(sd, cfg, img)
    srcs, poss = [], []
    for l, f in enumerate(feats):
        s = F.conv2d(f, sd[f"input_proj.{l}.0.weight"], sd[f"input_proj.{l}.0.bias"])
        srcs.append(F.group_norm(s, 32, sd[f"input_proj.{l}.1.weight"], sd[f"input_proj.{l}.1.bias"]))
    for l in range(len(feats), cfg.num_feature_levels):
        inp = feats[-1] if l == len(feats) else srcs[-1]
        s = F.conv2d(inp, sd[f"input_proj.{l}.0.weight"], sd[f"input_proj.{l}.0.bias"], stride=2, padding=1)
        srcs.append(F.group_norm(s, 32, sd[f"input_proj.{l}.1.weight"], sd[f"input_proj.{l}.1.bias"]))
    shapes = [(s.shape[2], s.shape[3]) for s in srcs]
    for l, (h, w) in enumerate(shapes):
        poss.append(pos_sine_hw(cfg, B, h, w))
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)                       # B,S,D
    pos = torch.cat([p.flatten(2).transpose(1, 2) + sd["transformer.level_embed"][l].view(1, 1, -1)
                     for l, p in enumerate(poss)], 1)
    if stages is not None:
        stages["src"], stages["pos"], stages["feats"] = src.clone(), pos.clone(), feats
    # ---- encoder (transformer.py:481-595)
    ref2 = enc_reference_points(shapes)[None, :, None, :].expand(B, -1, cfg.num_feature_levels, -1)
    text = encoded_text[None].expand(B, -1, -1)
    pos_text = sine_pos_embed_1d(position_ids.float())[None].expand(B, -1, -1)
    out = src
    t = "transformer."
    for i in range(cfg.enc_layers):
        out, text = fusion_layer(sd, f"{t}encoder.fusion_layers.{i}.", out, text)
        text = text_layer(sd, f"{t}encoder.text_layers.{i}.", text, pos_text, text_self_mask)
        out = deform_enc_layer(sd, f"{t}encoder.layers.{i}.", cfg, out, pos, ref2, shapes)
    memory = out
    if stages is not None:
        stages["memory"], stages["memory_text"] = memory.clone(), text.clone()
    # ---- two-stage query selection (transformer.py:284-327)
    props, valid = encoder_proposals(shapes)
    om = memory.masked_fill(~valid[None, :, None], 0.0)
    om = _ln(F.linear(om, sd[t + "enc_output.weight"], sd[t + "enc_output.bias"]), sd, t + "enc_output_norm")
    logits = (om @ text.transpose(-1, -2)).max(-1)[0]                                       # B,S
    coord_unsel = mlp(sd, t + "enc_out_bbox_embed.", om, 3) + props[None]
    # top-k by value, ties -> lower index first (torch.topk's tie order is unspecified)
    order = torch.sort(logits, dim=1, descending=True, stable=True)[1][:, :cfg.num_queries]
    if stages is not None and "force_topk" in stages:     # test hook: pin the query selection (sensitivity probes)
        order = stages["force_topk"].long()
    ref_unsig = torch.gather(coord_unsel, 1, order[..., None].expand(-1, -1, 4))
    tgt = sd[t + "tgt_embed.weight"][None].expand(B, -1, -1)
    if stages is not None:
        stages["topk"], stages["topk_logits"], stages["ref_unsig"] = order.clone(), logits.clone(), ref_unsig.clone()
    # ---- decoder (transformer.py:665-735, 868-927)
    ref = ref_unsig.sigmoid()
    output = tgt
    hs, refs = [], [ref]
    for i in range(cfg.dec_layers):
        p = f"{t}decoder.layers.{i}."
        ref_in = ref[:, :, None, :].expand(-1, -1, cfg.num_feature_levels, -1)           # valid_ratios == 1
        qse = sine_embed_4d(ref_in[:, :, 0, :])
        qpos = mlp(sd, t + "decoder.ref_point_head.", qse, 2)
        q = output + qpos
        output = _ln(output + mha(sd, p + "self_attn.", q, q, output, cfg.nheads), sd, p + "norm2")
        output = _ln(output + mha(sd, p + "ca_text.", output + qpos, text, text, cfg.nheads), sd, p + "catext_norm")
        output = _ln(output + msda_module(sd, p + "cross_attn.", cfg, output + qpos, ref_in, memory, shapes),
                     sd, p + "norm1")
        f = F.linear(F.relu(F.linear(output, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                     sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        output = _ln(output + f, sd, p + "norm3")
        ref = (mlp(sd, "bbox_embed.0.", output, 3) + inverse_sigmoid(ref)).sigmoid()
        refs.append(ref)
        hs.append(_ln(output, sd, t + "decoder.norm"))
    # ---- heads (groundingdino.py:331-349): last layer only
    boxes = (mlp(sd, "bbox_embed.0.", hs[-1], 3) + inverse_sigmoid(refs[-2])).sigmoid()
    logits = hs[-1] @ text.transpose(-1, -2)
    if stages is not None:
        stages["hs"], stages["refs"] = [h.clone() for h in hs], [r.clone() for r in refs]
    return logits, boxes


def postprocess_detections(logits: torch.Tensor, boxes: torch.Tensor, box_threshold: float = 0.2):
    """predict() post-processing (GD/util/inference.py:70-75) + cxcywh_to_xyxy
    (InkLayer/utils/processing.py:56-63) for ONE image: ([n,4] float64 xyxy normalised, [n] scores)."""
    prob = logits.sigmoid()
    score = prob.max(dim=1)[0]
    keep = score > box_threshold
    b = boxes[keep].double().numpy().reshape(-1, 4)
    cx, cy, w, h = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1), score[keep].numpy()
