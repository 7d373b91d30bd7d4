"""Random-init weights with the reference's state_dict layout (no checkpoints exist offline).

Used by bench.py / smoke() / tools: generated directly on the GPU (637 M parameters in
milliseconds), every parameter non-zero.  Real checkpoints go through the same engines."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch


def _fill(shapes: Dict[str, Tuple[int, ...]], device, seed: int) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in shapes.items():
        x = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "weight" and len(shape) == 1:
            x = 1.0 + 0.1 * x
        elif leaf == "bias":
            x = 0.1 * x
        elif leaf == "weight" and len(shape) >= 2 and not any(
                e in name for e in ("point_embeddings", "not_a_point_embed", "no_mask_embed", "iou_token",
                                    "mask_tokens", "tgt_embed", "level_embed")):
            fan_in = shape[0] if "output_upscaling" in name else int(math.prod(shape[1:]))
            x = x / math.sqrt(fan_in)
        elif "rel_pos" in name or "relative_position_bias_table" in name:
            x = 0.2 * x
        elif "gaussian_matrix" in name:
            pass
        else:
            x = 0.5 * x
        sd[name] = x
    return sd


def sam_param_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """Parameter names/shapes of segment_anything's Sam (SA/build_sam.py:66-101)."""
    D, P, E = cfg.embed_dim, cfg.patch_size, cfg.prompt_embed_dim
    hd, mlp, g = D // cfg.num_heads, int(D * cfg.mlp_ratio), cfg.grid
    s: Dict[str, Tuple[int, ...]] = {
        "image_encoder.pos_embed": (1, g, g, D),
        "image_encoder.patch_embed.proj.weight": (D, 3, P, P),
        "image_encoder.patch_embed.proj.bias": (D,),
    }
    for i in range(cfg.depth):
        p = f"image_encoder.blocks.{i}."
        S = g if i in cfg.global_attn_indexes else cfg.window_size
        s.update({p + "norm1.weight": (D,), p + "norm1.bias": (D,), p + "attn.rel_pos_h": (2 * S - 1, hd),
                  p + "attn.rel_pos_w": (2 * S - 1, hd), p + "attn.qkv.weight": (3 * D, D),
                  p + "attn.qkv.bias": (3 * D,), p + "attn.proj.weight": (D, D), p + "attn.proj.bias": (D,),
                  p + "norm2.weight": (D,), p + "norm2.bias": (D,), p + "mlp.lin1.weight": (mlp, D),
                  p + "mlp.lin1.bias": (mlp,), p + "mlp.lin2.weight": (D, mlp), p + "mlp.lin2.bias": (D,)})
    s.update({"image_encoder.neck.0.weight": (E, D, 1, 1), "image_encoder.neck.1.weight": (E,),
              "image_encoder.neck.1.bias": (E,), "image_encoder.neck.2.weight": (E, E, 3, 3),
              "image_encoder.neck.3.weight": (E,), "image_encoder.neck.3.bias": (E,),
              "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix": (2, E // 2),
              "prompt_encoder.no_mask_embed.weight": (1, E)})
    for i in range(4):
        s[f"prompt_encoder.point_embeddings.{i}.weight"] = (1, E)
    t = "mask_decoder.transformer."
    def attn(prefix, internal):
        for n in ("q_proj", "k_proj", "v_proj"):
            s[f"{prefix}.{n}.weight"] = (internal, E)
            s[f"{prefix}.{n}.bias"] = (internal,)
        s[f"{prefix}.out_proj.weight"] = (E, internal)
        s[f"{prefix}.out_proj.bias"] = (E,)
    for i in range(cfg.dec_depth):
        p = f"{t}layers.{i}."
        attn(p + "self_attn", E)
        attn(p + "cross_attn_token_to_image", E // 2)
        attn(p + "cross_attn_image_to_token", E // 2)
        for n in ("norm1", "norm2", "norm3", "norm4"):
            s[p + n + ".weight"] = (E,)
            s[p + n + ".bias"] = (E,)
        s.update({p + "mlp.lin1.weight": (cfg.dec_mlp_dim, E), p + "mlp.lin1.bias": (cfg.dec_mlp_dim,),
                  p + "mlp.lin2.weight": (E, cfg.dec_mlp_dim), p + "mlp.lin2.bias": (E,)})
    attn(t + "final_attn_token_to_image", E // 2)
    s.update({t + "norm_final_attn.weight": (E,), t + "norm_final_attn.bias": (E,),
              "mask_decoder.iou_token.weight": (1, E),
              "mask_decoder.mask_tokens.weight": (cfg.num_mask_tokens, E),
              "mask_decoder.output_upscaling.0.weight": (E, E // 4, 2, 2),
              "mask_decoder.output_upscaling.0.bias": (E // 4,),
              "mask_decoder.output_upscaling.1.weight": (E // 4,),
              "mask_decoder.output_upscaling.1.bias": (E // 4,),
              "mask_decoder.output_upscaling.3.weight": (E // 4, E // 8, 2, 2),
              "mask_decoder.output_upscaling.3.bias": (E // 8,)})
    dims = [E, E, E, E // 8]
    idims = [E, 256, 256, cfg.num_mask_tokens]
    for j in range(3):
        for i in range(cfg.num_mask_tokens):
            s[f"mask_decoder.output_hypernetworks_mlps.{i}.layers.{j}.weight"] = (dims[j + 1], dims[j])
            s[f"mask_decoder.output_hypernetworks_mlps.{i}.layers.{j}.bias"] = (dims[j + 1],)
        s[f"mask_decoder.iou_prediction_head.layers.{j}.weight"] = (idims[j + 1], idims[j])
        s[f"mask_decoder.iou_prediction_head.layers.{j}.bias"] = (idims[j + 1],)
    return s


def random_sam_state_dict(cfg, device, seed: int = 0) -> Dict[str, torch.Tensor]:
    return _fill(sam_param_shapes(cfg), device, seed)


def gdino_param_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """Parameter names/shapes of the GroundingDINO Swin-T model minus BERT/feat_map
    (GD/models/GroundingDINO/groundingdino.py module tree; models/GroundingDINO_SwinT_OGC.py)."""
    s: Dict[str, Tuple[int, ...]] = {}
    C0, ws, D, Fd = cfg.embed_dim, cfg.window_size, cfg.hidden_dim, cfg.dim_feedforward
    bb = "backbone.0."
    s.update({bb + "patch_embed.proj.weight": (C0, 3, 4, 4), bb + "patch_embed.proj.bias": (C0,),
              bb + "patch_embed.norm.weight": (C0,), bb + "patch_embed.norm.bias": (C0,)})
    for i, (dep, nh) in enumerate(zip(cfg.depths, cfg.num_heads)):
        C = C0 * 2 ** i
        for b in range(dep):
            p = f"{bb}layers.{i}.blocks.{b}."
            s.update({p + "norm1.weight": (C,), p + "norm1.bias": (C,),
                      p + "attn.relative_position_bias_table": ((2 * ws - 1) ** 2, nh),
                      p + "attn.qkv.weight": (3 * C, C), p + "attn.qkv.bias": (3 * C,),
                      p + "attn.proj.weight": (C, C), p + "attn.proj.bias": (C,),
                      p + "norm2.weight": (C,), p + "norm2.bias": (C,),
                      p + "mlp.fc1.weight": (4 * C, C), p + "mlp.fc1.bias": (4 * C,),
                      p + "mlp.fc2.weight": (C, 4 * C), p + "mlp.fc2.bias": (C,)})
        if i < len(cfg.depths) - 1:
            p = f"{bb}layers.{i}.downsample."
            s.update({p + "reduction.weight": (2 * C, 4 * C), p + "norm.weight": (4 * C,), p + "norm.bias": (4 * C,)})
        if i in cfg.out_indices:
            s.update({f"{bb}norm{i}.weight": (C,), f"{bb}norm{i}.bias": (C,)})
    chans = [C0 * 2 ** i for i in cfg.out_indices]
    for l in range(cfg.num_feature_levels):
        s[f"input_proj.{l}.0.weight"] = (D, chans[l], 1, 1) if l < len(chans) else (D, chans[-1], 3, 3)
        s.update({f"input_proj.{l}.0.bias": (D,), f"input_proj.{l}.1.weight": (D,), f"input_proj.{l}.1.bias": (D,)})
    t = "transformer."
    s[t + "level_embed"] = (cfg.num_feature_levels, D)
    M, L, P = cfg.nheads, cfg.num_feature_levels, cfg.n_points
    E = Fd // 2

    def lin(p, o, i):
        s[p + ".weight"], s[p + ".bias"] = (o, i), (o,)

    def ln(p, n=D):
        s[p + ".weight"], s[p + ".bias"] = (n,), (n,)

    def msda(p):
        lin(p + "sampling_offsets", M * L * P * 2, D); lin(p + "attention_weights", M * L * P, D)
        lin(p + "value_proj", D, D); lin(p + "output_proj", D, D)

    def mha(p):
        s[p + "in_proj_weight"], s[p + "in_proj_bias"] = (3 * D, D), (3 * D,)
        lin(p + "out_proj", D, D)

    for i in range(cfg.enc_layers):
        p = f"{t}encoder.layers.{i}."
        msda(p + "self_attn."); ln(p + "norm1"); lin(p + "linear1", Fd, D); lin(p + "linear2", D, Fd); ln(p + "norm2")
        p = f"{t}encoder.text_layers.{i}."
        mha(p + "self_attn."); lin(p + "linear1", Fd // 2, D); lin(p + "linear2", D, Fd // 2); ln(p + "norm1"); ln(p + "norm2")
        p = f"{t}encoder.fusion_layers.{i}."
        ln(p + "layer_norm_v"); ln(p + "layer_norm_l")
        for n in ("v_proj", "l_proj", "values_v_proj", "values_l_proj"):
            lin(p + "attn." + n, E, D)
        lin(p + "attn.out_v_proj", D, E); lin(p + "attn.out_l_proj", D, E)
        s[p + "gamma_v"], s[p + "gamma_l"] = (D,), (D,)
    for i in range(cfg.dec_layers):
        p = f"{t}decoder.layers.{i}."
        msda(p + "cross_attn."); ln(p + "norm1"); mha(p + "ca_text."); ln(p + "catext_norm")
        mha(p + "self_attn."); ln(p + "norm2"); lin(p + "linear1", Fd, D); lin(p + "linear2", D, Fd); ln(p + "norm3")
    ln(t + "decoder.norm")
    lin(t + "decoder.ref_point_head.layers.0", D, 2 * D); lin(t + "decoder.ref_point_head.layers.1", D, D)
    s[t + "tgt_embed.weight"] = (cfg.num_queries, D)
    lin(t + "enc_output", D, D); ln(t + "enc_output_norm")
    for j, (o, i_) in enumerate(((D, D), (D, D), (4, D))):
        lin(f"{t}enc_out_bbox_embed.layers.{j}", o, i_)
        lin(f"bbox_embed.0.layers.{j}", o, i_)
    return s


def random_gdino_state_dict(cfg, device, seed: int = 1) -> Dict[str, torch.Tensor]:
    sd = _fill(gdino_param_shapes(cfg), device, seed)
    for k in sd:                                   # in_proj_weight has no ".weight" leaf
        if k.endswith("in_proj_weight"):
            sd[k] = sd[k] / math.sqrt(sd[k].shape[1]) / 0.5
        elif k.endswith("in_proj_bias"):
            sd[k] = 0.2 * sd[k]
        elif k.endswith("gamma_v") or k.endswith("gamma_l"):
            sd[k] = 0.2 + 0.1 * sd[k]
    return sd


def random_text_features(cfg, device, n_tokens: int = 4, seed: int = 2) -> torch.Tensor:
    """Stand-in for feat_map(BERT("object.")) [n_tokens, 256] (no bert-base-uncased offline)."""
    g = torch.Generator(device=device).manual_seed(seed)
    return 0.5 * torch.randn((n_tokens, cfg.hidden_dim), generator=g, device=device)


def depth_param_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """Parameter names/shapes of DepthAnythingV2("vitb") (DA/dpt.py:153-176, DA/dinov2.py:397-415)."""
    D, P, Fe = cfg.embed_dim, cfg.patch_size, cfg.features
    g = cfg.img_size // P
    s: Dict[str, Tuple[int, ...]] = {
        "pretrained.cls_token": (1, 1, D), "pretrained.pos_embed": (1, g * g + 1, D), "pretrained.mask_token": (1, D),
        "pretrained.patch_embed.proj.weight": (D, 3, P, P), "pretrained.patch_embed.proj.bias": (D,),
        "pretrained.norm.weight": (D,), "pretrained.norm.bias": (D,),
    }
    for i in range(cfg.depth):
        p = f"pretrained.blocks.{i}."
        s.update({p + "norm1.weight": (D,), p + "norm1.bias": (D,), p + "attn.qkv.weight": (3 * D, D),
                  p + "attn.qkv.bias": (3 * D,), p + "attn.proj.weight": (D, D), p + "attn.proj.bias": (D,),
                  p + "ls1.gamma": (D,), p + "norm2.weight": (D,), p + "norm2.bias": (D,),
                  p + "mlp.fc1.weight": (cfg.mlp_ratio * D, D), p + "mlp.fc1.bias": (cfg.mlp_ratio * D,),
                  p + "mlp.fc2.weight": (D, cfg.mlp_ratio * D), p + "mlp.fc2.bias": (D,), p + "ls2.gamma": (D,)})
    oc, h = cfg.out_channels, "depth_head."
    for i, c in enumerate(oc):
        s[f"{h}projects.{i}.weight"], s[f"{h}projects.{i}.bias"] = (c, D, 1, 1), (c,)
        s[f"{h}scratch.layer{i + 1}_rn.weight"] = (Fe, c, 3, 3)
    s[h + "resize_layers.0.weight"], s[h + "resize_layers.0.bias"] = (oc[0], oc[0], 4, 4), (oc[0],)
    s[h + "resize_layers.1.weight"], s[h + "resize_layers.1.bias"] = (oc[1], oc[1], 2, 2), (oc[1],)
    s[h + "resize_layers.3.weight"], s[h + "resize_layers.3.bias"] = (oc[3], oc[3], 3, 3), (oc[3],)
    for r in (1, 2, 3, 4):
        p = f"{h}scratch.refinenet{r}."
        s[p + "out_conv.weight"], s[p + "out_conv.bias"] = (Fe, Fe, 1, 1), (Fe,)
        for u in (1, 2):
            for c in (1, 2):
                s[f"{p}resConfUnit{u}.conv{c}.weight"], s[f"{p}resConfUnit{u}.conv{c}.bias"] = (Fe, Fe, 3, 3), (Fe,)
    s[h + "scratch.output_conv1.weight"], s[h + "scratch.output_conv1.bias"] = (Fe // 2, Fe, 3, 3), (Fe // 2,)
    s[h + "scratch.output_conv2.0.weight"], s[h + "scratch.output_conv2.0.bias"] = (32, Fe // 2, 3, 3), (32,)
    s[h + "scratch.output_conv2.2.weight"], s[h + "scratch.output_conv2.2.bias"] = (1, 32, 1, 1), (1,)
    return s


def random_depth_state_dict(cfg, device, seed: int = 3) -> Dict[str, torch.Tensor]:
    sd = _fill(depth_param_shapes(cfg), device, seed)
    for k in sd:
        if k.endswith("gamma"):
            sd[k] = 0.5 + 0.2 * sd[k]          # LayerScale
    return sd
