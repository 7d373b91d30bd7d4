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
