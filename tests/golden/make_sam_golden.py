"""Generate tests/golden/sam_small.npz by running the REFERENCE's own SAM modules
(/root/reference/InkLayer/third_party/segment-anything/segment_anything/modeling) on CPU.

Build-container only (the reference does not travel to the GPU box).  The reference package
is imported read-only, exactly as SURVEY §8c records: `modeling/` is put on sys.path as a
top-level package so that torchvision-dependent siblings are never touched.

The weights are NOT stored: they come from oracle.sam_ref.seeded_state_dict (numpy's frozen
MT19937 stream), and `load_state_dict(strict=True)` into the reference modules below is what
pins the oracle's parameter names and shapes to the reference.
"""
import sys
from functools import partial
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
SA = "/root/reference/InkLayer/third_party/segment-anything/segment_anything"
sys.path.insert(0, SA)

import modeling as ref  # noqa: E402  (the reference's package)
from oracle import sam_ref  # noqa: E402

SMALL = sam_ref.SamConfig(embed_dim=160, depth=4, num_heads=2, global_attn_indexes=(1, 3),
                          window_size=14, img_size=512, prompt_embed_dim=64, dec_depth=2,
                          dec_heads=2, dec_mlp_dim=128, iou_head_hidden=64, mask_in_chans=16)
SEED = 1234


def build_reference(cfg):
    E, g = cfg.prompt_embed_dim, cfg.grid
    return ref.Sam(
        image_encoder=ref.ImageEncoderViT(
            depth=cfg.depth, embed_dim=cfg.embed_dim, img_size=cfg.img_size, mlp_ratio=cfg.mlp_ratio,
            norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=cfg.num_heads,
            patch_size=cfg.patch_size, qkv_bias=True, use_rel_pos=True,
            global_attn_indexes=list(cfg.global_attn_indexes), window_size=cfg.window_size,
            out_chans=E),
        prompt_encoder=ref.PromptEncoder(embed_dim=E, image_embedding_size=(g, g),
                                         input_image_size=(cfg.img_size, cfg.img_size),
                                         mask_in_chans=cfg.mask_in_chans),
        mask_decoder=ref.MaskDecoder(
            num_multimask_outputs=3,
            transformer=ref.TwoWayTransformer(depth=cfg.dec_depth, embedding_dim=E,
                                              mlp_dim=cfg.dec_mlp_dim, num_heads=cfg.dec_heads),
            transformer_dim=E, iou_head_depth=3, iou_head_hidden_dim=cfg.iou_head_hidden),
    ).eval()


@torch.no_grad()
def main():
    cfg = SMALL
    sd = sam_ref.seeded_state_dict(sam_ref.sam_param_shapes(cfg), SEED)
    model = build_reference(cfg)
    model.load_state_dict(sd, strict=True)   # pins names + shapes

    rs = np.random.RandomState(7)
    # a non-square "resized" image (so the zero padding of Sam.preprocess is exercised)
    ih, iw = 512, 384
    img = torch.from_numpy(rs.randint(0, 256, size=(3, ih, iw)).astype(np.uint8))
    x = model.preprocess(img)[None]
    emb = model.image_encoder(x)
    # token map after block 0 (windowed) and block 1 (global), for stage-level pinning
    t = model.image_encoder.patch_embed(x) + model.image_encoder.pos_embed
    t0 = model.image_encoder.blocks[0](t)
    t1 = model.image_encoder.blocks[1](t0)

    boxes = torch.tensor([[30.5, 40.25, 200.0, 310.75], [0.0, 0.0, 383.0, 511.0],
                          [100.0, 17.0, 140.5, 90.0]])
    sparse, dense = model.prompt_encoder(points=None, boxes=boxes, masks=None)
    low, iou = model.mask_decoder(image_embeddings=emb, image_pe=model.prompt_encoder.get_dense_pe(),
                                  sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense,
                                  multimask_output=False)
    orig_hw = (700, 525)
    logits = model.postprocess_masks(low, (ih, iw), orig_hw)

    out = dict(
        seed=np.int64(SEED), image=img.numpy(), boxes=boxes.numpy(),
        input_hw=np.array([ih, iw]), orig_hw=np.array(orig_hw),
        tokens_b0=t0[0, ::4, ::4, ::8].numpy(), tokens_b1=t1[0, ::4, ::4, ::8].numpy(),
        image_embedding=emb[0].numpy().astype(np.float32),
        dense_pe=model.prompt_encoder.get_dense_pe()[0, ::4].numpy(),
        sparse=sparse.numpy(), low_res=low.numpy(), iou=iou.numpy(),
        logits_sub=logits[:, :, ::5, ::5].numpy(),
        mask_counts=(logits > 0).flatten(1).sum(1).numpy(),
    )
    path = Path(__file__).with_name("sam_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, path.stat().st_size >> 10, "KiB")


if __name__ == "__main__":
    main()
