"""Pins oracle/sam_ref.py to the reference: golden vectors were produced by the reference's own
SAM modules (tests/golden/make_sam_golden.py) from the same seeded weights."""
from pathlib import Path

import numpy as np
import torch

from oracle import sam_ref

GOLD = Path(__file__).parent / "golden" / "sam_small.npz"
SMALL = sam_ref.SamConfig(embed_dim=160, depth=4, num_heads=2, global_attn_indexes=(1, 3),
                          window_size=14, img_size=512, prompt_embed_dim=64, dec_depth=2,
                          dec_heads=2, dec_mlp_dim=128, iou_head_hidden=64, mask_in_chans=16)


def _close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    err = np.abs(a - b).max()
    assert err <= tol * max(1.0, np.abs(b).max()), (err, np.abs(b).max())


@torch.no_grad()
def test_sam_oracle_matches_reference_golden():
    g = np.load(GOLD)
    sd = sam_ref.seeded_state_dict(sam_ref.sam_param_shapes(SMALL), int(g["seed"]))
    img = torch.from_numpy(g["image"])
    ih, iw = map(int, g["input_hw"])
    x = sam_ref.preprocess(SMALL, img)[None]
    assert x.shape == (1, 3, 512, 512) and float(x[0, :, :, iw:].abs().max()) == 0.0
    _close(sam_ref.image_encoder(sd, SMALL, x, upto=1)[0, ::4, ::4, ::8], g["tokens_b0"], 2e-5)
    _close(sam_ref.image_encoder(sd, SMALL, x, upto=2)[0, ::4, ::4, ::8], g["tokens_b1"], 2e-5)
    emb = sam_ref.image_encoder(sd, SMALL, x)
    _close(emb[0], g["image_embedding"], 5e-5)
    pe = sam_ref.dense_pe(sd, SMALL)
    _close(pe[0, ::4], g["dense_pe"], 1e-5)
    boxes = torch.from_numpy(g["boxes"])
    sparse = sam_ref.embed_boxes(sd, SMALL, boxes)
    _close(sparse, g["sparse"], 1e-5)
    low, iou = sam_ref.mask_decoder(sd, SMALL, emb, pe, sparse)
    _close(low, g["low_res"], 1e-4)
    _close(iou, g["iou"], 1e-4)
    logits = sam_ref.postprocess_masks(SMALL, low, (ih, iw), tuple(int(v) for v in g["orig_hw"]))
    _close(logits[:, :, ::5, ::5], g["logits_sub"], 1e-4)
    counts = (logits > 0).flatten(1).sum(1).numpy()
    assert np.abs(counts - g["mask_counts"]).max() <= 2   # threshold-at-0 pixels may flip by rounding


def test_preprocess_shape_and_box_transform():
    # ResizeLongestSide rounding: int(x + 0.5) (SA/utils/transforms.py:93-102)
    assert sam_ref.preprocess_shape(750, 750, 1024) == (1024, 1024)
    assert sam_ref.preprocess_shape(600, 800, 1024) == (768, 1024)
    assert sam_ref.preprocess_shape(333, 1001, 1024) == (341, 1024)
    b = sam_ref.apply_boxes(torch.tensor([[10.0, 20.0, 30.0, 40.0]]), (600, 800), 1024)
    assert torch.allclose(b, torch.tensor([[12.8, 25.6, 38.4, 51.2]]))


def test_param_inventory_vit_h():
    shapes = sam_ref.sam_param_shapes(sam_ref.SamConfig())
    n = {k.split(".")[0]: 0 for k in shapes}
    for k, s in shapes.items():
        n[k.split(".")[0]] += int(np.prod(s))
    # parameter counts of sam_vit_h (637.0 M encoder; 4.06 M decoder)
    assert n["image_encoder"] == 637_026_048
    assert n["mask_decoder"] == 4_058_340
    assert n["prompt_encoder"] == 6_220 + 256  # + the (2,128) gaussian buffer
