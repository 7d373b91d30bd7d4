"""Generate tests/golden/gdino_small.npz by running the REFERENCE's own GroundingDINO modules on CPU
(build container only).  Follows SURVEY §8c: transformers is imported first, then in-memory stub
modules stand in for packages the image lacks (timm / torchvision / cv2 / pycocotools / addict /
yapf) — only trivial helpers (DropPath = identity at inference, to_2tuple, trunc_normal_) are
touched by the code under test.  Nothing is written into /root/reference.

GroundingDINO.__init__ itself needs bert-base-uncased from the network, so the whole-model forward
(groundingdino.py:227-365) is assembled here from the reference's sub-modules (SwinTransformer,
PositionEmbeddingSineHW, Transformer incl. encoder/decoder/fusion/text layers/MSDeformAttn, MLP,
ContrastiveEmbed) with a synthetic 4-token text_dict.  Weights come from
oracle.sam_ref.seeded_state_dict over oracle.gdino_ref.gdino_param_shapes; loading them into the
reference modules pins names and shapes.
"""
import importlib.machinery
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
GDROOT = "/root/reference/InkLayer/third_party/GroundingDINO"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    import transformers  # noqa: F401  (must come before the torchvision stub)

    class DropPath(nn.Identity):
        def __init__(self, *a, **k):
            super().__init__()

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def trunc_normal_(t, std=1.0, **k):
        return nn.init.normal_(t, std=std)

    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.layers", DropPath=DropPath, to_2tuple=to_2tuple, trunc_normal_=trunc_normal_)
    tv = _stub("torchvision", __version__="0.25.0", _is_tracing=lambda: False)
    _stub("torchvision.ops")
    _stub("torchvision.ops.boxes", nms=None, box_area=None)
    _stub("torchvision.ops.misc", FrozenBatchNorm2d=nn.BatchNorm2d)
    _stub("torchvision.models", resnet50=None, resnet101=None)
    _stub("torchvision.models._utils", IntermediateLayerGetter=nn.Module)
    _stub("torchvision.transforms")
    _stub("torchvision.transforms.functional")
    tv.ops = sys.modules["torchvision.ops"]
    tv.models = sys.modules["torchvision.models"]
    _stub("cv2")
    _stub("pycocotools")
    _stub("pycocotools.mask")
    _stub("addict", Dict=dict)
    _stub("yapf")
    _stub("yapf.yapflib")
    _stub("yapf.yapflib.yapf_api", FormatCode=lambda *a, **k: ("", False))
    sys.path.insert(0, GDROOT)


install_stubs()
from groundingdino.models.GroundingDINO.backbone.swin_transformer import build_swin_transformer  # noqa: E402
from groundingdino.models.GroundingDINO.backbone.position_encoding import PositionEmbeddingSineHW  # noqa: E402
from groundingdino.models.GroundingDINO.transformer import build_transformer  # noqa: E402
from groundingdino.models.GroundingDINO.utils import MLP, ContrastiveEmbed  # noqa: E402
from groundingdino.models.GroundingDINO.bertwarper import generate_masks_with_special_tokens_and_transfer_map  # noqa: E402
from groundingdino.models.GroundingDINO.ms_deform_attn import multi_scale_deformable_attn_pytorch  # noqa: E402
from groundingdino.util.misc import NestedTensor, inverse_sigmoid  # noqa: E402

from oracle import gdino_ref, sam_ref  # noqa: E402

SMALL = gdino_ref.GDinoConfig(enc_layers=2, dec_layers=2, num_queries=60)
SEED = 4321
TOKEN_IDS = [101, 4874, 1012, 102]      # "[CLS] object . [SEP]" (id of "object" is data, not checked)


def load_prefixed(module, sd, prefix):
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    res = module.load_state_dict(sub, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    bad = [k for k in res.missing_keys if "relative_position_index" not in k]
    assert not bad, bad
    return module


@torch.no_grad()
def main():
    cfg = SMALL
    sd = sam_ref.seeded_state_dict(gdino_ref.gdino_param_shapes(cfg), SEED)
    # make the fusion gates non-trivial (reference init 1e-4 would hide the fusion branch)
    for k in sd:
        if k.endswith("gamma_v") or k.endswith("gamma_l"):
            sd[k] = 0.3 * torch.ones_like(sd[k]) + 0.05 * sd[k]
    args = types.SimpleNamespace(
        hidden_dim=cfg.hidden_dim, dropout=0.0, nheads=cfg.nheads, num_queries=cfg.num_queries,
        dim_feedforward=cfg.dim_feedforward, enc_layers=cfg.enc_layers, dec_layers=cfg.dec_layers,
        pre_norm=False, query_dim=4, transformer_activation="relu", num_patterns=0,
        num_feature_levels=4, enc_n_points=4, dec_n_points=4, two_stage_type="standard",
        embed_init_tgt=True, use_text_enhancer=True, use_fusion_layer=True, use_checkpoint=False,
        use_transformer_ckpt=False, use_text_cross_attention=True, text_dropout=0.0,
        fusion_dropout=0.0, fusion_droppath=0.1)
    swin = build_swin_transformer("swin_T_224_1k", 224, out_indices=(1, 2, 3), dilation=False, use_checkpoint=False)
    swin.eval()
    load_prefixed(swin, sd, "backbone.0.")
    posemb = PositionEmbeddingSineHW(128, temperatureH=20, temperatureW=20, normalize=True)
    tr = build_transformer(args)
    tr.eval()
    bbox = MLP(256, 256, 4, 3)
    load_prefixed(bbox, sd, "bbox_embed.0.")
    enc_bbox = MLP(256, 256, 4, 3)
    tr.enc_out_bbox_embed = enc_bbox
    tr.enc_out_class_embed = ContrastiveEmbed()
    tr.decoder.bbox_embed = nn.ModuleList([bbox for _ in range(cfg.dec_layers)])
    tr.decoder.class_embed = nn.ModuleList([ContrastiveEmbed() for _ in range(cfg.dec_layers)])
    sub = {k[len("transformer."):]: v for k, v in sd.items() if k.startswith("transformer.")}
    for i in range(cfg.dec_layers):
        for j in range(3):
            for n in ("weight", "bias"):
                sub[f"decoder.bbox_embed.{i}.layers.{j}.{n}"] = sd[f"bbox_embed.0.layers.{j}.{n}"]
    res = tr.load_state_dict(sub, strict=True)
    inproj = nn.ModuleList([nn.Sequential(nn.Conv2d(c, 256, 1), nn.GroupNorm(32, 256)) for c in (192, 384, 768)]
                           + [nn.Sequential(nn.Conv2d(768, 256, 3, stride=2, padding=1), nn.GroupNorm(32, 256))])
    load_prefixed(inproj, sd, "input_proj.")

    rs = np.random.RandomState(9)
    h, w = 200, 264          # not multiples of 4*7 nor even at every stage: pads, odd merges, 13-like level
    img = torch.from_numpy(rs.standard_normal((1, 3, h, w)).astype(np.float32))
    enc_text = torch.from_numpy((0.5 * rs.standard_normal((4, 256))).astype(np.float32))

    tok = {"input_ids": torch.tensor([TOKEN_IDS])}
    self_mask, pos_ids, _ = generate_masks_with_special_tokens_and_transfer_map(tok, [101, 102, 1012, 1029], None)
    text_dict = {"encoded_text": enc_text[None].clone(), "text_token_mask": torch.ones(1, 4, dtype=torch.bool),
                 "position_ids": pos_ids, "text_self_attention_masks": self_mask}

    # ---- GroundingDINO.forward, assembled from the reference's modules (groundingdino.py:300-349)
    mask = torch.zeros((1, h, w), dtype=torch.bool)
    feats = swin(NestedTensor(img, mask))
    features = [feats[i] for i in range(3)]
    poss = [posemb(f).to(f.tensors.dtype) for f in features]
    srcs, masks = [], []
    for l, f in enumerate(features):
        s, m = f.decompose()
        srcs.append(inproj[l](s))
        masks.append(m)
    s = inproj[3](features[-1].tensors)
    m = torch.nn.functional.interpolate(mask[None].float(), size=s.shape[-2:]).to(torch.bool)[0]
    poss.append(posemb(NestedTensor(s, m)).to(s.dtype))
    srcs.append(s)
    masks.append(m)
    hs, reference, hs_enc, ref_enc, init_box = tr(srcs, masks, None, poss, None, None, text_dict)
    layer_hs, layer_ref = hs[-1], reference[:-1][-1]
    boxes = (bbox(layer_hs) + inverse_sigmoid(layer_ref)).sigmoid()
    logits = ContrastiveEmbed()(layer_hs, text_dict)

    # ---- MSDA core on its own (the reference's CPU form of the CUDA op)
    shapes = [(12, 9), (6, 5), (3, 3), (2, 2)]
    S = sum(a * b for a, b in shapes)
    v = torch.from_numpy(rs.standard_normal((2, S, 8, 32)).astype(np.float32))
    loc = torch.from_numpy(rs.uniform(-0.15, 1.15, size=(2, 37, 8, 4, 4, 2)).astype(np.float32))
    aw = torch.from_numpy(rs.uniform(0, 1, size=(2, 37, 8, 4, 4)).astype(np.float32))
    msda_out = multi_scale_deformable_attn_pytorch(v, torch.tensor(shapes), loc, aw)

    out = dict(
        seed=np.int64(SEED), image=img.numpy(), encoded_text=enc_text.numpy(),
        token_ids=np.array(TOKEN_IDS), self_mask=self_mask[0].numpy(), position_ids=pos_ids[0].numpy(),
        feat1=features[0].tensors[0, ::8].numpy(), feat3=features[2].tensors[0, ::16].numpy(),
        pos0=poss[0][0, ::16].numpy(), pos3=poss[3][0, ::16].numpy(),
        src3=srcs[3][0, ::8].numpy(),
        memory_text=text_dict["encoded_text"][0].numpy(),
        hs_last=layer_hs[0].numpy(), ref_init=reference[0][0].numpy(), ref_last=layer_ref[0].numpy(),
        pred_boxes=boxes[0].numpy(), pred_logits=logits[0, :, :4].numpy(),
        logits_pad_is_neginf=np.bool_(torch.isinf(logits[0, :, 4:]).all().item()),
        msda_value=v.numpy(), msda_loc=loc.numpy(), msda_w=aw.numpy(), msda_out=msda_out.numpy(),
        msda_shapes=np.array(shapes),
    )
    path = Path(__file__).with_name("gdino_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, path.stat().st_size >> 10, "KiB")


if __name__ == "__main__":
    main()
