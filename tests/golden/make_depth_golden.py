"""Generate tests/golden/depth_small.npz by running the REFERENCE's own Depth-Anything-V2 modules on CPU (build
container only).  The model code (DA/dpt.py DepthAnythingV2 / DPTHead, DA/dinov2.py, DA/dinov2_layers,
DA/util/blocks.py) is torch-only; `cv2` and `torchvision.transforms.Compose` are imported at module level for the
image2tensor helper only (which is NOT run here: cv2 does not exist offline), so two in-memory stub modules stand in for
them exactly as tests/golden/make_gdino_golden.py does for GroundingDINO.  Nothing is written into /root/reference.

Weights: oracle.depth_ref.seeded_state_dict (numpy MT19937), loaded STRICTLY into the reference module - that pins the
parameter names and shapes.  Stored: a few slices of the intermediate features and the depth maps for two input sizes
(one square 266x266 -> 19x19 patches, one non-square 252x322 -> the position-embedding interpolation path).
"""
import importlib.machinery
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
DAROOT = "/root/reference/InkLayer/third_party/Depth_Anything_V2"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    if "cv2" not in sys.modules:
        _stub("cv2", INTER_AREA=3, INTER_CUBIC=2, INTER_NEAREST=0)
    if "torchvision" not in sys.modules:
        _stub("torchvision", __version__="0.0")
        _stub("torchvision.transforms", Compose=lambda ts: ts)
    sys.path.insert(0, DAROOT)
    from depth_anything_v2.dpt import DepthAnythingV2
    return DepthAnythingV2


def main():
    from oracle import depth_ref
    DepthAnythingV2 = load_reference()
    cfg = depth_ref.DepthConfig()
    sd = depth_ref.seeded_state_dict(cfg, 31)
    torch.manual_seed(0)
    model = DepthAnythingV2(encoder="vitb", features=128, out_channels=[96, 192, 384, 768]).eval()
    missing = model.load_state_dict(sd, strict=True)
    print("strict load ok:", missing)
    out = {"seed": np.int64(31)}
    rs = np.random.RandomState(7)
    for tag, (h, w) in (("sq", (266, 266)), ("ns", (252, 322))):
        x = torch.from_numpy(rs.standard_normal((1, 3, h, w)).astype(np.float32))
        with torch.no_grad():
            feats = model.pretrained.get_intermediate_layers(x, model.intermediate_layer_idx["vitb"], return_class_token=True)
            depth = model(x)
            mine = depth_ref.forward(sd, cfg, x)
        print(tag, "depth", tuple(depth.shape), "oracle-vs-reference max abs diff", (mine - depth).abs().max().item(),
              "depth max", depth.max().item())
        out[f"{tag}_x"] = x.numpy()
        out[f"{tag}_depth"] = depth.numpy()
        for i, (pt, cls) in enumerate(feats):
            out[f"{tag}_feat{i}"] = pt[0, ::7, ::16].numpy()
            out[f"{tag}_cls{i}"] = cls[0].numpy()
    # full-size input (518 x 518: the pos-embed identity path): checksums only
    x = torch.from_numpy(rs.standard_normal((1, 3, 518, 518)).astype(np.float32))
    with torch.no_grad():
        depth = model(x)
    out["full_x_seed"] = np.int64(7)
    out["full_depth_rows"] = depth[0, ::37].numpy()
    out["full_depth_mean"] = np.float64(depth.double().mean().item())
    np.savez_compressed(Path(__file__).resolve().parent / "depth_small.npz", **out)
    print("wrote depth_small.npz")


if __name__ == "__main__":
    main()
