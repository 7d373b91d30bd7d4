"""Pins oracle/depth_ref.py (Depth-Anything-V2 ViT-B) to the reference's own modules through
tests/golden/depth_small.npz (tests/golden/make_depth_golden.py: DA/dpt.py + DA/dinov2.py run on CPU, strict load)."""
from pathlib import Path

import numpy as np
import torch

GOLD = Path(__file__).resolve().parent / "golden" / "depth_small.npz"


def _setup():
    from oracle import depth_ref
    g = np.load(GOLD)
    cfg = depth_ref.DepthConfig()
    return depth_ref, g, cfg, depth_ref.seeded_state_dict(cfg, int(g["seed"]))


def test_network_matches_reference_golden_square_and_interpolated_pos_embed():
    depth_ref, g, cfg, sd = _setup()
    for tag in ("sq", "ns"):
        st = {}
        d = depth_ref.forward(sd, cfg, torch.from_numpy(g[f"{tag}_x"]), stages=st)
        assert np.abs(d.numpy() - g[f"{tag}_depth"]).max() < 2e-5 * max(1.0, np.abs(g[f"{tag}_depth"]).max())
        assert g[f"{tag}_depth"].max() > 0.1                      # not a dead (all-ReLU'd) map
        for i, (pt, cls) in enumerate(st["feats"]):
            assert np.abs(pt[0, ::7, ::16].numpy() - g[f"{tag}_feat{i}"]).max() < 2e-4
            assert np.abs(cls[0].numpy() - g[f"{tag}_cls{i}"]).max() < 2e-4


def test_full_size_518_matches_reference_rows():
    depth_ref, g, cfg, sd = _setup()
    rs = np.random.RandomState(int(g["full_x_seed"]))
    rs.standard_normal((1, 3, 266, 266)); rs.standard_normal((1, 3, 252, 322))      # the generator's stream position
    x = torch.from_numpy(rs.standard_normal((1, 3, 518, 518)).astype(np.float32))
    d = depth_ref.forward(sd, cfg, x)
    assert np.abs(d[0, ::37].numpy() - g["full_depth_rows"]).max() < 5e-5
    assert abs(d.double().mean().item() - float(g["full_depth_mean"])) < 1e-5


def test_resize_rule_and_cubic_properties():
    depth_ref, g, cfg, sd = _setup()
    assert depth_ref.resize_shape(750, 750) == (518, 518)
    assert depth_ref.resize_shape(1024, 1024) == (518, 518)
    assert depth_ref.resize_shape(600, 800) == (518, 686)          # 690.67 -> nearest multiple of 14
    assert depth_ref.resize_shape(512, 512) == (518, 518)
    assert depth_ref.resize_shape(300, 1000) == (518, 1722)
    rs = np.random.RandomState(0)
    img = rs.rand(40, 56, 3)
    assert np.array_equal(depth_ref.resize_cubic(img, 40, 56), img)
    const = np.full((30, 30, 3), 0.37)
    assert np.abs(depth_ref.resize_cubic(const, 77, 51) - 0.37).max() < 1e-12      # weights sum to 1, replicated border
    # the four taps are Keys' cubic convolution kernel with a = -0.75 evaluated at the tap distances
    def keys(t, a=-0.75):
        t = abs(t)
        if t <= 1:
            return (a + 2) * t ** 3 - (a + 3) * t ** 2 + 1
        return a * t ** 3 - 5 * a * t ** 2 + 8 * a * t - 4 * a if t < 2 else 0.0
    row = rs.rand(1, 23, 3)
    up = depth_ref.resize_cubic(np.tile(row, (4, 1, 1)), 4, 57)
    for d in (0, 5, 28, 56):
        fx = (d + 0.5) * 23 / 57 - 0.5
        sx = int(np.floor(fx))
        want = sum(row[0, min(max(sx - 1 + k, 0), 22), 1] * keys(fx - (sx - 1 + k)) for k in range(4))
        assert abs(up[2, d, 1] - want) < 1e-12
    x, (h, w) = depth_ref.image2tensor((rs.rand(75, 100, 3) * 255).astype(np.uint8), cfg)
    assert tuple(x.shape) == (1, 3, 518, 686) and (h, w) == (75, 100) and x.dtype == torch.float32
