"""Pillow-exact bilinear resize: the coefficient tables (host) and the HIP passes against PIL itself."""
import numpy as np
import pytest
from PIL import Image

SIZES = [((1024, 1024), (800, 800)),      # detector, square sketch
         ((750, 750), (800, 800)),        # bunny_cook: up-scaling
         ((750, 750), (1024, 1024)),      # SAM longest side, up-scaling
         ((1000, 2000), (666, 1332)),     # max_size=1333 branch of the detector resize
         ((1536, 1100), (1024, 733)),     # SAM, down-scaling, odd sizes
         ((512, 640), (512, 800)),        # one axis only
         ((3000, 900), (800, 240))]       # strong down-scaling: wide kernels


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    a[: h // 3] = 255                      # sketch-like: large saturated areas exercise the clipping
    a[h // 3: h // 2, ::7] = 0
    return a


@pytest.mark.parametrize("src,dst", SIZES)
def test_coefficients_and_emulation_match_pillow(src, dst):
    from inklayer_amd.resize import pil_bilinear_coeffs
    from oracle.pil_resize_ref import resize_bilinear_u8
    img = _img(*src, seed=src[0] + dst[1])
    want = np.asarray(Image.fromarray(img).resize((dst[1], dst[0]), Image.BILINEAR))
    got = resize_bilinear_u8(img, dst[0], dst[1], pil_bilinear_coeffs)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_detector_and_sam_shapes_use_the_reference_rounding():
    from inklayer_amd.gdino import resize_shape
    from inklayer_amd.sam import preprocess_shape
    assert resize_shape(1024, 1024) == (800, 800)
    assert preprocess_shape(750, 750, 1024) == (1024, 1024)
    assert preprocess_shape(1536, 1100, 1024) == (1024, 733)


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", SIZES)
def test_hip_resize_matches_pillow_bit_for_bit(src, dst):
    import torch
    from inklayer_amd import ops
    img = _img(*src, seed=src[1] + dst[0])
    want = np.asarray(Image.fromarray(img).resize((dst[1], dst[0]), Image.BILINEAR))
    got = ops.resize_bilinear_u8(torch.from_numpy(img).cuda(), dst[0], dst[1]).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_pipeline_prepare_equals_host_pil_path():
    """prepare() (GPU resizes) hands the engines exactly the bytes the host PIL path produced."""
    import torch
    from inklayer_amd import gdino, sam
    from inklayer_amd.pipeline import gpu_preprocess
    img = _img(750, 760, seed=5)
    det_ref = gdino.resize_for_detector(img)
    sam_ref = sam.resize_longest_side(np.ascontiguousarray(img), 1024)
    det, sm = gpu_preprocess(torch.from_numpy(img).cuda(), 1024)
    assert np.array_equal(det.cpu().numpy(), det_ref)
    assert np.array_equal(sm.cpu().numpy(), sam_ref)
