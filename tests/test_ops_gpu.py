"""Op-level parity of the HIP kernels against plain PyTorch fp32/fp64 math (GPU box only)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_gemm(a, w, bias, act, col_scale, residual, row_map, out_rows):
    y = a.double() @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    if act == "gelu":
        y = torch.nn.functional.gelu(y)
    elif act == "relu":
        y = torch.relu(y)
    if col_scale is not None:
        y = y * col_scale.double()
    if row_map is None:
        if residual is not None:
            y = y + residual.double()
        return y
    out = torch.zeros((out_rows, y.shape[1]), dtype=torch.float64, device=a.device)
    keep = row_map >= 0
    idx = row_map[keep].long()
    out[idx] = y[keep] + (residual.double()[idx] if residual is not None else 0)
    return out


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (4096, 3840, 1280),
                                   (4900, 1280, 1280), (1000, 96, 96), (77, 256, 32),
                                   (13294, 2048, 256), (112, 128, 2048), (300, 36, 160)])
@pytest.mark.parametrize("act", [None, "gelu", "relu"])
def test_gemm_matches_fp64(dev, M, N, K, act):
    from inklayer_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 31 + N * 7 + K)
    a = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).half().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    out = ops.gemm(a, w, bias, act=act)
    ref = _ref_gemm(a, w, bias, act, None, None, None, M)
    err = (out.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-5 * scale + 1e-5, (err, scale)


def test_gemm_epilogue_scatter_residual_scale_f16(dev):
    from inklayer_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K, R = 700, 256, 192, 650
    a = torch.randn(M, K, generator=g).half().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).half().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    cs = torch.randn(N, generator=g).to(dev)
    res = torch.randn(R, N, generator=g).to(dev)
    perm = torch.randperm(M, generator=g)
    row_map = torch.full((M,), -1, dtype=torch.int32)
    row_map[perm[:R]] = torch.arange(R, dtype=torch.int32)
    row_map = row_map.to(dev)
    out = torch.full((R, N), float("nan"), device=dev)
    ops.gemm(a, w, bias, col_scale=cs, residual=res, row_map=row_map, out=out)
    ref = _ref_gemm(a, w, bias, None, cs, res, row_map, R)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 1e-4
    # f16 output + strided A (a view into a wider buffer)
    big = torch.randn(M, 3 * K, generator=g).half().to(dev)
    a2 = big[:, K:2 * K]
    o16 = ops.gemm(a2, w, bias, out_dtype=torch.float16)
    ref2 = _ref_gemm(a2, w, bias, None, None, None, None, M)
    assert o16.dtype == torch.float16
    assert (o16.double() - ref2).abs().max().item() < 2e-3 * ref2.abs().max().item()


@pytest.mark.parametrize("C", [32, 64, 96, 256, 768, 1280, 2048])
def test_layernorm_rows(dev, C):
    from inklayer_amd import ops
    g = torch.Generator(device="cpu").manual_seed(C)
    R = 333
    x = (torch.randn(R, C, generator=g) * 3 + 0.5).to(dev)
    gamma = torch.randn(C, generator=g).to(dev)
    beta = torch.randn(C, generator=g).to(dev)
    ref = torch.nn.functional.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-6)
    o32 = ops.layernorm_rows(x, gamma, beta, 1e-6, out_dtype=torch.float32)
    assert (o32.double() - ref).abs().max().item() < 2e-5
    gather = torch.randint(-1, R, (500,), generator=g, dtype=torch.int32).to(dev)
    o16 = ops.layernorm_rows(x, gamma, beta, 1e-6, gather=gather)
    refg = ref[gather.clamp(min=0).long()] * (gather >= 0).unsqueeze(1)
    assert (o16.double() - refg).abs().max().item() < 4e-3 * max(1.0, refg.abs().max().item())
    assert (o16[gather < 0] == 0).all()


def test_add_cvt(dev):
    from inklayer_amd import ops
    a = torch.randn(1000, 256, device=dev)
    b = torch.randn(1000, 256, device=dev)
    assert torch.equal(ops.add_cvt_f16(a, b), (a + b).half())
    assert torch.equal(ops.add_cvt_f16(a), a.half())
