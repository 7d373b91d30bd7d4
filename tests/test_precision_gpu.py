"""Split-f16 operands and f32-I/O attention (the precise tail of the SAM path) against fp64 math.
GPU box only."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K,scale_a,scale_w", [(512, 256, 256, 1.0, 0.06), (300, 128, 2048, 1.0, 0.02),
                                                   (4096, 256, 1280, 3.0, 0.03), (640, 32, 256, 1e-2, 1e-2),
                                                   (1024, 128, 64, 0.5, 0.12), (200, 768, 256, 30.0, 1.0)])
def test_split_gemm_is_fp32_grade(dev, M, N, K, scale_a, scale_w):
    """gemm(add_split_f16(A), split_weight(W)) vs float64: ~2^-21 relative, against 2^-11 for plain f16 operands.
    The (1e-2, 1e-2) case puts every low part far below f16's normal range without the 64x segment scaling."""
    from inklayer_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * scale_a).to(dev)
    w = (torch.randn(N, K, generator=g) * scale_w).to(dev)
    bias = torch.randn(N, generator=g).to(dev) * scale_a * scale_w
    addend = (torch.randn(K, generator=g) * scale_a).to(dev)
    ref = (a.double() + addend.double()) @ w.double().t() + bias.double()
    out = ops.gemm(ops.add_split_f16(a, addend), ops.split_weight(w), bias)
    rel = ((out.double() - ref).norm() / ref.norm()).item()
    plain = ops.gemm(ops.add_cvt_f16(a, addend), w.half(), bias)
    rel16 = ((plain.double() - ref).norm() / ref.norm()).item()
    print(f"M{M} N{N} K{K}: split {rel:.2e}  plain f16 {rel16:.2e}")
    assert rel < 2e-6 and rel < rel16 / 50


def test_layernorm_split_rows(dev):
    from inklayer_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(1000, 256, generator=g).to(dev) * 3 + 1
    gam, bet = torch.randn(256, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
    ref = torch.nn.functional.layer_norm(x.double(), (256,), gam.double(), bet.double(), 1e-6)
    s = ops.layernorm_rows(x, gam, bet, 1e-6, split=True)
    assert s.shape == (1000, 768) and s.dtype == torch.float16
    hi, lo, hs = s[:, :256].double(), s[:, 256:512].double(), s[:, 512:].double()
    assert ((hi + lo / 64 - ref).abs().max() / ref.abs().max()).item() < 1e-6
    assert torch.equal(hs, (s[:, :256].float() / 64).half().double())
    # GELU variant (LayerNorm2d + GELU of the upscaler) and the gather / zero-row case
    s2 = ops.layernorm_rows(x, gam, bet, 1e-6, split=True, act="gelu")
    ref2 = torch.nn.functional.gelu(ref)
    assert (((s2[:, :256].double() + s2[:, 256:512].double() / 64) - ref2).abs().max()).item() < 2e-6
    gather = torch.tensor([5, -1, 7, 0], dtype=torch.int32, device=dev)
    s3 = ops.layernorm_rows(x, gam, bet, 1e-6, split=True, gather=gather)
    assert torch.equal(s3[1], torch.zeros(768, dtype=torch.float16, device=dev)) and torch.equal(s3[0], s[5])


def _attn_ref(q, k, v, scale):
    a = (q.double() @ k.double().transpose(-1, -2)) * scale
    return a.softmax(-1) @ v.double()


@pytest.mark.parametrize("hd,heads,nq,nk", [(16, 8, 4096, 7), (32, 8, 7, 7)])
def test_attn_fewkeys_f32_rows(dev, hd, heads, nq, nk):
    """SAM decoder image -> token attention (4096 x 7, 8 heads x 16) and token self-attention (7 x 7, 8 x 32) on f32
    rows, incl. q_batch_rows (several batch entries reading the same query rows)."""
    from inklayer_amd import ops
    g = torch.Generator(device="cpu").manual_seed(hd + nq)
    n_img, n = 2, 5
    E = heads * hd
    q_img = torch.randn(n_img * nq, E, generator=g)
    k = torch.randn(n * nk, E, generator=g)
    v = torch.randn(n * nk, E, generator=g)
    img_of = [0, 1, 1, 0, 1]
    rows = torch.tensor([i * nq for i in img_of], dtype=torch.int32, device=dev)
    out = ops.attn_fewkeys(q_img.to(dev), k.to(dev), v.to(dev), B=n, n_heads=heads, head_dim=hd,
                           scale=1 / math.sqrt(hd), n_q=nq, q_batch_rows=rows)
    assert out.dtype == torch.float32 and out.shape == (n * nq, E)
    for b in range(n):
        qb = q_img[img_of[b] * nq:(img_of[b] + 1) * nq].view(nq, heads, hd).transpose(0, 1)
        kb = k[b * nk:(b + 1) * nk].view(nk, heads, hd).transpose(0, 1)
        vb = v[b * nk:(b + 1) * nk].view(nk, heads, hd).transpose(0, 1)
        ref = _attn_ref(qb, kb, vb, 1 / math.sqrt(hd)).transpose(0, 1).reshape(nq, E)
        got = out[b * nq:(b + 1) * nq].double().cpu()
        assert (got - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())


def test_attn_fewq_f32_rows(dev):
    """SAM decoder token -> image attention (7 x 4096, 8 heads x 16) on f32 rows, shared keys via kv_batch_rows and a
    ragged key count (4096 and 4000: the masked tail of the last 64-key tile)."""
    from inklayer_amd import ops
    g = torch.Generator(device="cpu").manual_seed(9)
    heads, hd, nq = 8, 16, 7
    E = heads * hd
    for nk in (4096, 4000):
        n_img, n = 2, 3
        q = torch.randn(n * nq, E, generator=g)
        k = torch.randn(n_img * nk, E, generator=g) * 2
        v = torch.randn(n_img * nk, E, generator=g)
        img_of = [1, 0, 1]
        rows = torch.tensor([i * nk for i in img_of], dtype=torch.int32, device=dev)
        out = ops.attn_fewq(q.to(dev), k.to(dev), v.to(dev), n_batch=n, n_heads=heads, head_dim=hd,
                            scale=1 / math.sqrt(hd), n_q=nq, n_k=nk, kv_batch_rows=rows)
        assert out.dtype == torch.float32
        for b in range(n):
            qb = q[b * nq:(b + 1) * nq].view(nq, heads, hd).transpose(0, 1)
            kb = k[img_of[b] * nk:(img_of[b] + 1) * nk].view(nk, heads, hd).transpose(0, 1)
            vb = v[img_of[b] * nk:(img_of[b] + 1) * nk].view(nk, heads, hd).transpose(0, 1)
            ref = _attn_ref(qb, kb, vb, 1 / math.sqrt(hd)).transpose(0, 1).reshape(nq, E)
            got = out[b * nq:(b + 1) * nq].double().cpu()
            assert (got - ref).abs().max().item() < 5e-6 * max(1.0, ref.abs().max().item())


def test_position_constants_and_gathered_residual(dev):
    """(x + pe) W = x W + pe W: attn_fewq's k_add / attn_fewkeys' q_add (per-position constants added inside the
    kernels) and layernorm_rows' gathered residual add (the image keys shared by the boxes of an image)."""
    from inklayer_amd import ops
    g = torch.Generator(device="cpu").manual_seed(17)
    heads, hd, nq, nk, n = 8, 16, 7, 4096, 3
    E = heads * hd
    q = torch.randn(n * nq, E, generator=g)
    k = torch.randn(n * nk, E, generator=g)
    v = torch.randn(n * nk, E, generator=g)
    kadd = torch.randn(nk, E, generator=g)
    a = ops.attn_fewq(q.to(dev), k.to(dev), v.to(dev), n_batch=n, n_heads=heads, head_dim=hd, scale=0.25, n_q=nq, n_k=nk,
                      k_add=kadd.to(dev))
    b = ops.attn_fewq(q.to(dev), (k.view(n, nk, E) + kadd).reshape(n * nk, E).contiguous().to(dev), v.to(dev), n_batch=n,
                      n_heads=heads, head_dim=hd, scale=0.25, n_q=nq, n_k=nk)
    assert (a - b).abs().max().item() < 1e-6
    T = 512
    qi = torch.randn(2 * T, E, generator=g)
    kk = torch.randn(n * nq, E, generator=g)
    vv = torch.randn(n * nq, E, generator=g)
    qadd = torch.randn(T, E, generator=g)
    rows = torch.tensor([T, 0, T], dtype=torch.int32, device=dev)
    a = ops.attn_fewkeys(qi.to(dev), kk.to(dev), vv.to(dev), B=n, n_heads=heads, head_dim=hd, scale=0.25, n_q=T,
                         q_batch_rows=rows, q_add=qadd.to(dev))
    b = ops.attn_fewkeys((qi.view(2, T, E) + qadd).reshape(2 * T, E).contiguous().to(dev), kk.to(dev), vv.to(dev), B=n,
                         n_heads=heads, head_dim=hd, scale=0.25, n_q=T, q_batch_rows=rows)
    assert (a - b).abs().max().item() < 1e-6
    # LayerNorm(x[r] + add[rows[r // T] + r % T])
    x = torch.randn(n * T, 256, generator=g)
    add = torch.randn(2 * T, 256, generator=g)
    gam, bet = torch.randn(256, generator=g), torch.randn(256, generator=g)
    got = ops.layernorm_rows(x.to(dev), gam.to(dev), bet.to(dev), 1e-5, out_dtype=torch.float32, add=add.to(dev),
                             add_batch_rows=rows, rows_per_batch=T)
    full = x.view(n, T, 256) + add.view(2, T, 256)[[1, 0, 1]]
    ref = torch.nn.functional.layer_norm(full.double(), (256,), gam.double(), bet.double(), 1e-5).reshape(n * T, 256)
    assert (got.double().cpu() - ref).abs().max().item() < 2e-5
