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


@pytest.mark.parametrize("variant", [-1, 0, 10, 45])
@pytest.mark.parametrize("M,N,K", [(300, 36, 160), (77, 252, 128), (513, 324, 192)])
def test_gemm_f16_out_ragged_narrow_rows(dev, variant, M, N, K):
    """f16 output whose row stride is not a multiple of 8 elements (8-byte stores) and whose last 16-B chunk is
    cut by N, with GELU and a late (non-linear) residual - the epilogue paths the big shapes never take."""
    from inklayer_amd import ops, _lib
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).half().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    _lib.lib().ink_gemm_set_variant(variant)
    try:
        o1 = ops.gemm(a, w, bias, out_dtype=torch.float16)
        o2 = ops.gemm(a, w, bias, act="gelu", residual=res, out_dtype=torch.float16)
    finally:
        _lib.lib().ink_gemm_set_variant(-1)
    r1 = _ref_gemm(a, w, bias, None, None, None, None, M)
    r2 = _ref_gemm(a, w, bias, "gelu", None, res, None, M)
    assert (o1.double() - r1).abs().max().item() < 2e-3 * r1.abs().max().item()
    assert (o2.double() - r2).abs().max().item() < 2e-3 * r2.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(300, 320, 128), (256 * 70 + 9, 1280, 256), (256 * 300 + 5, 320, 192)])
def test_gemm_persistent_variant_matches_one_tile_kernel(dev, M, N, K):
    """Variant 55 = the 256x320 ping-pong kernel with persistent workgroups (the next tile's pipeline fill is requested
    before the stores of the current one): same arithmetic in the same order as variant 45, so bit-equal outputs, for
    every epilogue it takes (f16, GELU -> f16, f32 with and without residual), with fewer tiles than CUs, several tiles
    per workgroup, a ragged last row tile; forms it does not take are rejected."""
    from inklayer_amd import ops, _lib
    g = torch.Generator(device="cpu").manual_seed(M + K)
    a = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).half().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    calls = [lambda: ops.gemm(a, w, bias, out_dtype=torch.float16), lambda: ops.gemm(a, w, bias, act="gelu", out_dtype=torch.float16),
             lambda: ops.gemm(a, w, bias, residual=res), lambda: ops.gemm(a, w, None)]
    outs = {}
    try:
        for var in (445, 455):
            _lib.lib().ink_gemm_set_variant(var)
            outs[var] = [c() for c in calls]
        with pytest.raises(Exception):
            ops.gemm(a, w, bias, col_scale=torch.ones(N, device=dev), residual=res)      # layer scale: one-tile kernel only
    finally:
        _lib.lib().ink_gemm_set_variant(-1)
    for o45, o55 in zip(outs[445], outs[455]):
        assert torch.equal(o45, o55)
    ref = _ref_gemm(a, w, bias, None, None, res, None, M)
    assert (outs[455][2].double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("M,N,K", [(1100, 640, 1280), (515, 320, 2560)])
def test_gemm_residual_through_mfma_variant(dev, M, N, K):
    """Variant 54 of the 256x320 ping-pong kernel feeds the f32 residual through the MFMA pipe during the K loop
    (hi + lo f16 halves against an identity operand) instead of preloading it into the accumulators: same result as
    the preload to f32 rounding, in place, with a ragged last row tile; shapes it cannot take are rejected."""
    from inklayer_amd import ops, _lib
    g = torch.Generator(device="cpu").manual_seed(M + K)
    a = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).half().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = (torch.randn(M, N, generator=g) * 4).to(dev)
    ref = _ref_gemm(a, w, bias, None, None, res, None, M)
    _lib.lib().ink_gemm_set_variant(454)
    try:
        out = res.clone()
        ops.gemm(a, w, bias, residual=out, out=out)
        with pytest.raises(Exception):
            ops.gemm(a[:, :320].contiguous(), w[:, :320].contiguous(), bias, residual=res)      # K / 32 = 10
        with pytest.raises(Exception):
            ops.gemm(a, w, bias, out_dtype=torch.float32)                                       # no residual
    finally:
        _lib.lib().ink_gemm_set_variant(-1)
    assert (out.double() - ref).abs().max().item() < 1e-4
    pre = ops.gemm(a, w, bias, residual=res)
    assert (out - pre).abs().max().item() < 1e-4


@pytest.mark.parametrize("variant", [-1, 0, 10, 40, 42, 45, 47])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.float16])
def test_gemm_residual_preload_rowmap(dev, variant, out_dtype):
    """Linear residual (no activation / layer scale) is preloaded into the accumulators before the K loop and the
    epilogue is load-free: check it with a scattering row_map that drops rows, ragged M/N tiles, K deep enough for
    several K-tiles, on every tile family (auto, 128x128, 16-wave 256x256, ping-pong 256x256 and 256x320)."""
    from inklayer_amd import ops, _lib
    g = torch.Generator(device="cpu").manual_seed(11)
    M, N, K, R = 1100, 712, 320, 1000
    a = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).half().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(R, N, generator=g).to(dev)
    perm = torch.randperm(M, generator=g)
    row_map = torch.full((M,), -1, dtype=torch.int32)
    row_map[perm[:R]] = torch.arange(R, dtype=torch.int32)
    row_map = row_map.to(dev)
    ref = _ref_gemm(a, w, bias, None, None, res, row_map, R)
    _lib.lib().ink_gemm_set_variant(variant)
    try:
        if out_dtype == torch.float32:
            out = res.clone()                                  # in place: out IS the residual stream
            ops.gemm(a, w, bias, residual=out, row_map=row_map, out=out)
            tol = 1e-4
        else:
            out = torch.full((R, N), float("nan"), device=dev, dtype=torch.float16)
            ops.gemm(a, w, bias, residual=res, row_map=row_map, out=out)
            tol = 2e-3 * ref.abs().max().item()
        # plain residual, no row map, exact tile multiples
        a2 = a[:1024, :256].contiguous(); w2 = w[:512, :256].contiguous(); r2 = res[:1000].repeat(2, 1)[:1024, :512].contiguous()
        o2 = ops.gemm(a2, w2, bias[:512].contiguous(), residual=r2, out_dtype=out_dtype)
    finally:
        _lib.lib().ink_gemm_set_variant(-1)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < tol
    ref2 = _ref_gemm(a2, w2, bias[:512], None, None, r2, None, 1024)
    assert (o2.double() - ref2).abs().max().item() < (1e-4 if out_dtype == torch.float32 else 2e-3 * ref2.abs().max().item())


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


def _ref_sam_attn(qkv, B, H, hd, S, rph, rpw, scale):
    """fp64 restatement of SA/modeling/image_encoder.py:224-240 + 325-361 on packed f16 qkv."""
    N = S * S
    x = qkv.double().reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)  # 3,B,H,N,hd
    q, k, v = x[0], x[1], x[2]
    attn = (q * scale) @ k.transpose(-2, -1)
    if rph is not None:
        idx = (torch.arange(S)[:, None] - torch.arange(S)[None, :] + (S - 1)).to(qkv.device)
        Rh, Rw = rph.double()[idx], rpw.double()[idx]          # S,S,hd
        rq = q.reshape(B, H, S, S, hd)
        rel_h = torch.einsum("bnhwc,hkc->bnhwk", rq, Rh)
        rel_w = torch.einsum("bnhwc,wkc->bnhwk", rq, Rw)
        attn = (attn.view(B, H, S, S, S, S) + rel_h[..., :, None] + rel_w[..., None, :]).view(B, H, N, N)
    o = attn.softmax(-1) @ v
    return o.permute(0, 2, 1, 3).reshape(B * N, H * hd)


@pytest.mark.parametrize("S,B,H,relpos", [(14, 5, 2, True), (64, 1, 2, True), (14, 3, 1, False),
                                          (64, 2, 1, False)])
def test_flash_attn_sam(dev, S, B, H, relpos):
    from inklayer_amd import ops
    hd = 80
    g = torch.Generator(device="cpu").manual_seed(S + B)
    qkv = (torch.randn(B * S * S, 3 * H * hd, generator=g) * 1.5).half().to(dev)
    rph = (torch.randn(2 * S - 1, hd, generator=g) * 0.3).to(dev) if relpos else None
    rpw = (torch.randn(2 * S - 1, hd, generator=g) * 0.3).to(dev) if relpos else None
    scale = hd ** -0.5
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
    kw = {}
    if relpos:
        r = ops.relpos_bias(q, rph, rpw, S=S, n_batch=B, n_heads=H, head_dim=hd, scale=scale)
        if S == 64:
            kw = dict(rel_h=r[0], rel_w=r[1], grid_w=64)
        else:
            kw = dict(rel_aug=r, grid_w=S)
    out = ops.flash_attn(q, k, v, n_batch=B, n_heads=H, head_dim=hd, scale=scale, **kw)
    ref = _ref_sam_attn(qkv, B, H, hd, S, rph, rpw, scale)
    err = (out.double() - ref).abs().max().item()
    assert torch.isfinite(out).all()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), err
    if relpos and S == 64:
        # the same with f16 rel-pos tables (round 3: what the engine uses at SAM's own grid)
        r16 = ops.relpos_bias(q, rph, rpw, S=S, n_batch=B, n_heads=H, head_dim=hd, scale=scale, f16_tables=True)
        assert r16[0].dtype == torch.float16
        assert (r16[0].float() - r[0]).abs().max().item() <= 2 ** -10 * r[0].abs().max().item()
        out16 = ops.flash_attn(q, k, v, n_batch=B, n_heads=H, head_dim=hd, scale=scale, rel_h=r16[0], rel_w=r16[1], grid_w=64)
        err16 = (out16.double() - ref).abs().max().item()
        print(f"global attention vs float64: f32 tables {err:.2e}, f16 tables {err16:.2e}")
        assert torch.isfinite(out16).all() and err16 < 4e-3 * max(1.0, ref.abs().max().item()), err16


@pytest.mark.parametrize("plant", [(70,), (150,), (194,), (70, 150, 194), (3, 194)])
def test_window_attention_deferred_max_rescale(dev, plant):
    """The window kernel defers the running max (rescale only when a row's max grows by more than 2^12 between key
    tiles).  Planted keys whose scores exceed everything before them by far more than that, in the second / third /
    tail tile and in sequence, drive the rare rescale branch of both subtiles; against float64."""
    from inklayer_amd import ops
    S, B, H, hd = 14, 3, 2, 80
    g = torch.Generator(device="cpu").manual_seed(sum(plant))
    qkv = torch.randn(B * S * S, 3 * H * hd, generator=g) * 0.7
    qkv[:, :H * hd] += 0.5                                  # q . ones = 40 + noise
    x = qkv.view(B, S * S, 3, H, hd)
    for i, key in enumerate(plant):
        x[:, key, 1] = 1.5 * (i + 1)                        # k = 1.5, 3, 4.5 x ones: scores +60, +120, +180 (x scale)
    qkv = qkv.half().to(dev)
    rph = (torch.randn(2 * S - 1, hd, generator=g) * 0.3).to(dev)
    rpw = (torch.randn(2 * S - 1, hd, generator=g) * 0.3).to(dev)
    scale = hd ** -0.5
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
    r = ops.relpos_bias(q, rph, rpw, S=S, n_batch=B, n_heads=H, head_dim=hd, scale=scale)
    out = ops.flash_attn(q, k, v, n_batch=B, n_heads=H, head_dim=hd, scale=scale, rel_aug=r, grid_w=S)
    ref = _ref_sam_attn(qkv, B, H, hd, S, rph, rpw, scale)
    assert torch.isfinite(out).all()
    err = (out.double() - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("plant", [(70,), (1500,), (4090,), (70, 1500, 4090), (3, 130)])
def test_global_attention_deferred_max_rescale(dev, plant):
    """Same as the window test for the global kernel (64 x 64 tokens, 64 key tiles): planted keys far above everything
    before them, in the second tile, in the middle, in the last tile and in sequence."""
    from inklayer_amd import ops
    S, B, H, hd = 64, 1, 2, 80
    g = torch.Generator(device="cpu").manual_seed(sum(plant))
    qkv = torch.randn(B * S * S, 3 * H * hd, generator=g) * 0.7
    qkv[:, :H * hd] += 0.5
    x = qkv.view(B, S * S, 3, H, hd)
    for i, key in enumerate(plant):
        x[:, key, 1] = 1.5 * (i + 1)
    qkv = qkv.half().to(dev)
    rph = (torch.randn(2 * S - 1, hd, generator=g) * 0.3).to(dev)
    rpw = (torch.randn(2 * S - 1, hd, generator=g) * 0.3).to(dev)
    scale = hd ** -0.5
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
    rh, rw = ops.relpos_bias(q, rph, rpw, S=S, n_batch=B, n_heads=H, head_dim=hd, scale=scale)
    out = ops.flash_attn(q, k, v, n_batch=B, n_heads=H, head_dim=hd, scale=scale, rel_h=rh, rel_w=rw, grid_w=S)
    ref = _ref_sam_attn(qkv, B, H, hd, S, rph, rpw, scale)
    assert torch.isfinite(out).all()
    err = (out.double() - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("nq,nk", [(512, 1024), (256, 256), (768, 4096), (320, 1024), (512, 192)])
def test_global_attention_rectangular(dev, nq, nk):
    """bias_mode 1 with given rel_h / rel_w tables on rectangular problems: n_q % 256 == 0 and n_k % 128 == 0 take the
    one-wave-per-SIMD kernel (2 .. 64 key tiles, 1 .. 3 query tiles per head), the others the streamed one."""
    from inklayer_amd import ops
    B, H, hd = 2, 2, 80
    g = torch.Generator(device="cpu").manual_seed(nq + nk)
    q = (torch.randn(B * nq, H * hd, generator=g) * 1.2).half().to(dev)
    k = (torch.randn(B * nk, H * hd, generator=g) * 1.2).half().to(dev)
    v = torch.randn(B * nk, H * hd, generator=g).half().to(dev)
    rel_h = (torch.randn(B * H, nq, 64, generator=g) * 3).to(dev)
    rel_w = (torch.randn(B * H, nq, 64, generator=g) * 3).to(dev)
    scale = hd ** -0.5
    out = ops.flash_attn(q, k, v, n_batch=B, n_heads=H, head_dim=hd, scale=scale, n_q=nq, n_k=nk,
                         rel_h=rel_h, rel_w=rel_w, grid_w=64)
    qd = q.double().view(B, nq, H, hd).transpose(1, 2)
    kd = k.double().view(B, nk, H, hd).transpose(1, 2)
    vd = v.double().view(B, nk, H, hd).transpose(1, 2)
    kk = torch.arange(nk, device=dev)
    bias = rel_h.double().view(B, H, nq, 64)[..., kk // 64] + rel_w.double().view(B, H, nq, 64)[..., kk % 64]
    ref = (((qd @ kd.transpose(-1, -2)) + bias) * scale).softmax(-1) @ vd
    ref = ref.transpose(1, 2).reshape(B * nq, H * hd)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 4e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("nq,nk", [(900, 900), (49, 49), (100, 77)])
def test_flash_attn_hd32(dev, nq, nk):
    from inklayer_amd import ops
    B, H, hd = 2, 8, 32
    g = torch.Generator(device="cpu").manual_seed(nq)
    q = torch.randn(B * nq, H * hd, generator=g).half().to(dev)
    k = torch.randn(B * nk, H * hd, generator=g).half().to(dev)
    v = torch.randn(B * nk, H * hd, generator=g).half().to(dev)
    out = ops.flash_attn(q, k, v, n_batch=B, n_heads=H, head_dim=hd, scale=hd ** -0.5)
    qd = q.double().view(B, nq, H, hd).transpose(1, 2)
    kd = k.double().view(B, nk, H, hd).transpose(1, 2)
    vd = v.double().view(B, nk, H, hd).transpose(1, 2)
    ref = ((qd @ kd.transpose(-1, -2)) * hd ** -0.5).softmax(-1) @ vd
    ref = ref.transpose(1, 2).reshape(B * nq, H * hd)
    assert (out.double() - ref).abs().max().item() < 3e-3


def test_window_attention_token_rows_equals_padded_layout(dev):
    """tok_rows folds window_partition / window_unpartition (image_encoder.py:243-289) into the attention:
    gathering q/k/v rows, substituting qkv(0) for padded keys and scattering the output must give exactly what
    the padded window layout gives (same arithmetic, same order)."""
    from inklayer_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    B, grid, S, H, hd = 2, 20, 14, 2, 80
    D, T, nwin = H * hd, grid * grid, 2
    Mw = nwin * nwin * S * S
    r = torch.arange(B * Mw)
    b, rr = r // Mw, r % Mw
    win, pos = rr // (S * S), rr % (S * S)
    y, x = (win // nwin) * S + pos // S, (win % nwin) * S + pos % S
    m = torch.where((y < grid) & (x < grid), b * T + y * grid + x, torch.full_like(r, -1))
    assert int((m >= 0).sum()) == B * T
    wm = m.to(torch.int32).to(dev)
    qkv_tok = (torch.randn(B * T, 3 * D, generator=g) * 0.5).half().to(dev)
    pad_k = (torch.randn(D, generator=g) * 0.5).half().to(dev)
    pad_v = (torch.randn(D, generator=g) * 0.5).half().to(dev)
    rel_h = (torch.randn(2 * S - 1, hd, generator=g) * 0.2).to(dev)
    rel_w = (torch.randn(2 * S - 1, hd, generator=g) * 0.2).to(dev)
    scale = hd ** -0.5
    # padded layout: padded rows are [q = 0, pad_k, pad_v]
    pad_row = torch.cat([torch.zeros(D, device=dev, dtype=torch.float16), pad_k, pad_v])
    qkv_win = torch.where((wm >= 0)[:, None], qkv_tok[wm.clamp(min=0).long()], pad_row[None, :]).contiguous()
    nb = B * nwin * nwin
    kw = dict(n_batch=nb, n_heads=H, head_dim=hd, scale=scale)
    aug_ref = ops.relpos_bias(qkv_win[:, :D], rel_h, rel_w, S=S, **kw)
    o_ref = ops.flash_attn(qkv_win[:, :D], qkv_win[:, D:2 * D], qkv_win[:, 2 * D:], rel_aug=aug_ref, grid_w=S, **kw)
    aug = ops.relpos_bias(qkv_tok[:, :D], rel_h, rel_w, S=S, tok_rows=wm, **kw,
                          out=torch.zeros(nb * H * S * S, 32, device=dev, dtype=torch.float16))
    out = torch.full((B * T, D), float("nan"), device=dev, dtype=torch.float16)
    ops.flash_attn(qkv_tok[:, :D], qkv_tok[:, D:2 * D], qkv_tok[:, 2 * D:], n_q=S * S, n_k=S * S, rel_aug=aug,
                   grid_w=S, tok_rows=wm, pad_k=pad_k, pad_v=pad_v, out=out, **kw)
    assert torch.isfinite(out).all()                       # every token row written exactly once
    keep = wm >= 0
    assert torch.equal(out[wm[keep].long()], o_ref[keep])
    # the rel-pos rows of real queries agree too
    a4 = aug.view(nb, H, S * S, 32); r4 = aug_ref.view(nb, H, S * S, 32)
    k4 = keep.view(nb, 1, S * S, 1).expand_as(a4)
    assert torch.equal(a4[k4], r4[k4])


@pytest.mark.parametrize("n_heads,n_k,shared", [(8, 4096, False), (8, 4096, True), (8, 1000, False), (2, 300, False)])
def test_attn_fewq_forms(dev, n_heads, n_k, shared):
    """SAM decoder token->image attention (7 queries x many keys, head_dim 16): the LDS-tiled kernel (n_heads % 4
    == 0) and the generic one (n_heads = 2), per-entry keys or keys shared through kv_batch_rows, ragged key counts."""
    import math
    from inklayer_amd import ops
    g = torch.Generator(device="cpu").manual_seed(n_k + n_heads)
    n, nq, hd = 12, 7, 16
    nkv = 3 if shared else n
    q = (torch.randn(n * nq, n_heads * hd, generator=g) * 0.5).half().to(dev)
    k = (torch.randn(nkv * n_k, n_heads * hd, generator=g) * 0.5).half().to(dev)
    v = (torch.randn(nkv * n_k, n_heads * hd, generator=g) * 0.5).half().to(dev)
    owner = torch.arange(n, device=dev) // 4 if shared else torch.arange(n, device=dev)
    rows = (owner * n_k).to(torch.int32) if shared else None
    out = ops.attn_fewq(q, k, v, n_batch=n, n_heads=n_heads, head_dim=hd, scale=1 / math.sqrt(hd), n_q=nq, n_k=n_k,
                        kv_batch_rows=rows)
    qf = q.double().view(n, nq, n_heads, hd).transpose(1, 2)
    kf = k.double().view(nkv, n_k, n_heads, hd).transpose(1, 2)[owner]
    vf = v.double().view(nkv, n_k, n_heads, hd).transpose(1, 2)[owner]
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) / math.sqrt(hd), -1) @ vf).transpose(1, 2).reshape(n * nq, n_heads * hd)
    assert (out.double() - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-5


@torch.no_grad()
@pytest.mark.parametrize("M", [1000, 24576 + 100])
def test_gemm_split_stream_and_layernorm_fold(dev, M):
    """ABI-4 forms of ink_gemm_f16 (the SAM ViT-H block on the split-f16 residual stream), on the 128x128 tile (M = 1000)
    and on the specialised ping-pong kernels (M = 24676: N % 320 == 0 and >= 384 tiles, with a ragged last tile):
      * split residual in, split C + per-chunk row statistics out (proj / lin2);
      * LayerNorm folded into the projection (qkv; lin1 + GELU), fed by those statistics."""
    from inklayer_amd import ops
    g = torch.Generator().manual_seed(5)
    D = 1280
    x = (torch.randn(M, D, generator=g) * 1.5 + 0.7 * torch.randn(M, 1, generator=g)).to(dev)
    chunk = ops.gemm_stats_chunk(M, D, D)
    assert chunk == (80 if M > 20000 else 64)
    hi, lo = torch.empty(M, D, device=dev, dtype=torch.float16), torch.empty(M, D, device=dev, dtype=torch.float16)
    st = torch.empty(M, D // chunk, 2, device=dev)
    ops.hilo_split_stats(x, hi, lo, st, chunk)
    xd = x.double()
    rec = hi.double() + lo.double()
    assert bool(((rec - xd).abs() <= 2e-6 * xd.abs() + 4e-8).all())        # ~22 significant bits (f16 subnormal floor for the lo part)
    assert torch.equal(ops.hilo_join(hi, lo), (hi.float() + lo.float()))
    parts = xd.view(M, D // chunk, chunk)
    assert (st[..., 0].double() - parts.sum(-1)).abs().max().item() < 1e-3
    assert ((st[..., 1].double() - (parts ** 2).sum(-1)).abs() / (parts ** 2).sum(-1)).max().item() < 1e-5
    # ---- folded LayerNorm: qkv form (f16 out) and lin1 form (GELU)
    gam = (1.0 + 0.1 * torch.randn(D, generator=g)).to(dev)
    bet = (0.1 * torch.randn(D, generator=g)).to(dev)
    for N, act in ((3840, None), (2560, "gelu")):
        W = (torch.randn(N, D, generator=g) / D ** 0.5).to(dev)
        b = (0.1 * torch.randn(N, generator=g)).to(dev)
        wl = (W * gam[None]).half().contiguous()
        colsum = wl.double().sum(1).float().contiguous()
        bias_ln = (W.double() @ bet.double() + b.double()).float().contiguous()
        out = ops.gemm(hi, wl, bias_ln, act=act, out_dtype=torch.float16, ln=(st, D, 1e-6, colsum))
        mean = xd.mean(1, keepdim=True)
        rstd = 1.0 / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-6)
        want = rstd * ((hi.double() - mean) @ wl.double().t()) + bias_ln.double()       # the fold's own arithmetic, in f64
        if act == "gelu":
            want = torch.nn.functional.gelu(want)
        err = (out.double() - want).abs().max().item() / want.abs().max().item()
        # ... and against LayerNorm -> Linear on the unrounded f32 operands (what the reference computes)
        ref = torch.nn.functional.layer_norm(xd, (D,), gam.double(), bet.double(), 1e-6) @ W.double().t() + b.double()
        if act == "gelu":
            ref = torch.nn.functional.gelu(ref)
        l2 = ((out.double() - ref).norm() / ref.norm()).item()
        print(f"M={M} N={N} act={act}: vs fold-in-f64 max-rel {err:.2e} (f16 output rounding), vs fp32 LN+Linear l2-rel {l2:.2e}")
        assert err < 1.5e-3 and l2 < 1.5e-3
    # ---- projection on the split stream: residual in, C + statistics out, in place
    Wp = (torch.randn(D, D, generator=g) / D ** 0.5).to(dev).half()
    bp = (0.1 * torch.randn(D, generator=g)).to(dev)
    a = torch.randn(M, D, generator=g).to(dev).half()
    want = rec + a.double() @ Wp.double().t() + bp.double()
    st2 = torch.zeros_like(st)
    ops.gemm(a, Wp, bp, residual_hilo=(hi, lo), out_hilo=(hi, lo), stats_out=st2)
    got = hi.double() + lo.double()
    assert ((got - want).abs().max() / want.abs().max()).item() < 5e-6
    wp = want.view(M, D // chunk, chunk)
    assert (st2[..., 0].double() - wp.sum(-1)).abs().max().item() < 2e-3
    assert ((st2[..., 1].double() - (wp ** 2).sum(-1)).abs() / (wp ** 2).sum(-1)).max().item() < 2e-5


@torch.no_grad()
@pytest.mark.parametrize("M,hid", [(1000, 2048), (128, 2048), (37, 128), (13294 + 5, 1024)])
def test_fused_ffn_equals_the_two_projection_form(dev, M, hid):
    """ink_ffn256_fused = LayerNorm(res + linear2(relu(linear1(x)))) (transformer.py:780-799) against float64 with the
    same two rounding points as the two-GEMM form (f16 operand x, f16 hidden activations) and against that form itself
    (ops.gemm x 2 + layernorm_rows).  Ragged M: the last workgroup's rows are clamped / masked."""
    from inklayer_amd import ops
    g = torch.Generator().manual_seed(M + hid)
    x32 = torch.randn(M, 256, generator=g) * 1.5 + 0.3
    w1 = (torch.randn(hid, 256, generator=g) / 16).half()
    w2 = (torch.randn(256, hid, generator=g) / 45).half()
    b1, b2 = torch.randn(hid, generator=g) * 0.2, torch.randn(256, generator=g) * 0.2
    lg, lb = 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    x16 = x32.half()
    h = torch.relu(x16.double() @ w1.double().T + b1.double()).float().half()
    y = x32.double() + h.double() @ w2.double().T + b2.double()
    want = torch.nn.functional.layer_norm(y, (256,), lg.double(), lb.double(), 1e-5)
    d = lambda t: t.to(dev).contiguous()
    blob = ops.ffn256_pack(d(w1), d(b1), d(w2))
    xs = d(x32)
    got = ops.ffn256_fused(d(x16), xs, blob, hid, d(b2), d(lg), d(lb), 1e-5)
    ea = (got.double().cpu() - want).abs()
    err, mean_err = ea.max().item(), ea.mean().item()
    print(f"M={M} hid={hid}: fused FFN vs float64 max abs {err:.2e}, mean abs {mean_err:.2e}")
    # the f16 rounding of a hidden activation near a tie can go the other way than in float64 (f32 accumulation): one
    # f16 ulp of one hidden unit moves its row by up to ~3e-4; everything else is f32 accumulation noise
    assert torch.isfinite(got).all() and err < 1e-3 and mean_err < 5e-6
    ff = ops.gemm(d(x16), d(w1), d(b1), act="relu", out_dtype=torch.float16)
    y2 = ops.gemm(ff, d(w2), d(b2), residual=xs.clone())
    two = ops.layernorm_rows(y2, d(lg), d(lb), 1e-5, out_dtype=torch.float32)
    assert (got - two).abs().max().item() < 1e-3 and (got - two).abs().mean().item() < 5e-6
    # in place: out aliases res
    got2 = ops.ffn256_fused(d(x16), xs, blob, hid, d(b2), d(lg), d(lb), 1e-5, out=xs)
    assert got2.data_ptr() == xs.data_ptr() and torch.equal(got2, got)


@torch.no_grad()
@pytest.mark.parametrize("M", [1000, 13294 + 5])
def test_fused_ffn_with_the_preceding_projection(dev, M):
    """ops.ffn256_fused(pre=...): s = LN1(res + x Wpre^T + b), out = LN2(s + linear2(relu(linear1(f16(s))))) - the tail
    of DeformableTransformerEncoderLayer.forward (transformer.py:780-799) from the attention's output projection on -
    against float64 with the same rounding points and against the three-launch form (GEMM + LayerNorm + fused FFN)."""
    from inklayer_amd import ops
    hid = 2048
    g = torch.Generator().manual_seed(M)
    x16 = (torch.randn(M, 256, generator=g) * 0.8).half()
    res = torch.randn(M, 256, generator=g) * 1.5 + 0.3
    wp = (torch.randn(256, 256, generator=g) / 16).half()
    w1 = (torch.randn(hid, 256, generator=g) / 16).half()
    w2 = (torch.randn(256, hid, generator=g) / 45).half()
    bp, b1, b2 = torch.randn(256, generator=g) * 0.2, torch.randn(hid, generator=g) * 0.2, torch.randn(256, generator=g) * 0.2
    g1, e1 = 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    g2, e2 = 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    ln = torch.nn.functional.layer_norm
    s_ = ln(res.double() + x16.double() @ wp.double().T + bp.double(), (256,), g1.double(), e1.double(), 1e-5)
    s16 = s_.float().half()
    h = torch.relu(s16.double() @ w1.double().T + b1.double()).float().half()
    want = ln(s_ + h.double() @ w2.double().T + b2.double(), (256,), g2.double(), e2.double(), 1e-5)
    d = lambda t: t.to(dev).contiguous()
    blob = ops.ffn256_pack(d(w1), d(b1), d(w2), d(wp))
    got = ops.ffn256_fused(d(x16), d(res), blob, hid, d(b2), d(g2), d(e2), 1e-5, pre=(d(bp), d(g1), d(e1)))
    ea = (got.double().cpu() - want).abs()
    print(f"M={M}: out_proj + norm1 + FFN + norm2 fused vs float64 max abs {ea.max().item():.2e}, mean abs {ea.mean().item():.2e}")
    assert torch.isfinite(got).all() and ea.max().item() < 2e-3 and ea.mean().item() < 1e-5
    y = ops.gemm(d(x16), d(wp), d(bp), residual=d(res))
    sx = torch.empty_like(y)
    s16d = torch.empty(y.shape, device=dev, dtype=torch.float16)
    ops.layernorm_rows(y, d(g1), d(e1), 1e-5, out=sx, out2=s16d)
    three = ops.ffn256_fused(s16d, sx, ops.ffn256_pack(d(w1), d(b1), d(w2)), hid, d(b2), d(g2), d(e2), 1e-5)
    e3 = (got - three).abs()
    assert e3.max().item() < 2e-3 and e3.mean().item() < 1e-5
