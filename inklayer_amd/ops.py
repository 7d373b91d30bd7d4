"""Thin torch-tensor wrappers over the C ABI (PyTorch = device memory + streams only).

Every function enqueues hand-written HIP kernels on torch's current stream.
Tensors must live on the GPU; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import InkAttn, InkGemm, check

ACT = {None: 0, "none": 0, "gelu": 1, "relu": 2}
F16, F32 = torch.float16, torch.float32


_GEMM_TRACE = None


def set_gemm_trace(lst) -> None:
    """bench.py instrumentation: when `lst` is a list, every gemm() launch is bracketed by two HIP events
    on the launch stream and (flops, start, end) is appended; None switches it off."""
    global _GEMM_TRACE
    _GEMM_TRACE = lst


_ATTN_TRACE = None


def set_attn_trace(lst) -> None:
    """bench.py instrumentation: when `lst` is a list, every flash_attn() launch is bracketed by two HIP events on
    the launch stream and (n_batch, n_heads, n_q, n_k, head_dim, bias_mode, rows_q, start, end) is appended."""
    global _ATTN_TRACE
    _ATTN_TRACE = lst


def tracing_off() -> bool:
    """No per-launch event bracketing requested (a captured graph would bypass it)."""
    return _GEMM_TRACE is None and _ATTN_TRACE is None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """Raw handle of torch's current HIP stream.  Called once per kernel launch: the public
    `torch.cuda.current_stream()` builds a Stream object (~8 us); the raw accessor is ~0.3 us."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def own_f32(t: torch.Tensor, dev) -> torch.Tensor:
    """An engine-owned contiguous f32 copy of a parameter on `dev`.  A tensor that is only a VIEW into a larger
    storage (e.g. a slice of dist.broadcast_state_dict's 1 GiB flat buckets) is cloned, so that the engines do not
    keep those buckets alive through a few kilobytes of biases and norm weights."""
    t = t.detach().to(dev, F32).contiguous()
    if t.untyped_storage().nbytes() > t.numel() * t.element_size():
        t = t.clone()
    return t


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda, "InkLayer HIP ops need GPU tensors (no CPU fallback)"
    return t.data_ptr()


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
         act: Optional[str] = None, residual: Optional[torch.Tensor] = None,
         col_scale: Optional[torch.Tensor] = None, row_map: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None, out_dtype: torch.dtype = F32,
         out_rows: Optional[int] = None, residual_hilo=None, out_hilo=None,
         stats_out: Optional[torch.Tensor] = None, ln=None) -> torch.Tensor:
    """out[row_map[m]] = residual[row_map[m]] + col_scale * act(a[m] @ w.T + bias).

    a: f16 [M, K] (row stride arbitrary, multiple of 8), w: f16 [N, K].
    Split-f16 stream forms (SAM ViT-H blocks): residual_hilo = (hi, lo) f16 planes instead of `residual`;
    out_hilo = (hi, lo) f16 planes instead of `out` (returns hi); stats_out f32 [rows, N / chunk, 2] receives the
    per-chunk (sum, sum of squares) of every output row (chunk = gemm_stats_chunk(M, N, K));
    ln = (stats [M, parts, 2], dim, eps, colsum [N]): LayerNorm over the `dim` source columns of a's rows folded in
    (w carries gamma, bias carries beta w^T + b)."""
    assert a.dtype == F16 and w.dtype == F16
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    p = InkGemm()
    if out_hilo is not None:
        out, out_lo = out_hilo
        assert out.dtype == F16 and out_lo.dtype == F16 and out_lo.shape == out.shape and out_lo.stride() == out.stride()
        p.C_lo = out_lo.data_ptr()
    if out is None:
        rows = out_rows if out_rows is not None else M
        out = torch.empty((rows, N), device=a.device, dtype=out_dtype)
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape[1] == N
    p.A, p.W, p.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    p.bias, p.col_scale = _p(bias), _p(col_scale)
    p.residual, p.row_map = _p(residual), _p(row_map)
    if bias is not None:
        assert bias.dtype == F32 and bias.numel() == N
    if col_scale is not None:
        assert col_scale.dtype == F32 and col_scale.numel() == N
    if residual is not None:
        assert residual.dtype == F32 and residual.stride(1) == 1 and residual.shape[1] == N
        p.ldr = residual.stride(0)
    if row_map is not None:
        assert row_map.dtype == torch.int32 and row_map.numel() == M
    if residual_hilo is not None:
        rh, rl = residual_hilo
        assert residual is None and rh.dtype == F16 and rl.dtype == F16 and rh.stride(1) == 1 and rh.shape[1] == N
        assert rl.shape == rh.shape and rl.stride() == rh.stride()
        p.res_hi, p.res_lo, p.ldr = rh.data_ptr(), rl.data_ptr(), rh.stride(0)
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldw, p.ldc = a.stride(0), w.stride(0), out.stride(0)
    p.act = ACT[act]
    p.c_f16 = 2 if out_hilo is not None else (1 if out.dtype == F16 else 0)
    assert out.dtype in (F16, F32)
    if stats_out is not None:
        chunk = gemm_stats_chunk(M, N, K)
        assert out_hilo is not None and N % chunk == 0 and stats_out.dtype == F32 and stats_out.is_contiguous()
        assert tuple(stats_out.shape) == (out.shape[0], N // chunk, 2)
        p.stats_out, p.stats_parts = stats_out.data_ptr(), N // chunk
    if ln is not None:
        st, dim, eps, colsum = ln
        assert st.dtype == F32 and st.is_contiguous() and st.dim() == 3 and st.shape[0] == M and st.shape[2] == 2
        assert colsum.dtype == F32 and colsum.numel() == N and row_map is None
        p.ln_stats, p.ln_parts, p.ln_dim, p.ln_eps, p.ln_colsum = st.data_ptr(), st.shape[1], int(dim), float(eps), colsum.data_ptr()
    if _GEMM_TRACE is None:
        check(_lib.lib().ink_gemm_f16(C.byref(p), _stream()), "ink_gemm_f16")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.lib().ink_gemm_f16(C.byref(p), _stream()), "ink_gemm_f16")
        e1.record()
        _GEMM_TRACE.append((2.0 * M * N * K, e0, e1, (M, N, K, act or '-', 'res' if residual is not None else '-',
                                                      'map' if row_map is not None else '-', 'f16' if p.c_f16 else 'f32')))
    return out


def gemm_stats_chunk(M: int, N: int, K: int) -> int:
    """Columns per row-statistics chunk of the kernel ink_gemm_f16 picks for this (split-output) shape."""
    return int(_lib.lib().ink_gemm_query_stats_chunk(M, N, K))


def hilo_split_stats(x: torch.Tensor, hi: torch.Tensor, lo: torch.Tensor, stats: torch.Tensor, chunk: int) -> None:
    """f32 rows -> the split-f16 stream (hi = f16(x), lo = f16(x - hi)) + per-chunk (sum, sum of squares) [rows, C/chunk, 2]."""
    assert x.dtype == F32 and x.dim() == 2 and x.stride(1) == 1 and hi.dtype == F16 and lo.dtype == F16
    rows, Cdim = x.shape
    assert hi.shape == x.shape and lo.shape == x.shape and hi.stride() == lo.stride() and hi.stride(1) == 1
    assert stats.dtype == F32 and stats.is_contiguous() and tuple(stats.shape) == (rows, Cdim // chunk, 2)
    check(_lib.lib().ink_hilo_split_stats(x.data_ptr(), x.stride(0), rows, Cdim, hi.data_ptr(), lo.data_ptr(), hi.stride(0),
                                          stats.data_ptr(), chunk, _stream()), "ink_hilo_split_stats")


def hilo_join(hi: torch.Tensor, lo: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f32 (hi + lo) of two contiguous f16 planes."""
    assert hi.dtype == F16 and lo.dtype == F16 and hi.is_contiguous() and lo.is_contiguous() and hi.shape == lo.shape
    if out is None:
        out = torch.empty(hi.shape, device=hi.device, dtype=F32)
    assert out.dtype == F32 and out.is_contiguous() and out.shape == hi.shape
    check(_lib.lib().ink_hilo_join(hi.data_ptr(), lo.data_ptr(), hi.numel(), out.data_ptr(), _stream()), "ink_hilo_join")
    return out


def layernorm_rows(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor],
                   eps: float, *, gather: Optional[torch.Tensor] = None,
                   out_dtype: torch.dtype = F16, out: Optional[torch.Tensor] = None,
                   out2: Optional[torch.Tensor] = None, act: Optional[str] = None,
                   split: bool = False, add: Optional[torch.Tensor] = None,
                   add_batch_rows: Optional[torch.Tensor] = None, rows_per_batch: int = 0,
                   split_f32: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm over the last dim of f32 x [R, C]; optional row gather (-1 -> zero row).
    `out2` (the other of f16/f32, same shape/stride) receives a second copy in the same pass.
    split=True: the f16 output is a split-f16 GEMM operand [R, 3C] (see add_split_f16); split_f32 (contiguous f32
    [R, C]) then receives the f32 result in the same pass."""
    assert x.dtype == F32 and x.dim() == 2 and x.stride(1) == 1
    Cdim = x.shape[1]
    rows = gather.numel() if gather is not None else x.shape[0]
    wcols = 3 * Cdim if split else Cdim
    if split:
        assert out_dtype == F16 and out2 is None
    if out is None:
        out = torch.empty((rows, wcols), device=x.device, dtype=out_dtype)
    assert out.shape == (rows, wcols) and out.stride(1) == 1
    oh = out.data_ptr() if out.dtype == F16 else None
    of = out.data_ptr() if out.dtype == F32 else None
    if out2 is not None:
        assert out2.shape == out.shape and out2.stride(0) == out.stride(0) and out2.dtype != out.dtype
        if out2.dtype == F16:
            oh = out2.data_ptr()
        else:
            of = out2.data_ptr()
    if split_f32 is not None:
        assert split and out.stride(0) == 3 * Cdim and split_f32.dtype == F32 and split_f32.is_contiguous()
        assert tuple(split_f32.shape) == (rows, Cdim)
        of = split_f32.data_ptr()
    if gather is not None:
        assert gather.dtype == torch.int32
    if add is not None:
        # LN(x[r] + add[add_batch_rows[r // rows_per_batch] + r % rows_per_batch]) (add_batch_rows None: add[r])
        assert gather is None and add.dtype == F32 and add.dim() == 2 and add.stride(1) == 1 and add.shape[1] == Cdim
        if add_batch_rows is not None:
            assert add_batch_rows.dtype == torch.int32 and add_batch_rows.is_cuda and rows_per_batch > 0
            assert add_batch_rows.numel() * rows_per_batch == rows
        else:
            rows_per_batch = rows
    check(_lib.lib().ink_layernorm_rows(x.data_ptr(), x.stride(0), _p(gamma), _p(beta), eps,
                                        _p(gather), rows, Cdim, oh, of, out.stride(0), ACT[act], int(split),
                                        _p(add), add.stride(0) if add is not None else 0, _p(add_batch_rows),
                                        rows_per_batch, _stream()), "ink_layernorm_rows")
    return out


def add_cvt_f16(a: torch.Tensor, b: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f16(a + b) for contiguous f32 tensors; b may be a (leading-dim) broadcast of a."""
    assert a.dtype == F32 and a.is_contiguous()
    nb = 0
    if b is not None:
        assert b.dtype == F32 and b.is_contiguous() and a.numel() % b.numel() == 0
        nb = b.numel()
    if out is None:
        out = torch.empty(a.shape, device=a.device, dtype=F16)
    assert out.dtype == F16 and out.is_contiguous() and out.numel() == a.numel()
    check(_lib.lib().ink_add_cvt_f16(a.data_ptr(), _p(b), nb, out.data_ptr(), a.numel(), _stream()),
          "ink_add_cvt_f16")
    return out


def add_split_f16(a: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Split-f16 GEMM operand of v = a + b (contiguous f32 [R, C]; b a leading-dim broadcast): f16 [R, 3C] =
    [hi | (v - hi) * 64 | hi / 64].  gemm() against `split_weight(W)` then yields v @ W.T at fp32-grade accuracy."""
    assert a.dtype == F32 and a.is_contiguous() and a.dim() == 2
    nb = 0
    if b is not None:
        assert b.dtype == F32 and b.is_contiguous() and a.numel() % b.numel() == 0
        nb = b.numel()
    R, Cn = a.shape
    out = torch.empty((R, 3 * Cn), device=a.device, dtype=F16)
    check(_lib.lib().ink_add_split_f16(a.data_ptr(), _p(b), nb, out.data_ptr(), a.numel(), Cn, _stream()),
          "ink_add_split_f16")
    return out


def split_weight(w32: torch.Tensor) -> torch.Tensor:
    """f32 [..., K] -> f16 [..., 3K] = [W_hi | W_hi / 64 | (W - W_hi) * 64]: the weight side of a split-f16 GEMM
    (load-time re-layout, like the f16 cast of the other matrices)."""
    w32 = w32.to(torch.float32)
    hi = w32.to(F16)
    lo = ((w32 - hi.to(torch.float32)) * 64.0).to(F16)
    return torch.cat([hi, (hi.to(torch.float32) / 64.0).to(F16), lo], dim=-1).contiguous()


def add_f32(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a + b (f32), b broadcast along the leading dims of a (periodic in flat index)."""
    assert a.dtype == F32 and b.dtype == F32 and a.is_contiguous() and b.is_contiguous()
    assert a.numel() % b.numel() == 0
    if out is None:
        out = torch.empty(a.shape, device=a.device, dtype=F32)
    assert out.is_contiguous() and out.numel() == a.numel()
    check(_lib.lib().ink_add_f32(a.data_ptr(), b.data_ptr(), b.numel(), out.data_ptr(), a.numel(),
                                 _stream()), "ink_add_f32")
    return out


def flash_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, n_batch: int, n_heads: int,
               head_dim: int, scale: float, n_q: Optional[int] = None, n_k: Optional[int] = None,
               rel_h: Optional[torch.Tensor] = None, rel_w: Optional[torch.Tensor] = None,
               rel_aug: Optional[torch.Tensor] = None, grid_w: int = 0,
               dense_bias: Optional[torch.Tensor] = None, dense_mask: Optional[torch.Tensor] = None,
               q_batch_rows: Optional[torch.Tensor] = None,
               kv_batch_rows: Optional[torch.Tensor] = None,
               tok_rows: Optional[torch.Tensor] = None, pad_k: Optional[torch.Tensor] = None,
               pad_v: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(scale*q@k^T + bias)@v for f16 row views q [.., >=H*hd], k/v [.., >=H*hd].

    q/k/v may be column slices of one packed qkv buffer (only the row stride matters).
    Batch entry b uses rows q_batch_rows[b] + [0, n_q) of q and kv_batch_rows[b] + [0, n_k) of
    k/v (defaults b*n_q, b*n_k); the output is dense [n_batch*n_q, H*hd].
    With tok_rows (int32 [n_batch, n_q], rel_aug mode) token i of window b is row tok_rows[b, i] of
    q, k, v and `out` (which is then required); -1 marks window padding, whose keys are pad_k / pad_v.
    """
    for t in (q, k, v):
        assert t.dtype == F16 and t.dim() == 2 and t.stride(1) == 1 and t.is_cuda
    if n_q is None:
        n_q = q.shape[0] // n_batch
    if n_k is None:
        n_k = k.shape[0] // n_batch
    if tok_rows is not None:
        assert rel_aug is not None and out is not None and n_q == n_k
        assert tok_rows.dtype == torch.int32 and tok_rows.is_cuda and tok_rows.numel() == n_batch * n_q
        for t in (pad_k, pad_v):
            assert t is not None and t.dtype == F16 and t.is_contiguous() and t.numel() == n_heads * head_dim
        assert out.dtype == F16 and out.stride(1) == 1
        # (the window kernel addresses O, and the gathered q / k / v rows, through 32-bit byte offsets)
        assert out.shape[0] * out.stride(0) * 2 < 2 ** 31 and q.shape[0] * q.stride(0) * 2 < 2 ** 32
    else:
        if out is None:
            out = torch.empty((n_batch * n_q, n_heads * head_dim), device=q.device, dtype=F16)
        assert out.dtype == F16 and out.shape[0] == n_batch * n_q and out.stride(1) == 1
    p = InkAttn()
    p.Q, p.K, p.V, p.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    p.ldq, p.ldk, p.ldv, p.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    p.n_batch, p.n_heads, p.n_q, p.n_k, p.head_dim = n_batch, n_heads, n_q, n_k, head_dim
    p.scale = scale
    p.grid_w = grid_w
    for name, t in (("q_batch_rows", q_batch_rows), ("kv_batch_rows", kv_batch_rows)):
        if t is not None:
            assert t.dtype == torch.int32 and t.numel() == n_batch and t.is_cuda
            setattr(p, name, t.data_ptr())
    if dense_bias is not None:
        assert dense_bias.dtype == F32 and dense_bias.is_contiguous() and dense_bias.shape == (n_heads, n_q, 64)
        p.bias_mode, p.dense_bias = 3, dense_bias.data_ptr()
        if dense_mask is not None:
            assert dense_mask.dtype == F32 and dense_mask.is_contiguous() and dense_mask.shape[1:] == (n_q, 64)
            p.dense_mask, p.n_mask = dense_mask.data_ptr(), dense_mask.shape[0]
    elif rel_aug is not None:
        assert rel_aug.dtype == F16 and rel_aug.is_contiguous()
        p.bias_mode, p.rel_aug = 2, rel_aug.data_ptr()
        if tok_rows is not None:
            p.tok_rows, p.pad_k, p.pad_v = tok_rows.data_ptr(), pad_k.data_ptr(), pad_v.data_ptr()
    elif rel_h is not None:
        assert rel_h.dtype == rel_w.dtype and rel_h.dtype in (F32, F16)      # f16 tables: relpos_bias(..., f16_tables=True)
        assert rel_h.is_contiguous() and rel_w.is_contiguous()
        p.bias_mode, p.rel_h, p.rel_w = 1, rel_h.data_ptr(), rel_w.data_ptr()
        p.rel_f16 = int(rel_h.dtype == F16)
    else:
        p.bias_mode = 0
    if _ATTN_TRACE is None:
        check(_lib.lib().ink_flash_attn(C.byref(p), _stream()), "ink_flash_attn")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.lib().ink_flash_attn(C.byref(p), _stream()), "ink_flash_attn")
        e1.record()
        _ATTN_TRACE.append((n_batch, n_heads, n_q, n_k, head_dim, int(p.bias_mode), int(out.shape[0]), e0, e1))
    return out


def relpos_bias(q: torch.Tensor, rel_pos_h: torch.Tensor, rel_pos_w: torch.Tensor, *, S: int,
                n_batch: int, n_heads: int, head_dim: int, scale: float, out=None,
                tok_rows: Optional[torch.Tensor] = None, f16_tables: bool = False):
    """SAM decomposed rel-pos terms / scale.  S == 64 -> (rel_h, rel_w) f32 (f16 with f16_tables: SAM's own attention shape
    only); S <= 16 -> rel_aug f16."""
    assert q.dtype == F16 and q.stride(1) == 1
    assert rel_pos_h.dtype == F32 and rel_pos_h.is_contiguous()
    assert rel_pos_w.dtype == F32 and rel_pos_w.is_contiguous()
    assert rel_pos_h.shape == (2 * S - 1, head_dim)
    n = n_batch * n_heads * S * S
    fn = _lib.lib().ink_relpos_bias
    if S == 64 and f16_tables:
        oh, ow = out if out is not None else (torch.empty((n, 64), device=q.device, dtype=F16),
                                              torch.empty((n, 64), device=q.device, dtype=F16))
        assert oh.dtype == F16 and ow.dtype == F16 and oh.numel() >= n * 64 and ow.numel() >= n * 64
        check(_lib.lib().ink_relpos_bias64_f16(q.data_ptr(), q.stride(0), rel_pos_h.data_ptr(), rel_pos_w.data_ptr(), n_batch,
                                               n_heads, head_dim, scale, oh.data_ptr(), ow.data_ptr(), _stream()),
              "ink_relpos_bias64_f16")
        return oh, ow
    if S == 64:
        oh, ow = out if out is not None else (torch.empty((n, 64), device=q.device, dtype=F32),
                                              torch.empty((n, 64), device=q.device, dtype=F32))
        assert oh.numel() >= n * 64 and ow.numel() >= n * 64
        check(fn(q.data_ptr(), q.stride(0), rel_pos_h.data_ptr(), rel_pos_w.data_ptr(), S, n_batch,
                 n_heads, head_dim, scale, None, oh.data_ptr(), ow.data_ptr(), None, _stream()),
              "ink_relpos_bias")
        return oh, ow
    aug = out if out is not None else torch.empty((n, 32), device=q.device, dtype=F16)
    assert aug.numel() >= n * 32
    if tok_rows is not None:
        assert tok_rows.dtype == torch.int32 and tok_rows.is_cuda and tok_rows.numel() == n_batch * S * S
    check(fn(q.data_ptr(), q.stride(0), rel_pos_h.data_ptr(), rel_pos_w.data_ptr(), S, n_batch,
             n_heads, head_dim, scale, tok_rows.data_ptr() if tok_rows is not None else None,
             None, None, aug.data_ptr(), _stream()), "ink_relpos_bias")
    return aug


def resize_bilinear_u8(image_u8: torch.Tensor, oh: int, ow: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PIL `Image.resize((ow, oh), BILINEAR)` of an HWC uint8 RGB CUDA tensor, bit for bit (antialiased)."""
    from .resize import plan_for
    assert image_u8.dtype == torch.uint8 and image_u8.is_cuda and image_u8.is_contiguous()
    h, w, c = image_u8.shape
    assert c == 3
    if (oh, ow) == (h, w):
        return image_u8
    pl = plan_for(h, w, oh, ow, image_u8.device)
    if out is None:
        out = torch.empty((oh, ow, 3), device=image_u8.device, dtype=torch.uint8)
    assert out.dtype == torch.uint8 and out.is_contiguous() and tuple(out.shape) == (oh, ow, 3)
    ptr = lambda t: t.data_ptr() if t is not None else None
    check(_lib.lib().ink_resize_bilinear_u8(image_u8.data_ptr(), h, w, out.data_ptr(), oh, ow, ptr(pl.xb), ptr(pl.xk),
                                            pl.kx, ptr(pl.yb), ptr(pl.yk), pl.ky, ptr(pl.tmp), _stream()),
          "ink_resize_bilinear_u8")
    return out


def sam_patchify(image_u8: torch.Tensor, L: int, P: int, mean: Sequence[float],
                 std: Sequence[float], chan_reverse: bool, out: torch.Tensor, split: bool = False) -> torch.Tensor:
    """uint8 HWC (h,w <= L) -> normalised, zero-padded f16 im2col [ (L/P)^2, 3*P*P ] (split: [.., 3*3*P*P])."""
    assert image_u8.dtype == torch.uint8 and image_u8.is_cuda and image_u8.is_contiguous()
    h, w, c = image_u8.shape
    assert c == 3 and out.dtype == F16 and out.is_contiguous()
    assert out.numel() == (L // P) ** 2 * 3 * P * P * (3 if split else 1)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    check(_lib.lib().ink_sam_patchify(image_u8.data_ptr(), h, w, L, P, m, s, int(chan_reverse), int(split),
                                      out.data_ptr(), _stream()), "ink_sam_patchify")
    return out


def im2col3x3(x: torch.Tensor, B: int, H: int, W: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f16 NHWC [B*H*W, C] -> [B*H*W, 9*C] (pad 1)."""
    assert x.dtype == F16 and x.is_contiguous() and x.shape[0] == B * H * W
    Cn = x.shape[1]
    if out is None:
        out = torch.empty((B * H * W, 9 * Cn), device=x.device, dtype=F16)
    check(_lib.lib().ink_im2col3x3_f16(x.data_ptr(), B, H, W, Cn, out.data_ptr(), _stream()),
          "ink_im2col3x3_f16")
    return out


def sam_pe_encode(coords01: torch.Tensor, gauss: torch.Tensor,
                  add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[sin, cos](2*pi*((2c-1) @ G)) (+ add[n % n_add]) -> f32 [N, 2F]."""
    assert coords01.dtype == F32 and coords01.is_contiguous() and coords01.shape[-1] == 2
    assert gauss.dtype == F32 and gauss.is_contiguous() and gauss.shape[0] == 2
    N, Fd = coords01.numel() // 2, gauss.shape[1]
    out = torch.empty((N, 2 * Fd), device=coords01.device, dtype=F32)
    n_add = 0
    if add is not None:
        assert add.dtype == F32 and add.is_contiguous() and add.shape[-1] == 2 * Fd
        n_add = add.numel() // (2 * Fd)
    check(_lib.lib().ink_sam_pe_encode(coords01.data_ptr(), gauss.data_ptr(), N, Fd, _p(add), n_add,
                                       out.data_ptr(), _stream()), "ink_sam_pe_encode")
    return out


def sam_mask_logits(up: torch.Tensor, hyper: torch.Tensor, n: int, g: int) -> torch.Tensor:
    """hyper[n,C] . up[(((b*g*g + tok)*4 + s1)*4 + s2), C] -> pixel-shuffled [n, 4g, 4g] f32."""
    assert up.dtype == F32 and up.is_contiguous() and hyper.dtype == F32 and hyper.is_contiguous()
    Cn = hyper.shape[1]
    assert up.numel() == n * g * g * 16 * Cn
    out = torch.empty((n, 4 * g, 4 * g), device=up.device, dtype=F32)
    check(_lib.lib().ink_sam_mask_logits(up.data_ptr(), hyper.data_ptr(), n, g, Cn, out.data_ptr(),
                                         _stream()), "ink_sam_mask_logits")
    return out


def sam_postprocess(low: torch.Tensor, L: int, input_hw: Tuple[int, int], orig_hw: Tuple[int, int],
                    thr: float = 0.0, want_logits: bool = False):
    """low [n,S,S] f32 -> uint8 masks [n, H, W] (and optionally the f32 logits)."""
    assert low.dtype == F32 and low.is_contiguous() and low.dim() == 3
    n, S, _ = low.shape
    oh, ow = orig_hw
    out = torch.empty((n, oh, ow), device=low.device, dtype=torch.uint8)
    lg = torch.empty((n, oh, ow), device=low.device, dtype=F32) if want_logits else None
    check(_lib.lib().ink_sam_postprocess(low.data_ptr(), n, S, L, input_hw[0], input_hw[1], oh, ow,
                                         thr, out.data_ptr(), _p(lg), _stream()),
          "ink_sam_postprocess")
    return (out, lg) if want_logits else out


# ---------------------------------------------------------------------------------------------
# GroundingDINO-side ops
# ---------------------------------------------------------------------------------------------
def ms_deform_attn_forward(value: torch.Tensor, spatial_shapes, level_start_index,
                           sampling_loc: torch.Tensor, attn_weight: torch.Tensor,
                           im2col_step: int = 64) -> torch.Tensor:
    """Same argument list as groundingdino._C.ms_deform_attn_forward (GD/.../csrc/vision.cpp:53-56)."""
    assert value.dtype == F32 and value.is_contiguous() and value.is_cuda
    assert sampling_loc.dtype == F32 and sampling_loc.is_contiguous()
    assert attn_weight.dtype == F32 and attn_weight.is_contiguous()
    B, S, M, Cn = value.shape
    _, Q, _, L, P, _ = sampling_loc.shape
    ss = [int(v) for v in torch.as_tensor(spatial_shapes).reshape(-1).tolist()]
    ls = [int(v) for v in torch.as_tensor(level_start_index).reshape(-1).tolist()]
    out = torch.empty((B, Q, M * Cn), device=value.device, dtype=F32)
    check(_lib.lib().ink_ms_deform_attn_forward(
        value.data_ptr(), (C.c_int64 * len(ss))(*ss), (C.c_int64 * len(ls))(*ls), sampling_loc.data_ptr(),
        attn_weight.data_ptr(), B, S, M, Cn, Q, L, P, im2col_step, out.data_ptr(), _stream()),
        "ink_ms_deform_attn_forward")
    return out


def msda_fused(value16: torch.Tensor, proj: torch.Tensor, ref: torch.Tensor, shapes: Sequence[Tuple[int, int]],
               B: int, Q: int, ref_batched: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """value16 f16 [B*S, 256]; proj f32 [B*Q, >=384]; ref f32 [Q, d] (shared) or [B*Q, d], d in {2,4}."""
    assert value16.dtype == F16 and value16.is_contiguous() and proj.dtype == F32 and proj.stride(1) == 1
    assert ref.dtype == F32 and ref.is_contiguous()
    S = value16.shape[0] // B
    d = ref.shape[-1]
    if out is None:
        out = torch.empty((B * Q, 256), device=value16.device, dtype=F16)
    flat = [v for hw in shapes for v in hw]
    check(_lib.lib().ink_msda_fused(value16.data_ptr(), proj.data_ptr(), proj.stride(0), ref.data_ptr(), d, d,
                                    Q * d if ref_batched else 0, (C.c_int32 * 8)(*flat), B, S, Q,
                                    out.data_ptr(), _stream()), "ink_msda_fused")
    return out


def swin_patchify(image_u8: torch.Tensor, mean: Sequence[float], std: Sequence[float], out: torch.Tensor):
    assert image_u8.dtype == torch.uint8 and image_u8.is_cuda and image_u8.is_contiguous()
    h, w, c = image_u8.shape
    assert c == 3 and out.dtype == F16 and out.is_contiguous() and out.numel() == -(-h // 4) * -(-w // 4) * 64
    check(_lib.lib().ink_swin_patchify(image_u8.data_ptr(), h, w, (C.c_float * 3)(*mean), (C.c_float * 3)(*std),
                                       out.data_ptr(), _stream()), "ink_swin_patchify")
    return out


def layernorm_merge4(x: torch.Tensor, gamma, beta, eps: float, gather4: torch.Tensor) -> torch.Tensor:
    assert x.dtype == F32 and x.stride(1) == 1 and gather4.dtype == torch.int32 and gather4.is_contiguous()
    rows, Cn = gather4.shape[0], x.shape[1]
    out = torch.empty((rows, 4 * Cn), device=x.device, dtype=F16)
    check(_lib.lib().ink_layernorm_merge4(x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr(), eps,
                                          gather4.data_ptr(), rows, Cn, out.data_ptr(), _stream()),
          "ink_layernorm_merge4")
    return out


def groupnorm_nhwc(x: torch.Tensor, B: int, T: int, G: int, gamma, beta, eps: float, out: torch.Tensor,
                   out_batch_stride: int) -> None:
    """x f32 [B*T, C] -> out (f32) at out + b*out_batch_stride + t*C (writes into the level's slice of
    the flattened multi-scale source)."""
    assert x.dtype == F32 and x.is_contiguous() and out.dtype == F32
    Cn = x.shape[1]
    ws = torch.empty(B * G * 2, device=x.device, dtype=F32)
    check(_lib.lib().ink_groupnorm_nhwc(x.data_ptr(), B, T, Cn, G, gamma.data_ptr(), beta.data_ptr(), eps,
                                        ws.data_ptr(), out.data_ptr(), out_batch_stride, _stream()),
          "ink_groupnorm_nhwc")


def gather_rows(x: torch.Tensor, idx: torch.Tensor, B: int, rows_per_batch: int, *, x_batch_rows: int,
                idx_batch_stride: int, out_dtype=F16) -> torch.Tensor:
    assert x.dtype == F32 and x.stride(1) == 1 and idx.dtype == torch.int32 and idx.is_cuda
    Cn = x.shape[1]
    out = torch.empty((B * rows_per_batch, Cn), device=x.device, dtype=out_dtype)
    oh = out.data_ptr() if out_dtype == F16 else None
    of = out.data_ptr() if out_dtype == F32 else None
    check(_lib.lib().ink_gather_rows(x.data_ptr(), x.stride(0), x_batch_rows, idx.data_ptr(), idx_batch_stride,
                                     rows_per_batch, B, Cn, oh, of, _stream()), "ink_gather_rows")
    return out


def biattn_fusion(qv16: torch.Tensor, kl16: torch.Tensor, B: int, S: int, T: int, scale: float,
                  chunk: int = 128):
    assert qv16.dtype == F16 and qv16.is_contiguous() and kl16.dtype == F16 and kl16.is_contiguous()
    E = qv16.shape[1] // 2
    dev = qv16.device
    nchunk = -(-S // chunk)
    scores = torch.empty(B * S * 4 * T, device=dev, dtype=F32)
    stats = torch.empty(B * 4 * T * 2, device=dev, dtype=F32)
    partial = torch.empty(B * 4 * nchunk * T * 256, device=dev, dtype=F32)
    out_v = torch.empty((B * S, E), device=dev, dtype=F16)
    out_l = torch.empty((B * T, E), device=dev, dtype=F16)
    check(_lib.lib().ink_biattn_fusion(qv16.data_ptr(), kl16.data_ptr(), B, S, T, E, scale, scores.data_ptr(),
                                       stats.data_ptr(), partial.data_ptr(), chunk, out_v.data_ptr(),
                                       out_l.data_ptr(), _stream()), "ink_biattn_fusion")
    return out_v, out_l


def fusion_fold(v: torch.Tensor, B: int, S: int, lnv_g: torch.Tensor, lnv_b: torch.Tensor, eps: float,
                text_kv: torch.Tensor, T: int, Wqv: torch.Tensor, bqv: torch.Tensor, Wo: torch.Tensor, bo: torch.Tensor,
                gamma_v: torch.Tensor, scale: float, pos: Optional[torch.Tensor] = None,
                out16_pos: Optional[torch.Tensor] = None, out16: Optional[torch.Tensor] = None) -> torch.Tensor:
    """BiAttentionBlock with the <= 4 caption tokens folded through it (csrc/fusion_fold.hip): v f32 [B*S, 256] is
    updated IN PLACE; returns the text-side attention output f16 [B*T, 1024].  text_kv: f32 [B*T, 2048] =
    [l_proj | values_l_proj] of LN_l(l); Wqv f16 [2048, 256] / bqv f32 [2048] = [v_proj ; values_v_proj].  out16 /
    out16_pos (f16 [B*S, 256], optional): f16(v) and f16(v + pos[s]) of the UPDATED v, pos f32 [S, 256]."""
    for t in (out16, out16_pos):
        assert t is None or (t.dtype == F16 and t.is_contiguous() and tuple(t.shape) == (B * S, 256))
    assert out16_pos is None or (pos is not None and pos.dtype == F32 and pos.is_contiguous() and tuple(pos.shape) == (S, 256))
    assert v.dtype == F32 and v.is_contiguous() and tuple(v.shape) == (B * S, 256)
    assert text_kv.dtype == F32 and text_kv.stride(1) == 1 and tuple(text_kv.shape) == (B * T, 2048)
    assert Wqv.dtype == F16 and Wqv.is_contiguous() and tuple(Wqv.shape) == (2048, 256) and bqv.dtype == F32
    assert Wo.dtype == F16 and Wo.is_contiguous() and tuple(Wo.shape) == (256, 1024)
    need = C.c_int64(0)
    check(_lib.lib().ink_fusion_fold_workspace(B, S, C.byref(need)), "ink_fusion_fold_workspace")
    ws = torch.empty(need.value, device=v.device, dtype=F32)
    out_l = torch.empty((B * T, 1024), device=v.device, dtype=F16)
    check(_lib.lib().ink_fusion_fold(v.data_ptr(), B, S, lnv_g.data_ptr(), lnv_b.data_ptr(), eps, text_kv.data_ptr(),
                                     text_kv[:, 1024:].data_ptr(), text_kv.stride(0), T, Wqv.data_ptr(), bqv.data_ptr(),
                                     Wqv[1024:].data_ptr(), bqv[1024:].data_ptr(), Wo.data_ptr(), bo.data_ptr(),
                                     gamma_v.data_ptr(), scale, ws.data_ptr(), out_l.data_ptr(), _p(pos), _p(out16_pos),
                                     _p(out16), _stream()),
          "ink_fusion_fold")
    return out_l


def proj256_ln_pack(ws: torch.Tensor) -> torch.Tensor:
    """A [128 -> 256] projection as the split-f16 matrix [256, 384] -> the LDS image of proj256_ln (load time)."""
    assert ws.dtype == F16 and ws.is_contiguous() and tuple(ws.shape) == (256, 384)
    blob = torch.empty(3 * 64 * 64 * 8, device=ws.device, dtype=F16)
    check(_lib.lib().ink_proj256_ln_pack(ws.data_ptr(), blob.data_ptr(), _stream()), "ink_proj256_ln_pack")
    return blob


def proj256_ln(a: torch.Tensor, blob: torch.Tensor, bias: torch.Tensor, res: torch.Tensor, ln_g: torch.Tensor,
               ln_b: torch.Tensor, eps: float, *, res_batch_rows: Optional[torch.Tensor] = None, rows_per_batch: int = 0,
               want_f32: bool = True, want_split: bool = True):
    """LayerNorm(res + a W^T + bias) in one kernel (transformer.py:175-182: out_proj + residual + norm4 of the image
    tokens; csrc/proj_ln.hip).  a f32 [R, 128]; res f32 [R, 256] or, with res_batch_rows (int32 per box), a tensor shared
    by the boxes of an image.  -> (f32 [R, 256] or None, split-f16 operand [R, 768] or None)."""
    R = int(a.shape[0])
    assert a.dtype == F32 and a.is_contiguous() and tuple(a.shape) == (R, 128)
    assert blob.dtype == F16 and blob.numel() == 3 * 64 * 64 * 8
    assert res.dtype == F32 and res.is_contiguous() and res.shape[1] == 256
    assert all(t.dtype == F32 and t.is_contiguous() and t.numel() == 256 for t in (bias, ln_g, ln_b))
    if res_batch_rows is not None:
        assert res_batch_rows.dtype == torch.int32 and res_batch_rows.is_cuda and rows_per_batch > 0
        assert res_batch_rows.numel() * rows_per_batch == R
    else:
        assert res.shape[0] == R
    of = torch.empty((R, 256), device=a.device, dtype=F32) if want_f32 else None
    osp = torch.empty((R, 768), device=a.device, dtype=F16) if want_split else None
    check(_lib.lib().ink_proj256_ln(a.data_ptr(), blob.data_ptr(), bias.data_ptr(), res.data_ptr(), _p(res_batch_rows),
                                    rows_per_batch, ln_g.data_ptr(), ln_b.data_ptr(), eps, R, _p(of), _p(osp), _stream()),
          "ink_proj256_ln")
    return of, osp


def sam_upscale_pack(ws: torch.Tensor) -> torch.Tensor:
    """output_upscaling.3 as the split-f16 matrix [128, 192] -> the LDS image of sam_upscale_tail (load time)."""
    assert ws.dtype == F16 and ws.is_contiguous() and tuple(ws.shape) == (128, 192)
    blob = torch.empty(4 * 12 * 64 * 8, device=ws.device, dtype=F16)
    check(_lib.lib().ink_sam_upscale_pack(ws.data_ptr(), blob.data_ptr(), _stream()), "ink_sam_upscale_pack")
    return blob


def sam_upscale_tail(u0: torch.Tensor, n: int, g: int, ln_g: torch.Tensor, ln_b: torch.Tensor, eps: float,
                     blob: torch.Tensor, b3: torch.Tensor, hyper: torch.Tensor) -> torch.Tensor:
    """LayerNorm2d + GELU + ConvTranspose2d(k2 s2) + GELU + hyper-network product of the mask decoder in one kernel
    (mask_decoder.py:54-60, 138-145): u0 f32 [n*g*g, 256] (4 sub-pixels x 64 channels per token; rows may be a column
    block of a wider tensor) or [n*g*g*4, 64] contiguous -> low-res mask logits f32 [n, 4g, 4g]."""
    if u0.shape[1] == 64:
        assert u0.is_contiguous() and u0.shape[0] == n * g * g * 4
        u0 = u0.view(n * g * g, 256)
    assert u0.dtype == F32 and u0.stride(1) == 1 and tuple(u0.shape) == (n * g * g, 256) and u0.stride(0) % 4 == 0
    assert blob.dtype == F16 and blob.numel() == 4 * 12 * 64 * 8 and b3.dtype == F32 and b3.numel() == 128
    assert hyper.dtype == F32 and hyper.is_contiguous() and tuple(hyper.shape) == (n, 32)
    assert ln_g.dtype == F32 and ln_b.dtype == F32 and ln_g.numel() == 64 and ln_b.numel() == 64
    low = torch.empty((n, 4 * g, 4 * g), device=u0.device, dtype=F32)
    check(_lib.lib().ink_sam_upscale_tail(u0.data_ptr(), u0.stride(0), n, g, ln_g.data_ptr(), ln_b.data_ptr(), eps, blob.data_ptr(),
                                          b3.data_ptr(), hyper.data_ptr(), low.data_ptr(), _stream()), "ink_sam_upscale_tail")
    return low


def ffn256_pack(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, w_pre: Optional[torch.Tensor] = None) -> torch.Tensor:
    """linear1.weight f16 [hid, 256] + linear1.bias f32 [hid] + linear2.weight f16 [256, hid] (+ the weight f16 [256, 256] of
    a preceding projection, see ffn256_fused) -> the packed weight blob (done once at load time; csrc/ffn_fused.hip)."""
    hid = int(w1.shape[0])
    assert w1.dtype == F16 and w2.dtype == F16 and w1.is_contiguous() and w2.is_contiguous()
    assert tuple(w1.shape) == (hid, 256) and tuple(w2.shape) == (256, hid)
    assert b1.dtype == F32 and b1.is_contiguous() and b1.numel() == hid
    assert w_pre is None or (w_pre.dtype == F16 and w_pre.is_contiguous() and tuple(w_pre.shape) == (256, 256))
    need = C.c_int64(0)
    check(_lib.lib().ink_ffn256_pack_bytes(hid, int(w_pre is not None), C.byref(need)), "ink_ffn256_pack_bytes")
    blob = torch.empty(need.value // 2, device=w1.device, dtype=F16)
    check(_lib.lib().ink_ffn256_pack(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), hid, _p(w_pre), blob.data_ptr(), _stream()),
          "ink_ffn256_pack")
    return blob


def ffn256_fused(x16: torch.Tensor, res: torch.Tensor, blob: torch.Tensor, hid: int, b2: torch.Tensor,
                 ln_g: torch.Tensor, ln_b: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None,
                 pre: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """LayerNorm(res + linear2(relu(linear1(x16)))) for d_model 256 in one kernel (transformer.py:780-799).  x16 f16
    [M, 256] (row stride free), res f32 [M, 256]; blob = ffn256_pack(...) of a d_ffn = hid layer; out f32 [M, 256] (may be
    `res`).  pre = (bias, ln_weight, ln_bias) of a preceding projection whose weight is in the blob: then
    s = LayerNorm_pre(res + x16 W_pre^T + bias) is formed first and the block runs on s (x16 = that projection's input)."""
    M = int(x16.shape[0])
    assert x16.dtype == F16 and x16.stride(1) == 1 and x16.shape[1] == 256
    assert res.dtype == F32 and res.is_contiguous() and tuple(res.shape) == (M, 256)
    need = C.c_int64(0)
    check(_lib.lib().ink_ffn256_pack_bytes(hid, int(pre is not None), C.byref(need)), "ink_ffn256_pack_bytes")
    assert blob.dtype == F16 and blob.numel() * 2 == need.value and b2.dtype == F32
    assert b2.numel() == 256 and ln_g.numel() == 256 and ln_b.numel() == 256 and ln_g.dtype == F32 and ln_b.dtype == F32
    if pre is not None:
        assert all(t.dtype == F32 and t.is_contiguous() and t.numel() == 256 for t in pre)
    if out is None:
        out = torch.empty_like(res)
    assert out.dtype == F32 and out.is_contiguous() and tuple(out.shape) == (M, 256)
    pb, pg, pe = pre if pre is not None else (None, None, None)
    check(_lib.lib().ink_ffn256_fused(x16.data_ptr(), x16.stride(0), res.data_ptr(), blob.data_ptr(), b2.data_ptr(),
                                      ln_g.data_ptr(), ln_b.data_ptr(), eps, M, hid, _p(pb), _p(pg), _p(pe), out.data_ptr(),
                                      _stream()), "ink_ffn256_fused")
    return out


def attn_fewkeys(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, B: int, n_heads: int, head_dim: int,
                 scale: float, blocked: Optional[torch.Tensor] = None, n_q: Optional[int] = None,
                 q_batch_rows: Optional[torch.Tensor] = None, q_add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attention against n_k <= 16 keys per batch entry; q/k/v/out are all f16 or all f32 rows (f32 math either
    way).  q_batch_rows: first q row of each batch entry (then n_q must be given)."""
    io = q.dtype
    for t in (q, k, v):
        assert t.dtype == io and io in (F16, F32) and t.dim() == 2 and t.stride(1) == 1
    if n_q is None:
        n_q = q.shape[0] // B
    n_k = k.shape[0] // B
    out = torch.empty((B * n_q, n_heads * head_dim), device=q.device, dtype=io)
    if blocked is not None:
        assert blocked.dtype == torch.uint8 and blocked.is_contiguous() and blocked.shape == (n_q, n_k)
    if q_batch_rows is not None:
        assert q_batch_rows.dtype == torch.int32 and q_batch_rows.numel() == B and q_batch_rows.is_cuda
    if q_add is not None:
        assert q_add.dtype == F32 and q_add.is_contiguous() and tuple(q_add.shape) == (n_q, n_heads * head_dim)
    check(_lib.lib().ink_attn_fewkeys(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(),
                                      v.stride(0), B, n_q, n_k, n_heads, head_dim, scale, _p(blocked),
                                      _p(q_batch_rows), _p(q_add), int(io == F32), out.data_ptr(), out.stride(0), _stream()),
          "ink_attn_fewkeys")
    return out


def attn_fewq(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, n_batch: int, n_heads: int, head_dim: int,
              scale: float, n_q: int, n_k: int, q_batch_rows: Optional[torch.Tensor] = None,
              kv_batch_rows: Optional[torch.Tensor] = None, k_add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Few queries (<= 8) against many keys; same row conventions as flash_attn.  q/k/v/out all f16 or all f32."""
    io = q.dtype
    for t in (q, k, v):
        assert t.dtype == io and io in (F16, F32) and t.dim() == 2 and t.stride(1) == 1
    out = torch.empty((n_batch * n_q, n_heads * head_dim), device=q.device, dtype=io)
    if k_add is not None:
        assert io == F32 and k_add.dtype == F32 and k_add.is_contiguous() and tuple(k_add.shape) == (n_k, n_heads * head_dim)
    check(_lib.lib().ink_attn_fewq(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                   n_batch, n_q, n_k, n_heads, head_dim, scale, _p(q_batch_rows),
                                   _p(kv_batch_rows), _p(k_add), int(io == F32), out.data_ptr(), out.stride(0), _stream()),
          "ink_attn_fewq")
    return out


def topk_rowmax(logits: torch.Tensor, K: int, want_values: bool = False):
    """logits f32 [B,S,T] -> int32 [B,K] indices of the K largest row maxima (descending, stable)."""
    assert logits.dtype == F32 and logits.is_contiguous() and logits.dim() == 3
    B, S, T = logits.shape
    idx = torch.empty((B, K), device=logits.device, dtype=torch.int32)
    val = torch.empty((B, K), device=logits.device, dtype=F32) if want_values else None
    nchunk = -(-S // 16384)
    ws = torch.empty((B * nchunk * K,), device=logits.device, dtype=torch.int64) if nchunk > 1 else None
    check(_lib.lib().ink_topk_rowmax(logits.data_ptr(), B, S, T, K, idx.data_ptr(), _p(val), _p(ws), _stream()),
          "ink_topk_rowmax")
    return (idx, val) if want_values else idx


def sine_embed4(ref: torch.Tensor, dim_t: torch.Tensor) -> torch.Tensor:
    assert ref.dtype == F32 and ref.is_contiguous() and ref.shape[-1] == 4 and dim_t.numel() == 128
    N = ref.numel() // 4
    out = torch.empty((N, 512), device=ref.device, dtype=F16)
    check(_lib.lib().ink_sine_embed4(ref.data_ptr(), dim_t.data_ptr(), N, out.data_ptr(), _stream()),
          "ink_sine_embed4")
    return out


def box_refine(delta: torch.Tensor, ref: torch.Tensor, ref_is_logit: bool = False) -> torch.Tensor:
    assert delta.dtype == F32 and delta.stride(1) == 1 and ref.dtype == F32 and ref.is_contiguous()
    N = ref.numel() // 4
    out = torch.empty_like(ref)
    check(_lib.lib().ink_box_refine(delta.data_ptr(), delta.stride(0), ref.data_ptr(), N, int(ref_is_logit),
                                    out.data_ptr(), _stream()), "ink_box_refine")
    return out


# ---------------------------------------------------------------------------------------------
# refinement hand-off ops
# ---------------------------------------------------------------------------------------------
def mask_cleanup(masks_u8: torch.Tensor, k: int, area_threshold: int = 500, aspect_threshold: float = 1.1,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """clean_up_mask (InkLayer/refinement/mask_cleaner.py:11-36) for [n, H, W] uint8 masks (> 127 = foreground) on
    the GPU: k x k closing, 8-connected components, area / aspect filter.  -> uint8 0/255 [n, H, W]."""
    assert masks_u8.dtype == torch.uint8 and masks_u8.is_cuda and masks_u8.is_contiguous() and masks_u8.dim() == 3
    n, H, W = masks_u8.shape
    if out is None:
        out = torch.empty_like(masks_u8)
    if n == 0:
        return out
    assert out.dtype == torch.uint8 and out.is_contiguous() and out.shape == masks_u8.shape
    need = C.c_int64(0)
    check(_lib.lib().ink_mask_cleanup_workspace_ints(n, H, W, k, C.byref(need)), "ink_mask_cleanup_workspace_ints")
    ws = torch.empty(need.value, device=masks_u8.device, dtype=torch.int32)
    ta, tb = torch.empty_like(masks_u8), torch.empty_like(masks_u8)
    check(_lib.lib().ink_mask_cleanup(masks_u8.data_ptr(), n, H, W, k, area_threshold, float(aspect_threshold),
                                      ta.data_ptr(), tb.data_ptr(), ws.data_ptr(), out.data_ptr(), _stream()),
          "ink_mask_cleanup")
    out._ink_overflow_flag = ws[:1]        # stays 0 by construction (run bound of a closed image); tests read it
    return out


def mask_sketch_iou_counts(masks_u8: torch.Tensor, sketch_rgb_u8: torch.Tensor) -> torch.Tensor:
    """int32 [n, n, 2] = (|r_i & r_j|, |r_i | r_j|), r = (mask > 0) & (PIL-luma(sketch) < 250)
    (InkLayer/refinement/nms_sketch.py:62-78, 186-234)."""
    assert masks_u8.dtype == torch.uint8 and masks_u8.is_cuda and masks_u8.is_contiguous() and masks_u8.dim() == 3
    n, H, W = masks_u8.shape
    assert sketch_rgb_u8.dtype == torch.uint8 and sketch_rgb_u8.is_cuda and sketch_rgb_u8.is_contiguous()
    assert tuple(sketch_rgb_u8.shape) == (H, W, 3), "masks and sketch have the same size on this path"
    counts = torch.empty((n, n, 2), device=masks_u8.device, dtype=torch.int32)
    if n == 0:
        return counts
    bits = torch.empty((n, (H * W + 63) // 64), device=masks_u8.device, dtype=torch.int64)
    check(_lib.lib().ink_mask_sketch_iou_counts(masks_u8.data_ptr(), sketch_rgb_u8.data_ptr(), n, H, W,
                                                bits.data_ptr(), counts.data_ptr(), _stream()),
          "ink_mask_sketch_iou_counts")
    return counts


# ---------------------------------------------------------------------------------------------
# Depth-Anything-V2 pixel-side ops
# ---------------------------------------------------------------------------------------------
def depth_patchify(image_u8: torch.Tensor, nh: int, nw: int, P: int, KP: int, mean: Sequence[float],
                   std: Sequence[float], chan_reverse: bool) -> torch.Tensor:
    """image2tensor (cv2 cubic resize to (nh, nw), normalise) + 14x14 patch gather -> split-f16 [T, 3*KP]."""
    assert image_u8.dtype == torch.uint8 and image_u8.is_cuda and image_u8.is_contiguous() and image_u8.shape[2] == 3
    H, W = int(image_u8.shape[0]), int(image_u8.shape[1])
    out = torch.empty(((nh // P) * (nw // P), 3 * KP), device=image_u8.device, dtype=F16)
    check(_lib.lib().ink_depth_patchify(image_u8.data_ptr(), H, W, nh, nw, P, KP, (C.c_double * 3)(*mean),
                                        (C.c_double * 3)(*std), int(chan_reverse), out.data_ptr(), _stream()),
          "ink_depth_patchify")
    return out


def resize_bilinear_ac(x: torch.Tensor, B: int, h: int, w: int, H: int, W: int, out_dtype=F32) -> torch.Tensor:
    """F.interpolate(bilinear, align_corners=True) of an NHWC f32 map [B*h*w, C] -> [B*H*W, C] (f32 or f16)."""
    assert x.dtype == F32 and x.is_contiguous() and x.shape[0] == B * h * w
    Cn = x.shape[1]
    out = torch.empty((B * H * W, Cn), device=x.device, dtype=out_dtype)
    check(_lib.lib().ink_resize_bilinear_ac_nhwc(x.data_ptr(), B, h, w, Cn, H, W,
                                                 out.data_ptr() if out_dtype == F32 else None,
                                                 out.data_ptr() if out_dtype == F16 else None, _stream()),
          "ink_resize_bilinear_ac_nhwc")
    return out


def im2col3x3_ex(x: torch.Tensor, B: int, H: int, W: int, stride: int = 1, relu: bool = False) -> torch.Tensor:
    """f16 NHWC [B*H*W, C] -> [B*OH*OW, 9*C] (pad 1, stride 1 or 2, optional ReLU on the gathered values)."""
    assert x.dtype == F16 and x.is_contiguous() and x.shape[0] == B * H * W
    Cn = x.shape[1]
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((B * OH * OW, 9 * Cn), device=x.device, dtype=F16)
    check(_lib.lib().ink_im2col3x3_ex_f16(x.data_ptr(), B, H, W, Cn, stride, int(relu), out.data_ptr(), _stream()),
          "ink_im2col3x3_ex_f16")
    return out
