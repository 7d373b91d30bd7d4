"""Thin torch-tensor wrappers over the C ABI (PyTorch = device memory + streams only).

Every function enqueues one hand-written HIP kernel on torch's current stream.
Tensors must live on the GPU; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import InkAttn, InkGemm, check

ACT = {None: 0, "none": 0, "gelu": 1, "relu": 2}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda, "InkLayer HIP ops need GPU tensors (no CPU fallback)"
    return t.data_ptr()


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
         act: Optional[str] = None, residual: Optional[torch.Tensor] = None,
         col_scale: Optional[torch.Tensor] = None, row_map: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.float32,
         out_rows: Optional[int] = None) -> torch.Tensor:
    """out[row_map[m]] = residual[row_map[m]] + col_scale * act(a[m] @ w.T + bias).

    a: f16 [M, K] (row stride arbitrary, multiple of 8), w: f16 [N, K].
    """
    assert a.dtype == torch.float16 and w.dtype == torch.float16
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        rows = out_rows if out_rows is not None else M
        out = torch.empty((rows, N), device=a.device, dtype=out_dtype)
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape[1] == N
    p = InkGemm()
    p.A, p.W, p.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    p.bias, p.col_scale = _p(bias), _p(col_scale)
    p.residual, p.row_map = _p(residual), _p(row_map)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
    if col_scale is not None:
        assert col_scale.dtype == torch.float32 and col_scale.numel() == N
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(1) == 1 and residual.shape[1] == N
        p.ldr = residual.stride(0)
    if row_map is not None:
        assert row_map.dtype == torch.int32 and row_map.numel() == M
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldw, p.ldc = a.stride(0), w.stride(0), out.stride(0)
    p.act = ACT[act]
    p.c_f16 = 1 if out.dtype == torch.float16 else 0
    assert out.dtype in (torch.float16, torch.float32)
    check(_lib.lib().ink_gemm_f16(C.byref(p), _stream()), "ink_gemm_f16")
    return out


def layernorm_rows(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor],
                   eps: float, *, gather: Optional[torch.Tensor] = None,
                   out_dtype: torch.dtype = torch.float16,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm over the last dim of f32 x [R, C]; optional row gather (-1 -> zero row)."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    Cdim = x.shape[1]
    rows = gather.numel() if gather is not None else x.shape[0]
    if out is None:
        out = torch.empty((rows, Cdim), device=x.device, dtype=out_dtype)
    assert out.shape == (rows, Cdim) and out.stride(1) == 1
    oh = out.data_ptr() if out.dtype == torch.float16 else None
    of = out.data_ptr() if out.dtype == torch.float32 else None
    if gather is not None:
        assert gather.dtype == torch.int32
    check(_lib.lib().ink_layernorm_rows(x.data_ptr(), x.stride(0), _p(gamma), _p(beta), eps,
                                        _p(gather), rows, Cdim, oh, of, out.stride(0), _stream()),
          "ink_layernorm_rows")
    return out


def add_cvt_f16(a: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f16(a + b) for contiguous f32 tensors."""
    assert a.dtype == torch.float32 and a.is_contiguous()
    if b is not None:
        assert b.dtype == torch.float32 and b.is_contiguous() and b.shape == a.shape
    out = torch.empty(a.shape, device=a.device, dtype=torch.float16)
    check(_lib.lib().ink_add_cvt_f16(a.data_ptr(), _p(b), out.data_ptr(), a.numel(), _stream()),
          "ink_add_cvt_f16")
    return out


def flash_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, n_batch: int, n_heads: int,
               head_dim: int, scale: float, rel_h: Optional[torch.Tensor] = None,
               rel_w: Optional[torch.Tensor] = None, rel_aug: Optional[torch.Tensor] = None,
               grid_w: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(scale*q@k^T + bias)@v for f16 row views q [B*nq, >=H*hd], k/v [B*nk, ...].

    q/k/v may be column slices of one packed qkv buffer (only the row stride matters).
    """
    for t in (q, k, v):
        assert t.dtype == torch.float16 and t.dim() == 2 and t.stride(1) == 1 and t.is_cuda
    n_q, n_k = q.shape[0] // n_batch, k.shape[0] // n_batch
    if out is None:
        out = torch.empty((q.shape[0], n_heads * head_dim), device=q.device, dtype=torch.float16)
    p = InkAttn()
    p.Q, p.K, p.V, p.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    p.ldq, p.ldk, p.ldv, p.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    p.n_batch, p.n_heads, p.n_q, p.n_k, p.head_dim = n_batch, n_heads, n_q, n_k, head_dim
    p.scale = scale
    p.grid_w = grid_w
    if rel_aug is not None:
        assert rel_aug.dtype == torch.float16 and rel_aug.is_contiguous()
        p.bias_mode, p.rel_aug = 2, rel_aug.data_ptr()
    elif rel_h is not None:
        assert rel_h.dtype == torch.float32 and rel_w.dtype == torch.float32
        assert rel_h.is_contiguous() and rel_w.is_contiguous()
        p.bias_mode, p.rel_h, p.rel_w = 1, rel_h.data_ptr(), rel_w.data_ptr()
    else:
        p.bias_mode = 0
    check(_lib.lib().ink_flash_attn(C.byref(p), _stream()), "ink_flash_attn")
    return out


def relpos_bias(q: torch.Tensor, rel_pos_h: torch.Tensor, rel_pos_w: torch.Tensor, *, S: int,
                n_batch: int, n_heads: int, head_dim: int, scale: float):
    """SAM decomposed rel-pos terms / scale.  S == 64 -> (rel_h, rel_w) f32; S <= 16 -> rel_aug f16."""
    assert q.dtype == torch.float16 and q.stride(1) == 1
    assert rel_pos_h.dtype == torch.float32 and rel_pos_h.is_contiguous()
    assert rel_pos_w.dtype == torch.float32 and rel_pos_w.is_contiguous()
    assert rel_pos_h.shape == (2 * S - 1, head_dim)
    n = n_batch * n_heads * S * S
    fn = _lib.lib().ink_relpos_bias
    if S == 64:
        oh = torch.empty((n, 64), device=q.device, dtype=torch.float32)
        ow = torch.empty((n, 64), device=q.device, dtype=torch.float32)
        check(fn(q.data_ptr(), q.stride(0), rel_pos_h.data_ptr(), rel_pos_w.data_ptr(), S, n_batch,
                 n_heads, head_dim, scale, oh.data_ptr(), ow.data_ptr(), None, _stream()),
              "ink_relpos_bias")
        return oh, ow
    aug = torch.empty((n, 32), device=q.device, dtype=torch.float16)
    check(fn(q.data_ptr(), q.stride(0), rel_pos_h.data_ptr(), rel_pos_w.data_ptr(), S, n_batch,
             n_heads, head_dim, scale, None, None, aug.data_ptr(), _stream()), "ink_relpos_bias")
    return aug
