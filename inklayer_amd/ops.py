"""Thin torch-tensor wrappers over the C ABI (PyTorch = device memory + streams only).

Every function enqueues one hand-written HIP kernel on torch's current stream.
Tensors must live on the GPU; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import InkGemm, check

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
