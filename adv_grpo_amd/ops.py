"""Thin Python wrappers over the C-ABI kernels (device tensors in, device tensors out).

Host plumbing only: allocation via torch, pointers + sizes handed to libadvgrpo_hip.so."""
import torch

from . import _lib

ACT = {None: 0, "none": 0, "gelu_tanh": 1, "gelu": 2, "gelu_erf": 2, "silu": 3}


def gemm(a, w, bias=None, act=None, alpha=1.0, gate=None, gate_rows=0, residual=None, out=None,
         out_dtype=torch.bfloat16, seg=None):
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T).  a, w: bf16, last dim contiguous.
    gate: [G, N] bf16 with row m using gate[m // gate_rows].  residual: [rows, N] bf16 (indexed with the
    output row map).  seg = (seg_rows, seg_stride, seg_off) scatters row m to
    (m // seg_rows) * seg_stride + seg_off + m % seg_rows of `out` (which must then be given)."""
    lib = _lib.load()
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    assert a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1]
    assert a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        assert seg is None
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    assert out.stride(-1) == 1
    seg_rows, seg_stride, seg_off = seg if seg is not None else (0, 0, 0)
    _lib.check(lib.advgrpo_gemm_bf16(
        a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(-2),
        _lib.dtype_code(out.dtype), M, N, K, _lib.ptr(bias), ACT[act], float(alpha),
        gate.data_ptr() if gate is not None else None, gate.stride(0) if gate is not None else 0, int(gate_rows),
        residual.data_ptr() if residual is not None else None, residual.stride(-2) if residual is not None else 0,
        int(seg_rows), int(seg_stride), int(seg_off), 1, 0, 0, 0, _lib.stream_ptr()))
    return out


def bmm_nt(a, w, out=None, out_dtype=torch.bfloat16, alpha=1.0):
    """Batched out[b] = alpha * a[b] @ w[b]^T; a [B,M,K], w [B,N,K] bf16."""
    lib = _lib.load()
    B, M, K = a.shape
    N = w.shape[1]
    assert a.stride(2) == 1 and w.stride(2) == 1
    if out is None:
        out = torch.empty(B, M, N, dtype=out_dtype, device=a.device)
    _lib.check(lib.advgrpo_gemm_bf16(
        a.data_ptr(), a.stride(1), w.data_ptr(), w.stride(1), out.data_ptr(), out.stride(1),
        _lib.dtype_code(out.dtype), M, N, K, None, 0, float(alpha), None, 0, 0, None, 0, 0, 0, 0,
        B, a.stride(0), w.stride(0), out.stride(0), _lib.stream_ptr()))
    return out


def attention(q, k, v, num_heads, scale=None, causal=False, out=None):
    """q [B,Sq,H*D], k/v [B,Skv,H*D] bf16 views (last dim contiguous, any row/batch pitch) -> [B,Sq,H*D]."""
    lib = _lib.load()
    B, Sq, HD = q.shape
    Skv = k.shape[1]
    D = HD // num_heads
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    if out is None:
        out = torch.empty(B, Sq, HD, dtype=torch.bfloat16, device=q.device)
    if scale is None:
        scale = D ** -0.5
    _lib.check(lib.advgrpo_attention_fwd(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q.stride(1), k.stride(1), v.stride(1),
        out.stride(1), q.stride(0), k.stride(0), v.stride(0), out.stride(0), B, num_heads, Sq, Skv, D,
        float(scale), int(causal), _lib.stream_ptr()))
    return out
