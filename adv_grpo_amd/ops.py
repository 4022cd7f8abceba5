"""Thin Python wrappers over the C-ABI kernels (device tensors in, device tensors out).

Host plumbing only: allocation via torch, pointers + sizes handed to libadvgrpo_hip.so."""
import ctypes

import threading

import torch

from . import _lib

ACT = {None: 0, "none": 0, "gelu_tanh": 1, "gelu_new": 1, "gelu": 2, "gelu_erf": 2, "silu": 3, "quick_gelu": 4}

GEMM_VARIANTS = {0: "gemm_bf16_kernel<128,128,2,2,false>", 1: "gemm_bf16_kernel<128,64,2,2,false>",
                 2: "gemm_bf16_kernel<64,128,2,2,false>", 4: "gemm_bf16_kernel<128,128,2,2,true>",
                 5: "gemm_bf16_kernel<128,64,2,2,true>", 11: "gemm_bf16_pipe_kernel<128,128,3,4,2>",
                 14: "gemm_bf16_kernel<128,128,2,4,false>", 15: "gemm_bf16_kernel<128,128,4,2,false>",
                 17: "gemm_bf16_pipe_kernel<256,128,3,4,2>", 18: "gemm_bf16_kernel<128,128,4,2,true>",
                 20: "gemm_bf16_pp_kernel<256,256,4,2,4>", 21: "gemm_bf16_pp_kernel<256,128,4,4,2>",
                 23: "gemm_bf16_pp_kernel<256,128,3,4,2>", 26: "gemm_bf16_kernel<192,128,4,2,false>",
                 27: "gemm_bf16_kernel<128,192,2,4,false>", 30: "gemm8p_kernel", 31: "gemm4w_kernel"}
# bench.py sets this to a list to collect (kernel name, flops, start event, end event) per GEMM launch;
# events are recorded on the stream the kernel is launched on (torch's current stream).
PROFILE = None
PROFILE_STRIDE = 1      # bench.py: HIP events around every PROFILE_STRIDE-th launch OF EACH GEMM KERNEL (1 = all of them)
PROFILE_COUNTS = {}     # kernel name -> launches seen while PROFILE is on (sampled or not)


class _Prof:
    def __init__(self, M, N, K, batch, conv):
        self.on = PROFILE is not None
        if self.on:
            lib = _lib.load()
            self.name = GEMM_VARIANTS[lib.advgrpo_gemm_variant(M, N, K, batch, conv)]
            n = PROFILE_COUNTS[self.name] = PROFILE_COUNTS.get(self.name, 0) + 1
            self.on = n % PROFILE_STRIDE == 0
        if self.on:
            self.flops = 2.0 * M * N * K * batch
            self.shape = (M, N, K, batch, conv)
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.s.record()

    def __exit__(self, *a):
        if self.on:
            self.e.record()
            PROFILE.append((self.name, self.flops, self.s, self.e, self.shape))


def gemm(a, w, bias=None, act=None, alpha=1.0, gate=None, gate_rows=0, residual=None, out=None,
         out_dtype=torch.bfloat16, seg=None, a_seg=None, M=None):
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T).  a, w: bf16, last dim contiguous.
    gate: [G, N] bf16 with row m using gate[m // gate_rows].  residual: [rows, N] bf16 (indexed with the
    output row map).  seg = (seg_rows, seg_stride, seg_off) scatters row m to
    (m // seg_rows) * seg_stride + seg_off + m % seg_rows of `out` (which must then be given)."""
    lib = _lib.load()
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    assert a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1]
    assert a.stride(1) == 1 and w.stride(1) == 1
    K = a.shape[1]
    M = a.shape[0] if M is None else M
    N = w.shape[0]
    if out is None:
        assert seg is None
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    assert out.stride(-1) == 1
    seg_rows, seg_stride, seg_off = seg if seg is not None else (0, 0, 0)
    a_rows, a_stride, a_off = a_seg if a_seg is not None else (0, 0, 0)
    with _Prof(M, N, K, 1, 0):
      _lib.check(lib.advgrpo_gemm_bf16(
        a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(-2),
        _lib.dtype_code(out.dtype), M, N, K, _lib.ptr(bias), ACT[act], float(alpha),
        gate.data_ptr() if gate is not None else None, gate.stride(0) if gate is not None else 0, int(gate_rows),
        residual.data_ptr() if residual is not None else None, residual.stride(-2) if residual is not None else 0,
        int(seg_rows), int(seg_stride), int(seg_off), int(a_rows), int(a_stride), int(a_off), 1, 0, 0, 0,
        _lib.stream_ptr()))
    return out


def gemm_desc(a, w, bias=None, act=None, alpha=1.0, gate=None, gate_rows=0, residual=None, out=None,
              out_dtype=torch.bfloat16, seg=None, a_seg=None, M=None, aux_out=None, aux_in=None, rms=None):
    """Descriptor of one Linear for gemm_grouped (arguments as in gemm / gemm_train).
    rms = (weight [n_w, 64] bf16, nheads, heads_per_weight, eps, rs_out or None): fused per-head RMSNorm of the first
    `nheads` 64-wide output column groups (QK-norm of a fused QKV projection).  Returns (descriptor, out)."""
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.shape[1] == w.shape[1]
    assert a.stride(1) == 1 and w.stride(1) == 1
    K = a.shape[1]
    M = a.shape[0] if M is None else M
    N = w.shape[0]
    if out is None:
        assert seg is None
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    assert out.stride(-1) == 1
    d = _lib.GemmDesc()
    d.A, d.W, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.lda, d.ldw, d.ldc = a.stride(0), w.stride(0), out.stride(-2)
    d.out_dtype, d.M, d.N, d.K = _lib.dtype_code(out.dtype), M, N, K
    d.bias = bias.data_ptr() if bias is not None else None
    d.act = ACT_D[act] if act in ACT_D else ACT[act]
    d.alpha = float(alpha)
    if gate is not None:
        d.gate, d.gate_stride, d.gate_rows = gate.data_ptr(), gate.stride(0), int(gate_rows)
    if residual is not None:
        d.residual, d.ldr = residual.data_ptr(), residual.stride(-2)
    if seg is not None:
        d.seg_rows, d.seg_stride, d.seg_off = (int(v) for v in seg)
    if a_seg is not None:
        d.a_seg_rows, d.a_seg_stride, d.a_seg_off = (int(v) for v in a_seg)
    aux = aux_out if aux_out is not None else aux_in
    if aux is not None:
        d.aux_out = aux_out.data_ptr() if aux_out is not None else None
        d.aux_in = aux_in.data_ptr() if aux_in is not None else None
        d.ld_aux = aux.stride(0)
    if rms is not None:
        rw, nheads, hpw, eps, rs_out = rms
        d.rms_weight, d.rms_nheads, d.rms_heads_per_weight, d.rms_eps = rw.data_ptr(), int(nheads), int(hpw), float(eps)
        d.rms_rs_out = rs_out.data_ptr() if rs_out is not None else None
    return d, out


def gemm_grouped(descs):
    """One launch for one or two Linears ((descriptor, out) pairs from gemm_desc); returns the outputs."""
    lib = _lib.load()
    arr = (_lib.GemmDesc * len(descs))(*[d for d, _ in descs])
    d0 = descs[0][0]
    flops_extra = sum(2.0 * d.M * d.N * d.K for d, _ in descs[1:])
    prof = _Prof(d0.M, d0.N, d0.K, 1, 0)
    if prof.on and len(descs) == 2:
        prof.flops += flops_extra
        # the grouped launch runs the *_pair_kernel instantiation of the first problem's tile variant (15 / 17 / 26);
        # any other variant falls back to two launches and is recorded under the first one's name
        prof.name = {GEMM_VARIANTS[15]: "gemm_bf16_pair_kernel<128,128,4,2>",
                     GEMM_VARIANTS[17]: "gemm_bf16_pipe_pair_kernel<256,128,3,4,2>",
                     GEMM_VARIANTS[26]: "gemm_bf16_pair_kernel<192,128,4,2>",
                     GEMM_VARIANTS[27]: "gemm_bf16_pair_kernel<128,192,2,4>",
                     GEMM_VARIANTS[30]: "gemm8p_kernel"}.get(prof.name, prof.name)
    with prof:
        _lib.check(lib.advgrpo_gemm_grouped(arr, len(descs), _lib.stream_ptr()))
    return [o for _, o in descs]


class Fp8Rows:
    """A matrix quantised row by row to fp8 e4m3 (quantize.hip): q [M, K] uint8 codes, scale [M] f32, x ~= scale[:, None] * q."""
    __slots__ = ("q", "scale")

    def __init__(self, q, scale):
        self.q, self.scale = q, scale

    def rows(self, start, stop):
        return Fp8Rows(self.q[start:stop], self.scale[start:stop])

    def dequant(self):
        return self.q.view(torch.float8_e4m3fn).float() * self.scale[:, None]


def quant_fp8_rows(x, out=None, split=None):
    """Per-row symmetric e4m3 quantisation of a bf16 matrix x [M, K] (row pitch free).  split = (first, period): the rows of a
    joint [B, period] layout are compacted, the `first` leading rows of every period to the front (advgrpo.h)."""
    lib = _lib.load()
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    M, K = x.shape
    if out is None:
        out = Fp8Rows(torch.empty(M, K, dtype=torch.uint8, device=x.device), torch.empty(M, dtype=torch.float32, device=x.device))
    assert out.q.shape == (M, K) and out.q.stride(1) == 1 and out.scale.is_contiguous() and out.scale.numel() == M
    sf, sp = (int(v) for v in split) if split is not None else (0, 0)
    _lib.check(lib.advgrpo_quant_fp8_rows(x.data_ptr(), x.stride(0), out.q.data_ptr(), out.q.stride(0), out.scale.data_ptr(),
                                          M, K, sf, sp, _lib.stream_ptr()))
    return out


class _As16:
    """Presents a uint8 code matrix to gemm_desc (which checks for bf16 operands and takes pointers / pitches in elements)."""
    dtype = torch.bfloat16

    def __init__(self, t):
        self.t, self.shape, self.device = t, t.shape, t.device

    def stride(self, i):
        return self.t.stride(i)

    def data_ptr(self):
        return self.t.data_ptr()


def gemm_desc_fp8(a, w, **kw):
    """gemm_desc for fp8 operands: a, w are Fp8Rows (activation rows / weight output channels).  Returns
    (descriptor, out, scales): the triple gemm_grouped_fp8 takes."""
    assert a.q.dtype == torch.uint8 and w.q.dtype == torch.uint8 and a.q.shape[1] == w.q.shape[1]
    assert "a_seg" not in kw and "aux_in" not in kw
    d, out = gemm_desc(_As16(a.q), _As16(w.q), **kw)
    sc = _lib.Fp8Scales()
    sc.a_scale, sc.w_scale = a.scale.data_ptr(), w.scale.data_ptr()
    assert a.scale.numel() >= d.M and w.scale.numel() == d.N
    return d, out, sc


def gemm_grouped_fp8(descs):
    """One launch of the eight-phase kernel on fp8 operands for one or two Linears (triples from gemm_desc_fp8)."""
    lib = _lib.load()
    arr = (_lib.GemmDesc * len(descs))(*[d for d, _, _ in descs])
    scs = (_lib.Fp8Scales * len(descs))(*[s for _, _, s in descs])
    d0 = descs[0][0]
    prof = _Prof(d0.M, d0.N, d0.K, 1, 0)
    if prof.on:
        prof.flops += sum(2.0 * d.M * d.N * d.K for d, _, _ in descs[1:])
        prof.name = "gemm8p_kernel_fp8"
    with prof:
        _lib.check(lib.advgrpo_gemm_fp8_grouped(arr, scs, len(descs), _lib.stream_ptr()))
    return [o for _, o, _ in descs]


def bmm_nt(a, w, out=None, out_dtype=torch.bfloat16, alpha=1.0):
    """Batched out[b] = alpha * a[b] @ w[b]^T; a [B,M,K], w [B,N,K] bf16."""
    lib = _lib.load()
    B, M, K = a.shape
    N = w.shape[1]
    assert a.stride(2) == 1 and w.stride(2) == 1
    if out is None:
        out = torch.empty(B, M, N, dtype=out_dtype, device=a.device)
    with _Prof(M, N, K, B, 0):
      _lib.check(lib.advgrpo_gemm_bf16(
        a.data_ptr(), a.stride(1), w.data_ptr(), w.stride(1), out.data_ptr(), out.stride(1),
        _lib.dtype_code(out.dtype), M, N, K, None, 0, float(alpha), None, 0, 0, None, 0, 0, 0, 0, 0, 0, 0,
        B, a.stride(0), w.stride(0), out.stride(0), _lib.stream_ptr()))
    return out


def attention_bias(q, k, v, num_heads, bias, scale=1.0, causal=False, out=None):
    """softmax(q k^T * scale + bias) v with bias [H,Sq,Skv] f32 shared over the batch (head dim 64)."""
    lib = _lib.load()
    B, Sq, HD = q.shape
    Skv = k.shape[1]
    assert bias.dtype == torch.float32 and bias.is_contiguous() and tuple(bias.shape) == (num_heads, Sq, Skv)
    if out is None:
        out = torch.empty(B, Sq, HD, dtype=torch.bfloat16, device=q.device)
    _lib.check(lib.advgrpo_attention_fwd_bias(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q.stride(1), k.stride(1), v.stride(1), out.stride(1),
        q.stride(0), k.stride(0), v.stride(0), out.stride(0), B, num_heads, Sq, Skv, HD // num_heads, float(scale),
        int(causal), bias.data_ptr(), _lib.stream_ptr()))
    return out


def attention(q, k, v, num_heads, scale=None, causal=False, out=None, lse=None):
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
        float(scale), int(causal), lse.data_ptr() if lse is not None else None, _lib.stream_ptr()))
    return out


def attention_fallback_count(reset=True):
    """Workgroups of the pipelined attention forwards that took the running-maximum fallback since the last reset (synchronises)."""
    import ctypes
    n = ctypes.c_longlong(0)
    _lib.check(_lib.load().advgrpo_attention_fallback_count(ctypes.byref(n), int(reset)))
    return n.value


def attention_bwd(q, k, v, o, d_o, lse, num_heads, dq, dk, dv, scale=None):
    """Gradients of ops.attention (head dim 64 or 128).  q,k,v,o,d_o: [B,S,H*hd] bf16 views; lse f32 [B,H,Sq] from the forward;
    dq/dk/dv: bf16 views of ONE packed buffer (same row / batch pitch)."""
    lib = _lib.load()
    B, Sq, HD = q.shape
    Skv = k.shape[1]
    D = HD // num_heads
    if scale is None:
        scale = D ** -0.5
    assert dq.stride(1) == dk.stride(1) == dv.stride(1) and dq.stride(0) == dk.stride(0) == dv.stride(0)
    work = torch.empty(B, num_heads, (Sq + 31) // 32, 64, dtype=torch.float32, device=q.device)    # (-L/c | -D) per 32 queries
    _lib.check(lib.advgrpo_attention_bwd(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), work.data_ptr(),
        dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), q.stride(1), k.stride(1), v.stride(1), o.stride(1),
        d_o.stride(1), dq.stride(1), q.stride(0), k.stride(0), v.stride(0), o.stride(0), d_o.stride(0), dq.stride(0),
        B, num_heads, Sq, Skv, D, float(scale), _lib.stream_ptr()))
    return dq, dk, dv


def layernorm_mod(x, out=None, w=None, b=None, scale=None, shift=None, scale2=None, shift2=None,
                  rows_per_batch=0, eps=1e-6):
    """x [M,D] bf16.  scale/shift [Bt,D] views (row pitch = stride(0)); returns out (and out2 if scale2 given)."""
    lib = _lib.load()
    M, D = x.shape
    out = torch.empty(M, D, dtype=torch.bfloat16, device=x.device) if out is None else out
    # (out2 shares out's row pitch: one ldo for both in the C entry)
    out2 = torch.empty_strided(out.shape, out.stride(), dtype=out.dtype, device=out.device) if scale2 is not None else None
    ms = scale.stride(0) if scale is not None else 0
    if scale is not None:
        assert shift.stride(0) == ms and scale.stride(1) == 1
    if scale2 is not None:
        assert scale2.stride(0) == ms and shift2.stride(0) == ms
    dp = lambda t: t.data_ptr() if t is not None else None
    _lib.check(lib.advgrpo_layernorm_mod(x.data_ptr(), x.stride(0), out.data_ptr(), dp(out2), out.stride(0), dp(w),
                                         dp(b), dp(scale), dp(shift), dp(scale2), dp(shift2), ms, int(rows_per_batch),
                                         M, D, float(eps), _lib.stream_ptr()))
    return (out, out2) if scale2 is not None else out


def layernorm_mod_fp8(x, q, q2=None, out=None, out2=None, w=None, b=None, scale=None, shift=None, scale2=None, shift2=None,
                      rows_per_batch=0, eps=1e-6):
    """layernorm_mod whose output(s) leave as fp8 rows (q, q2: Fp8Rows of M rows) for the fp8 Linears; out / out2: the bf16
    tensors to write as well, or None to skip them.  The codes equal quant_fp8_rows(layernorm_mod(...)) bit for bit."""
    lib = _lib.load()
    M, D = x.shape
    assert q.q.shape == (M, D) and q.q.stride(1) == 1 and q.scale.numel() == M and (q2 is None) == (scale2 is None)
    assert q2 is None or (q2.q.shape == (M, D) and q2.q.stride(0) == q.q.stride(0))
    assert out2 is None or (out is not None and out2.stride() == out.stride())
    ms = scale.stride(0) if scale is not None else 0
    if scale is not None:
        assert shift.stride(0) == ms and scale.stride(1) == 1
    if scale2 is not None:
        assert scale2.stride(0) == ms and shift2.stride(0) == ms
    dp = lambda t: t.data_ptr() if t is not None else None
    _lib.check(lib.advgrpo_layernorm_mod_fp8(x.data_ptr(), x.stride(0), dp(out), dp(out2), out.stride(0) if out is not None else D,
                                             dp(w), dp(b), dp(scale), dp(shift), dp(scale2), dp(shift2), ms, int(rows_per_batch),
                                             M, D, float(eps), q.q.data_ptr(), q.scale.data_ptr(),
                                             q2.q.data_ptr() if q2 is not None else None,
                                             q2.scale.data_ptr() if q2 is not None else None, q.q.stride(0), _lib.stream_ptr()))
    return out, out2


def layernorm_mod_pair(a, b):
    """Two layernorm_mod (or two layernorm_mod_fp8) problems in ONE launch: a, b are dicts of that function's keyword arguments
    plus "x" (and "q" / "q2" for the fp8 form: then "out" / "out2" are optional and may stay None).  Returns the two results in the
    single call's form.  Bit-identical to two calls; saves the launch-bound text-stream launch of an MMDiT block."""
    lib = _lib.load()
    descs, keep, rets = [], [], []
    for kw in (a, b):
        x = kw["x"]
        M, D = x.shape
        fp8 = kw.get("q") is not None
        scale, shift, scale2, shift2 = kw.get("scale"), kw.get("shift"), kw.get("scale2"), kw.get("shift2")
        out, out2 = kw.get("out"), kw.get("out2")
        q, q2 = kw.get("q"), kw.get("q2")
        if not fp8:
            out = torch.empty(M, D, dtype=torch.bfloat16, device=x.device) if out is None else out
            if scale2 is not None and out2 is None:
                out2 = torch.empty_strided(out.shape, out.stride(), dtype=out.dtype, device=out.device)
        else:
            assert q.q.shape == (M, D) and q.q.stride(1) == 1 and q.scale.numel() == M and (q2 is None) == (scale2 is None)
            assert q2 is None or (q2.q.shape == (M, D) and q2.q.stride(0) == q.q.stride(0))
        assert out2 is None or (out is not None and out2.stride() == out.stride())
        ms = scale.stride(0) if scale is not None else 0
        if scale is not None:
            assert shift.stride(0) == ms and scale.stride(1) == 1
        if scale2 is not None:
            assert scale2.stride(0) == ms and shift2.stride(0) == ms
        dp = lambda t: t.data_ptr() if t is not None else None
        d = _lib.LnDesc(x.data_ptr(), x.stride(0), dp(out), dp(out2), out.stride(0) if out is not None else D, dp(kw.get("w")),
                        dp(kw.get("b")), dp(scale), dp(shift), dp(scale2), dp(shift2), ms, int(kw.get("rows_per_batch", 0)), M, D,
                        float(kw.get("eps", 1e-6)), q.q.data_ptr() if fp8 else None, q.scale.data_ptr() if fp8 else None,
                        q2.q.data_ptr() if q2 is not None else None, q2.scale.data_ptr() if q2 is not None else None,
                        q.q.stride(0) if fp8 else 0)
        descs.append(d)
        keep.append((x, out, out2, scale, shift, scale2, shift2, q, q2, kw.get("w"), kw.get("b")))
        rets.append((out, out2) if (fp8 or scale2 is not None) else out)
    _lib.check(lib.advgrpo_layernorm_mod_pair(ctypes.byref(descs[0]), ctypes.byref(descs[1]), _lib.stream_ptr()))
    del keep
    return rets[0], rets[1]


def rmsnorm_rows(x, w, eps=1e-6, out=None):
    """T5LayerNorm over the rows of x [M,D] bf16 (D <= 4096)."""
    lib = _lib.load()
    M, D = x.shape
    out = torch.empty(M, D, dtype=torch.bfloat16, device=x.device) if out is None else out
    _lib.check(lib.advgrpo_rmsnorm_rows(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), w.data_ptr(), M, D,
                                        float(eps), _lib.stream_ptr()))
    return out


def rmsnorm_heads(buf, col0, nheads, weight, heads_per_weight, eps=1e-6, seg=None, M=None, rs_out=None):
    """In place on buf [rows, ld] bf16; weight [(nheads/heads_per_weight), 64] bf16.
    rs_out: optional f32 [rows, nheads] receiving 1/rms (indexed by the mapped row) for the backward."""
    lib = _lib.load()
    seg_rows, seg_stride, seg_off = seg if seg is not None else (0, 0, 0)
    M = buf.shape[0] if M is None else M
    _lib.check(lib.advgrpo_rmsnorm_heads(buf.data_ptr(), buf.stride(0), M, col0, nheads, weight.data_ptr(),
                                         heads_per_weight, float(eps), int(seg_rows), int(seg_stride), int(seg_off),
                                         rs_out.data_ptr() if rs_out is not None else None, _lib.stream_ptr()))
    return buf


def qk_norm_rope(buf, S, n_first, nheads, head_dim, w_first, w_rest, heads_per_weight, rope=None, eps=1e-6, col0=0, rs_out=None):
    """In place on the joint QKV buffer buf [B * S, ld] bf16: per-head RMSNorm (weights w_first for the first n_first token
    rows of every sample, w_rest for the others; [nheads / heads_per_weight, head_dim] bf16 each) of the nheads heads at
    columns [col0, col0 + nheads * head_dim), then the rotary embedding rope [S, head_dim] f32 ((cos, sin) pairs; None: none)."""
    lib = _lib.load()
    assert buf.dtype == torch.bfloat16 and buf.dim() == 2 and buf.stride(1) == 1 and buf.shape[0] % S == 0
    assert w_first.is_contiguous() and w_rest.is_contiguous() and w_first.dtype == torch.bfloat16 and w_rest.dtype == torch.bfloat16
    assert rope is None or (rope.dtype == torch.float32 and rope.is_contiguous() and tuple(rope.shape) == (S, head_dim))
    _lib.check(lib.advgrpo_qk_norm_rope(buf.data_ptr(), buf.stride(0), buf.shape[0], int(S), int(n_first), int(col0), int(nheads),
                                        int(head_dim), w_first.data_ptr(), w_rest.data_ptr(), int(heads_per_weight), float(eps),
                                        rope.data_ptr() if rope is not None else None,
                                        rs_out.data_ptr() if rs_out is not None else None, _lib.stream_ptr()))
    return buf


def timestep_embedding(t, dim=256):
    lib = _lib.load()
    t = t.float().contiguous()
    out = torch.empty(t.shape[0], dim, dtype=torch.bfloat16, device=t.device)
    _lib.check(lib.advgrpo_timestep_embedding(t.data_ptr(), out.data_ptr(), t.shape[0], dim, _lib.stream_ptr()))
    return out


def unary(x, act=None, x2=None):
    lib = _lib.load()
    y = torch.empty_like(x)
    _lib.check(lib.advgrpo_unary(x.data_ptr(), x2.data_ptr() if x2 is not None else None, y.data_ptr(), x.numel(),
                                 ACT[act], _lib.stream_ptr()))
    return y


def patchify(x):
    lib = _lib.load()
    B, C, H, W = x.shape
    out = torch.empty(B * (H // 2) * (W // 2), C * 4, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.advgrpo_patchify(_lib.ptr(x), _lib.dtype_code(x.dtype), out.data_ptr(), B, C, H, W,
                                    _lib.stream_ptr()))
    return out


def unpatchify(tokens, B, C, H, W, out_dtype=torch.bfloat16):
    lib = _lib.load()
    out = torch.empty(B, C, H, W, dtype=out_dtype, device=tokens.device)
    _lib.check(lib.advgrpo_unpatchify(_lib.ptr(tokens), out.data_ptr(), _lib.dtype_code(out_dtype), B, C, H, W,
                                      _lib.stream_ptr()))
    return out


def cached(cache, key, make):
    """Fill-once device-tensor caches that are read from several HIP streams (rollouts of two prompt groups, the scoring
    stream): the tensor is built on the first caller's stream and that stream is drained ONCE before the entry is
    published, so a reader on another stream never sees it half written.  Entries are never evicted."""
    t = cache.get(key)
    if t is None:
        t = make()
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
        cache[key] = t
    return t


# ---- side streams that really run beside their partners.  HIP maps streams onto a few hardware queues (4 by default,
# GPU_MAX_HW_QUEUES) in the order of their first use, round-robin: whether a side stream shares a queue with the stream it is
# meant to overlap -- and is then simply serialised behind it -- depends on how many streams the process used before.
# Measured (round 4): bench.py's epoch leg ran its G-step micro-steps in 105 ms where a fresh process took 94 ms, because the
# rollout leg in front had used one more stream and the adapter-gradient stream landed on the main stream's queue.

def _overlaps(a, b, dev):
    """Do single-workgroup kernels on streams a and b run at the same time?  (group-advantage launches: ~0.15 ms each, one CU)"""
    from . import stat_tracking
    r = torch.rand(768, 2, device=dev)
    g = (torch.arange(768, device=dev) // 8).to(torch.int32)
    n = 6

    def burst(streams):
        torch.cuda.synchronize(dev)
        t0 = torch.cuda.Event(enable_timing=True)
        ends = []
        t0.record(torch.cuda.current_stream(dev))
        for st in streams:
            st.wait_event(t0)
            with torch.cuda.stream(st):
                for _ in range(n):
                    stat_tracking.group_advantage(r, g, True)
                e = torch.cuda.Event(enable_timing=True)
                e.record(st)
                ends.append(e)
        torch.cuda.synchronize(dev)
        return max(t0.elapsed_time(e) for e in ends)
    burst([a, b])                                   # first use of both streams (queue assignment), code load
    one, both = min(burst([a]) for _ in range(2)), min(burst([a, b]) for _ in range(2))
    return both < 1.5 * one


_CONCURRENT_LOCK = threading.Lock()


def concurrent_stream(device, partners=()):
    """A torch stream on `device` that is MEASURED to run concurrently with every stream in `partners` (default: the current
    stream): candidates from torch's pool are tried until one does (at most 8).  With more live streams than hardware queues
    not every pair can be concurrent -- name the partners that matter.

    The measurement drains the DEVICE (torch.cuda.synchronize) around micro-bursts of single-workgroup kernels, so it is only
    meaningful -- and only cheap -- while nothing else is enqueuing work: call it at construction time (Trainer / scorer / model
    __init__), never from a worker thread beside running rollouts.  Calls are serialised by a module lock; if no
    candidate overlaps, a warning says so and the last candidate is returned (work on it is then serialised behind a partner: slower,
    not wrong).  The result is NOT cached across calls: a stream measured early in a process did not stay concurrent with the
    launch stream once other streams had come and gone (bench.py's epoch leg behind its pricing legs: G-step micro-steps of 109 ms
    with the reused stream against 88 ms with one measured when the model was built; round 5, same box)."""
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        return None
    partners = list(partners or ()) or [torch.cuda.current_stream(dev)]
    with _CONCURRENT_LOCK:
        cand, ok = None, False
        for _ in range(8):
            cand = torch.cuda.Stream(device=dev)
            if all(_overlaps(p, cand, dev) for p in partners):
                ok = True
                break
        if not ok:
            import warnings
            warnings.warn(f"adv_grpo_amd.ops.concurrent_stream: none of 8 candidate streams ran beside {len(partners)} partner stream(s) on {dev} "
                          "(all hardware queues shared? GPU_MAX_HW_QUEUES is read when the HIP runtime starts); side work will be serialised")
        return cand


_ZERO_PAGE = {}


def zero_page(device):
    return cached(_ZERO_PAGE, str(device), lambda: torch.zeros(256, dtype=torch.bfloat16, device=device))


def conv3x3(x, w, bias=None, upsample=False, act=None, residual=None, out_dtype=torch.bfloat16):
    """x NHWC bf16 [B,Hin,Win,Cin]; w [Cout, 9*Cin] bf16 (k = (ky*3+kx)*Cin + c) -> [B,Hout,Wout,Cout]."""
    lib = _lib.load()
    B, Hin, Win, Cin = x.shape
    Cout = w.shape[0]
    Hout, Wout = (Hin * 2, Win * 2) if upsample else (Hin, Win)
    y = torch.empty(B, Hout, Wout, Cout, dtype=out_dtype, device=x.device)
    with _Prof(B * Hout * Wout, Cout, 9 * Cin, 1, 1):
      _lib.check(lib.advgrpo_conv3x3_nhwc(_lib.ptr(x), _lib.ptr(w), y.data_ptr(), _lib.dtype_code(out_dtype), B, Hout, Wout,
                                        Cin, Cout, int(upsample), _lib.ptr(bias), ACT[act], _lib.ptr(residual),
                                        zero_page(x.device).data_ptr(), _lib.stream_ptr()))
    return y


def groupnorm_nhwc(x, weight, bias, groups=32, eps=1e-6, silu=False):
    lib = _lib.load()
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    stats = torch.empty(lib.advgrpo_groupnorm_scratch_bytes(B, HW, groups) // 8, dtype=torch.float64, device=x.device)
    _lib.check(lib.advgrpo_groupnorm_nhwc(_lib.ptr(x), y.data_ptr(), stats.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), B,
                                          HW, C, groups, float(eps), int(silu), _lib.stream_ptr()))
    return y


# ---- split-bf16 ("bf16x3") VAE mode: f32 between the matrix products, operands as [hi|hi|lo] x [hi|lo|hi] (include/advgrpo.h)
def split_x3(x, order=0, bias=None):
    """f32 [..., K] (+ bias[K]) -> bf16 [..., 3K]; order 0 = left operand (activations), 1 = right operand (weights),
    2 = [hi | unwritten | lo] for the activations of conv3x3_x3 with Cout >= 128 only."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous()
    K = x.shape[-1]
    out = torch.empty(*x.shape[:-1], 3 * K, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.advgrpo_split_bf16x3(x.data_ptr(), _lib.ptr(bias), out.data_ptr(), x.numel() // K, K, int(order),
                                        _lib.stream_ptr()))
    return out


def conv3x3_x3(x3, w3, bias=None, upsample=False, act=None, residual=None):
    """x3 NHWC split bf16 [B,Hin,Win,3C]; w3 [Cout, 9*3C]; bias / residual f32 -> f32 [B,Hout,Wout,Cout]."""
    lib = _lib.load()
    B, Hin, Win, Cin3 = x3.shape
    Cout = w3.shape[0]
    Hout, Wout = (Hin * 2, Win * 2) if upsample else (Hin, Win)
    assert bias is None or bias.dtype == torch.float32
    assert residual is None or (residual.dtype == torch.float32 and residual.is_contiguous())
    y = torch.empty(B, Hout, Wout, Cout, dtype=torch.float32, device=x3.device)
    with _Prof(B * Hout * Wout, Cout, 9 * Cin3, 1, 1):
      _lib.check(lib.advgrpo_conv3x3_nhwc_x3(_lib.ptr(x3), _lib.ptr(w3), y.data_ptr(), B, Hout, Wout, Cin3, Cout, int(upsample),
                                           _lib.ptr(bias), ACT[act], _lib.ptr(residual), zero_page(x3.device).data_ptr(),
                                           _lib.stream_ptr()))
    return y


def groupnorm_nhwc_x3(x, weight, bias, groups=32, eps=1e-6, silu=False, pair_only=False):
    """f32 NHWC in, f32 affine -> split bf16 [..., 3C]; pair_only: the middle third stays unwritten (the output feeds
    conv3x3_x3 with Cout >= 128, which reads hi and lo only)."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and weight.dtype == torch.float32
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    y = torch.empty(*x.shape[:-1], 3 * C, dtype=torch.bfloat16, device=x.device)
    stats = torch.empty(lib.advgrpo_groupnorm_scratch_bytes(B, HW, groups) // 8, dtype=torch.float64, device=x.device)
    _lib.check(lib.advgrpo_groupnorm_nhwc_x3(x.data_ptr(), y.data_ptr(), stats.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), B,
                                             HW, C, groups, float(eps), int(silu), int(pair_only), _lib.stream_ptr()))
    return y


def latents_mix_to_nhwc(z, cpad, inv_std, mean, P, bias, x3=False):
    """Qwen-Image VAE: (z / inv_std + mean) through the 1x1 post_quant_conv (P [C,C], bias [C], f32) -> NHWC bf16 [B,H,W,cpad]
    (x3: split rows [B,H,W,3 cpad])."""
    lib = _lib.load()
    B, C, H, W = z.shape
    out = torch.empty(B, H, W, (3 if x3 else 1) * cpad, dtype=torch.bfloat16, device=z.device)
    z = z.contiguous()
    for t in (inv_std, mean, P, bias):
        assert t.dtype == torch.float32 and t.is_contiguous()
    _lib.check(lib.advgrpo_latents_mix_to_nhwc(_lib.ptr(z), _lib.dtype_code(z.dtype), out.data_ptr(), int(x3), B, C, H, W, cpad,
                                               inv_std.data_ptr(), mean.data_ptr(), P.data_ptr(), bias.data_ptr(), _lib.stream_ptr()))
    return out


def rmsnorm_nhwc(x, gamma, mult, silu=False, out="bf16"):
    """Per-pixel RMS norm over the last (channel) axis: x / max(||x||, 1e-12) * mult * gamma (+ SiLU).  x f32 or bf16 [..., C];
    out "bf16" -> bf16 [..., C]; "x3" -> split rows [..., 3C] ([hi | hi | lo]); "x3pair" -> [hi | unwritten | lo]."""
    lib = _lib.load()
    assert x.is_contiguous() and gamma.dtype == torch.float32 and gamma.numel() == x.shape[-1]
    C = x.shape[-1]
    mode = {"bf16": 0, "x3": 1, "x3pair": 2}[out]
    y = torch.empty(*x.shape[:-1], C if mode == 0 else 3 * C, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.advgrpo_rmsnorm_nhwc(x.data_ptr(), _lib.dtype_code(x.dtype), y.data_ptr(), gamma.data_ptr(), x.numel() // C, C,
                                        float(mult), int(silu), mode, _lib.stream_ptr()))
    return y


# ---- "f16x2": fp16-exact decoder weights (include/advgrpo.h) -- fp16 pair activations, one-piece fp16 weights, two products
def split_f16x2(x, prescale=1.0, bias=None):
    """f32 [..., K] (+ bias[K]) -> [..., 3K] 16-bit: thirds [f16 hi | unwritten | f16 lo] of prescale * x (prescale: a power of two)."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous()
    K = x.shape[-1]
    out = torch.empty(*x.shape[:-1], 3 * K, dtype=torch.bfloat16, device=x.device)      # (16-bit container; the values are fp16)
    _lib.check(lib.advgrpo_split_f16x2(x.data_ptr(), _lib.ptr(bias), out.data_ptr(), x.numel() // K, K, float(prescale), _lib.stream_ptr()))
    return out


CONV_F16X2_STAT_ROWS = 16         # ADVGRPO_CONV_F16X2_STAT_ROWS


def groupnorm_nhwc_f16x2(x, weight, bias, groups=32, eps=1e-6, silu=False, prescale=1.0, tile_stats=None):
    """tile_stats: the block sums conv3x3_f16x2(..., gn_stats=True) attached to x (`x.gn_tile_stats`): the statistics kernel's
    pass over x is skipped."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and weight.dtype == torch.float32
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    y = torch.empty(*x.shape[:-1], 3 * C, dtype=torch.bfloat16, device=x.device)
    stats = torch.empty(lib.advgrpo_groupnorm_scratch_bytes(B, HW, groups) // 8, dtype=torch.float64, device=x.device)
    assert tile_stats is None or (tile_stats.dtype == torch.float32 and tile_stats.numel() == B * HW // CONV_F16X2_STAT_ROWS * (C // 4) * 2)
    _lib.check(lib.advgrpo_groupnorm_nhwc_f16x2(x.data_ptr(), y.data_ptr(), stats.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), B, HW, C,
                                                groups, float(eps), int(silu), float(prescale), _lib.ptr(tile_stats),
                                                CONV_F16X2_STAT_ROWS, _lib.stream_ptr()))
    return y


def conv3x3_f16x2_pair(x2, w16, prescale, bias=None, upsample=False, act=None, residual=None, alpha=1.0, single=False):
    """conv3x3_f16x2 whose output leaves as the fp16-pair rows of prescale * y ([B,Hout,Wout,3 Cout], = split_f16x2(y, prescale) bit for
    bit) instead of f32 y: for an output that only the next f16x2 convolution reads."""
    lib = _lib.load()
    B, Hin, Win, Cin3 = x2.shape
    Cout = w16.shape[0]
    assert w16.dtype == torch.float16 and w16.is_contiguous() and w16.shape[1] == 3 * Cin3
    assert bias is None or bias.dtype == torch.float32
    assert residual is None or (residual.dtype == torch.float32 and residual.is_contiguous())
    Hout, Wout = (Hin * 2, Win * 2) if upsample else (Hin, Win)
    out = torch.empty(B, Hout, Wout, 3 * Cout, dtype=torch.bfloat16, device=x2.device)      # (16-bit container; the values are fp16)
    with _Prof(B * Hout * Wout, Cout, 2 * 3 * Cin3, 1, 1):
        _lib.check((lib.advgrpo_conv3x3_nhwc_f16x1_pair if single else lib.advgrpo_conv3x3_nhwc_f16x2_pair)(_lib.ptr(x2), _lib.ptr(w16), out.data_ptr(), float(prescale), B, Hout, Wout, Cin3, Cout,
                                                       int(upsample), _lib.ptr(bias), ACT[act], _lib.ptr(residual),
                                                       zero_page(x2.device).data_ptr(), float(alpha), _lib.stream_ptr()))
    return out


def conv3x3_f16x2(x2, w16, bias=None, upsample=False, act=None, residual=None, alpha=1.0, gn_stats=False, bf16_pieces=False, single=False):
    """x2 NHWC fp16-pair rows [B,Hin,Win,3C]; w16 [Cout, 9C] fp16 (k = (ky*3+kx)*C + c); bias / residual f32 -> f32 [B,Hout,Wout,Cout].
    bf16_pieces: the "bf16x2" form -- x2 = bf16 [hi | unwritten | lo] rows, w16 one bf16 piece (weights exact in bf16).
    gn_stats: the epilogue also leaves the 16-pixel x 4-channel block sums of the GroupNorm that reads the output, as `y.gn_tile_stats` (an attribute
    of THIS tensor object: views do not carry it)."""
    lib = _lib.load()
    B, Hin, Win, Cin3 = x2.shape
    Cout = w16.shape[0]
    assert w16.dtype == (torch.bfloat16 if bf16_pieces else torch.float16) and w16.is_contiguous() and w16.shape[1] == 3 * Cin3
    assert not (single and bf16_pieces)
    # single: the TF32-class "f16x1" form (the hi product only; include/advgrpo.h)
    fn = lib.advgrpo_conv3x3_nhwc_bf16x2 if bf16_pieces else (lib.advgrpo_conv3x3_nhwc_f16x1 if single else lib.advgrpo_conv3x3_nhwc_f16x2)
    Hout, Wout = (Hin * 2, Win * 2) if upsample else (Hin, Win)
    assert bias is None or bias.dtype == torch.float32
    assert residual is None or (residual.dtype == torch.float32 and residual.is_contiguous())
    y = torch.empty(B, Hout, Wout, Cout, dtype=torch.float32, device=x2.device)
    part = None
    if gn_stats and (Hout * Wout) % CONV_F16X2_STAT_ROWS == 0:
        part = torch.empty(B * Hout * Wout // CONV_F16X2_STAT_ROWS, Cout // 4, 2, dtype=torch.float32, device=x2.device)
    with _Prof(B * Hout * Wout, Cout, 2 * 3 * Cin3, 1, 1):                # two products per tap over C channels: K_eff = 2 x 9 C
        _lib.check(fn(_lib.ptr(x2), _lib.ptr(w16), y.data_ptr(), B, Hout, Wout, Cin3, Cout, int(upsample),
                      _lib.ptr(bias), ACT[act], _lib.ptr(residual), zero_page(x2.device).data_ptr(),
                      float(alpha), _lib.ptr(part), _lib.stream_ptr()))
    if part is not None:
        y.gn_tile_stats = part
    return y


def softmax_rows_x3(s):
    """f32 [..., n] -> softmax rows as split bf16 [..., 3n]."""
    lib = _lib.load()
    assert s.dtype == torch.float32 and s.is_contiguous()
    n = s.shape[-1]
    out = torch.empty(*s.shape[:-1], 3 * n, dtype=torch.bfloat16, device=s.device)
    _lib.check(lib.advgrpo_softmax_rows_x3(s.data_ptr(), out.data_ptr(), s.numel() // n, n, _lib.stream_ptr()))
    return out


def add_rows_f32(a, b=None, bias=None):
    lib = _lib.load()
    assert a.dtype == torch.float32 and a.is_contiguous() and (b is None or (b.dtype == torch.float32 and b.is_contiguous()))
    C = a.shape[-1]
    y = torch.empty_like(a)
    _lib.check(lib.advgrpo_add_rows_f32(a.data_ptr(), _lib.ptr(b), _lib.ptr(bias), y.data_ptr(), a.numel() // C, C,
                                        _lib.stream_ptr()))
    return y


# ---- bf16x3 ViT towers (csrc/x3.hip): row kernels between two split-bf16 matrix products
X3_ACT = {None: 0, "gelu": 1, "quick_gelu": 2}


def layernorm_x3(x, w, b, eps=1e-5):
    """x f32 [M, D] -> LayerNorm with f32 affine -> split rows bf16 [M, 3D] (left operand)."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and w.dtype == torch.float32 and b.dtype == torch.float32
    M, D = x.shape
    out = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.advgrpo_layernorm_x3(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, D, float(eps), _lib.stream_ptr()))
    return out


def split_act_x3(x, order=0, bias=None, act=None):
    """split(act(x + bias)): x f32 [..., K] -> bf16 [..., 3K]; act None | "gelu" (erf) | "quick_gelu"."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and (bias is None or bias.dtype == torch.float32)
    K = x.shape[-1]
    out = torch.empty(*x.shape[:-1], 3 * K, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.advgrpo_split_act_bf16x3(x.data_ptr(), _lib.ptr(bias), out.data_ptr(), x.numel() // K, K, int(order), X3_ACT[act],
                                            _lib.stream_ptr()))
    return out


def softmax_rows_x3_masked(s, n_valid, causal_period=0, alpha=1.0):
    """softmax(alpha * s) over the first n_valid keys (causal: row r keeps r % causal_period + 1 of them) -> split rows [rows, 3n]."""
    lib = _lib.load()
    assert s.dtype == torch.float32 and s.is_contiguous()
    n = s.shape[-1]
    out = torch.empty(*s.shape[:-1], 3 * n, dtype=torch.bfloat16, device=s.device)
    _lib.check(lib.advgrpo_softmax_rows_x3_masked(s.data_ptr(), out.data_ptr(), s.numel() // n, n, int(n_valid), int(causal_period),
                                                  float(alpha), _lib.stream_ptr()))
    return out


def latents_to_nhwc_x3(z, cpad, scaling_factor, shift_factor):
    lib = _lib.load()
    B, C, H, W = z.shape
    out = torch.empty(B, H, W, 3 * cpad, dtype=torch.bfloat16, device=z.device)
    z = z.contiguous()
    _lib.check(lib.advgrpo_latents_to_nhwc_x3(_lib.ptr(z), _lib.dtype_code(z.dtype), out.data_ptr(), B, C, H, W,
                                              cpad, float(scaling_factor), float(shift_factor), _lib.stream_ptr()))
    return out


def softmax_rows_(s):
    lib = _lib.load()
    n = s.shape[-1]
    _lib.check(lib.advgrpo_softmax_rows(_lib.ptr(s), s.numel() // n, n, _lib.stream_ptr()))
    return s


def latents_to_nhwc(z, cpad, scaling_factor, shift_factor):
    lib = _lib.load()
    B, C, H, W = z.shape
    out = torch.empty(B, H, W, cpad, dtype=torch.bfloat16, device=z.device)
    z = z.contiguous()
    _lib.check(lib.advgrpo_latents_to_nhwc(_lib.ptr(z), _lib.dtype_code(z.dtype), out.data_ptr(), B, C, H, W,
                                           cpad, float(scaling_factor), float(shift_factor), _lib.stream_ptr()))
    return out


def image_postprocess(y):
    """y NHWC [B,H,W,ldc] (bf16/f32) -> [B,3,H,W] f32 in [0,1]."""
    lib = _lib.load()
    B, H, W, ldc = y.shape
    img = torch.empty(B, 3, H, W, dtype=torch.float32, device=y.device)
    _lib.check(lib.advgrpo_image_postprocess(_lib.ptr(y), _lib.dtype_code(y.dtype), ldc, img.data_ptr(), B, H, W,
                                             _lib.stream_ptr()))
    return img


# ------------------------------------------------------------------ G-step (training) ops
ACT_D = {"dgelu_tanh": 5, "dgelu": 6, "mul_aux": 7}   # "mul_aux": y *= aux_in (gated feed-forward)


def gemm_train(a, w, bias=None, act=None, alpha=1.0, gate=None, gate_rows=0, residual=None, out=None,
               out_dtype=torch.bfloat16, aux_out=None, aux_in=None, splitk=1, M=None):
    """GEMM with the training extras (see advgrpo_gemm_bf16_train): aux_out = pre-activation copy,
    act in {"dgelu_tanh","dgelu"} multiplies by the activation derivative at aux_in, splitk > 1 accumulates
    atomically into an f32 `out` (which must be given and hold the running sum)."""
    lib = _lib.load()
    K = a.shape[1]
    M = a.shape[0] if M is None else M
    N = w.shape[0]
    if out is None:
        assert splitk == 1
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    aux = aux_out if aux_out is not None else aux_in
    code = ACT_D[act] if act in ACT_D else ACT[act]
    with _Prof(M, N, K, splitk, 0):
      _lib.check(lib.advgrpo_gemm_bf16_train(
        a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), _lib.dtype_code(out.dtype),
        M, N, K, _lib.ptr(bias), code, float(alpha), gate.data_ptr() if gate is not None else None,
        gate.stride(0) if gate is not None else 0, int(gate_rows), residual.data_ptr() if residual is not None else None,
        residual.stride(0) if residual is not None else 0, aux_out.data_ptr() if aux_out is not None else None,
        aux_in.data_ptr() if aux_in is not None else None, aux.stride(0) if aux is not None else 0, int(splitk),
        _lib.stream_ptr()))
    return out


def gemm_tn(P, Q, out, alpha=1.0, M=None, p_seg=None, q_seg=None, transpose_out=False):
    """out[n1, n2] (or out[n2, n1] with transpose_out) += alpha * sum_m P[row_p(m), n1] * Q[row_q(m), n2]; P [.., N1] and
    Q [.., 64] bf16 token-major views (last dim contiguous), out f32 holding the running sum."""
    lib = _lib.load()
    assert P.dtype == torch.bfloat16 and Q.dtype == torch.bfloat16 and out.dtype == torch.float32
    assert P.stride(1) == 1 and Q.stride(1) == 1 and out.stride(1) == 1
    M = P.shape[0] if M is None else M
    ps = p_seg if p_seg is not None else (0, 0, 0)
    qs = q_seg if q_seg is not None else (0, 0, 0)
    need = lib.advgrpo_gemm_tn_workspace_bytes(M, P.shape[1])
    wkey = (str(P.device), _lib.stream_ptr())         # one workspace per launch stream (the adapter gradients run on a side
    ws = _TN_WS.get(wkey)                             # stream): allocated, reused and re-grown in that stream's order only
    if ws is None or ws.numel() < need:
        ws = _TN_WS[wkey] = torch.empty(need, dtype=torch.uint8, device=P.device)
    _lib.check(lib.advgrpo_gemm_tn_f32acc(P.data_ptr(), P.stride(0), int(ps[0]), int(ps[1]), int(ps[2]), Q.data_ptr(),
                                          Q.stride(0), int(qs[0]), int(qs[1]), int(qs[2]), out.data_ptr(), out.stride(0),
                                          int(transpose_out), M, P.shape[1], Q.shape[1], float(alpha), ws.data_ptr(),
                                          _lib.stream_ptr()))
    return out


_TN_WS = {}
_TNG_WS = {}


def tn_desc(P, Q, out, alpha=1.0, M=None, p_seg=None, q_seg=None, transpose_out=False):
    """One problem of gemm_tn_grouped: out[n1, n2] (or out[n2, n1] with transpose_out) += alpha * sum_m P[row_p(m), n1] * Q[row_q(m), n2];
    P [.., N1] and Q [.., 64] bf16 token-major views, out f32 holding the running sum."""
    assert P.dtype == torch.bfloat16 and Q.dtype == torch.bfloat16 and out.dtype == torch.float32
    assert P.stride(1) == 1 and Q.stride(1) == 1 and out.stride(1) == 1 and Q.shape[1] == 64
    d = _lib.TnDesc()
    ps, qs = p_seg or (0, 0, 0), q_seg or (0, 0, 0)
    d.P, d.ldp, d.p_seg_rows, d.p_seg_stride, d.p_seg_off = P.data_ptr(), P.stride(0), int(ps[0]), int(ps[1]), int(ps[2])
    d.Q, d.ldq, d.q_seg_rows, d.q_seg_stride, d.q_seg_off = Q.data_ptr(), Q.stride(0), int(qs[0]), int(qs[1]), int(qs[2])
    d.C, d.ldc, d.transpose_out = out.data_ptr(), out.stride(0), int(transpose_out)
    d.M, d.N1, d.alpha = (P.shape[0] if M is None else M), P.shape[1], float(alpha)
    d._keep = (P, Q, out)
    return d


def gemm_tn_grouped(descs):
    """All token-contracted products of an adapter group in one launch (csrc/gemm_tn.hip, grouped form): up to 12 descs from tn_desc.
    One workspace per launch stream, used by nothing else (its arrival counters stay zero between launches)."""
    lib = _lib.load()
    arr = (_lib.TnDesc * len(descs))(*descs)
    need = int(lib.advgrpo_gemm_tn_grouped_workspace_bytes(arr, len(descs)))
    dev = descs[0]._keep[0].device
    wkey = (str(dev), _lib.stream_ptr())
    ws = _TNG_WS.get(wkey)
    if ws is None or ws.numel() < need:
        ws = _TNG_WS[wkey] = torch.zeros(max(need, 1 << 20), dtype=torch.uint8, device=dev)      # (zeroed ONCE, on this stream)
    _lib.check(lib.advgrpo_gemm_tn_grouped(arr, len(descs), ws.data_ptr(), ws.numel(), 1, _lib.stream_ptr()))


def transpose(x, R=None, seg=None, pad_to=64, out=None):
    """x [rows, C] bf16 (row pitch = stride(0)) -> [C, Rpad] with Rpad = R rounded up to `pad_to` (zero filled).
    R rows are taken through the row-segment map `seg` = (seg_rows, seg_stride, seg_off) when given."""
    lib = _lib.load()
    C = x.shape[1]
    R = x.shape[0] if R is None else R
    Rpad = (R + pad_to - 1) // pad_to * pad_to
    out = torch.empty(C, Rpad, dtype=torch.bfloat16, device=x.device) if out is None else out
    sr, ss, so = seg if seg is not None else (0, 0, 0)
    _lib.check(lib.advgrpo_transpose_bf16(x.data_ptr(), out.data_ptr(), R, C, x.stride(0), out.stride(0), Rpad, int(sr),
                                          int(ss), int(so), _lib.stream_ptr()))
    return out


def layernorm_mod_bwd(x, dy0, scale0=None, dy1=None, scale1=None, dres=None, rows_per_batch=0, eps=1e-6, out=None, gates=None):
    """gates: up to two [G, D] bf16 gate views (rows pitch stride(0), the same for both) -> returns (dx, [gate_k * dx ...]): the gated copies
    the data-gradient GEMMs of the next gated projections read, written by the same pass (bit for bit gate_mul(dx, gate_k))."""
    lib = _lib.load()
    M, D = x.shape
    out = torch.empty(M, D, dtype=torch.bfloat16, device=x.device) if out is None else out
    ms = scale0.stride(0) if scale0 is not None else 0
    dp = lambda t: t.data_ptr() if t is not None else None
    if gates:
        assert 1 <= len(gates) <= 2 and all(g.stride(0) == gates[0].stride(0) and g.stride(1) == 1 for g in gates)
        gouts = [torch.empty(M, D, dtype=torch.bfloat16, device=x.device) for _ in gates]
        _lib.check(lib.advgrpo_layernorm_mod_bwd_gated(
            x.data_ptr(), x.stride(0), dy0.data_ptr(), dp(dy1), dy0.stride(0), dp(scale0), dp(scale1), ms, int(rows_per_batch), dp(dres),
            out.data_ptr(), out.stride(0), M, D, float(eps), gates[0].data_ptr(), gouts[0].data_ptr(),
            gates[1].data_ptr() if len(gates) > 1 else None, gouts[1].data_ptr() if len(gates) > 1 else None, gates[0].stride(0),
            _lib.stream_ptr()))
        return out, gouts
    _lib.check(lib.advgrpo_layernorm_mod_bwd(x.data_ptr(), x.stride(0), dy0.data_ptr(), dp(dy1), dy0.stride(0), dp(scale0),
                                             dp(scale1), ms, int(rows_per_batch), dp(dres), out.data_ptr(), out.stride(0),
                                             M, D, float(eps), _lib.stream_ptr()))
    return out


def rmsnorm_heads_bwd(dy, y, rs, col0, nheads, weight, heads_per_weight, seg=None, M=None):
    lib = _lib.load()
    seg_rows, seg_stride, seg_off = seg if seg is not None else (0, 0, 0)
    M = dy.shape[0] if M is None else M
    _lib.check(lib.advgrpo_rmsnorm_heads_bwd(dy.data_ptr(), dy.stride(0), y.data_ptr(), y.stride(0), rs.data_ptr(), M, col0,
                                             nheads, weight.data_ptr(), heads_per_weight, int(seg_rows), int(seg_stride),
                                             int(seg_off), _lib.stream_ptr()))
    return dy


def qk_norm_rope_bwd(dy, y, rs, S, n_first, nheads, head_dim, w_first, w_rest, heads_per_weight, rope=None, col0=0):
    """In place on dy [B * S, ld] bf16: gradient w.r.t. the output of ops.qk_norm_rope (same arguments; y = its saved output,
    rs = its rs_out [B * S, nheads] f32) -> gradient w.r.t. its input."""
    lib = _lib.load()
    assert dy.dtype == torch.bfloat16 and y.dtype == torch.bfloat16 and dy.stride(1) == 1 and y.stride(1) == 1 and rs.dtype == torch.float32
    _lib.check(lib.advgrpo_qk_norm_rope_bwd(dy.data_ptr(), dy.stride(0), y.data_ptr(), y.stride(0), rs.data_ptr(), dy.shape[0], int(S),
                                            int(n_first), int(col0), int(nheads), int(head_dim), w_first.data_ptr(), w_rest.data_ptr(),
                                            int(heads_per_weight), rope.data_ptr() if rope is not None else None, _lib.stream_ptr()))
    return dy


def gate_mul(x, gate, rows_per_batch, out=None):
    """out[m,:] = gate[m // rows_per_batch, :] * x[m,:]; gate a [G,D] bf16 view (row pitch stride(0))."""
    lib = _lib.load()
    M, D = x.shape
    out = torch.empty_like(x) if out is None else out
    _lib.check(lib.advgrpo_gate_mul(x.data_ptr(), gate.data_ptr(), out.data_ptr(), M, D, int(rows_per_batch), gate.stride(0),
                                    _lib.stream_ptr()))
    return out
