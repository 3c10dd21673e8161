"""Tensor-level wrappers over the C ABI (include/vitmi.h).

PyTorch is used here only for device memory and the current HIP stream; every
function enqueues hand-written HIP kernels from libvitmi.so and nothing else.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (BF16, EPI_BIAS_GELU, EPI_DGELU, EPI_PATCH_POS, EPI_RESIDUAL, EPI_STORE, F32,
                   GEMM_AUTO, GEMM_FAST, GEMM_GENERIC, FoldDesc, GemmDesc, check, load)

_WS: dict = {}


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"vitmi: unsupported dtype {t.dtype}")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.VitmiError("vitmi ops run on the GPU only (tensor is on %s); there is no CPU fallback" % t.device)


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream)."""
    key = (str(device), _stream())
    w = _WS.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = w
    return w


def gemm(A, B, C_out, *, a_kmajor=True, b_kmajor=True, epilogue=EPI_STORE, bias=None, R=None,
         gamma=None, aux=None, C2=None, pos=None, n_tok=0, cls=None, alpha=1.0, accumulate=False,
         impl=GEMM_AUTO, rowscale=None, rows_per_group=0, colsum_part=None, aux_deriv=False,
         launch_flags=0):
    """C = epilogue(op(A) @ op(B)^T); see vitmi_gemm in include/vitmi.h."""
    d = _gemm_desc(A, B, C_out, a_kmajor=a_kmajor, b_kmajor=b_kmajor, epilogue=epilogue, bias=bias, R=R, gamma=gamma,
                   aux=aux, C2=C2, pos=pos, n_tok=n_tok, cls=cls, alpha=alpha, accumulate=accumulate, impl=impl,
                   rowscale=rowscale, rows_per_group=rows_per_group, colsum_part=colsum_part, aux_deriv=aux_deriv,
                   launch_flags=launch_flags)
    lib = load()
    need = lib.vitmi_gemm_workspace(C.byref(d))
    if need:
        ws = workspace(need, A.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    check(lib.vitmi_gemm(C.byref(d), _stream()), "vitmi_gemm")
    return C_out


def split3(x, *, b_pattern: bool, stacked: bool):
    """The bf16 image of an fp32 matrix for the "bf16x3" mode (vitmi_split3): [rows, 3 cols] (k-major operand) or
    [3 rows, cols] (k-minor operand), A pattern hi|lo|hi or B pattern hi|hi|lo."""
    _need_cuda(x)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    R, Cn = x.shape
    out = torch.empty((3 * R, Cn) if stacked else (R, 3 * Cn), dtype=torch.bfloat16, device=x.device)
    check(load().vitmi_split3(x.data_ptr(), x.stride(0), R, Cn, out.data_ptr(), out.stride(0), int(b_pattern), int(stacked),
                              _stream()), "vitmi_split3")
    return out


def gelu_fwd(pre, out):
    _need_cuda(pre, out)
    M, N = pre.shape
    check(load().vitmi_gelu_fwd(pre.data_ptr(), pre.stride(0), out.data_ptr(), out.stride(0), M, N, _stream()), "vitmi_gelu_fwd")
    return out


def gelu_bwd(dh, pre, out):
    _need_cuda(dh, pre, out)
    M, N = pre.shape
    check(load().vitmi_gelu_bwd(dh.data_ptr(), dh.stride(0), pre.data_ptr(), pre.stride(0), out.data_ptr(), out.stride(0),
                                M, N, _stream()), "vitmi_gelu_bwd")
    return out


def gemm_split3(A, B, C_out, *, a_kmajor=True, b_kmajor=True, epilogue=EPI_STORE, bias=None, aux=None, C2=None,
                aux_deriv=False, colsum_part=None, **k):
    """`gemm` for fp32 operands in the "bf16x3" mode: both operands are written as their three-part bf16 images
    (`split3`) and ONE bf16 product over K' = 3K runs on the tile kernels with the fp32 epilogue asked for.  The two GELU
    epilogues (bf16-only on the tile kernel) become a plain fp32 product + the element-wise fp32 kernel."""
    assert A.dtype == torch.float32 and B.dtype == torch.float32 and C_out.dtype == torch.float32
    assert colsum_part is None and not aux_deriv
    A3 = split3(A, b_pattern=False, stacked=not a_kmajor)
    B3 = split3(B, b_pattern=True, stacked=not b_kmajor)
    if epilogue == EPI_BIAS_GELU:
        pre = C2 if C2 is not None else torch.empty_like(C_out)
        gemm(A3, B3, pre, a_kmajor=a_kmajor, b_kmajor=b_kmajor, bias=bias, **k)
        return gelu_fwd(pre, C_out)
    if epilogue == EPI_DGELU:
        gemm(A3, B3, C_out, a_kmajor=a_kmajor, b_kmajor=b_kmajor, **k)
        return gelu_bwd(C_out, aux, C_out)
    return gemm(A3, B3, C_out, a_kmajor=a_kmajor, b_kmajor=b_kmajor, epilogue=epilogue, bias=bias, C2=C2, **k)


def gemm_pair(A0, B0, C0, A1, B1, C1, launch_flags=0):
    """Two weight-gradient products C_i = A_i^T @ B_i (k-minor operands [tokens, features], fp32 C) in one launch where
    the library can pair them (vitmi_gemm_pair), else one after the other; same results."""
    d0 = _gemm_desc(A0, B0, C0, a_kmajor=False, b_kmajor=False, launch_flags=launch_flags)
    d1 = _gemm_desc(A1, B1, C1, a_kmajor=False, b_kmajor=False, launch_flags=launch_flags)
    lib = load()
    need = max(lib.vitmi_gemm_pair_workspace(C.byref(d0), C.byref(d1)), lib.vitmi_gemm_workspace(C.byref(d0)),
               lib.vitmi_gemm_workspace(C.byref(d1)))
    ws = workspace(need, A0.device)
    for d in (d0, d1):                       # the fallback (two vitmi_gemm calls) splits K through the descriptors' own fields
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    check(lib.vitmi_gemm_pair(C.byref(d0), C.byref(d1), ws.data_ptr(), ws.numel(), _stream()), "vitmi_gemm_pair")


def gemm_pair_shares_a_launch(M0, N0, M1, N1, K) -> bool:
    """Host-only query: would the two k-minor bf16 products [M0,N0] and [M1,N1] over K rows share one launch?"""
    ds = []
    for M, N in ((M0, N0), (M1, N1)):
        d = GemmDesc()
        d.M, d.N, d.K = M, N, K
        d.A = d.B = d.C = 256
        d.lda, d.ldb, d.ldc = M, N, N
        d.a_kmajor = d.b_kmajor = 0
        d.in_dtype, d.c_dtype, d.epilogue = BF16, F32, EPI_STORE
        ds.append(d)
    return load().vitmi_gemm_pair_workspace(C.byref(ds[0]), C.byref(ds[1])) > 0


def _gemm_desc(A, B, C_out, *, a_kmajor=True, b_kmajor=True, epilogue=EPI_STORE, bias=None, R=None,
               gamma=None, aux=None, C2=None, pos=None, n_tok=0, cls=None, alpha=1.0, accumulate=False,
               impl=GEMM_AUTO, rowscale=None, rows_per_group=0, colsum_part=None, aux_deriv=False,
               launch_flags=0):
    _need_cuda(A, B, C_out)
    assert A.dim() == 2 and B.dim() == 2 and C_out.dim() == 2
    assert A.stride(1) == 1 and B.stride(1) == 1 and C_out.stride(1) == 1
    M, K = (A.shape if a_kmajor else (A.shape[1], A.shape[0]))
    N, Kb = (B.shape if b_kmajor else (B.shape[1], B.shape[0]))
    assert K == Kb, f"gemm: K mismatch {K} vs {Kb}"
    assert tuple(C_out.shape) == (M, N), f"gemm: C shape {tuple(C_out.shape)} != {(M, N)}"
    assert A.dtype == B.dtype
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.A, d.lda, d.a_kmajor = A.data_ptr(), A.stride(0), int(a_kmajor)
    d.B, d.ldb, d.b_kmajor = B.data_ptr(), B.stride(0), int(b_kmajor)
    d.in_dtype = dtype_code(A)
    d.epilogue = epilogue
    d.C, d.ldc, d.c_dtype = C_out.data_ptr(), C_out.stride(0), dtype_code(C_out)
    if C2 is not None:
        want = A.dtype if epilogue == EPI_RESIDUAL else C_out.dtype
        assert C2.dtype == want and tuple(C2.shape) == (M, N) and C2.stride(1) == 1
        d.C2, d.ldc2 = C2.data_ptr(), C2.stride(0)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
        d.bias = bias.data_ptr()
    if R is not None:
        assert tuple(R.shape) == (M, N) and R.stride(1) == 1
        d.R, d.ldr, d.r_dtype = R.data_ptr(), R.stride(0), dtype_code(R)
    if gamma is not None:
        assert gamma.dtype == torch.float32 and gamma.numel() == N and gamma.is_contiguous()
        d.gamma = gamma.data_ptr()
    if aux is not None:
        assert aux.dtype == A.dtype and tuple(aux.shape) == (M, N) and aux.stride(1) == 1
        d.AUX, d.ldaux = aux.data_ptr(), aux.stride(0)
    if pos is not None:
        assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.numel() == n_tok * N
        d.pos, d.n_tok = pos.data_ptr(), n_tok
    if cls is not None:
        assert cls.dtype == torch.float32 and cls.numel() == N and cls.is_contiguous()
        d.cls = cls.data_ptr()
    d.alpha = float(alpha)
    d.accumulate = int(accumulate)
    d.impl = impl
    if rowscale is not None:
        assert rowscale.dtype == torch.float32 and rows_per_group > 0 and rowscale.numel() * rows_per_group >= M
        d.rowscale, d.rows_per_group = rowscale.data_ptr(), rows_per_group
    if colsum_part is not None:
        assert colsum_part.dtype == torch.float32 and colsum_part.is_contiguous()
        assert tuple(colsum_part.shape) == ((M + 127) // 128, N)
        d.colsum_part = colsum_part.data_ptr()
    d.aux_is_derivative = int(bool(aux_deriv))
    d.launch_flags = int(launch_flags)
    return d


def gemm_batched(A, B, C_out, *, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, batch, batch_inner,
                 a_bs, b_bs, c_bs, a_off=0, b_off=0, c_off=0, alpha=1.0, impl=GEMM_AUTO):
    """batch independent products on strided views of A/B/C storage (element offsets/strides):
    C_z = alpha * op(A_z) op(B_z)^T, z = zo*batch_inner + zi.  One workgroup per problem for
    small bf16 problems (gemm_small.hip), else the generic MFMA kernel (impl=GEMM_GENERIC forces it)."""
    _need_cuda(A, B, C_out)
    assert A.dtype == B.dtype
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.A, d.lda, d.a_kmajor = A.data_ptr() + a_off * A.element_size(), lda, int(a_kmajor)
    d.B, d.ldb, d.b_kmajor = B.data_ptr() + b_off * B.element_size(), ldb, int(b_kmajor)
    d.in_dtype = dtype_code(A)
    d.epilogue = EPI_STORE
    d.C, d.ldc, d.c_dtype = C_out.data_ptr() + c_off * C_out.element_size(), ldc, dtype_code(C_out)
    d.alpha = float(alpha)
    d.impl = impl
    d.batch, d.batch_inner = batch, batch_inner
    d.a_bs[0], d.a_bs[1] = a_bs
    d.b_bs[0], d.b_bs[1] = b_bs
    d.c_bs[0], d.c_bs[1] = c_bs
    check(load().vitmi_gemm(C.byref(d), _stream()), "vitmi_gemm(batched)")
    return C_out


def gemm_uses_fast(M, N, K, *, a_kmajor=True, b_kmajor=True, in_dtype=BF16, c_dtype=BF16,
                   epilogue=EPI_STORE, lda=None, ldb=None, ldc=None, colsum_part=False) -> bool:
    """Host-only query (no GPU needed): would this problem take the fast kernel?"""
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.A = d.B = d.C = 256  # any non-null, 256-B aligned address
    d.R = d.AUX = d.pos = 256
    d.lda = lda or (K if a_kmajor else M)
    d.ldb = ldb or (K if b_kmajor else N)
    d.ldc = ldc or N
    d.ldr = d.ldaux = N
    d.n_tok = 1
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.in_dtype, d.c_dtype, d.r_dtype, d.epilogue = in_dtype, c_dtype, c_dtype, epilogue
    if colsum_part:
        d.colsum_part = 256
    return bool(load().vitmi_gemm_uses_fast(C.byref(d)))


def layernorm_fwd(x, gamma, beta, y, mean, rstd, eps, *, M=None, D=None, x_stride=None, y_stride=None):
    """Rows of x (stride x_stride elements) -> y; mean/rstd fp32 [M] (may be None)."""
    _need_cuda(x, y)
    D = D or x.shape[-1]
    M = M or x.numel() // D
    xs = x_stride if x_stride is not None else D
    ys = y_stride if y_stride is not None else D
    check(load().vitmi_layernorm_fwd(x.data_ptr(), dtype_code(x), xs, gamma.data_ptr(), beta.data_ptr(),
                                     y.data_ptr(), dtype_code(y), ys, _ptr(mean), _ptr(rstd),
                                     M, D, float(eps), _stream()), "vitmi_layernorm_fwd")
    return y


class FoldQueue:
    """Deferred folds of fp32 partial column sums (vitmi_fold_many): the engines queue the ~50 small folds of a backward
    pass (LayerNorm dgamma | dbeta | bias sums, the bias partials of the GEMM / attention epilogues) and run them in one
    launch per flush — at the end of backward, or before a gradient-bucket section is handed to the reducer.  The queue
    keeps the partial buffers alive until then.  Results are bit-identical to the single folds."""

    def __init__(self):
        self._descs, self._keep = [], []

    def add(self, part, S, N, ld, outs, keep=()):
        d = FoldDesc()
        d.part, d.S, d.nseg, d.N, d.ld = part.data_ptr(), int(S), len(outs), int(N), int(ld)
        for k, o in enumerate(outs):
            assert o.dtype == torch.float32 and o.numel() >= N
            d.out[k] = o.data_ptr()
        self._descs.append(d)
        self._keep.append((part, outs, keep))

    def add_desc(self, d, keep):
        self._descs.append(d)
        self._keep.append(keep)

    def __len__(self):
        return len(self._descs)

    def flush(self):
        if not self._descs:
            return
        arr = (FoldDesc * len(self._descs))(*self._descs)
        try:
            check(load().vitmi_fold_many(arr, len(self._descs), _stream()), "vitmi_fold_many")
        finally:
            self._descs, self._keep = [], []

    def clear(self):
        self._descs, self._keep = [], []


def layernorm_bwd(dy, x, mean, rstd, gamma, g_in, g_out, gb_out, dgamma, dbeta, *, gsum=None,
                  gb_scale=None, gb_rowscale=None, rows_per_group=0, M=None, D=None, dy_stride=None,
                  x_stride=None, g_stride=None, gb_stride=None, fold=None):
    """fold (a FoldQueue): leave the fold of the dgamma | dbeta | gsum partials to the queue's next flush; the partial
    rows then live in a buffer of their own (kept by the queue) instead of the shared workspace."""
    _need_cuda(dy, x, g_out)
    D = D or x.shape[-1]
    M = M or dy.numel() // D
    lib = load()
    nbytes = lib.vitmi_layernorm_bwd_workspace(M, D)
    args = (
        dy.data_ptr(), dtype_code(dy), dy_stride if dy_stride is not None else D,
        x.data_ptr(), dtype_code(x), x_stride if x_stride is not None else D,
        mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
        _ptr(g_in), g_out.data_ptr(), dtype_code(g_out), g_stride if g_stride is not None else D,
        _ptr(gb_out), dtype_code(gb_out) if gb_out is not None else dtype_code(dy),
        gb_stride if gb_stride is not None else D,
        dgamma.data_ptr(), dbeta.data_ptr(), _ptr(gsum), _ptr(gb_scale), _ptr(gb_rowscale), rows_per_group,
        M, D)
    if fold is None:
        ws = workspace(nbytes, dy.device)
        check(lib.vitmi_layernorm_bwd(*args, ws.data_ptr(), ws.numel(), _stream()), "vitmi_layernorm_bwd")
        return
    part = torch.empty(int(nbytes), dtype=torch.uint8, device=dy.device)
    d = FoldDesc()
    check(lib.vitmi_layernorm_bwd_deferred(*args, part.data_ptr(), part.numel(), C.byref(d), _stream()),
          "vitmi_layernorm_bwd_deferred")
    fold.add_desc(d, (part, dgamma, dbeta, gsum))


def attn_fwd(qkv, out, lse, B, N, H, hd, scale):
    _need_cuda(qkv, out, lse)
    assert qkv.is_contiguous() and out.is_contiguous() and lse.dtype == torch.float32
    check(load().vitmi_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), dtype_code(qkv),
                                B, N, H, hd, float(scale), _stream()), "vitmi_attn_fwd")
    return out


def attn_bwd_dbias_rows(B, N) -> int:
    return int(load().vitmi_attn_bwd_dbias_rows(B, N))


def attn_bwd(qkv, out, dout, lse, dqkv, B, N, H, hd, scale, dbias_part=None, launch_flags=0):
    """dbias_part (bf16 only): fp32 [attn_bwd_dbias_rows(B, N), 3*H*hd] partial column sums of dqkv."""
    _need_cuda(qkv, out, dout, dqkv)
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and dqkv.is_contiguous()
    lib = load()
    if dbias_part is not None:
        assert dbias_part.dtype == torch.float32 and dbias_part.is_contiguous()
        assert tuple(dbias_part.shape) == (attn_bwd_dbias_rows(B, N), 3 * H * hd)
    ws = workspace(lib.vitmi_attn_bwd_workspace(B, N, H), qkv.device)
    check(lib.vitmi_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                             dqkv.data_ptr(), dtype_code(qkv), B, N, H, hd, float(scale), _ptr(dbias_part),
                             int(launch_flags), ws.data_ptr(), ws.numel(), _stream()), "vitmi_attn_bwd")
    return dqkv


def cast(src, dst):
    _need_cuda(src, dst)
    assert src.numel() == dst.numel() and src.is_contiguous() and dst.is_contiguous()
    check(load().vitmi_cast(src.data_ptr(), dtype_code(src), dst.data_ptr(), dtype_code(dst),
                            src.numel(), _stream()), "vitmi_cast")
    return dst


def axpy(x, y, a=1.0):
    """y += a * x (fp32, contiguous): gradient accumulation over a span of the flat gradient buffer."""
    _need_cuda(x, y)
    assert x.dtype == y.dtype == torch.float32 and x.numel() == y.numel() and x.is_contiguous() and y.is_contiguous()
    check(load().vitmi_axpy(x.data_ptr(), y.data_ptr(), float(a), x.numel(), _stream()), "vitmi_axpy")
    return y


def patchify(x, out, p, cls_rows):
    """x [B,C,H,W] fp32 (any strides) -> out [B*(cls_rows+gh*gw), ld] with ld >= C*p*p; the columns
    beyond C*p*p are written as zeros (K padding for the tile GEMM)."""
    _need_cuda(x, out)
    assert x.dtype == torch.float32 and x.dim() == 4 and out.dim() == 2 and out.is_contiguous()
    B, Cc, H, W = x.shape
    assert out.shape[1] >= Cc * p * p and out.shape[1] % 4 == 0
    sb, sc, sh, sw = x.stride()
    check(load().vitmi_patchify(x.data_ptr(), sb, sc, sh, sw, out.data_ptr(), dtype_code(out), out.shape[1],
                                B, Cc, H, W, p, int(cls_rows), _stream()), "vitmi_patchify")
    return out


def pos_resample(src, table, out=None):
    """out[r] = sum_e w[e] * src[col[e]] over row r's entries of `table` = (row_ptr, col, w, rows):
    the bicubic pos_embed resize, or with the transposed table its backward (posembed.py)."""
    row_ptr, col, w, rows = table
    _need_cuda(src, row_ptr, col, w)
    assert src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1
    assert row_ptr.dtype == torch.int32 and col.dtype == torch.int32 and w.dtype == torch.float32
    assert row_ptr.numel() == rows + 1
    D = src.shape[1]
    if out is None:
        out = torch.empty((rows, D), dtype=torch.float32, device=src.device)
    assert out.dtype == torch.float32 and tuple(out.shape) == (rows, D) and out.stride(1) == 1
    check(load().vitmi_pos_resample(src.data_ptr(), src.stride(0), row_ptr.data_ptr(), col.data_ptr(), w.data_ptr(),
                                    out.data_ptr(), out.stride(0), rows, D, _stream()), "vitmi_pos_resample")
    return out


FOLD_DIRECT_ROWS = 2048      # vitmi_colsum folds an fp32 matrix of at most this many rows in one reduce_rows launch


def colsum(x, out, *, M=None, N=None, ld=None, fold=None):
    """fold (a FoldQueue): a short fp32 matrix (a partial buffer) is queued instead of folded now — the same sum."""
    _need_cuda(x, out)
    N = N or x.shape[-1]
    M = M or x.numel() // N
    ld = ld or N
    assert out.dtype == torch.float32 and out.numel() >= N
    if fold is not None and x.dtype == torch.float32 and M <= FOLD_DIRECT_ROWS:
        fold.add(x, M, N, ld, [out])
        return out
    lib = load()
    ws = workspace(lib.vitmi_colsum_workspace(M, N), x.device)
    check(lib.vitmi_colsum(x.data_ptr(), dtype_code(x), M, N, ld, out.data_ptr(), ws.data_ptr(),
                           ws.numel(), _stream()), "vitmi_colsum")
    return out


def softmax_xent(logits, labels, loss_buf, dlogits, correct_buf):
    """loss_buf fp32 [1+B] (loss_buf[0] = mean loss), correct_buf int32 [1+B]."""
    _need_cuda(logits, labels, loss_buf, dlogits, correct_buf)
    B, K = logits.shape
    assert logits.dtype == torch.float32 and logits.is_contiguous() and labels.dtype == torch.int64
    assert loss_buf.numel() >= 1 + B and correct_buf.numel() >= 1 + B and correct_buf.dtype == torch.int32
    check(load().vitmi_softmax_xent(logits.data_ptr(), labels.data_ptr(), loss_buf.data_ptr(),
                                    dlogits.data_ptr(), correct_buf.data_ptr(), B, K, _stream()),
          "vitmi_softmax_xent")


def sgd_momentum(p, g, buf, shadow, lr, momentum, grad_scale=1.0):
    _need_cuda(p, g, buf)
    assert p.dtype == g.dtype == buf.dtype == torch.float32
    assert p.numel() == g.numel() == buf.numel()
    check(load().vitmi_sgd_momentum(p.data_ptr(), g.data_ptr(), buf.data_ptr(), _ptr(shadow),
                                    p.numel(), float(lr), float(momentum), float(grad_scale),
                                    _stream()), "vitmi_sgd_momentum")


def image_ingest(src, dst, off_y, off_x, flip, mean, std, pad, fill=128):
    """src uint8 [B,H,W,C] (NHWC) -> dst fp32 [B,C,S,S]; see vitmi_image_ingest."""
    _need_cuda(src, dst)
    assert src.dtype == torch.uint8 and src.is_contiguous() and dst.dtype == torch.float32 and dst.is_contiguous()
    B, H, W, C = src.shape
    assert dst.shape[0] == B and dst.shape[1] == C and dst.shape[2] == dst.shape[3]
    for t, dt in ((off_y, torch.int32), (off_x, torch.int32), (flip, torch.uint8), (mean, torch.float32), (std, torch.float32)):
        assert t is None or (t.is_cuda and t.dtype == dt and t.is_contiguous())
    check(load().vitmi_image_ingest(src.data_ptr(), dst.data_ptr(), _ptr(off_y), _ptr(off_x), _ptr(flip), _ptr(mean),
                                    _ptr(std), B, H, W, C, dst.shape[2], int(pad), int(fill), _stream()),
          "vitmi_image_ingest")
    return dst


def ingest_patchify(src, out, off_y, off_x, flip, mean, std, S, pad, p, cls_rows, fill=128):
    """src uint8 [B,H,W,C] (NHWC) -> out [B*(cls_rows + (S/p)^2), ld >= C*p*p] patch rows; see vitmi_ingest_patchify."""
    _need_cuda(src, out)
    assert src.dtype == torch.uint8 and src.is_contiguous() and out.dim() == 2 and out.is_contiguous()
    B, H, W, C = src.shape
    g = S // p
    assert out.shape[0] == B * (cls_rows + g * g) and out.shape[1] >= C * p * p
    for t, dt in ((off_y, torch.int32), (off_x, torch.int32), (flip, torch.uint8), (mean, torch.float32), (std, torch.float32)):
        assert t is None or (t.is_cuda and t.dtype == dt and t.is_contiguous())
    check(load().vitmi_ingest_patchify(src.data_ptr(), out.data_ptr(), dtype_code(out), out.shape[1], _ptr(off_y), _ptr(off_x),
                                       _ptr(flip), _ptr(mean), _ptr(std), B, H, W, C, int(S), int(pad), int(fill), int(p),
                                       int(cls_rows), _stream()), "vitmi_ingest_patchify")
    return out


def adam(p, g, m, v, shadow, state, lr, beta1, beta2, eps, weight_decay, decoupled, grad_scale=1.0):
    _need_cuda(p, g, m, v, state)
    assert p.dtype == g.dtype == m.dtype == v.dtype == state.dtype == torch.float32
    assert p.numel() == g.numel() == m.numel() == v.numel() and state.numel() >= 1
    check(load().vitmi_adam(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(shadow), state.data_ptr(),
                            p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                            int(bool(decoupled)), float(grad_scale), _stream()), "vitmi_adam")


def adagrad(p, g, acc, shadow, state, lr, lr_decay, eps, weight_decay, grad_scale=1.0):
    _need_cuda(p, g, acc, state)
    assert p.dtype == g.dtype == acc.dtype == state.dtype == torch.float32 and p.numel() == g.numel() == acc.numel()
    check(load().vitmi_adagrad(p.data_ptr(), g.data_ptr(), acc.data_ptr(), _ptr(shadow), state.data_ptr(), p.numel(),
                               float(lr), float(lr_decay), float(eps), float(weight_decay), float(grad_scale),
                               _stream()), "vitmi_adagrad")


def adadelta(p, g, square_avg, acc_delta, shadow, lr, rho, eps, weight_decay, grad_scale=1.0):
    _need_cuda(p, g, square_avg, acc_delta)
    assert p.dtype == g.dtype == square_avg.dtype == acc_delta.dtype == torch.float32
    assert p.numel() == g.numel() == square_avg.numel() == acc_delta.numel()
    check(load().vitmi_adadelta(p.data_ptr(), g.data_ptr(), square_avg.data_ptr(), acc_delta.data_ptr(), _ptr(shadow),
                                p.numel(), float(lr), float(rho), float(eps), float(weight_decay), float(grad_scale),
                                _stream()), "vitmi_adadelta")


def adabelief(p, g, m, s, shadow, state, lr, beta1, beta2, eps, weight_decay, decoupled, rectify, grad_scale=1.0):
    _need_cuda(p, g, m, s, state)
    assert p.dtype == g.dtype == m.dtype == s.dtype == state.dtype == torch.float32
    assert p.numel() == g.numel() == m.numel() == s.numel()
    check(load().vitmi_adabelief(p.data_ptr(), g.data_ptr(), m.data_ptr(), s.data_ptr(), _ptr(shadow), state.data_ptr(),
                                 p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                 int(bool(decoupled)), int(bool(rectify)), float(grad_scale), _stream()), "vitmi_adabelief")


# ------------------------------------------------------------------ CaiT ops ---
def th_softmax_fwd(S, Wl, bl, Ww, bw, P, Pm, B, H, N, Nk, ld):
    _need_cuda(S, P, Pm)
    check(load().vitmi_th_softmax_fwd(S.data_ptr(), Wl.data_ptr(), bl.data_ptr(), Ww.data_ptr(), bw.data_ptr(),
                                      P.data_ptr(), Pm.data_ptr(), dtype_code(S), B, H, N, Nk, ld, _stream()),
          "vitmi_th_softmax_fwd")


def th_softmax_bwd(S, P, dPm, Wl, Ww, dS, dWl, dbl, dWw, dbw, B, H, N, Nk, ld):
    _need_cuda(S, P, dPm, dS)
    lib = load()
    ws = workspace(lib.vitmi_th_softmax_bwd_workspace(B, H, N), S.device)
    check(lib.vitmi_th_softmax_bwd(S.data_ptr(), P.data_ptr(), dPm.data_ptr(), Wl.data_ptr(), Ww.data_ptr(),
                                   dS.data_ptr(), dWl.data_ptr(), dbl.data_ptr(), dWw.data_ptr(), dbw.data_ptr(),
                                   dtype_code(S), B, H, N, Nk, ld, ws.data_ptr(), ws.numel(), _stream()),
          "vitmi_th_softmax_bwd")


def th_attn_supported(t, H, N, hd) -> bool:
    """Host-only query: do the fused talking-heads kernels take this shape / dtype?"""
    code = (BF16 if t == torch.bfloat16 else F32 if t == torch.float32 else -1) if isinstance(t, torch.dtype) else dtype_code(t)
    return bool(load().vitmi_th_attn_supported(code, H, N, hd))


def _th_ws(B, H, N, hd, device):
    w = workspace(load().vitmi_th_attn_workspace(B, H, N, hd) + 256, device)
    off = (-w.data_ptr()) % 256
    return w.data_ptr() + off, w.numel() - off


def th_attn_fwd(qkv, Wl, bl, Ww, bw, out, B, H, N, hd, scale):
    """out [B,N,H,hd] = talking-heads attention of qkv [B,N,3,H,hd] (vitmi_th_attn_fwd: scores never leave the CU)."""
    _need_cuda(qkv, out)
    assert qkv.is_contiguous() and out.is_contiguous()
    ptr, nb = _th_ws(B, H, N, hd, qkv.device)
    check(load().vitmi_th_attn_fwd(qkv.data_ptr(), Wl.data_ptr(), bl.data_ptr(), Ww.data_ptr(), bw.data_ptr(), out.data_ptr(),
                                   dtype_code(qkv), B, H, N, hd, float(scale), ptr, nb, _stream()), "vitmi_th_attn_fwd")
    return out


def th_attn_bwd(qkv, dout, Wl, bl, Ww, bw, dqkv, dS, Pm, ld, dWl, dbl, dWw, dbw, B, H, N, hd, scale):
    """dQ into dqkv's q slots, dS and Pm ([B,H,N,ld]) for the caller's dK / dV products, the four mixing-parameter gradients."""
    _need_cuda(qkv, dout, dqkv, dS, Pm)
    assert qkv.is_contiguous() and dout.is_contiguous() and dqkv.is_contiguous() and dS.is_contiguous() and Pm.is_contiguous()
    ptr, nb = _th_ws(B, H, N, hd, qkv.device)
    check(load().vitmi_th_attn_bwd(qkv.data_ptr(), dout.data_ptr(), Wl.data_ptr(), bl.data_ptr(), Ww.data_ptr(), bw.data_ptr(),
                                   dqkv.data_ptr(), dS.data_ptr(), Pm.data_ptr(), ld, dWl.data_ptr(), dbl.data_ptr(), dWw.data_ptr(),
                                   dbw.data_ptr(), dtype_code(qkv), B, H, N, hd, float(scale), ptr, nb, _stream()), "vitmi_th_attn_bwd")


def class_attn_fwd(q, k, v, kv_stride, out, p_save, B, H, N, hd, scale):
    _need_cuda(q, k, v, out, p_save)
    check(load().vitmi_class_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), kv_stride, out.data_ptr(),
                                      p_save.data_ptr(), dtype_code(q), B, H, N, hd, float(scale), _stream()),
          "vitmi_class_attn_fwd")


def class_attn_bwd(q, k, v, kv_stride, dout, p_save, dq, dk, dv, dkv_stride, B, H, N, hd, scale):
    _need_cuda(q, k, v, dout, dq, dk, dv)
    check(load().vitmi_class_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), kv_stride, dout.data_ptr(),
                                      p_save.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dkv_stride,
                                      dtype_code(q), B, H, N, hd, float(scale), _stream()), "vitmi_class_attn_bwd")


def colsum_mul(x, y, out, *, M=None, N=None, ldx=None, ldy=None):
    _need_cuda(x, y, out)
    N = N or x.shape[-1]
    M = M or x.numel() // N
    lib = load()
    ws = workspace(lib.vitmi_colsum_mul_workspace(M, N), x.device)
    check(lib.vitmi_colsum_mul(x.data_ptr(), dtype_code(x), ldx or N, y.data_ptr(), dtype_code(y), ldy or N,
                               M, N, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "vitmi_colsum_mul")
    return out


def scale_cast(x, out, scale=None, *, M, N, ldx=None, ldo=None, rowscale=None, rows_per_group=0):
    """out[m, :N] = cast(x[m, :N] * scale * rowscale[m // rows_per_group]); rows at strides ldx / ldo."""
    _need_cuda(x, out)
    check(load().vitmi_scale_cast(x.data_ptr(), dtype_code(x), ldx or N, _ptr(scale), _ptr(rowscale),
                                  rows_per_group, out.data_ptr(), dtype_code(out), ldo or N, M, N, _stream()),
          "vitmi_scale_cast")
    return out


# ------------------------------------------------------------------ Swin ops ---
def win_attn_fwd(qkv, out, lse, bias, mask, Bw, H, N, hd, Himg, Wimg, ws, shift, scale):
    _need_cuda(qkv, out, lse, bias)
    check(load().vitmi_win_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), bias.data_ptr(), _ptr(mask),
                                    dtype_code(qkv), Bw, H, N, hd, Himg, Wimg, ws, shift, float(scale), _stream()),
          "vitmi_win_attn_fwd")


def win_attn_bwd_fuses_qkv_bias(t, hd) -> bool:
    return bool(load().vitmi_win_attn_bwd_fuses_qkv_bias(dtype_code(t), hd))


def win_attn_bwd(qkv, dout, lse, bias, mask, dqkv, dbias, Bw, H, N, hd, Himg, Wimg, ws, shift, scale,
                 dqkv_bias=None):
    """dqkv_bias: fp32 [3*H*hd] <- column sums of dqkv (only where win_attn_bwd_fuses_qkv_bias)."""
    _need_cuda(qkv, dout, dqkv, dbias)
    lib = load()
    ws_buf = workspace(lib.vitmi_win_attn_bwd_workspace(Bw, H, N), qkv.device)
    check(lib.vitmi_win_attn_bwd(qkv.data_ptr(), dout.data_ptr(), lse.data_ptr(), bias.data_ptr(), _ptr(mask),
                                 dqkv.data_ptr(), dbias.data_ptr(), _ptr(dqkv_bias), dtype_code(qkv), Bw, H, N, hd, Himg, Wimg, ws,
                                 shift, float(scale), ws_buf.data_ptr(), ws_buf.numel(), _stream()),
          "vitmi_win_attn_bwd")


def relpos_bias_gather(table, index, bias, T, H, N):
    check(load().vitmi_relpos_bias(table.data_ptr(), index.data_ptr(), bias.data_ptr(), None, None, T, H, N,
                                   _stream()), "vitmi_relpos_bias(gather)")


def relpos_bias_scatter(dbias, index, dtable, T, H, N):
    check(load().vitmi_relpos_bias(None, index.data_ptr(), None, dbias.data_ptr(), dtable.data_ptr(), T, H, N,
                                   _stream()), "vitmi_relpos_bias(scatter)")


def patch_merge(src, dst, B, Hh, Ww, Cc, inverse=False):
    _need_cuda(src, dst)
    assert src.dtype == dst.dtype
    check(load().vitmi_patch_merge(src.data_ptr(), dst.data_ptr(), dtype_code(src), B, Hh, Ww, Cc, int(inverse),
                                   _stream()), "vitmi_patch_merge")


def token_mean_fwd(x, out, B, L, Cc):
    check(load().vitmi_token_mean(x.data_ptr(), out.data_ptr(), None, None, dtype_code(x), B, L, Cc, _stream()),
          "vitmi_token_mean(fwd)")


def token_mean_bwd(dout, dx, B, L, Cc):
    check(load().vitmi_token_mean(None, None, dout.data_ptr(), dx.data_ptr(), dtype_code(dx), B, L, Cc, _stream()),
          "vitmi_token_mean(bwd)")
