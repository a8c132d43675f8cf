"""Thin tensor-level wrappers over the C ABI (include/skg.h).

Every function takes torch CUDA tensors only to get at device pointers, strides and the current HIP
stream; all arithmetic happens inside libskg.so.  A 2-D "matrix view" is any tensor with
``stride(1) == 1`` - column slices of wider buffers are passed as (pointer, leading dimension).
"""
from __future__ import annotations

import ctypes
import threading
from typing import List, Optional, Sequence, Tuple

import torch

from ._lib import SkgError, SkgTap, check, lib

EPI_RELU, EPI_OUT_F32, EPI_GEGLU = 1, 2, 4
CONV_S1, CONV_S2, CONV_UP2, CONV_S2T, CONV_S2A = 0, 1, 2, 3, 4


_workspace = {}
_workspace_lock = threading.Lock()      # (two host threads on one stream: allocate + register is one step)
WORKSPACE_BYTES = 128 << 20
_capture_owner = None      # set by private_buffers(): scratch / workspace owned by one hipGraph set


def _stream() -> int:
    """Current HIP stream.  The first launch on a (device, stream) registers that stream's split-K workspace with
    libskg.so (skg_set_workspace keeps one slab per stream: launches on different streams may run concurrently -
    several pipelines in one process, a graph capture stream next to the eager stream - and never share a slab;
    launches on ONE stream are ordered, so they share their stream's slab and scratch buffers)."""
    st = torch.cuda.current_stream()
    key = (st.device.index, st.cuda_stream)
    if key not in _workspace:
        with _workspace_lock:
            if key not in _workspace:
                with torch.cuda.device(st.device):
                    ws = torch.empty(WORKSPACE_BYTES // 4, device=st.device, dtype=torch.float32)
                    check(lib.skg_set_workspace(ws.data_ptr(), WORKSPACE_BYTES, st.cuda_stream), "skg_set_workspace")
                    _workspace[key] = ws
    return st.cuda_stream


def release_stream(stream: "torch.cuda.Stream"):
    """Drop the split-K slab, the library's registry entry and the scratch buffers of a stream that will not launch again (a
    sampler's side stream, a retired graph set): they are otherwise held for the life of the process."""
    key = (stream.device.index, stream.cuda_stream)
    with _workspace_lock:
        if _workspace.pop(key, None) is not None:
            check(lib.skg_set_workspace(None, 0, stream.cuda_stream), "skg_set_workspace")
    # (scratch keys carry (str(device), stream handle): the same handle value - 0, the default stream - exists on every device)
    for k in [k for k in _scratch if isinstance(k, tuple) and len(k) > 1 and isinstance(k[1], tuple) and k[1][1] == stream.cuda_stream
              and str(k[1][0]) == str(stream.device)]:
        _scratch.pop(k, None)


class private_buffers:
    """``with private_buffers(owner):`` inside a ``torch.cuda.graph`` capture - the launches captured in the block use a
    split-K workspace and scratch buffers that belong to `owner` (a dict that keeps them alive) instead of the capture
    stream's shared ones, so two graph sets captured on torch's one capture stream can be replayed concurrently on
    different streams.  The workspace must be allocated BEFORE the capture starts: ``prepare(owner, dev)``."""

    @staticmethod
    def prepare(owner: dict, dev):
        owner["ws"] = torch.empty(WORKSPACE_BYTES // 4, device=dev, dtype=torch.float32)
        owner["scratch"] = {}

    def __init__(self, owner: dict):
        self.owner = owner

    def __enter__(self):
        global _capture_owner
        st = torch.cuda.current_stream()
        self.key = (st.device.index, st.cuda_stream)
        self.prev_owner, _capture_owner = _capture_owner, self.owner
        with _workspace_lock:
            self.prev_ws = _workspace.get(self.key)
            _workspace[self.key] = self.owner["ws"]
            check(lib.skg_set_workspace(self.owner["ws"].data_ptr(), WORKSPACE_BYTES, st.cuda_stream), "skg_set_workspace")
        return self

    def __exit__(self, *exc):
        global _capture_owner
        _capture_owner = self.prev_owner
        with _workspace_lock:
            if self.prev_ws is None:
                del _workspace[self.key]
                check(lib.skg_set_workspace(None, 0, self.key[1]), "skg_set_workspace")
            else:
                _workspace[self.key] = self.prev_ws
                check(lib.skg_set_workspace(self.prev_ws.data_ptr(), WORKSPACE_BYTES, self.key[1]), "skg_set_workspace")


def _skey(dev):
    """Scratch-buffer key part: device + current stream (see _stream)."""
    return (str(dev), torch.cuda.current_stream().cuda_stream)


def _scratch_buf(key, n: int, dev) -> torch.Tensor:
    """fp32 scratch of n floats for `key`: the current stream's, or - inside private_buffers() - the graph set's own."""
    store = _scratch if _capture_owner is None else _capture_owner["scratch"]
    if key not in store:
        store[key] = torch.empty(n, device=dev, dtype=torch.float32)
    return store[key]


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), "matrix view must be row-major"
    return t.stride(0)


def _f16(*ts):
    for t in ts:
        assert t is None or (t.is_cuda and t.dtype == torch.float16), "fp16 CUDA tensor expected"


class GNPartial:
    """Per (sample, 128-row chunk, group) sum / sum of squares of a producer's output - what `groupnorm(partial=)` folds
    instead of reading the tensor a second time (skg_gemm_f16_gn / skg_conv3x3_f16_gn)."""
    __slots__ = ("buf", "nch", "rows", "groups")

    def __init__(self, rows: int, HW: int, groups: int, dev):
        self.rows, self.nch, self.groups = rows, HW // 128, groups
        self.buf = torch.empty(rows * self.nch * groups * 2, device=dev, dtype=torch.float32)


class Pair:
    """Accuracy mode: a tensor kept as (hi, lo) fp16 views with one pitch, value = hi + lo; `full` = the [M, 2C] tensor
    [hi | lo] when the two are its halves (then the pair is directly a K-doubled matmul operand)."""
    __slots__ = ("hi", "lo", "full")

    def __init__(self, hi, lo, full=None):
        self.hi, self.lo, self.full = hi, lo, full

    @staticmethod
    def empty(M: int, C: int, dev) -> "Pair":
        buf = torch.empty(M, 2 * C, device=dev, dtype=torch.float16)
        return Pair(buf[:, :C], buf[:, C:], buf)


def gn_fusable(M: int, N: int, HW: int, groups: int) -> bool:
    """Shapes the `gn_stats=` form of gemm / conv3x3 accepts (whole 128-row chunks per sample, even group width)."""
    return (HW % 128 == 0 and HW // 128 <= 128 and M % HW == 0 and N % groups == 0 and N % 8 == 0 and
            (N // groups) % 2 == 0 and N // groups >= 4 and N <= 4096)


def gn_concat_ok(CA: int, CB: int, groups: int, ga: int, gb: int) -> bool:
    """Can groupnorm(partial=(pa, CA, pb)) regroup the two producers' sums for a [CA | CB]-channel concatenation?"""
    C = CA + CB
    if C % groups or CA % ga or CB % gb:
        return False
    cpg = C // groups
    return CA % cpg == 0 and cpg % (CA // ga) == 0 and cpg % (CB // gb) == 0


def gemm(A: torch.Tensor, B: torch.Tensor, out: Optional[torch.Tensor] = None, *,
         bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         alpha: float = 1.0, relu: bool = False, out_f32: bool = False, geglu: bool = False, gn_stats=None,
         out_lo: Optional[torch.Tensor] = None, residual_lo: Optional[torch.Tensor] = None):
    """out[m][n] = epi(alpha*(A[m,:] . B[n,:] + bias[n]) + residual[m][n]);  A [M,K], B [N,K].
    geglu=True: B / bias are the interleaved FF1 pack and out is [M, N/2] = a * gelu(g).
    gn_stats=(HW, groups): also returns the GroupNorm partial sums of the output -> (out, GNPartial).
    out_lo / residual_lo (accuracy mode): the output / residual as (hi, lo) pairs of fp16 views with equal pitch."""
    _f16(A, B, bias, residual, out_lo, residual_lo)
    M, K = A.shape
    N = B.shape[0]
    assert B.shape[1] == K
    if out_lo is not None or residual_lo is not None:
        assert out is not None and not (out_f32 or geglu)
        assert out_lo is None or _ld(out_lo) == _ld(out)
        assert residual_lo is None or residual is None or _ld(residual_lo) == _ld(residual)
        r_any = residual if residual is not None else residual_lo
        if gn_stats is not None:      # the partial sums of the output's hi part come with it
            HW, groups = gn_stats
            part = GNPartial(M // HW, HW, groups, A.device)
            check(lib.skg_gemm_f16_hilo_gn(_p(A), _ld(A), _p(B), _ld(B), _p(out), _p(out_lo), _ld(out), M, N, K, _p(bias),
                                           _p(residual), _p(residual_lo), _ld(r_any) if r_any is not None else 0, alpha,
                                           EPI_RELU if relu else 0, _p(part.buf), HW, groups, _stream()), "skg_gemm_f16_hilo_gn")
            return out, part
        check(lib.skg_gemm_f16_hilo(_p(A), _ld(A), _p(B), _ld(B), _p(out), _p(out_lo), _ld(out), M, N, K, _p(bias),
                                    _p(residual), _p(residual_lo), _ld(r_any) if r_any is not None else 0, alpha,
                                    EPI_RELU if relu else 0, _stream()), "skg_gemm_f16_hilo")
        return out
    if out is None:
        out = torch.empty(M, N // 2 if geglu else N, device=A.device,
                          dtype=torch.float32 if out_f32 else torch.float16)
    flags = (EPI_RELU if relu else 0) | (EPI_OUT_F32 if out_f32 else 0) | (EPI_GEGLU if geglu else 0)
    if gn_stats is not None:
        HW, groups = gn_stats
        part = GNPartial(M // HW, HW, groups, A.device)
        check(lib.skg_gemm_f16_gn(_p(A), _ld(A), _p(B), _ld(B), _p(out), _ld(out), M, N, K, _p(bias),
                                  _p(residual), _ld(residual) if residual is not None else 0, alpha, flags,
                                  _p(part.buf), HW, groups, _stream()), "skg_gemm_f16_gn")
        return out, part
    check(lib.skg_gemm_f16(_p(A), _ld(A), _p(B), _ld(B), _p(out), _ld(out), M, N, K, _p(bias),
                           _p(residual), _ld(residual) if residual is not None else 0, alpha, flags,
                           _stream()), "skg_gemm_f16")
    return out


def gemm_geglu_keep(A: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor] = None, out=None, pre=None):
    """FF1 with the gate fused AND the pre-activation kept: returns (a * gelu(g) [M, N/2], H [M, N] interleaved pack)."""
    _f16(A, B, bias)
    M, K = A.shape
    N = B.shape[0]
    if out is None:
        out = torch.empty(M, N // 2, device=A.device, dtype=torch.float16)
    if pre is None:
        pre = torch.empty(M, N, device=A.device, dtype=torch.float16)
    check(lib.skg_gemm_f16_geglu_keep(_p(A), _ld(A), _p(B), _ld(B), _p(out), _ld(out), _p(pre), _ld(pre), M, N, K,
                                      _p(bias), _stream()), "skg_gemm_f16_geglu_keep")
    return out, pre


def ff_block(X, gamma, beta, eps: float, pack: torch.Tensor, bias1_pack: torch.Tensor, bias2: torch.Tensor,
             out=None, want_stats: bool = False, keep_from: Optional[int] = None):
    """Fused feed-forward sub-block at C = 320: out = X + b2 + W2 . geglu(W1 . LayerNorm(X) + b1) in one launch
    (skg_ff_block_f16; pack / bias1_pack from unet.pack_ff_block).  out may be X.
    keep_from=m0: rows >= m0 also store the FF1 output (interleaved pack, what geglu_bwd reads) -> returned last.
    X (and out) may be ops.Pair objects (accuracy mode: skg_ff_block_f16_hilo)."""
    pair = isinstance(X, Pair)
    Xh = X.hi if pair else X
    _f16(Xh, gamma, beta, pack, bias2)
    M, C = Xh.shape
    assert pack.is_contiguous() and pack.dim() == 3 and pack.shape[1:] == (60, 512) and bias1_pack.dtype == torch.float32
    F = pack.shape[0] * 32
    if out is None:
        out = Pair.empty(M, C, Xh.device) if pair else torch.empty(M, C, device=Xh.device, dtype=torch.float16)
    stats = torch.empty(M, 2, device=Xh.device, dtype=torch.float32) if want_stats else None
    pre = None if keep_from is None else torch.empty(M - keep_from, 2 * F, device=Xh.device, dtype=torch.float16)
    if pair:
        assert _ld(X.lo) == _ld(X.hi) and _ld(out.lo) == _ld(out.hi)
        check(lib.skg_ff_block_f16_hilo(_p(X.hi), _p(X.lo), _ld(X.hi), _p(out.hi), _p(out.lo), _ld(out.hi), M, C, F, _p(gamma), _p(beta),
                                        eps, _p(pack), _p(bias1_pack), _p(bias2), _p(stats), _p(pre), _ld(pre) if pre is not None else 0,
                                        keep_from or 0, _stream()), "skg_ff_block_f16_hilo")
    elif keep_from is None:
        check(lib.skg_ff_block_f16(_p(X), _ld(X), _p(out), _ld(out), M, C, F, _p(gamma), _p(beta), eps, _p(pack), _p(bias1_pack),
                                   _p(bias2), _p(stats), _stream()), "skg_ff_block_f16")
    else:
        check(lib.skg_ff_block_f16_keep(_p(X), _ld(X), _p(out), _ld(out), M, C, F, _p(gamma), _p(beta), eps, _p(pack), _p(bias1_pack),
                                        _p(bias2), _p(stats), _p(pre), _ld(pre), keep_from, _stream()), "skg_ff_block_f16_keep")
    if keep_from is None:
        return (out, stats) if want_stats else out
    return (out, stats, pre) if want_stats else (out, pre)


def ff_block_proj(X, gamma, beta, eps: float, pack: torch.Tensor, bias1_pack: torch.Tensor, bias2: torch.Tensor, bias_proj: torch.Tensor,
                  R: torch.Tensor, out=None, want_stats: bool = False, keep_from: Optional[int] = None, gn: Optional[Tuple[int, int]] = None):
    """ff_block followed by Transformer2DModel.proj_out and the outer residual in the same launch (skg_ff_block_proj_f16):
    out = R + bias_proj + W_proj . fp16(X + FF(LayerNorm(X))); pack = unet.pack_ff_block(..., w_proj) (five chunks more).
    gn = (HW, groups): also the GroupNorm partial sums of out.  Returns (out, stats | None, pre | None, GNPartial | None);
    out must not be X (it may be R)."""
    pair = isinstance(X, Pair)
    if pair:      # accuracy mode: X, R and out are pairs (skg_ff_block_proj_f16_hilo)
        _f16(X.hi, X.lo, gamma, beta, pack, bias2, bias_proj, R.hi, R.lo)
        M, C = X.hi.shape
        F = (pack.shape[0] - 5) * 32
        assert pack.is_contiguous() and pack.shape[1:] == (60, 512) and bias1_pack.shape[0] * 32 == F and out is not None
        assert _ld(X.lo) == _ld(X.hi) and _ld(out.lo) == _ld(out.hi) and _ld(R.lo) == _ld(R.hi)
        dev = X.hi.device
        stats = torch.empty(M, 2, device=dev, dtype=torch.float32) if want_stats else None
        pre = None if keep_from is None else torch.empty(M - keep_from, 2 * F, device=dev, dtype=torch.float16)
        part = GNPartial(M // gn[0], gn[0], gn[1], dev) if gn is not None else None
        check(lib.skg_ff_block_proj_f16_hilo(_p(X.hi), _p(X.lo), _ld(X.hi), _p(out.hi), _p(out.lo), _ld(out.hi), M, C, F, _p(gamma), _p(beta), eps,
                                             _p(pack), _p(bias1_pack), _p(bias2), _p(bias_proj), _p(R.hi), _p(R.lo), _ld(R.hi), _p(stats), _p(pre),
                                             _ld(pre) if pre is not None else 0, keep_from or 0, _p(part.buf) if part is not None else None,
                                             gn[0] if gn else 0, gn[1] if gn else 0, _stream()), "skg_ff_block_proj_f16_hilo")
        return out, stats, pre, part
    _f16(X, gamma, beta, pack, bias2, bias_proj, R)
    M, C = X.shape
    assert pack.is_contiguous() and pack.dim() == 3 and pack.shape[1:] == (60, 512) and bias1_pack.dtype == torch.float32
    F = (pack.shape[0] - 5) * 32
    assert bias1_pack.shape[0] * 32 == F
    if out is None:
        out = torch.empty(M, C, device=X.device, dtype=torch.float16)
    stats = torch.empty(M, 2, device=X.device, dtype=torch.float32) if want_stats else None
    pre = None if keep_from is None else torch.empty(M - keep_from, 2 * F, device=X.device, dtype=torch.float16)
    part = GNPartial(M // gn[0], gn[0], gn[1], X.device) if gn is not None else None
    check(lib.skg_ff_block_proj_f16(_p(X), _ld(X), _p(out), _ld(out), M, C, F, _p(gamma), _p(beta), eps, _p(pack), _p(bias1_pack), _p(bias2),
                                    _p(bias_proj), _p(R), _ld(R), _p(stats), _p(pre), _ld(pre) if pre is not None else 0, keep_from or 0,
                                    _p(part.buf) if part is not None else None, gn[0] if gn else 0, gn[1] if gn else 0, _stream()),
          "skg_ff_block_proj_f16")
    return out, stats, pre, part


def xattn_block(X, HW: int, heads: int, Nkv: int, gamma, beta, eps: float, wpack: torch.Tensor, kvpack: torch.Tensor,
                bias_out: torch.Tensor, scale: float, out=None, keep_from: Optional[int] = None):
    """Fused cross-attention sub-block at C = 320, 8 heads: out = X + bo + Wo . Attention(Wq . LayerNorm(X), K, V) over the text
    keys of each row's image, in one launch (skg_xattn_block_f16; packs from unet.pack_xattn_weights / pack_xattn_kv).
    X (and out) may be ops.Pair objects (accuracy mode: skg_xattn_block_f16_hilo)."""
    pair = isinstance(X, Pair)
    Xh = X.hi if pair else X
    _f16(Xh, gamma, beta, wpack, kvpack, bias_out)
    M, C = Xh.shape
    wp, kp = (60, 16) if heads == 8 else (80, 20)      # pieces per head: 8 x 40 (padded to 48) or 5 x 64
    assert wpack.is_contiguous() and wpack.shape == (heads, wp, 512) and kvpack.is_contiguous() and kvpack.shape == (M // HW, heads, kp, 512)
    if out is None:
        out = Pair.empty(M, C, Xh.device) if pair else torch.empty(M, C, device=Xh.device, dtype=torch.float16)
    if pair and keep_from is not None:      # accuracy mode, guided step: the stashing launch on pairs (skg_xattn_block_f16_hilo_keep)
        assert _ld(X.lo) == _ld(X.hi) and _ld(out.lo) == _ld(out.hi)
        Mk = M - keep_from
        dev = Xh.device
        st = torch.empty(Mk, 2, device=dev, dtype=torch.float32)
        q = torch.empty(Mk, C, device=dev, dtype=torch.float16)
        o = torch.empty(Mk, C, device=dev, dtype=torch.float16)
        lse = torch.empty(Mk // HW, heads, HW, device=dev, dtype=torch.float32)
        check(lib.skg_xattn_block_f16_hilo_keep(_p(X.hi), _p(X.lo), _ld(X.hi), _p(out.hi), _p(out.lo), _ld(out.hi), M, HW, C, heads, Nkv,
                                                _p(gamma), _p(beta), eps, _p(wpack), _p(kvpack), _p(bias_out), scale, _p(st), _p(q), _p(o), C,
                                                _p(lse), keep_from, _stream()), "skg_xattn_block_f16_hilo_keep")
        return out, st, q, o, lse
    if pair:
        assert _ld(X.lo) == _ld(X.hi) and _ld(out.lo) == _ld(out.hi)
        check(lib.skg_xattn_block_f16_hilo(_p(X.hi), _p(X.lo), _ld(X.hi), _p(out.hi), _p(out.lo), _ld(out.hi), M, HW, C, heads, Nkv,
                                           _p(gamma), _p(beta), eps, _p(wpack), _p(kvpack), _p(bias_out), scale, _stream()),
              "skg_xattn_block_f16_hilo")
        return out
    if keep_from is not None:      # stashing launch: -> (out, stats, q, o, lse) of the rows >= keep_from (skg_xattn_block_f16_keep)
        Mk = M - keep_from
        st = torch.empty(Mk, 2, device=X.device, dtype=torch.float32)
        q = torch.empty(Mk, C, device=X.device, dtype=torch.float16)
        o = torch.empty(Mk, C, device=X.device, dtype=torch.float16)
        lse = torch.empty(Mk // HW, heads, HW, device=X.device, dtype=torch.float32)
        check(lib.skg_xattn_block_f16_keep(_p(X), _ld(X), _p(out), _ld(out), M, HW, C, heads, Nkv, _p(gamma), _p(beta), eps, _p(wpack),
                                           _p(kvpack), _p(bias_out), scale, _p(st), _p(q), _p(o), C, _p(lse), keep_from, _stream()),
              "skg_xattn_block_f16_keep")
        return out, st, q, o, lse
    check(lib.skg_xattn_block_f16(_p(X), _ld(X), _p(out), _ld(out), M, HW, C, heads, Nkv, _p(gamma), _p(beta), eps, _p(wpack),
                                  _p(kvpack), _p(bias_out), scale, _stream()), "skg_xattn_block_f16")
    return out


def conv3x3(X: torch.Tensor, Wp: torch.Tensor, rows: int, IH: int, IW: int, mode: int = CONV_S1,
            out: Optional[torch.Tensor] = None, *, bias=None, residual=None, alpha: float = 1.0,
            relu: bool = False, gn_groups: Optional[int] = None, out_lo=None, residual_lo=None):
    """X [rows*IH*IW, Cin] (view), Wp [Cout, 9*Cin] tap-major.  Returns [rows*OH*OW, Cout];
    gn_groups=G: (out, GNPartial) - the GroupNorm partial sums of the output come with it.
    out_lo / residual_lo (accuracy mode): see gemm."""
    _f16(X, Wp, bias, residual, out_lo, residual_lo)
    Cin = X.shape[1]
    Cout = Wp.shape[0]
    assert Wp.shape[1] == 9 * Cin and Wp.is_contiguous() and X.shape[0] == rows * IH * IW
    if mode == CONV_S1:
        OH, OW = IH, IW
    elif mode in (CONV_S2, CONV_S2A):
        OH, OW = IH // 2, IW // 2
    else:
        OH, OW = IH * 2, IW * 2
    if out is None:
        out = torch.empty(rows * OH * OW, Cout, device=X.device, dtype=torch.float16)
    if out_lo is not None or residual_lo is not None:
        assert out_lo is None or _ld(out_lo) == _ld(out)
        assert residual_lo is None or residual is None or _ld(residual_lo) == _ld(residual)
        r_any = residual if residual is not None else residual_lo
        if gn_groups is not None:
            part = GNPartial(rows, OH * OW, gn_groups, X.device)
            check(lib.skg_conv3x3_f16_hilo_gn(_p(X), _ld(X), _p(Wp), _p(out), _p(out_lo), _ld(out), rows, IH, IW, Cin, Cout, mode,
                                              _p(bias), _p(residual), _p(residual_lo), _ld(r_any) if r_any is not None else 0,
                                              alpha, EPI_RELU if relu else 0, _p(part.buf), gn_groups, _stream()),
                  "skg_conv3x3_f16_hilo_gn")
            return out, part
        check(lib.skg_conv3x3_f16_hilo(_p(X), _ld(X), _p(Wp), _p(out), _p(out_lo), _ld(out), rows, IH, IW, Cin, Cout, mode,
                                       _p(bias), _p(residual), _p(residual_lo), _ld(r_any) if r_any is not None else 0,
                                       alpha, EPI_RELU if relu else 0, _stream()), "skg_conv3x3_f16_hilo")
        return out
    if gn_groups is not None:
        part = GNPartial(rows, OH * OW, gn_groups, X.device)
        check(lib.skg_conv3x3_f16_gn(_p(X), _ld(X), _p(Wp), _p(out), _ld(out), rows, IH, IW, Cin, Cout, mode,
                                     _p(bias), _p(residual), _ld(residual) if residual is not None else 0,
                                     alpha, EPI_RELU if relu else 0, _p(part.buf), gn_groups, _stream()),
              "skg_conv3x3_f16_gn")
        return out, part
    check(lib.skg_conv3x3_f16(_p(X), _ld(X), _p(Wp), _p(out), _ld(out), rows, IH, IW, Cin, Cout, mode,
                              _p(bias), _p(residual), _ld(residual) if residual is not None else 0,
                              alpha, EPI_RELU if relu else 0, _stream()), "skg_conv3x3_f16")
    return out


def conv3x3_sc(X: torch.Tensor, X2: torch.Tensor, Wcat: torch.Tensor, rows: int, IH: int, IW: int, out: Optional[torch.Tensor] = None, *,
               bias=None, relu: bool = False, gn_groups: Optional[int] = None, out_lo=None):
    """conv2 + conv_shortcut of a ResnetBlock as ONE implicit GEMM (skg_conv3x3_sc_f16): X [rows*IH*IW, Cin] the 3x3 input,
    X2 [rows*IH*IW, K2] the block's input (accuracy mode: the pair buffer [x_hi | x_lo]), Wcat [Cout, 9*Cin + K2], bias =
    conv2.bias + conv_shortcut.bias.  Returns out, or (out, GNPartial) with gn_groups.  Raises SkgError(rc = -2) when declined."""
    _f16(X, X2, Wcat, bias, out_lo)
    Cin, K2, Cout = X.shape[1], X2.shape[1], Wcat.shape[0]
    assert Wcat.shape[1] == 9 * Cin + K2 and Wcat.is_contiguous() and X.shape[0] == X2.shape[0] == rows * IH * IW
    if out is None:
        out = torch.empty(rows * IH * IW, Cout, device=X.device, dtype=torch.float16)
    assert out_lo is None or _ld(out_lo) == _ld(out)
    part = GNPartial(rows, IH * IW, gn_groups, X.device) if gn_groups is not None else None
    check(lib.skg_conv3x3_sc_f16(_p(X), _ld(X), _p(X2), _ld(X2), K2, _p(Wcat), _p(out), _p(out_lo), _ld(out), rows, IH, IW, Cin, Cout,
                                 _p(bias), EPI_RELU if relu else 0, _p(part.buf) if part is not None else None,
                                 gn_groups or 0, _stream()), "skg_conv3x3_sc_f16")
    return (out, part) if part is not None else out


def groupnorm_wino(X: torch.Tensor, rows: int, IH: int, IW: int, groups: int, eps: float, gamma, beta, silu: bool):
    """GroupNorm(+SiLU) of a small map straight into the Winograd input transform of its consumer (skg_groupnorm_wino_fwd):
    -> (V [rows*IH/2*IW/2, 16*C] for conv3x3_wino(None, ..., V=V), statistics [rows, groups, 2]).  Raises SkgError(rc = -2) when the
    (row, group) slice does not fit one workgroup (run groupnorm + conv3x3_wino on its output)."""
    _f16(X, gamma, beta)
    C = X.shape[1]
    V = torch.empty(rows * (IH // 2) * (IW // 2), 16 * C, device=X.device, dtype=torch.float16)
    st = torch.empty(rows, groups, 2, device=X.device, dtype=torch.float32)
    check(lib.skg_groupnorm_wino_fwd(_p(X), _ld(X), _p(V), rows, IH, IW, C, groups, eps, _p(gamma), _p(beta), int(silu), _p(st), _stream()),
          "skg_groupnorm_wino_fwd")
    return V, st


def conv3x3_wino(X: Optional[torch.Tensor], U: torch.Tensor, rows: int, IH: int, IW: int, out: Optional[torch.Tensor] = None, *, bias=None,
                 residual=None, relu: bool = False, out_lo=None, residual_lo=None, V: Optional[torch.Tensor] = None):
    """3x3 stride-1 convolution by Winograd F(2x2, 3x3) (skg_conv3x3_wino_f16): X [rows*IH*IW, Cin] (view), U [Cout, 16*Cin] =
    unet.pack_conv_wino(weight).  Returns [rows*IH*IW, Cout].  Raises SkgError(rc = -2) when declined (odd map, Cin % 64, no room for the
    16 fp32 slabs in the stream's workspace): the caller runs conv3x3.  X = None, V = groupnorm_wino(...)[0]: the input transform exists."""
    _f16(X, U, bias, residual, out_lo, residual_lo, V)
    Cout = U.shape[0]
    Cin = U.shape[1] // 16
    Mt = rows * (IH // 2) * (IW // 2)
    if X is None:
        assert V is not None and V.shape == (Mt, 16 * Cin) and V.is_contiguous() and U.is_contiguous()
        dev_ = V.device
    else:
        assert X.shape[1] == Cin and U.is_contiguous() and X.shape[0] == rows * IH * IW
        dev_ = X.device
        V = torch.empty(Mt, 16 * Cin, device=dev_, dtype=torch.float16)
    if out is None:
        out = torch.empty(rows * IH * IW, Cout, device=dev_, dtype=torch.float16)
    assert out_lo is None or _ld(out_lo) == _ld(out)
    assert residual_lo is None or residual is None or _ld(residual_lo) == _ld(residual)
    r_any = residual if residual is not None else residual_lo
    st = _stream()
    check(lib.skg_conv3x3_wino_f16(_p(X), _ld(X) if X is not None else 0, _p(U), _p(V), _p(out), _p(out_lo), _ld(out), rows, IH, IW, Cin, Cout, _p(bias),
                                   _p(residual), _p(residual_lo), _ld(r_any) if r_any is not None else 0, EPI_RELU if relu else 0, st),
          "skg_conv3x3_wino_f16")
    return out


def conv_up2(X: torch.Tensor, Wpp: torch.Tensor, rows: int, IH: int, IW: int, out: Optional[torch.Tensor] = None, *, bias=None, W9=None):
    """Nearest-2x upsample + 3x3 conv, polyphase (four 4-tap convs over the low-res input).  X [rows*IH*IW, Cin] (view),
    Wpp [4, Cout, 4*Cin] (unet.pack_conv_up2).  Returns [rows*2IH*2IW, Cout].  W9: the layer's ordinary 9-tap pack - the
    fall-back when the launch is declined (SKG_E_UNSUPPORTED)."""
    _f16(X, Wpp, bias)
    Cin, Cout = X.shape[1], Wpp.shape[1]
    assert Wpp.shape == (4, Cout, 4 * Cin) and Wpp.is_contiguous() and X.shape[0] == rows * IH * IW
    if out is None:
        out = torch.empty(rows * 4 * IH * IW, Cout, device=X.device, dtype=torch.float16)
    try:
        check(lib.skg_conv3x3_up2_f16(_p(X), _ld(X), _p(Wpp), _p(out), _ld(out), rows, IH, IW, Cin, Cout, _p(bias), _stream()),
              "skg_conv3x3_up2_f16")
    except SkgError as e:      # the LDS-DMA kernel declined the launch (an operand >= 2 GiB): the 9-tap gather form, when the caller has its pack
        if e.rc != -2 or W9 is None:
            raise
        return conv3x3(X, W9, rows, IH, IW, CONV_UP2, out=out, bias=bias)
    return out


def conv_up2_hilo(X2: torch.Tensor, Wpp3: torch.Tensor, rows: int, IH: int, IW: int, out: Pair, *, bias=None, W9x2=None):
    """Accuracy mode: conv_up2 on the pair buffer X2 = [x_hi | x_lo] ([rows*IH*IW, 2C]) with (hi, lo) pre-summed weights
    Wpp3 [4, Cout, 4 * 3C] (unet.pack_conv_up2_hilo); the output is the pair `out`.  W9x2 (a callable returning the layer's 9-tap
    [W | W] pack, built on first use): the fall-back when the polyphase launch is declined (an operand >= 2 GiB - the pair
    buffer is twice as wide, so the limit comes at half the batch size of conv_up2's): ADVICE r4."""
    _f16(X2, Wpp3, bias, out.hi, out.lo)
    C, Cout = X2.shape[1] // 2, Wpp3.shape[1]
    assert Wpp3.shape == (4, Cout, 12 * C) and Wpp3.is_contiguous() and X2.shape[0] == rows * IH * IW and _ld(out.hi) == _ld(out.lo)
    try:
        check(lib.skg_conv3x3_up2_f16_hilo(_p(X2), _ld(X2), _p(Wpp3), _p(out.hi), _p(out.lo), _ld(out.hi), rows, IH, IW, C, Cout, _p(bias),
                                           _stream()), "skg_conv3x3_up2_f16_hilo")
    except SkgError as e:
        if e.rc != -2 or W9x2 is None:
            raise
        conv3x3(X2, W9x2(), rows, IH, IW, CONV_UP2, out=out.hi, bias=bias, out_lo=out.lo)
    return out


def conv_up2_pairout(X: torch.Tensor, Wpp: torch.Tensor, rows: int, IH: int, IW: int, out: Pair, *, bias=None):
    """Accuracy mode, cheaper upsampler (round 5): the default polyphase launch on the hi part X [rows*IH*IW, C] (a view of the
    pair buffer) with the default pack Wpp [4, Cout, 4*C]; only the OUTPUT is a pair.  Raises SkgError(rc = -2) when declined."""
    _f16(X, Wpp, bias, out.hi, out.lo)
    Cin, Cout = X.shape[1], Wpp.shape[1]
    assert Wpp.shape == (4, Cout, 4 * Cin) and Wpp.is_contiguous() and X.shape[0] == rows * IH * IW and _ld(out.hi) == _ld(out.lo)
    check(lib.skg_conv3x3_up2_f16_pairout(_p(X), _ld(X), _p(Wpp), _p(out.hi), _p(out.lo), _ld(out.hi), rows, IH, IW, Cin, Cout, _p(bias),
                                          _stream()), "skg_conv3x3_up2_f16_pairout")
    return out


def gemm_rows(A: torch.Tensor, B: torch.Tensor, out: torch.Tensor, seg_rows: int, seg_stride: int, *, bias=None):
    """A [M, K] @ B[N, K]^T with product row m stored to row (m // seg_rows) * seg_stride + m % seg_rows of `out`
    (out: [(M // seg_rows) * seg_stride, >= N] view): every batch row's block into its slot of a longer per-row buffer."""
    _f16(A, B, out, bias)
    M, K = A.shape
    N = B.shape[0]
    assert B.shape[1] == K and M % seg_rows == 0 and out.shape[0] >= (M // seg_rows - 1) * seg_stride + seg_rows and out.shape[1] >= N
    try:
        check(lib.skg_gemm_f16_rows(_p(A), _ld(A), _p(B), _ld(B), _p(out), _ld(out), M, N, K, _p(bias), seg_rows, seg_stride, _stream()),
              "skg_gemm_f16_rows")
    except SkgError as e:      # declined: one GEMM per batch row straight into its slot
        if e.rc != -2:
            raise
        for b in range(M // seg_rows):
            gemm(A[b * seg_rows:(b + 1) * seg_rows], B, out=out[b * seg_stride:b * seg_stride + seg_rows, :N], bias=bias)
    return out


def conv4x4s2(X: torch.Tensor, W16: torch.Tensor, rows: int, IH: int, IW: int, out: Optional[torch.Tensor] = None, *, bias=None, W9T=None):
    """4 x 4 stride-2 convolution, padding 1 (the data gradient of conv_up2).  X [rows*IH*IW, Cin] (view), W16 [Cout, 16*Cin]
    (unet.pack_conv_up2_dgrad).  Returns [rows*(IH/2)*(IW/2), Cout]."""
    _f16(X, W16, bias)
    Cin, Cout = X.shape[1], W16.shape[0]
    assert W16.shape == (Cout, 16 * Cin) and W16.is_contiguous() and X.shape[0] == rows * IH * IW
    if out is None:
        out = torch.empty(rows * (IH // 2) * (IW // 2), Cout, device=X.device, dtype=torch.float16)
    try:
        check(lib.skg_conv4x4s2_f16(_p(X), _ld(X), _p(W16), _p(out), _ld(out), rows, IH, IW, Cin, Cout, _p(bias), _stream()),
              "skg_conv4x4s2_f16")
    except SkgError as e:      # declined: the 9-tap dgrad at the upsampled size + 2 x 2 sum-pool it replaced
        if e.rc != -2 or W9T is None or bias is not None:
            raise
        return sumpool2x2(conv3x3(X, W9T, rows, IH, IW, bias=bias), rows, IH // 2, IW // 2, out=out)
    return out


_scratch = {}


def _gn_scratch(rows: int, groups: int, dev) -> torch.Tensor:
    n = lib.skg_groupnorm_scratch_floats(rows, groups)
    return _scratch_buf(("gn", _skey(dev), n), n, dev)


def groupnorm_stats(X, rows, HW, groups, eps, stats=None):
    _f16(X)
    C = X.shape[1]
    if stats is None:
        stats = torch.empty(rows, groups, 2, device=X.device, dtype=torch.float32)
    check(lib.skg_groupnorm_stats(_p(X), _ld(X), rows, HW, C, groups, eps, _p(stats),
                                  _p(_gn_scratch(rows, groups, X.device)), _stream()), "skg_groupnorm_stats")
    return stats


def groupnorm_apply(X, rows, HW, groups, stats, gamma, beta, silu: bool, out=None):
    _f16(X, gamma, beta)
    C = X.shape[1]
    if out is None:
        out = torch.empty(X.shape[0], C, device=X.device, dtype=torch.float16)
    check(lib.skg_groupnorm_apply(_p(X), _ld(X), _p(out), _ld(out), rows, HW, C, groups, _p(stats), _p(gamma),
                                  _p(beta), int(silu), _stream()), "skg_groupnorm_apply")
    return out


def groupnorm(X, rows, HW, groups, eps, gamma, beta, silu: bool, out=None, partial: Optional[GNPartial] = None):
    """Forward GroupNorm.  partial=: the producer of X already left the chunk sums behind (gemm / conv3x3 with
    gn_stats= / gn_groups=): ONE launch that folds them and applies, X is read once.
    Otherwise:  Up to 32x32 maps: two launches (chunk partial sums; apply, which folds the partials itself
    and publishes the statistics) - at 64x64 the 86 chunk partials per group make the in-kernel fold dearer than the
    4.7 us finalize launch it replaces, so the three-launch path stays (measured: tools/ew_bench.py)."""
    if isinstance(partial, tuple):
        # X = [A | B], each half written by its own producer: (GNPartial of A, channels of A, GNPartial of B)
        pa, CA, pb = partial
        assert pa.rows == rows and pb.rows == rows and pa.nch == pb.nch == HW // 128
        _f16(X, gamma, beta)
        C = X.shape[1]
        if out is None:
            out = torch.empty(X.shape[0], C, device=X.device, dtype=torch.float16)
        st = torch.empty(rows, groups, 2, device=X.device, dtype=torch.float32)
        check(lib.skg_groupnorm_from_partial2(_p(X), _ld(X), _p(out), _ld(out), rows, HW, C, CA, groups, eps, _p(gamma),
                                              _p(beta), int(silu), _p(st), _p(pa.buf), pa.groups, _p(pb.buf),
                                              pb.groups, pa.nch, _stream()), "skg_groupnorm_from_partial2")
        return out, st
    if partial is not None:
        assert partial.rows == rows and partial.groups == groups and partial.nch == HW // 128
        _f16(X, gamma, beta)
        C = X.shape[1]
        if out is None:
            out = torch.empty(X.shape[0], C, device=X.device, dtype=torch.float16)
        st = torch.empty(rows, groups, 2, device=X.device, dtype=torch.float32)
        check(lib.skg_groupnorm_from_partial(_p(X), _ld(X), _p(out), _ld(out), rows, HW, C, groups, eps, _p(gamma),
                                             _p(beta), int(silu), _p(st), _p(partial.buf), partial.nch, _stream()),
              "skg_groupnorm_from_partial")
        return out, st
    if HW >= 4096:
        st = groupnorm_stats(X, rows, HW, groups, eps)
        return groupnorm_apply(X, rows, HW, groups, st, gamma, beta, silu, out), st
    _f16(X, gamma, beta)
    C = X.shape[1]
    if out is None:
        out = torch.empty(X.shape[0], C, device=X.device, dtype=torch.float16)
    st = torch.empty(rows, groups, 2, device=X.device, dtype=torch.float32)
    check(lib.skg_groupnorm_fwd(_p(X), _ld(X), _p(out), _ld(out), rows, HW, C, groups, eps, _p(gamma), _p(beta),
                                int(silu), _p(st), _p(_gn_scratch(rows, groups, X.device)), _stream()),
          "skg_groupnorm_fwd")
    return out, st


def groupnorm_hilo(X, X_lo, rows, HW, groups, eps, gamma, beta, silu: bool, out=None, want_stats=False, partial=None, out_lo=None):
    """GroupNorm(+SiLU) of the pair X + X_lo (accuracy mode): statistics from the hi part (the producer's epilogue sums when
    `partial` - a GNPartial, or (GNPartial of A, channels of A, GNPartial of B) for a concatenation - is given, else an own
    pass; small maps: one launch on the pair's sum), apply on the sum.  out_lo: the output as a pair too (pitch of `out`)."""
    _f16(X, X_lo, gamma, beta, out_lo)
    assert _ld(X) == _ld(X_lo) and (out_lo is None or (out is not None and _ld(out_lo) == _ld(out)))
    C = X.shape[1]
    if out is None:
        out = torch.empty(X.shape[0], C, device=X.device, dtype=torch.float16)
    st = torch.empty(rows, groups, 2, device=X.device, dtype=torch.float32)
    if partial is not None:
        pa, CA, pb = partial if isinstance(partial, tuple) else (partial, 0, None)
        assert pa.rows == rows and pa.nch == HW // 128 and (pb is not None or pa.groups == groups)
        check(lib.skg_groupnorm_from_partial_hilo(_p(X), _p(X_lo), _ld(X), _p(out), _p(out_lo), _ld(out), rows, HW, C, CA, groups, eps,
                                                  _p(gamma), _p(beta), int(silu), _p(st), _p(pa.buf), pa.groups,
                                                  None if pb is None else _p(pb.buf), 0 if pb is None else pb.groups, pa.nch,
                                                  _stream()), "skg_groupnorm_from_partial_hilo")
    else:
        check(lib.skg_groupnorm_fwd_hilo(_p(X), _p(X_lo), _ld(X), _p(out), _p(out_lo), _ld(out), rows, HW, C, groups, eps, _p(gamma),
                                         _p(beta), int(silu), _p(st), _p(_gn_scratch(rows, groups, X.device)), _stream()),
              "skg_groupnorm_fwd_hilo")
    return (out, st) if want_stats else out


def layernorm_hilo(X, X_lo, gamma, beta, eps=1e-5, out=None, want_stats=False):
    _f16(X, X_lo, gamma, beta)
    assert _ld(X) == _ld(X_lo)
    M, C = X.shape
    if out is None:
        out = torch.empty(M, C, device=X.device, dtype=torch.float16)
    stats = torch.empty(M, 2, device=X.device, dtype=torch.float32) if want_stats else None
    check(lib.skg_layernorm_fwd_hilo(_p(X), _p(X_lo), _ld(X), _p(out), _ld(out), M, C, _p(gamma), _p(beta), eps, _p(stats),
                                     _stream()), "skg_layernorm_fwd_hilo")
    return (out, stats) if want_stats else out


def groupnorm_bwd(X, dY, rows, HW, groups, stats, gamma, beta, silu: bool, residual=None, out=None):
    _f16(X, dY, gamma, beta, residual)
    C = X.shape[1]
    if out is None:
        out = torch.empty(X.shape[0], C, device=X.device, dtype=torch.float16)
    check(lib.skg_groupnorm_bwd(_p(X), _ld(X), _p(dY), _ld(dY), _p(out), _ld(out), _p(residual),
                                _ld(residual) if residual is not None else 0, rows, HW, C, groups, _p(stats),
                                _p(gamma), _p(beta), int(silu), _p(_gn_scratch(rows, groups, X.device)),
                                _stream()), "skg_groupnorm_bwd")
    return out


def layernorm(X, gamma, beta, eps=1e-5, out=None, want_stats=False):
    _f16(X, gamma, beta)
    M, C = X.shape
    if out is None:
        out = torch.empty(M, C, device=X.device, dtype=torch.float16)
    stats = torch.empty(M, 2, device=X.device, dtype=torch.float32) if want_stats else None
    check(lib.skg_layernorm_fwd(_p(X), _ld(X), _p(out), _ld(out), M, C, _p(gamma), _p(beta), eps, _p(stats),
                                _stream()), "skg_layernorm_fwd")
    return (out, stats) if want_stats else out


def layernorm_bwd(X, dY, gamma, stats, residual=None, out=None):
    _f16(X, dY, gamma, residual)
    M, C = X.shape
    if out is None:
        out = torch.empty(M, C, device=X.device, dtype=torch.float16)
    check(lib.skg_layernorm_bwd(_p(X), _ld(X), _p(dY), _ld(dY), _p(out), _ld(out), _p(residual),
                                _ld(residual) if residual is not None else 0, M, C, _p(gamma), _p(stats),
                                _stream()), "skg_layernorm_bwd")
    return out


def geglu(H, out=None, interleaved: bool = False):
    _f16(H)
    M, F2 = H.shape
    F = F2 // 2
    if out is None:
        out = torch.empty(M, F, device=H.device, dtype=torch.float16)
    check(lib.skg_geglu_fwd(_p(H), _ld(H), _p(out), _ld(out), M, F, int(interleaved), _stream()), "skg_geglu_fwd")
    return out


def geglu_bwd(H, dY, out=None, interleaved: bool = False):
    _f16(H, dY)
    M, F2 = H.shape
    if out is None:
        out = torch.empty(M, F2, device=H.device, dtype=torch.float16)
    check(lib.skg_geglu_bwd(_p(H), _ld(H), _p(dY), _ld(dY), _p(out), _ld(out), M, F2 // 2, int(interleaved),
                            _stream()), "skg_geglu_bwd")
    return out


def geglu_interleave_index(F: int) -> torch.Tensor:
    """Row permutation that turns diffusers' FF1 weight [a rows (F) ; g rows (F)] into the interleaved pack:
    packed row 4t+e is a_{2t+e} for e < 2 and g_{2t+e-2} for e >= 2."""
    r = torch.arange(2 * F)
    t, e = r // 4, r % 4
    return torch.where(e < 2, 2 * t + e, F + 2 * t + (e - 2))


def transpose(X, out=None):
    _f16(X)
    M, C = X.shape
    if out is None:
        out = torch.empty(C, M, device=X.device, dtype=torch.float16)
    check(lib.skg_transpose_f16(_p(X), _ld(X), _p(out), _ld(out), M, C, _stream()), "skg_transpose_f16")
    return out


def axpby(A, B=None, out=None, alpha=1.0, beta=1.0):
    _f16(A, B)
    M, C = A.shape
    if out is None:
        out = torch.empty(M, C, device=A.device, dtype=torch.float16)
    check(lib.skg_axpby_f16(_p(A), _ld(A), _p(B), _ld(B) if B is not None else 0, _p(out), _ld(out), M, C,
                            alpha, beta, _stream()), "skg_axpby_f16")
    return out


def batch_copy(X, in_batch_rows, out, out_batch_rows, batches, rows_per_batch):
    """out[b*out_batch_rows + r] = X[b*in_batch_rows + r] for r < rows_per_batch."""
    _f16(X, out)
    check(lib.skg_batch_copy_f16(_p(X), _ld(X), in_batch_rows, _p(out), _ld(out), out_batch_rows, batches,
                                 rows_per_batch, X.shape[1], _stream()), "skg_batch_copy_f16")
    return out


def silu(X, out=None):
    _f16(X)
    M, C = X.shape
    if out is None:
        out = torch.empty(M, C, device=X.device, dtype=torch.float16)
    check(lib.skg_silu_f16(_p(X), _ld(X), _p(out), _ld(out), M, C, _stream()), "skg_silu_f16")
    return out


def quick_gelu(X, out=None):
    _f16(X)
    M, C = X.shape
    if out is None:
        out = torch.empty(M, C, device=X.device, dtype=torch.float16)
    check(lib.skg_quick_gelu_f16(_p(X), _ld(X), _p(out), _ld(out), M, C, _stream()), "skg_quick_gelu_f16")
    return out


def gelu(X, out=None):
    _f16(X)
    M, C = X.shape
    if out is None:
        out = torch.empty(M, C, device=X.device, dtype=torch.float16)
    check(lib.skg_gelu_f16(_p(X), _ld(X), _p(out), _ld(out), M, C, _stream()), "skg_gelu_f16")
    return out


def sumpool2x2(X, rows, H, W, out=None):
    """X [rows*2H*2W, C] -> [rows*H*W, C]."""
    _f16(X)
    C = X.shape[1]
    if out is None:
        out = torch.empty(rows * H * W, C, device=X.device, dtype=torch.float16)
    check(lib.skg_sumpool2x2_f16(_p(X), _ld(X), _p(out), _ld(out), rows, H, W, C, _stream()),
          "skg_sumpool2x2_f16")
    return out


def nchw_to_nhwc(X: torch.Tensor, Cpad: int, out=None):
    """float32 [rows, C, H, W] -> fp16 [rows*H*W, Cpad]."""
    assert X.is_cuda and X.dtype == torch.float32 and X.is_contiguous()
    rows, C, H, W = X.shape
    if out is None:
        out = torch.empty(rows * H * W, Cpad, device=X.device, dtype=torch.float16)
    check(lib.skg_nchw_f32_to_nhwc_f16(_p(X), _p(out), rows, C, H * W, Cpad, _stream()),
          "skg_nchw_f32_to_nhwc_f16")
    return out


def nhwc_to_nchw(X, rows: int, C: int, H: int, W: int):
    if isinstance(X, Pair):      # accuracy mode: hi + lo in fp32
        return nhwc_to_nchw(X.hi, rows, C, H, W) + nhwc_to_nchw(X.lo, rows, C, H, W)
    _f16(X)
    out = torch.empty(rows, C, H, W, device=X.device, dtype=torch.float32)
    check(lib.skg_nhwc_f16_to_nchw_f32(_p(X), _ld(X), _p(out), rows, C, H * W, _stream()),
          "skg_nhwc_f16_to_nchw_f32")
    return out


def attn_fwd(Q, K, Vt, batch, heads, Nq, Nkv, kv_stride, dh, scale, out=None, want_lse=False, causal=False,
             v_rows=False):
    """v_rows=True: `Vt` is the row-major V ([batch*kv_stride, heads*dh] view, any row pitch) instead of its transpose."""
    _f16(Q, K, Vt)
    assert not (causal and v_rows)
    if out is None:
        out = torch.empty(batch * Nq, heads * dh, device=Q.device, dtype=torch.float16)
    lse = torch.empty(batch, heads, Nq, device=Q.device, dtype=torch.float32) if want_lse else None
    fn = lib.skg_attn_fwd_causal if causal else lib.skg_attn_fwd_rowv if v_rows else lib.skg_attn_fwd
    check(fn(_p(Q), _ld(Q), _p(K), _ld(K), _p(Vt), _ld(Vt), _p(out), _ld(out), _p(lse), batch,
             heads, Nq, Nkv, kv_stride, dh, scale, _stream()), "skg_attn_fwd")
    return (out, lse) if want_lse else out


def attn_bwd_delta(O, dO, batch, heads, Nq, dh):
    _f16(O, dO)
    delta = torch.empty(batch, heads, Nq, device=O.device, dtype=torch.float32)
    check(lib.skg_attn_bwd_delta(_p(O), _ld(O), _p(dO), _ld(dO), _p(delta), batch, heads, Nq, dh, _stream()),
          "skg_attn_bwd_delta")
    return delta


def attn_bwd_dq(Q, K, V, dO, lse, delta, batch, heads, Nq, Nkv, kv_stride, dh, scale, out=None):
    _f16(Q, K, V, dO)
    if out is None:
        out = torch.empty(batch * Nq, heads * dh, device=Q.device, dtype=torch.float16)
    check(lib.skg_attn_bwd_dq(_p(Q), _ld(Q), _p(K), _ld(K), _p(V), _ld(V), _p(dO), _ld(dO),
                              _p(lse), _p(delta), _p(out), _ld(out), batch, heads, Nq, Nkv, kv_stride, dh,
                              scale, _stream()), "skg_attn_bwd_dq")
    return out


def attn_bwd_dq_delta(Q, K, V, dO, O, lse, batch, heads, Nq, Nkv, kv_stride, dh, scale, out=None):
    """attn_bwd_delta + attn_bwd_dq in one launch (skg_attn_bwd_dq_delta) -> (dQ, delta)."""
    _f16(Q, K, V, dO, O)
    if out is None:
        out = torch.empty(batch * Nq, heads * dh, device=Q.device, dtype=torch.float16)
    delta = torch.empty(batch, heads, Nq, device=Q.device, dtype=torch.float32)
    check(lib.skg_attn_bwd_dq_delta(_p(Q), _ld(Q), _p(K), _ld(K), _p(V), _ld(V), _p(dO), _ld(dO), _p(O), _ld(O), _p(lse), _p(delta),
                                    _p(out), _ld(out), batch, heads, Nq, Nkv, kv_stride, dh, scale, _stream()), "skg_attn_bwd_dq_delta")
    return out, delta


def attn_bwd_dkv(Q, K, V, dO, lse, delta, batch, heads, Nq, Nkv, dh, scale, dK=None, dV=None):
    _f16(Q, K, V, dO)
    if dK is None:
        dK = torch.empty(batch * Nkv, heads * dh, device=Q.device, dtype=torch.float16)
    if dV is None:
        dV = torch.empty(batch * Nkv, heads * dh, device=Q.device, dtype=torch.float16)
    check(lib.skg_attn_bwd_dkv(_p(Q), _ld(Q), _p(K), _ld(K), _p(V), _ld(V), _p(dO), _ld(dO),
                               _p(lse), _p(delta), _p(dK), _ld(dK), _p(dV), _ld(dV), batch,
                               heads, Nq, Nkv, dh, scale, _stream()), "skg_attn_bwd_dkv")
    return dK, dV


# ---- LGP --------------------------------------------------------------------------------------------
def lgp_layer0_gather(P: Sequence[torch.Tensor], sizes: Sequence[int], Wextra, bias0, noise, sigma: float,
                      samples: int, h: int, H0: int, out=None, rows: Optional[int] = None):
    """rows defaults to 2*samples ([uncond ; cond] blocks of the sampler); the trainer passes rows = samples."""
    rows = 2 * samples if rows is None else rows
    arr = (SkgTap * len(P))()
    for i, (t, s) in enumerate(zip(P, sizes)):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == (rows * s * s, H0)
        arr[i].P, arr[i].s = t.data_ptr(), s
    if out is None:
        out = torch.empty(rows * h * h, H0, device=noise.device, dtype=torch.float16)
    check(lib.skg_lgp_layer0_gather(ctypes.addressof(arr), len(P), _p(Wextra), _ld(Wextra) if Wextra is not None else 0, _p(bias0),
                                    _p(noise), sigma, samples, _p(out), rows, h, H0, _stream()),
          "skg_lgp_layer0_gather")
    return out


def lgp_layer0_scatter(dZ, rows, h, s, H0):
    _f16(dZ)
    if s == h and dZ.stride(0) == H0:          # the adjoint of an identity resize: no copy
        return dZ
    out = torch.empty(rows * s * s, H0, device=dZ.device, dtype=torch.float16)
    check(lib.skg_lgp_layer0_scatter(_p(dZ), _ld(dZ), _p(out), rows, h, s, H0, _stream()),
          "skg_lgp_layer0_scatter")
    return out


def _bn_scratch(samples, C, dev):
    n = lib.skg_bn_scratch_floats(samples, C)
    return _scratch_buf(("bn", _skey(dev), n), n, dev)


def bn_stats(X, samples, segs, seg_rows, eps=1e-5, running_mean=None, running_var=None):
    _f16(X)
    C = X.shape[1]
    stats = torch.empty(samples, C, 2, device=X.device, dtype=torch.float32)
    check(lib.skg_bn_stats(_p(X), _ld(X), samples, segs, seg_rows, C, eps, _p(stats),
                           _p(_bn_scratch(samples, C, X.device)), _p(running_mean), _p(running_var), _stream()),
          "skg_bn_stats")
    return stats


def bn_stats_from_running(running_mean, running_var, samples, eps=1e-5):
    C = running_mean.numel()
    stats = torch.empty(samples, C, 2, device=running_mean.device, dtype=torch.float32)
    check(lib.skg_bn_stats_from_running(_p(running_mean), _p(running_var), samples, C, eps, _p(stats), _stream()),
          "skg_bn_stats_from_running")
    return stats


def bn_apply(X, samples, segs, seg_rows, stats, gamma, beta, out=None):
    _f16(X, gamma, beta)
    C = X.shape[1]
    if out is None:
        out = torch.empty(X.shape[0], C, device=X.device, dtype=torch.float16)
    check(lib.skg_bn_apply(_p(X), _ld(X), _p(out), _ld(out), samples, segs, seg_rows, C, _p(stats), _p(gamma),
                           _p(beta), _stream()), "skg_bn_apply")
    return out


def bn_relu_bwd(X, dY, samples, segs, seg_rows, stats, gamma, train: bool, out=None):
    _f16(X, dY, gamma)
    C = X.shape[1]
    if out is None:
        out = torch.empty(X.shape[0], C, device=X.device, dtype=torch.float16)
    check(lib.skg_bn_relu_bwd(_p(X), _ld(X), _p(dY), _ld(dY), _p(out), _ld(out), samples, segs, seg_rows, C,
                              _p(stats), _p(gamma), int(train), _p(_bn_scratch(samples, C, X.device)), _stream()),
          "skg_bn_relu_bwd")
    return out


def lgp_mse_seed(out16, target, samples, h, ldd, loss_scale):
    _f16(out16)
    dOut = torch.empty(2 * samples * h * h, ldd, device=out16.device, dtype=torch.float16)
    loss = torch.empty(samples, device=out16.device, dtype=torch.float32)
    check(lib.skg_lgp_mse_seed(_p(out16), _ld(out16), _p(target), _p(dOut), ldd, _p(loss), samples, h,
                               loss_scale, _stream()), "skg_lgp_mse_seed")
    return dOut, loss


# ---- LGP training ------------------------------------------------------------------------------------
def colsum(X, scale: float = 1.0):
    _f16(X)
    M, C = X.shape
    out = torch.empty(C, device=X.device, dtype=torch.float32)
    scr = _scratch_buf(("colsum", _skey(X.device), C), lib.skg_colsum_scratch_floats(C), X.device)
    check(lib.skg_colsum_f16(_p(X), _ld(X), M, C, scale, _p(out), _p(scr), _stream()), "skg_colsum_f16")
    return out


def bn_param_grads(X, dY, stats, scale: float = 1.0):
    """(dgamma, dbeta) of one BatchNorm1d batch = all rows of X; stats [1, C, 2] from bn_stats(samples=1)."""
    _f16(X, dY)
    M, C = X.shape
    dg = torch.empty(C, device=X.device, dtype=torch.float32)
    db = torch.empty(C, device=X.device, dtype=torch.float32)
    check(lib.skg_bn_param_grads(_p(X), _ld(X), _p(dY), _ld(dY), M, C, _p(stats), scale, _p(dg), _p(db),
                                 _p(_bn_scratch(1, C, X.device)), _stream()), "skg_bn_param_grads")
    return dg, db


def lgp_extra_features(noise, sigma: float, samples: int, rows: int, h: int, ld: int = 64):
    out = torch.empty(rows * h * h, ld, device=noise.device, dtype=torch.float16)
    check(lib.skg_lgp_extra_features(_p(noise), sigma, samples, rows, h, _p(out), ld, _stream()),
          "skg_lgp_extra_features")
    return out


def lgp_mse_train(out16, target, samples, h, ldd, loss_scale):
    _f16(out16)
    dOut = torch.empty(samples * h * h, ldd, device=out16.device, dtype=torch.float16)
    parts = torch.empty(samples, device=out16.device, dtype=torch.float32)
    check(lib.skg_lgp_mse_train(_p(out16), _ld(out16), _p(target), _p(dOut), ldd, _p(parts), samples, h, loss_scale,
                                _stream()), "skg_lgp_mse_train")
    return dOut, parts


def adamw_step(param, grad, exp_avg, exp_avg_sq, param_f16, lr, beta1, beta2, eps, weight_decay, step: int,
               inv_grad_scale: float = 1.0):
    n = param.numel()
    assert param.dtype == grad.dtype == exp_avg.dtype == exp_avg_sq.dtype == torch.float32 and grad.numel() == n
    check(lib.skg_adamw_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(param_f16), n, lr, beta1, beta2,
                             eps, weight_decay, step, inv_grad_scale, _stream()), "skg_adamw_step")


# ---- sampler ------------------------------------------------------------------------------------------
def softmax_rows(X, out=None):
    """Row softmax of fp16 scores (fp32 arithmetic); in place when out is X."""
    _f16(X)
    M, N = X.shape
    if out is None:
        out = torch.empty(M, N, device=X.device, dtype=torch.float16)
    check(lib.skg_softmax_rows_f16(_p(X), _ld(X), _p(out), _ld(out), M, N, _stream()), "skg_softmax_rows_f16")
    return out


def image_postprocess(X, pixels: int, C: int, scale: float = 0.5, shift: float = 0.5):
    """fp16 NHWC rows (first C channels of pitch ld) -> float [pixels, C] = clamp(x*scale + shift, 0, 1)."""
    _f16(X)
    out = torch.empty(pixels, C, device=X.device, dtype=torch.float32)
    check(lib.skg_image_postprocess(_p(X), _ld(X), _p(out), pixels, C, scale, shift, _stream()),
          "skg_image_postprocess")
    return out


def image_to_u8(X, pixels: int, C: int, scale: float = 0.5, shift: float = 0.5):
    """fp16 NHWC rows -> uint8 [pixels, C] = rint(clamp(x*scale + shift, 0, 1) * 255) (numpy_to_pil's quantisation)."""
    _f16(X)
    out = torch.empty(pixels, C, device=X.device, dtype=torch.uint8)
    check(lib.skg_image_to_u8(_p(X), _ld(X), _p(out), pixels, C, scale, shift, _stream()), "skg_image_to_u8")
    return out


def gaussian_sample(moments, samples: int, L: int, HW: int, noise=None, scale: float = 1.0):
    """(mean + exp(0.5*clamp(logvar))*noise)*scale from fp16 NHWC moments [samples*HW, >=2L] -> float [samples, L, HW]."""
    _f16(moments)
    out = torch.empty(samples, L, HW, device=moments.device, dtype=torch.float32)
    check(lib.skg_gaussian_sample(_p(moments), _ld(moments), _p(noise), _p(out), samples, L, HW, scale, _stream()),
          "skg_gaussian_sample")
    return out


def eps_halves(eps, samples, HW):
    """(uncond rows, cond rows, pair offset) of the UNet's eps for cfg_*_step: fp16 [2 S HW, >= 4], or - accuracy mode - a
    Pair whose lo part sits a fixed number of columns to the right of the hi part in the same buffer."""
    if isinstance(eps, Pair):
        off = (eps.lo.data_ptr() - eps.hi.data_ptr()) // 2
        assert _ld(eps.hi) == _ld(eps.lo) and 0 < off <= _ld(eps.hi) - 4
        return eps.hi[:samples * HW], eps.hi[samples * HW:], off
    return eps[:samples * HW], eps[samples * HW:], 0


def cfg_ddim_step(eps_u, eps_c, x, samples, HW, g, coeffs: Tuple[float, float, float, float],
                  want_eps=False, lo_off: int = 0, v_prediction: bool = False):
    """lo_off != 0: eps_u / eps_c are the hi parts of pairs whose lo parts sit lo_off columns to the right (eps_halves).
    v_prediction: the model output is v (eps = c0 v + c1 x, x0 = c0 x - c1 v); the returned eps is the derived one."""
    _f16(eps_u, eps_c)
    x_prev = torch.empty_like(x)
    eps_out = torch.empty_like(x) if want_eps else None
    c0, c1, c2, c3 = coeffs
    check(lib.skg_cfg_ddim_step(_p(eps_u), _p(eps_c), _ld(eps_u), lo_off, _p(x), _p(x_prev), _p(eps_out), samples, HW,
                                g, c0, c1, c2, c3, int(bool(v_prediction)), _stream()), "skg_cfg_ddim_step")
    return (x_prev, eps_out) if want_eps else x_prev


def cfg_dpmpp2m_step(eps_u, eps_c, x, x0_io, samples, HW, g, coeffs: Tuple[float, float, float, float, float],
                     want_eps=False, lo_off: int = 0, v_prediction: bool = False):
    """coeffs = (alpha_s, sigma_s, a, b, c); x0_io is updated in place (previous x0 in, this step's x0 out)."""
    _f16(eps_u, eps_c)
    x_prev = torch.empty_like(x)
    eps_out = torch.empty_like(x) if want_eps else None
    al, sg, a, b, c = coeffs
    check(lib.skg_cfg_dpmpp2m_step(_p(eps_u), _p(eps_c), _ld(eps_u), lo_off, _p(x), _p(x0_io), _p(x_prev), _p(eps_out),
                                   samples, HW, g, al, sg, a, b, c, int(bool(v_prediction)), _stream()), "skg_cfg_dpmpp2m_step")
    return (x_prev, eps_out) if want_eps else x_prev


def guidance_update(grad, x_in, x_prev, samples, HW, beta):
    """In place: x_prev += alpha * (-grad).  Returns aux [samples,4] = (alpha, ||g||, sqrt2*||dx||, 0)."""
    _f16(grad)
    aux = torch.empty(samples, 4, device=grad.device, dtype=torch.float32)
    check(lib.skg_guidance_update(_p(grad), _ld(grad), _p(x_in), _p(x_prev), _p(aux), samples, HW, beta,
                                  _stream()), "skg_guidance_update")
    return aux
