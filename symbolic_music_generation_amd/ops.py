"""Thin torch-tensor front-ends over the C ABI (include/musicxl.h).

torch is used only for device memory and the current HIP stream; every function here enqueues hand-written HIP kernels
from libmusicxl.so and raises if the library is missing or a launch fails.  No CPU / eager fallbacks.
"""
import ctypes as C
import math
import os
from typing import Optional, Sequence

import torch

from ._lib import lib, check, MusicXLError

GEMM_OUT_F32 = 0x01
GEMM_OUT_F32_ATOMIC = 0x02
GEMM_BIAS = 0x04
GEMM_RELU = 0x08
GEMM_DROPOUT = 0x10
GEMM_RELU_BWD = 0x20
GEMM_ADD_AUX = 0x40
GEMM_SAVE_RELU_MASK = 0x100
GEMM_RELU_BWD_BITS = 0x200


def mix_seed(base_seed: int, step: int) -> int:
    """Dropout / LSH-rotation seed of one step: a 63-bit mix of (base seed, data-parallel rank, step).  Under data parallelism
    every rank draws its own masks (as DDP ranks do in the reference stack) while the weight-initialisation seed stays
    common; no collisions between runs once `step` passes 2^20 (the former `(seed << 20) + step`)."""
    import torch.distributed as dist
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    x = (base_seed * 0x9E3779B97F4A7C15 + rank * 0xD1B54A32D192ED03 + step * 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 32
    x = (x * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 32
    return x & 0x7FFFFFFFFFFFFFFF


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise MusicXLError(f'{name}: expected a device tensor (the HIP path has no CPU fallback)')
    if t.dtype != dtype:
        raise MusicXLError(f'{name}: expected {dtype}, got {t.dtype}')


def _cut(cutoffs: Sequence[int]):
    n = len(cutoffs)
    arr = (C.c_int * max(n, 1))(*cutoffs) if n else None
    return n, arr


KT_NAMES = ('fwd', 'delta', 'dq8', 'dkv', 'drd', 'rowbias', 'fused', 'dqfin', 'chunk_fwd', 'chunk_bwd_q', 'chunk_bwd_kv')   # MXL_KT_* ids of include/musicxl.h


def ktime_enable(on: bool):
    """per-kernel hipEvent brackets inside the attention launch functions (bench.py's roofline); off while capturing graphs"""
    check(lib().mxl_ktime_enable(int(bool(on))), 'mxl_ktime_enable')


def ktime_collect():
    """{kernel: (milliseconds summed, launches)} since the last collect; synchronises the recorded events"""
    n = len(KT_NAMES)
    ms, cnt = (C.c_float * n)(), (C.c_int * n)()
    check(lib().mxl_ktime_collect(C.cast(ms, C.c_void_p), C.cast(cnt, C.c_void_p), n), 'mxl_ktime_collect')
    return {KT_NAMES[i]: (float(ms[i]), int(cnt[i])) for i in range(n)}


def gemm(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, M: int, N: int, K: int, *, trans_a=False, trans_b=False,
         flags=0, alpha=1.0, bias: Optional[torch.Tensor] = None, aux: Optional[torch.Tensor] = None, ksplits=1,
         lda=None, ldb=None, ldc=None, ldaux=None, drop_p=0.0, seed=0, site=0, colsum: Optional[torch.Tensor] = None):
    """C[M,N] (+)= alpha * op(A) op(B); see mxl_gemm_bf16.  Leading dimensions default to the last-dim stride.
    `colsum` (N,) f32: += the column sums of C in the same call (mxl_gemm_bf16_colsum; bf16 output, ksplits == 1)."""
    _req(a, torch.bfloat16, 'A'); _req(b, torch.bfloat16, 'B')
    lda = lda if lda is not None else a.stride(-2)
    ldb = ldb if ldb is not None else b.stride(-2)
    ldc = ldc if ldc is not None else c.stride(-2)
    if aux is not None and ldaux is None:
        ldaux = 0 if flags & (GEMM_SAVE_RELU_MASK | GEMM_RELU_BWD_BITS) else aux.stride(-2)
    if colsum is not None:
        assert ksplits == 1
        check(lib().mxl_gemm_bf16_colsum(_p(a), _p(b), _p(c), M, N, K, lda, ldb, ldc, int(trans_a), int(trans_b), flags,
                                         float(alpha), _p(bias), _p(aux), ldaux or 0, float(drop_p), seed, site, _p(colsum),
                                         _stream()), 'mxl_gemm_bf16_colsum')
        return c
    check(lib().mxl_gemm_bf16(_p(a), _p(b), _p(c), M, N, K, lda, ldb, ldc, int(trans_a), int(trans_b), flags,
                              float(alpha), _p(bias), _p(aux), ldaux or 0, ksplits, float(drop_p), seed, site,
                              _stream()), 'mxl_gemm_bf16')
    return c


def gemm_relu_mask_bytes(M: int, N: int) -> int:
    """bytes of the relu-mask bit buffer GEMM_SAVE_RELU_MASK / GEMM_RELU_BWD_BITS use for an (M, N) output; 0 = not available at
    these sizes (use GEMM_RELU_BWD with the activations)"""
    if os.environ.get('MXL_NO_RELU_BITS') == '1':
        return 0
    return int(lib().mxl_gemm_relu_mask_bytes(int(M), int(N)))


def gemm_skinny(a, w, c, M, N, K, *, flags=0, bias=None, lda=None, ldw=None, ldc=None):
    """C[M<=64, N] = A W^T (+bias)(relu): weight-streaming form used by the decode step"""
    check(lib().mxl_gemm_skinny_bf16(_p(a), _p(w), _p(c), M, N, K, lda if lda is not None else a.stride(-2),
                                     ldw if ldw is not None else w.stride(-2), ldc if ldc is not None else c.stride(-2),
                                     flags, _p(bias), _stream()), 'mxl_gemm_skinny_bf16')
    return c


def gemm_skinny_partial(a, w, slabs, M, N, K, KS):
    """K-sliced skinny product: slabs (KS, 64, N) f32 receive the partial sums (finish with ln_residual_fwd_partial)"""
    check(lib().mxl_gemm_skinny_partial(_p(a), _p(w), _p(slabs), M, N, K, a.stride(-2), w.stride(-2), KS, _stream()),
          'mxl_gemm_skinny_partial')


def ln_residual_fwd_partial(slabs, KS, bias, res, gamma, beta, y, eps=1e-5):
    """y = LayerNorm(res + bf16(sum of the KS slabs + bias)); slabs (KS, 64, d) f32, res / y (N <= 64, d) bf16"""
    N, d = res.shape
    check(lib().mxl_ln_residual_fwd_partial(_p(slabs), KS, slabs.stride(0), _p(bias), _p(res), _p(gamma), _p(beta), _p(y), N, d,
                                            float(eps), _stream()), 'mxl_ln_residual_fwd_partial')
    return y


def decode_qkv(x, wqkv, qkv, kc, vc, t_dev, rrb, qr_out, dh):
    """one decode step's qkv projection + K/V ring append + (q + r_r_bias), fused (batch <= 64)"""
    B, d = x.shape
    check(lib().mxl_decode_qkv(_p(x), _p(wqkv), _p(qkv), _p(kc), _p(vc), _p(t_dev), _p(rrb), _p(qr_out), B, d, dh, kc.shape[-2],
                               _stream()), 'mxl_decode_qkv')


def sample_step(scores, V, ids, t_dev, rng_ctr, seed, E, emb_out, scale, counter, do_sample=False, top_k=0, top_p=1.0,
                temperature=1.0, repetition_penalty=1.0, typical_p=1.0):
    """sampler + embedding row of the sampled token (-> emb_out (B, d) bf16) + counter advance, one launch (V <= 2048).
    scores (B, >= V) f32: log-probabilities, or raw logits when repetition_penalty == 1."""
    B = scores.shape[0]
    check(lib().mxl_sample_step(_p(scores), scores.stride(0), int(V), _p(ids), ids.stride(0), _p(t_dev), _p(rng_ctr), seed, B,
                                int(do_sample), int(top_k or 0), float(top_p if top_p is not None else 1.0), float(temperature),
                                float(repetition_penalty if repetition_penalty is not None else 1.0),
                                float(typical_p if typical_p is not None else 1.0), _p(E), _p(emb_out), emb_out.shape[1],
                                float(scale), _p(counter), _stream()), 'mxl_sample_step')


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, relu=False, out=None,
           out_f32=False, drop_p=0.0, seed=0, site=0) -> torch.Tensor:
    """y = x @ w.T (+bias)(relu)(dropout); x (N, K) bf16, w (O, K) bf16."""
    N, K = x.shape
    O = w.shape[0]
    if out is None:
        out = torch.empty(N, O, device=x.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    flags = (GEMM_OUT_F32 if out_f32 else 0) | (GEMM_BIAS if bias is not None else 0) | (GEMM_RELU if relu else 0)
    if drop_p > 0:
        flags |= GEMM_DROPOUT
    return gemm(x, w, out, N, O, K, flags=flags, bias=bias, drop_p=drop_p, seed=seed, site=site)


def sinusoid_table(M: int, d: int, clamp_len: int, device, drop_p=0.0, seed=0, site=0, out=None) -> torch.Tensor:
    if out is None:
        out = torch.empty(M, d, device=device, dtype=torch.bfloat16)
    check(lib().mxl_sinusoid_table(_p(out), M, d, clamp_len, float(drop_p), seed, site, _stream()), 'mxl_sinusoid_table')
    return out


def embed_fwd(ids: torch.Tensor, E: torch.Tensor, out: torch.Tensor, scale: float, drop_p=0.0, seed=0, site=0):
    _req(ids, torch.int64, 'ids'); _req(E, torch.bfloat16, 'E')
    N, d = ids.numel(), E.shape[1]
    check(lib().mxl_embed_fwd(_p(ids), _p(E), _p(out), N, d, E.shape[0], float(scale), float(drop_p), seed, site,
                              _stream()), 'mxl_embed_fwd')
    return out


def embed_bwd(ids: torch.Tensor, dout: torch.Tensor, dE: torch.Tensor, scale: float, drop_p=0.0, seed=0, site=0,
              dout2: Optional[torch.Tensor] = None):
    _req(ids, torch.int64, 'ids'); _req(dout, torch.bfloat16, 'dout'); _req(dE, torch.float32, 'dE')
    N, d = ids.numel(), dE.shape[1]
    check(lib().mxl_embed_bwd(_p(ids), _p(dout), _p(dout2), _p(dE), N, d, dE.shape[0], float(scale), float(drop_p), seed,
                              site, _stream()), 'mxl_embed_bwd')
    return dE


def dropout(x: torch.Tensor, y: torch.Tensor, drop_p: float, seed=0, site=0):
    check(lib().mxl_dropout_bf16(_p(x), _p(y), x.numel(), float(drop_p), seed, site, _stream()), 'mxl_dropout_bf16')
    return y


def ln_residual_fwd(x, res, gamma, beta, y, z=None, mean=None, rstd=None, eps=1e-5, drop_p=0.0, seed=0, site=0):
    _req(x, torch.bfloat16, 'x'); _req(gamma, torch.float32, 'gamma'); _req(beta, torch.float32, 'beta')
    d = x.shape[-1]
    N = x.numel() // d
    check(lib().mxl_ln_residual_fwd(_p(x), _p(res), _p(gamma), _p(beta), _p(y), _p(z), _p(mean), _p(rstd), N, d,
                                    float(eps), float(drop_p), seed, site, _stream()), 'mxl_ln_residual_fwd')
    return y


def ln_residual_bwd(dy, dy2, z, mean, rstd, gamma, dres, dx, dgamma, dbeta, drop_p=0.0, seed=0, site=0, dxsum=None):
    """`dxsum` (d,) f32: += the column sums of dx (the bias gradient of the linear layer that produced x), d <= 1024"""
    d = z.shape[-1]
    N = z.numel() // d
    if dxsum is not None and d <= 1024:
        check(lib().mxl_ln_residual_bwd_colsum(_p(dy), _p(dy2), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx),
                                               _p(dgamma), _p(dbeta), _p(dxsum), N, d, float(drop_p), seed, site, _stream()),
              'mxl_ln_residual_bwd_colsum')
        return
    check(lib().mxl_ln_residual_bwd(_p(dy), _p(dy2), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx),
                                    _p(dgamma), _p(dbeta), N, d, float(drop_p), seed, site, _stream()),
          'mxl_ln_residual_bwd')
    if dxsum is not None:
        colsum(dx, dxsum, N, d)


def ln_bwd_add(dy, dy2, z, mean, rstd, gamma, dadd, dres, dgamma, dbeta):
    """dres = LayerNorm-backward(dy + dy2) + dadd"""
    d = z.shape[-1]
    N = z.numel() // d
    check(lib().mxl_ln_residual_bwd_add(_p(dy), _p(dy2), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dadd), _p(dres), _p(dgamma),
                                        _p(dbeta), N, d, _stream()), 'mxl_ln_residual_bwd_add')


def ln_bwd_add_drop(dy, dy2, z, mean, rstd, gamma, dadd, dres, dx, dxsum, dgamma, dbeta, p: float, seed: int, site: int):
    """dres = LayerNorm-backward(dy + dy2) + dadd;  dx = dropout(dres) under (seed, site);  dxsum (optional) += column sums of dx"""
    d = z.shape[-1]
    N = z.numel() // d
    check(lib().mxl_ln_residual_bwd_add_drop(_p(dy), _p(dy2), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dadd), _p(dres), _p(dx),
                                             _p(dxsum), _p(dgamma), _p(dbeta), N, d, float(p), seed, site, _stream()),
          'mxl_ln_residual_bwd_add_drop')


def dropout_colsum(x: torch.Tensor, y: torch.Tensor, out: torch.Tensor, M: int, N: int, p: float, seed: int, site: int):
    """y = dropout(x) (the mask of mxl_dropout_bf16) and out[n] += column sums of y, one pass"""
    check(lib().mxl_dropout_colsum_bf16(_p(x), _p(y), _p(out), M, N, float(p), seed, site, _stream()), 'mxl_dropout_colsum_bf16')


def colsum(x: torch.Tensor, out: torch.Tensor, M: int, N: int, ld: Optional[int] = None):
    check(lib().mxl_colsum_bf16(_p(x), _p(out), M, N, ld if ld is not None else x.stride(-2), _stream()), 'mxl_colsum_bf16')
    return out


def mem_update(mem, hid, out):
    B, M, d = mem.shape
    T = hid.shape[1]
    check(lib().mxl_mem_update(_p(mem), _p(hid), _p(out), B, M, T, d, _stream()), 'mxl_mem_update')
    return out


def phantom_sum_applies(*, T, dh, M, Kc) -> bool:
    """the shapes at which the forward can hand the backward its phantom value-sum (mxl_relattn_fwd_phantom /
    mxl_relattn_bwd_sparse_dg_oph): the ones at which relattn_bwd leaves the all-phantom dG blocks unwritten"""
    return (dh == 64 and T % 32 == 0 and M % 256 == 0 and Kc < M + T and os.environ.get('MXL_NO_DQ8') != '1'
            and os.environ.get('MXL_DG_RECOMPUTE', '1') != '0' and os.environ.get('MXL_NO_OPH') != '1')


def set_reserved_cus(k: int) -> None:
    """compute units the persistent GEMM grids leave free (mxl_set_reserved_cus; dist.GradSync sets it from MXL_RESERVE_CUS)"""
    check(lib().mxl_set_reserved_cus(int(k)), 'mxl_set_reserved_cus')


def fused_bwd_applies(*, T, dh, M, Kc, B=None, H=None) -> bool:
    """the shapes mxl_relattn_bwd_fused (and, with zero memories, mxl_relattn_drd_phantom) take -- one pass over the score cells, no
    dG tensor: the training shapes of every BASELINE config; anything else stays on relattn_bwd's three kernels.  B, H (optional):
    also the 32-bit offset limits of the phantom-cell kernel's buffer addressing."""
    ok = (dh == 64 and T % 32 == 0 and M % 32 == 0 and M <= 8192 and Kc % 32 == 0 and (T - Kc) % 64 == 0
          and os.environ.get('MXL_NO_FUSED_BWD') != '1')
    if ok and B is not None and H is not None and Kc < M + T:
        ok = H * (T // 32) * 4352 < 2 ** 31 and B * H * T * 4 < 2 ** 31
    return ok


def relattn_fwd(q, k, v, rd, r_w_bias, r_r_bias, out, lse, *, B, T, H, dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs,
                o_bs, o_rs, scale=None, oph=None, mph=None, oph_all=False, ph_buf=None):
    """q/k/v/out may be strided views (e.g. slices of one (B, Kc, 3*H*dh) qkv buffer); strides in elements.
    `oph` (like out) / `mph` (B, H, T) f32: also write the phantom value-sum relattn_bwd(..., oph=, mph=) consumes
    (oph_all: over every phantom cell, the form relattn_bwd_fused consumes; then `ph_buf`, mxl_relattn_drd_phantom_ws_bytes bytes,
    also receives the records of the phantom cells' dRd kernel: pass the same buffer to relattn_bwd_fused(ph_buf=, ph_ready=True))."""
    scale = scale if scale is not None else 1.0 / math.sqrt(dh)
    if oph is not None and oph_all:
        check(lib().mxl_relattn_fwd_phantom2(_p(q), _p(k), _p(v), _p(rd), _p(r_w_bias), _p(r_r_bias), _p(out), _p(lse), _p(oph),
                                             _p(mph), 1, _p(ph_buf), B, T, H, dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs, o_rs,
                                             float(scale), _stream()), 'mxl_relattn_fwd_phantom2')
        return out
    if oph is not None:
        check(lib().mxl_relattn_fwd_phantom(_p(q), _p(k), _p(v), _p(rd), _p(r_w_bias), _p(r_r_bias), _p(out), _p(lse), _p(oph),
                                            _p(mph), B, T, H, dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs, o_rs, float(scale),
                                            _stream()), 'mxl_relattn_fwd_phantom')
        return out
    check(lib().mxl_relattn_fwd(_p(q), _p(k), _p(v), _p(rd), _p(r_w_bias), _p(r_r_bias), _p(out), _p(lse), B, T, H, dh,
                                M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs, o_rs, float(scale), _stream()),
          'mxl_relattn_fwd')
    return out


def label_guard(labels: torch.Tensor, eos: int):
    _req(labels, torch.int64, 'labels')
    check(lib().mxl_label_guard(_p(labels), labels.shape[1], int(eos), _stream()), 'mxl_label_guard')


def adaptive_nll_fwd(logits, labels, nll, lse, acc2, B, T, V, cutoffs=()):
    n, arr = _cut(cutoffs)
    check(lib().mxl_adaptive_nll_fwd(_p(logits), logits.stride(0), _p(labels), _p(nll), _p(lse), _p(acc2), B, T, V, n,
                                     C.cast(arr, C.c_void_p) if arr is not None else None, _stream()),
          'mxl_adaptive_nll_fwd')


def adaptive_nll_bwd(logits, labels, nll, lse, acc2, dlogits, B, T, V, cutoffs=(), grad_scale=1.0, dlogits_lo=None):
    n, arr = _cut(cutoffs)
    if dlogits_lo is not None:          # two-term form: dlogits + dlogits_lo carries the gradient to 2^-16
        assert dlogits_lo.shape == dlogits.shape and dlogits_lo.stride(0) == dlogits.stride(0)
        check(lib().mxl_adaptive_nll_bwd_split(_p(logits), logits.stride(0), _p(labels), _p(nll), _p(lse), _p(acc2), _p(dlogits),
                                               _p(dlogits_lo), dlogits.stride(0), B, T, V, n,
                                               C.cast(arr, C.c_void_p) if arr is not None else None, float(grad_scale),
                                               _stream()), 'mxl_adaptive_nll_bwd_split')
        return
    check(lib().mxl_adaptive_nll_bwd(_p(logits), logits.stride(0), _p(labels), _p(nll), _p(lse), _p(acc2), _p(dlogits),
                                     dlogits.stride(0), B, T, V, n,
                                     C.cast(arr, C.c_void_p) if arr is not None else None, float(grad_scale), _stream()),
          'mxl_adaptive_nll_bwd')


def adaptive_logprob(logits, out, N, V, cutoffs=()):
    n, arr = _cut(cutoffs)
    check(lib().mxl_adaptive_logprob(_p(logits), logits.stride(0), _p(out), out.stride(0), N, V, n,
                                     C.cast(arr, C.c_void_p) if arr is not None else None, _stream()),
          'mxl_adaptive_logprob')
    return out


# ---- large-vocabulary (bucketed) adaptive softmax: see csrc/head_large.hip
def cluster_bucket(labels, V, cutoffs, perm, counts, tgt_head, tgt_tail):
    B, T = labels.shape
    n, arr = _cut(cutoffs)
    check(lib().mxl_cluster_bucket(_p(labels), B, T, V, n, C.cast(arr, C.c_void_p), _p(perm), _p(counts), _p(tgt_head),
                                   _p(tgt_tail), _stream()), 'mxl_cluster_bucket')


def gather_rows(src, idx, dst, n):
    check(lib().mxl_gather_rows_bf16(_p(src), src.stride(-2), _p(idx), _p(dst), n, src.shape[-1], _stream()), 'mxl_gather_rows_bf16')
    return dst


def scatter_add_rows(src, idx, dst, n):
    check(lib().mxl_scatter_add_rows_bf16(_p(src), _p(idx), _p(dst), dst.stride(-2), n, src.shape[-1], _stream()),
          'mxl_scatter_add_rows_bf16')


def rows_lse_pick(logits, ncols, n_rows, tgt, lse_out, pick_out, rows_idx=None, row0=0):
    check(lib().mxl_rows_lse_pick(_p(logits), logits.stride(0), ncols, n_rows, _p(rows_idx), row0, _p(tgt), _p(lse_out),
                                  _p(pick_out), _stream()), 'mxl_rows_lse_pick')


def bucket_nll_finish(hlse, hpick, tlse, tpick, tgt_head, tgt_tail, nll, nll_tok, acc2, B, T):
    check(lib().mxl_bucket_nll_finish(_p(hlse), _p(hpick), _p(tlse), _p(tpick), _p(tgt_head), _p(tgt_tail), _p(nll), _p(nll_tok),
                                      _p(acc2), B, T, _stream()), 'mxl_bucket_nll_finish')


def rows_softmax_grad(logits, ncols, n_rows, tgt, lse, nll_tok, acc2, grad_scale, out_hi, out_lo=None, rows_idx=None, row0=0):
    check(lib().mxl_rows_softmax_grad(_p(logits), logits.stride(0), ncols, n_rows, _p(rows_idx), row0, _p(tgt), _p(lse), _p(nll_tok),
                                      _p(acc2), float(grad_scale), _p(out_hi), _p(out_lo), out_hi.stride(0), _stream()),
          'mxl_rows_softmax_grad')


def sumsq(x: torch.Tensor, out_accum: torch.Tensor):
    check(lib().mxl_sumsq_f32(_p(x), x.numel(), _p(out_accum), _stream()), 'mxl_sumsq_f32')


def adamw_step(p, g, m, v, w16, n_decay, lr, beta1, beta2, eps, weight_decay, step, sumsq_buf=None, max_norm=0.0,
               grad_scale=1.0):
    check(lib().mxl_adamw_step(_p(p), _p(g), _p(m), _p(v), _p(w16), p.numel(), n_decay, float(lr), float(beta1),
                               float(beta2), float(eps), float(weight_decay), int(step), _p(sumsq_buf), float(max_norm),
                               float(grad_scale), _stream()), 'mxl_adamw_step')


def cast_bf16(x: torch.Tensor, y: torch.Tensor):
    check(lib().mxl_cast_f32_bf16(_p(x), _p(y), x.numel(), _stream()), 'mxl_cast_f32_bf16')
    return y


def cast_f32(x: torch.Tensor, y: torch.Tensor):
    """y (f32) = x (bf16)"""
    _req(x, torch.bfloat16, 'x'); _req(y, torch.float32, 'y')
    check(lib().mxl_cast_bf16_f32(_p(x), _p(y), x.numel(), _stream()), 'mxl_cast_bf16_f32')
    return y


def center_columns(x: torch.Tensor, y: torch.Tensor):
    """y = x - mean over rows (bf16, (M, N))"""
    M, N = x.shape
    check(lib().mxl_center_columns_bf16(_p(x), _p(y), M, N, _stream()), 'mxl_center_columns_bf16')
    return y


def transpose(src: torch.Tensor, dst: torch.Tensor, rows: int, cols: int, *, ld_src=None, ld_dst=None, batch=1,
              src_bstride=0, dst_bstride=0):
    """dst[b][c][r] = src[b][r][c] (bf16)"""
    _req(src, torch.bfloat16, 'src'); _req(dst, torch.bfloat16, 'dst')
    check(lib().mxl_transpose_bf16(_p(src), _p(dst), rows, cols, ld_src if ld_src is not None else cols,
                                   ld_dst if ld_dst is not None else rows, batch, src_bstride, dst_bstride, _stream()),
          'mxl_transpose_bf16')
    return dst


def gemm_batched(a, b, c, M, N, K, *, lda, ldb, ldc, trans_a=False, trans_b=False, flags=0, alpha=1.0, ksplits=1,
                 batch=1, bdiv=1, sA=(0, 0), sB=(0, 0), sC=(0, 0)):
    check(lib().mxl_gemm_bf16_batched(_p(a), _p(b), _p(c), M, N, K, lda, ldb, ldc, int(trans_a), int(trans_b), flags,
                                      float(alpha), ksplits, batch, bdiv, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1],
                                      _stream()), 'mxl_gemm_bf16_batched')
    return c


def add_rowbias(x, x_bs, x_rs, bias, out, B, T, n):
    check(lib().mxl_add_rowbias_bf16(_p(x), x_bs, x_rs, _p(bias), _p(out), B, T, n, _stream()), 'mxl_add_rowbias_bf16')
    return out


def relattn_bwd(q, k, v, rd, r_w_bias, r_r_bias, out, dout, lse, delta, dq, dk, dv, dg, d_rwb, d_rrb, *, B, T, H, dh, M,
                Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs, o_rs, dq_bs, dq_rs, dkv_bs, dkv_rs, scale=None,
                d_rd: Optional[torch.Tensor] = None, qr_buf: Optional[torch.Tensor] = None, defer_drd: bool = False,
                oph: Optional[torch.Tensor] = None, mph: Optional[torch.Tensor] = None):
    """Backward of relattn_fwd.  If `d_rd` (M, H*dh) f32 is given, also contracts dG with (q + r_r_bias):
    d_rd[d, h, :] += sum_{b,i} dG[b,h,i,d] * (q + r_r_bias)[b,i,h,:]   (needs dg and a (B,T,H*dh) bf16 qr_buf).
    When that contraction runs as the streaming kernel (dh = 64, T % 32 == 0, M % 8 == 0) it also produces d_rrb, and the
    backward proper is launched without it -- its query-owner kernel then takes the faster 8-wave form.

    `dg` may hold FEWER sequences than B: the batch is then walked in chunks of dg.shape[0] sequences, each chunk's attention
    backward followed at once by its dRd contraction, so the un-skewed score gradient of a chunk (B_c * H * T * M * 2 bytes) is
    produced and consumed while it is still in the 256 MB Infinity Cache instead of making a round trip through HBM (per-sequence
    outputs are unaffected; the batch-summed ones are atomically accumulated in any case)."""
    scale = scale if scale is not None else 1.0 / math.sqrt(dh)
    fused_rrb = (d_rd is not None and dg is not None and dh == 64 and T % 32 == 0 and M % 8 == 0 and M >= 8
                 and os.environ.get('MXL_NO_DQ8') != '1')
    Bc = B if dg is None else min(B, dg.shape[0])
    # phantom distances (zero memories: k = v = 0) stay out of HBM: the backward leaves their dG blocks unwritten and the dRd
    # contraction rebuilds them from qr, rd, lse and delta (mxl_relattn_bwd_sparse_dg / mxl_relattn_drd_recompute)
    sparse = fused_rrb and M % 256 == 0 and Kc < M + T and os.environ.get('MXL_DG_RECOMPUTE', '1') != '0'
    assert oph is None or sparse, 'oph / mph are for the shapes phantom_sum_applies() accepts, with d_rd / dg given'

    def launch(b0, n):
        sl = slice(b0, b0 + n)
        if sparse and oph is not None:          # the forward's phantom value-sum: the all-phantom blocks are not walked at all
            check(lib().mxl_relattn_bwd_sparse_dg_oph(_p(q[sl]), _p(k[sl]), _p(v[sl]), _p(rd), _p(r_w_bias), _p(r_r_bias),
                                                      _p(out[sl]), _p(dout[sl]), _p(lse[sl]), _p(delta[sl]), _p(dq[sl]), _p(dk[sl]),
                                                      _p(dv[sl]), _p(dg), _p(d_rwb), _p(oph[sl]), _p(mph[sl]), n, T, H, dh, M, Kc,
                                                      q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs, o_rs, dq_bs, dq_rs, dkv_bs, dkv_rs,
                                                      float(scale), _stream()), 'mxl_relattn_bwd_sparse_dg_oph')
            return
        if sparse:
            check(lib().mxl_relattn_bwd_sparse_dg(_p(q[sl]), _p(k[sl]), _p(v[sl]), _p(rd), _p(r_w_bias), _p(r_r_bias), _p(out[sl]),
                                                  _p(dout[sl]), _p(lse[sl]), _p(delta[sl]), _p(dq[sl]), _p(dk[sl]), _p(dv[sl]),
                                                  _p(dg), _p(d_rwb), n, T, H, dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs,
                                                  o_rs, dq_bs, dq_rs, dkv_bs, dkv_rs, float(scale), _stream()),
                  'mxl_relattn_bwd_sparse_dg')
            return
        check(lib().mxl_relattn_bwd(_p(q[sl]), _p(k[sl]), _p(v[sl]), _p(rd), _p(r_w_bias), _p(r_r_bias), _p(out[sl]), _p(dout[sl]),
                                    _p(lse[sl]), _p(delta[sl]), _p(dq[sl]), _p(dk[sl]), _p(dv[sl]), _p(dg), _p(d_rwb),
                                    None if fused_rrb else _p(d_rrb), n, T, H, dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs,
                                    o_rs, dq_bs, dq_rs, dkv_bs, dkv_rs, float(scale), _stream()), 'mxl_relattn_bwd')

    def drd(b0, n):
        if d_rd is not None:
            relattn_drd(q[b0:b0 + n], r_r_bias, dg, d_rd, qr_buf, B=n, T=T, H=H, dh=dh, M=M, q_bs=q_bs, q_rs=q_rs,
                        rd=rd if fused_rrb else None, rd_rs=rd_rs, d_rrb=d_rrb if fused_rrb else None,
                        d_rwb=d_rwb if fused_rrb else None,
                        recompute=dict(lse=lse[b0:b0 + n], delta=delta[b0:b0 + n], scale=scale, Kc=Kc) if sparse else None)

    if Bc >= B:
        launch(0, B)
        if defer_drd:          # (bench.py brackets the attention-backward launches alone)
            return lambda: drd(0, B)
        drd(0, B)
        return None
    for b0 in range(0, B, Bc):
        n = min(Bc, B - b0)
        launch(b0, n)
        drd(b0, n)
    return (lambda: None) if defer_drd else None


def relattn_bwd_fused_ws_numel(B, T, H, dh, M) -> int:
    """fp32 elements of the partial-dq slabs mxl_relattn_bwd_fused needs"""
    return int(lib().mxl_relattn_bwd_fused_ws_bytes(B, T, H, dh, M)) // 4


_PH_WS = {}


def _phantom_ws(B, T, H, device):
    """scratch of mxl_relattn_drd_phantom_prep (one record per sequence, head and 32-query tile), one per shape and device, allocated on
    first use; callers that capture graphs or account for every byte pass their own `ph_buf`"""
    key = (B, T, H, str(device))
    buf = _PH_WS.get(key)
    if buf is None:
        buf = torch.empty(int(lib().mxl_relattn_drd_phantom_ws_bytes(B, T, H)), device=device, dtype=torch.uint8)
        _PH_WS[key] = buf
    return buf


def gemm_headdot(a, b, c, M, N, K, o, T, delta, lda=None, ldb=None, ldc=None, ldo=None):
    """c = a @ b^T (bf16) and delta[(b_, h, t)] = sum_e c[m, 64 h + e] * o[m, 64 h + e] in the same launch (mxl_gemm_bf16_headdot).
    Returns True when delta was written; False when the problem is not the four-wave large-tile kernel's (c is written either way)."""
    rc = lib().mxl_gemm_bf16_headdot(_p(a), _p(b), _p(c), M, N, K, lda or K, ldb or K, ldc or N, _p(o), ldo or N, T, _p(delta), _stream())
    if rc == -2:      # MXL_EUNSUPPORTED: not the four-wave kernel's problem (c is written, delta is not)
        return False
    check(rc, 'mxl_gemm_bf16_headdot')
    return True


def relattn_bwd_fused(q, k, v, rd, r_w_bias, r_r_bias, out, dout, lse, delta, dq, dk, dv, d_rd, d_rwb, d_rrb, ws, qr_buf, *,
                      B, T, H, dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs, o_rs, dq_bs, dq_rs, dkv_bs, dkv_rs,
                      scale=None, oph=None, mph=None, defer_drd=False, ph_buf=None, ph_ready=False, delta_ready=False):
    """Backward of relattn_fwd in one pass over the score cells (mxl_relattn_bwd_fused): dq, dk, dv written, d_rd (M, H*dh) f32 /
    d_rwb / d_rrb accumulated.  With zero memories (Kc < M + T) `oph` / `mph` must come from relattn_fwd(..., oph_all=True), and
    the phantom cells' part of d_rd is added by mxl_relattn_drd_phantom from per-tile records (`ph_buf`,
    mxl_relattn_drd_phantom_ws_bytes bytes; ph_ready: relattn_fwd(..., ph_buf=) has filled them, otherwise
    mxl_relattn_drd_phantom_prep does; without `ph_buf` one scratch per shape is cached; `qr_buf` is no longer used by this path);
    their part of d_rrb comes out of the dq finishing kernel.  delta_ready: `delta` was filled by gemm_headdot."""
    scale = scale if scale is not None else 1.0 / math.sqrt(dh)
    d = H * dh
    # (the slab sum on a side stream beside the phantom cells' dRd kernel measured 7.16 ms against 7.12 ms in sequence -- the HBM-bound
    # sum fills the chip by itself -- and left dq / d_rrb incomplete until the deferred call: removed in round 5)
    check(lib().mxl_relattn_bwd_fused(_p(q), _p(k), _p(v), _p(rd), _p(r_w_bias), _p(r_r_bias), _p(out), _p(dout), _p(lse), _p(delta),
                                      _p(dq), _p(dk), _p(dv), _p(d_rd), d_rd.stride(0), _p(d_rwb), _p(d_rrb), _p(oph), _p(mph), _p(ws),
                                      B, T, H, dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs, o_rs, dq_bs, dq_rs, dkv_bs, dkv_rs,
                                      float(scale), 2 if delta_ready else 0, _stream()), 'mxl_relattn_bwd_fused')

    def phantom():
        if Kc < M + T:
            ph = ph_buf if ph_buf is not None else _phantom_ws(B, T, H, q.device)
            if not (ph_ready and ph_buf is not None):
                check(lib().mxl_relattn_drd_phantom_prep(_p(q), q_bs, q_rs, _p(r_r_bias), _p(lse), _p(ph), B, T, H, dh, float(scale),
                                                         _stream()), 'mxl_relattn_drd_phantom_prep')
            check(lib().mxl_relattn_drd_phantom(_p(ph), _p(delta), _p(d_rd), B, T, H, dh, M, d_rd.stride(0), _p(rd), int(rd_rs), Kc,
                                                _stream()), 'mxl_relattn_drd_phantom')
    if defer_drd:
        return phantom
    phantom()
    return None


def relattn_drd(q, r_r_bias, dg, d_rd, qr_buf, *, B, T, H, dh, M, q_bs, q_rs, rd=None, rd_rs=0, d_rrb=None, d_rwb=None,
                recompute=None):
    """d_rd[delta, h*dh:(h+1)*dh] (M x dh, fp32, +=) = sum_b dG[b,h]^T (M x T) @ (q + r_r_bias)[b,:,h,:] (T x dh); with rd /
    d_rrb (/ d_rwb) also d_rrb += colsum(dG) . Rd (and d_rwb -= the same): see mxl_relattn_drd"""
    d = H * dh
    add_rowbias(q, q_bs, q_rs, r_r_bias.reshape(-1), qr_buf, B, T, d)
    if recompute is not None:        # dg came from mxl_relattn_bwd_sparse_dg: its all-phantom blocks are rebuilt, not read
        check(lib().mxl_relattn_drd_recompute(_p(dg), _p(qr_buf), _p(d_rd), B, T, H, dh, M, T * d, d, d, _p(rd), int(rd_rs),
                                              _p(d_rrb), _p(d_rwb), _p(recompute['lse']), _p(recompute['delta']),
                                              float(recompute['scale']), int(recompute['Kc']), _stream()),
              'mxl_relattn_drd_recompute')
        return
    rc = lib().mxl_relattn_drd(_p(dg), _p(qr_buf), _p(d_rd), B, T, H, dh, M, T * d, d, d, _p(rd), int(rd_rs), _p(d_rrb),
                               _p(d_rwb), _stream())
    if rc == -2:      # MXL_EUNSUPPORTED shape: the batched GEMM form
        assert d_rrb is None
        gemm_batched(dg, qr_buf, d_rd, M, dh, T, lda=M, ldb=d, ldc=d, trans_a=True, trans_b=True,
                     flags=GEMM_OUT_F32_ATOMIC, batch=B * H, bdiv=H, sA=(H * T * M, T * M), sB=(T * d, dh), sC=(0, dh))
    else:
        check(rc, 'mxl_relattn_drd')


# ------------------------------------------------------------------ decode
def decode_embed(ids, t_dev, E, out, scale):
    B, d = out.shape
    check(lib().mxl_decode_embed(_p(ids), ids.stride(0), _p(t_dev), _p(E), _p(out), B, d, E.shape[0], float(scale),
                                 _stream()), 'mxl_decode_embed')


def kv_append(qkv, kc, vc, t_dev, rrb=None, qr_out=None):
    """kc / vc: head-major rings (B, H, M, dh).  With `rrb` (H*dh f32) and `qr_out` (B, H*dh) bf16 the same launch also writes
    q + r_r_bias, the operand of the step's BD product (then pass qr_ready=True to relattn_decode)."""
    B, H, M, dh = kc.shape
    check(lib().mxl_kv_append(_p(qkv), _p(kc), _p(vc), _p(t_dev), B, M, H * dh, dh, _p(rrb), _p(qr_out), _stream()),
          'mxl_kv_append')


def kv_fill(qkv, kc, vc, T):
    B, H, M, dh = kc.shape
    check(lib().mxl_kv_fill(_p(qkv), _p(kc), _p(vc), B, T, M, H * dh, dh, _stream()), 'mxl_kv_fill')


def decode_ring_pieces(B: int, H: int, M: int) -> int:
    """workgroups per (sequence, head) ring of a decode step (mxl_relattn_decode_split).  With fewer rings than CUs a ring per
    workgroup leaves CUs idle and every workgroup streaming 512 KB alone: two pieces up to 256 rings, four up to 96 (measured on
    the C5 shape, scripts/perf_decode_attn.py: B = 4: 27.3 -> 14.4 us, B = 8: 28.4 -> 17.7, B = 16: 29.7 -> 25.0, B = 21: 31.8 -> 27.3;
    from 384 rings on the pieces cost more than they balance: B = 32: 39.4 -> 42.7).  One piece for short rings.
    MXL_DECODE_PIECES overrides."""
    env = os.environ.get('MXL_DECODE_PIECES')
    if env:
        return max(1, min(8, int(env)))
    if M < 1024 or B * H > 256:
        return 1
    return 4 if B * H <= 96 else 2


def relattn_decode_split_scratch(B: int, H: int, dh: int, pieces: int, dev):
    """(ws, arrived) of mxl_relattn_decode_split, or None for one piece"""
    if pieces <= 1:
        return None
    n = int(lib().mxl_relattn_decode_split_ws_bytes(B, H, dh, pieces)) // 4
    return torch.empty(n, device=dev, dtype=torch.float32), torch.zeros(B * H, device=dev, dtype=torch.int32)


def relattn_decode(qkv, kc, vc, rd, rwb, rrb, out, t_dev, H, dh, qr_buf, bd_buf, scale=None, qr_ready=False, split=None, pieces=1):
    """qr_buf (B, H*dh) bf16 and bd_buf (B, H, M) f32 are scratch: BD = (q + r_r_bias) . rd^T for the whole batch.
    qr_ready: qr_buf already holds q + r_r_bias (written by kv_append).  split = relattn_decode_split_scratch(...) with the same
    `pieces`: the ring of every (sequence, head) goes to that many workgroups."""
    B, _, M, _ = kc.shape
    d = H * dh
    scale = scale if scale is not None else 1.0 / math.sqrt(dh)
    if not qr_ready:
        add_rowbias(qkv, 3 * d, 3 * d, rrb.reshape(-1), qr_buf, B, 1, d)
    if dh == 64 and B <= 64:
        check(lib().mxl_decode_bd(_p(qr_buf), _p(rd), _p(bd_buf), B, H, dh, M, d, rd.stride(0), _stream()), 'mxl_decode_bd')
    else:
        gemm_batched(qr_buf, rd, bd_buf, B, M, dh, lda=d, ldb=d, ldc=H * M, flags=GEMM_OUT_F32, batch=H, bdiv=1,
                     sA=(dh, 0), sB=(dh, 0), sC=(M, 0))
    if split is not None and pieces > 1:
        check(lib().mxl_relattn_decode_split(_p(qkv), _p(kc), _p(vc), _p(bd_buf), _p(rwb), _p(out), _p(t_dev), B, H, dh, M,
                                             C.c_float(float(scale)), pieces, _p(split[0]), _p(split[1]), _stream()),
              'mxl_relattn_decode_split')
        return
    check(lib().mxl_relattn_decode(_p(qkv), _p(kc), _p(vc), _p(bd_buf), _p(rwb), _p(out), _p(t_dev), B, H, dh, M,
                                   float(scale), _stream()), 'mxl_relattn_decode')


_sample_scratch = {}


def sample(logprobs, ids, t_dev, rng_ctr, seed, do_sample=False, top_k=0, top_p=1.0, temperature=1.0,
           repetition_penalty=1.0, typical_p=1.0, out_probs=None):
    B, V = logprobs.shape
    if V > 2048 or os.environ.get('MXL_SAMPLE_LARGE') == '1':      # beyond the LDS sort: the bisection sampler (sample_large.hip)
        key = (logprobs.device, B, V)
        scratch = _sample_scratch.get(key)
        if scratch is None:
            _sample_scratch.clear()
            scratch = _sample_scratch[key] = torch.empty(B, 2 * V, device=logprobs.device, dtype=torch.float32)
        check(lib().mxl_sample_large(_p(logprobs), logprobs.stride(0), V, _p(ids), ids.stride(0), _p(t_dev), _p(rng_ctr), seed, B,
                                     int(do_sample), int(top_k or 0), float(top_p if top_p is not None else 1.0),
                                     float(temperature), float(repetition_penalty if repetition_penalty is not None else 1.0),
                                     float(typical_p if typical_p is not None else 1.0), _p(out_probs), _p(scratch), _stream()),
              'mxl_sample_large')
        return
    check(lib().mxl_sample(_p(logprobs), logprobs.stride(0), V, _p(ids), ids.stride(0), _p(t_dev), _p(rng_ctr), seed, B,
                           int(do_sample), int(top_k or 0), float(top_p if top_p is not None else 1.0),
                           float(temperature), float(repetition_penalty if repetition_penalty is not None else 1.0),
                           float(typical_p if typical_p is not None else 1.0), _p(out_probs), _stream()), 'mxl_sample')


def find_token(ids: torch.Tensor, token: int, which: int = -1) -> torch.Tensor:
    """(B,) int32: index of the last (which < 0) / which-th occurrence of `token` per row of ids (B, T) int64, -1 if none"""
    B, T = ids.shape
    out = torch.empty(B, device=ids.device, dtype=torch.int32)
    check(lib().mxl_find_token(_p(ids), ids.stride(0), B, T, int(token), int(which), _p(out), _stream()), 'mxl_find_token')
    return out


def decode_advance(t_dev, rng_ctr):
    check(lib().mxl_decode_advance(_p(t_dev), _p(rng_ctr), _stream()), 'mxl_decode_advance')


def row_inv_norm(x, out, n):
    """out[j] = 1 / ||x[j]||  (bf16 rows)"""
    check(lib().mxl_row_inv_norm_bf16(_p(x), x.stride(-2), n, x.shape[-1], _p(out), _stream()), 'mxl_row_inv_norm_bf16')
    return out


def contrastive_select(ctx, ctx_inv, S, hid, probs, alpha, score, sel):
    """see mxl_contrastive_select: ctx (B, Smax, d) bf16, ctx_inv (B, Smax) f32, hid (B*K, d) bf16, probs (B, K) f32"""
    B, K = probs.shape
    check(lib().mxl_contrastive_select(_p(ctx), ctx.stride(0), _p(ctx_inv), ctx_inv.stride(0), S, _p(hid), _p(probs), float(alpha),
                                       B, K, ctx.shape[-1], _p(score), _p(sel), _stream()), 'mxl_contrastive_select')
    return sel


# ------------------------------------------------------------------ reformer
def axial_embed_fwd(ids, E, W0, W1, out, A0, A1, drop_p=0.0, seed=0, site_emb=0, site_pos=1):
    B, T = ids.shape
    d, d0 = E.shape[1], W0.shape[-1]
    check(lib().mxl_axial_embed_fwd(_p(ids), _p(E), _p(W0), _p(W1), _p(out), B, T, d, E.shape[0], A0, A1, d0, float(drop_p),
                                    seed, site_emb, site_pos, _stream()), 'mxl_axial_embed_fwd')
    return out


def axial_embed_bwd(ids, dout, dE, dW0, dW1, A0, A1, drop_p=0.0, seed=0, site_emb=0, site_pos=1, dout2=None):
    B, T = ids.shape
    d, d0 = dE.shape[1], dW0.shape[-1]
    check(lib().mxl_axial_embed_bwd(_p(ids), _p(dout), _p(dout2), _p(dE), _p(dW0), _p(dW1), B, T, d, dE.shape[0], A0, A1, d0,
                                    float(drop_p), seed, site_emb, site_pos, _stream()), 'mxl_axial_embed_bwd')


def lsh_hash(qk, bs, rs, rotations, buckets, B, T, H, dh, n_h, factors):
    arr = (C.c_int * len(factors))(*factors)
    check(lib().mxl_lsh_hash(_p(qk), bs, rs, _p(rotations), _p(buckets), B, T, H, dh, n_h, len(factors),
                             C.cast(arr, C.c_void_p), _stream()), 'mxl_lsh_hash')
    return buckets


def lsh_sort(buckets, sorted_idx, sorted_pos, BH, S, T, n_buckets_total):
    check(lib().mxl_lsh_sort(_p(buckets), _p(sorted_idx), _p(sorted_pos), BH, S, T, n_buckets_total, _stream()), 'mxl_lsh_sort')


def chunk_attn_fwd(q, k, v, spos, out, lse, B, T, H, dh, n_h, lsh, bs, rs, drop_p=0.0, seed=0, site=0):
    check(lib().mxl_chunk_attn_fwd(_p(q), _p(k), _p(v), _p(spos), _p(out), _p(lse), B, T, H, dh, n_h, int(lsh), bs, rs,
                                   float(drop_p), seed, site, _stream()), 'mxl_chunk_attn_fwd')


def chunk_attn_bwd(q, k, v, spos, out, lse, dout, dlse, dq, dk, dv, B, T, H, dh, n_h, lsh, bs, rs, drop_p=0.0, seed=0, site=0,
                   dq16=None, dk16=None, dv16=None, ld16=0):
    """dq / dk / dv: f32 (B, n_h, T, d), one slab per hash round, every element written once (may be None when the matching bf16
    destination is given; n_h == 1 only)"""
    check(lib().mxl_chunk_attn_bwd(_p(q), _p(k), _p(v), _p(spos), _p(out), _p(lse), _p(dout), _p(dlse), _p(dq), _p(dk), _p(dv),
                                   _p(dq16), _p(dk16), _p(dv16), int(ld16),
                                   B, T, H, dh, n_h, int(lsh), bs, rs, float(drop_p), seed, site, _stream()),
          'mxl_chunk_attn_bwd')


def lsh_keynorm_bwd(qk, bs, rs, dq, dk_eff, dqk, B, T, H, dh, ld_dqk=None):
    check(lib().mxl_lsh_keynorm_bwd(_p(qk), bs, rs, _p(dq), _p(dk_eff), _p(dqk), int(ld_dqk or H * dh), B, T, H, dh, _stream()),
          'mxl_lsh_keynorm_bwd')


def lsh_keynorm_bwd_rounds(qk, bs, rs, dq, dk_eff, dv, dqk, dv16, B, T, H, dh, n_h, ld_dqk=None, ld_dv=None):
    """lsh_keynorm_bwd over per-round slabs (B, n_h, T, d) f32; dv (optional) summed over the rounds into dv16 (bf16)"""
    check(lib().mxl_lsh_keynorm_bwd_rounds(_p(qk), bs, rs, _p(dq), _p(dk_eff), _p(dv), _p(dqk), int(ld_dqk or H * dh), _p(dv16),
                                           int(ld_dv or H * dh), B, T, H, dh, n_h, _stream()), 'mxl_lsh_keynorm_bwd_rounds')


def lsh_combine(out_r, lse, out, B, T, H, dh, n_h):
    check(lib().mxl_lsh_combine(_p(out_r), _p(lse), _p(out), B, T, H, dh, n_h, _stream()), 'mxl_lsh_combine')


def lsh_combine_bwd(out_r, lse, out, dout, dout_r, dlse, B, T, H, dh, n_h):
    check(lib().mxl_lsh_combine_bwd(_p(out_r), _p(lse), _p(out), _p(dout), _p(dout_r), _p(dlse), B, T, H, dh, n_h, _stream()),
          'mxl_lsh_combine_bwd')


# ------------------------------------------------------------------ Reformer cached decoding
def rf_decode_embed(ids, t, E, W0, W1, out, A1):
    B, d = out.shape
    check(lib().mxl_rf_decode_embed(_p(ids), ids.stride(0), t, _p(E), _p(W0), _p(W1), _p(out), B, d, E.shape[0], A1, W0.shape[-1],
                                    _stream()), 'mxl_rf_decode_embed')
    return out


def lsh_fix_buckets(buckets, rows, n_h, T, T_real, NB):
    check(lib().mxl_lsh_fix_buckets(_p(buckets), rows, n_h, T, T_real, NB, _stream()), 'mxl_lsh_fix_buckets')


def rf_query_bucket(raw, cache, bkmax, rows, n_h, NB, Tmax, t):
    check(lib().mxl_rf_query_bucket(_p(raw), _p(cache), _p(bkmax), rows, n_h, NB, Tmax, t, _stream()), 'mxl_rf_query_bucket')


def rf_decode_attn(q, kcache, vcache, sorted_idx, out, B, H, dh, n_h, Tmax, n, t, start=0, count=0, lsh=False):
    check(lib().mxl_rf_decode_attn(_p(q), q.stride(0), _p(kcache), _p(vcache), _p(sorted_idx), _p(out), B, H, dh, n_h, Tmax, n, t,
                                   start, count, int(lsh), _stream()), 'mxl_rf_decode_attn')
    return out
