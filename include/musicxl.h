/*
 * libmusicxl -- C ABI of the MI355X-native Transformer-XL / Reformer hot path.
 *
 * The reference (StefanHeng/Symbolic-Music-Generation) has no FFI for this path: its boundary is the Python object
 * contract of `get_model_n_tokenizer` (musicnlp/trainer/train.py:31-59) and the arithmetic sits in HuggingFace
 * `transformers==4.25.1` (its transfo_xl and reformer model directories).  Each entry point below names the upstream /
 * reference operation it replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless said otherwise; tensors are row-major, batch-major (B, T, ...);
 *   - bf16 tensors are raw 16-bit words; "f32" means IEEE float;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work (no sync, no allocation) and is
 *     therefore safe to capture into a hipGraph;
 *   - return value: 0 ok, negative = argument error (MXL_E*), positive = hipError_t from the launch.
 */
#ifndef MUSICXL_H
#define MUSICXL_H

#ifdef __cplusplus
extern "C" {
#endif

#define MXL_ABI_VERSION 1

int mxl_abi_version(void);
/* human-readable text for a negative libmusicxl code or a positive hipError_t */
const char* mxl_error_string(int code);

/* ------------------------------------------------------------------------------------------------------------
 * GEMM (bf16 in, fp32 accumulate on MFMA).  Replaces F.linear / torch.einsum in upstream qkv_net, r_net, o_net,
 * CoreNet (PositionwiseFF), crit.out_layers (modeling_transfo_xl.py), and their autograd backward.
 *   C[M,N] (+)= alpha * op(A) * op(B);  transA=0: A[M][K]  transA=1: A[K][M];  transB=0: B[N][K]  transB=1: B[K][N]
 *   lda/ldb multiples of 8 elements, base pointers 16-byte aligned, K % 8 == 0 unless both operands are transposed.
 * flags: output type (default bf16) and fused epilogue.
 * ---------------------------------------------------------------------------------------------------------- */
#define MXL_GEMM_OUT_F32        0x01  /* C is f32, plain store                                   */
#define MXL_GEMM_OUT_F32_ATOMIC 0x02  /* C is f32, atomicAdd (required when ksplits > 1)         */
#define MXL_GEMM_BIAS           0x04  /* + bias[n]                                               */
#define MXL_GEMM_RELU           0x08  /* max(.,0)            (CoreNet.1)                         */
#define MXL_GEMM_DROPOUT        0x10  /* inverted dropout.  Alone or with BIAS: the (seed, site, m*N+n) keep-mask of mxl_dropout_bf16
                                         (dropout_keep: a backward pass may regenerate it).  Together with MXL_GEMM_RELU (the FFN's
                                         hidden activations) a DIFFERENT mask: one hash per pair of elements (m, n), (m, n + 1),
                                         n even, a 16-bit decision each, the drop probability quantised to 2^-16.  Nothing may
                                         regenerate that one: the backward takes it from the saved bits (MXL_GEMM_SAVE_RELU_MASK)
                                         or from the zeros of the stored activations */
#define MXL_GEMM_RELU_BWD       0x20  /* C = aux[m][n] > 0 ? acc : 0  (backward through relu+dropout) */
#define MXL_GEMM_ADD_AUX        0x40  /* C = epilogue(acc) + aux[m][n]  (residual add after bias/dropout)  */
/* Compute units to leave free of the persistent GEMM grids (0 .. 128; default 0): they launch one workgroup per remaining CU.  For
 * data-parallel training: RCCL's reduction kernels run beside the backward on another stream, and a grid that occupies every CU for its
 * whole duration leaves them nowhere to start until it ends (symbolic_music_generation_amd/dist.py sets it from MXL_RESERVE_CUS when
 * the process group has more than one rank).  Process-wide, not stream-ordered: set it between steps. */
int mxl_set_reserved_cus(int k);
/* which large-tile kernel the last K-contiguous mxl_gemm_bf16* call went to (0: none, 1 / 2: eight waves with 256- / 192-wide tiles,
 * 3: four waves of 128 x 128): for tests that mean to exercise one of them */
int mxl_gemm_last_nt_kernel(void);

/* The relu (+dropout) mask of C as bits instead of the bf16 activations, for the backward through CoreNet.1 / CoreNet.2:
 *   MXL_GEMM_SAVE_RELU_MASK  (with BIAS | RELU [| DROPOUT]): also writes, through `aux`, one bit per output element (> 0)
 *   MXL_GEMM_RELU_BWD_BITS   C = bit ? alpha * acc : 0, bits read through `aux` (the buffer a SAVE call of the same M, N filled)
 * The bits are in the large-tile kernel's accumulator layout (opaque; mxl_gemm_relu_mask_bytes(M, N) bytes, 0 = these M, N do not
 * take that kernel: use MXL_GEMM_RELU_BWD with the activations).  Calls that cannot honour the flags return MXL_EUNSUPPORTED. */
#define MXL_GEMM_SAVE_RELU_MASK 0x100
#define MXL_GEMM_RELU_BWD_BITS  0x200
size_t mxl_gemm_relu_mask_bytes(int M, int N);
int mxl_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                  int transA, int transB, int flags, float alpha, const float* bias,
                  const void* aux, int ldaux, int ksplits,
                  float drop_p, unsigned long long seed, unsigned site, void* stream);

/* mxl_gemm_bf16 (bf16 output, no K-split) and, in the same call, colsum[n] += sum_m C[m][n]: the bias gradient of the layer whose
 * output gradient C is (pos_ff.CoreNet.0.bias: C = the masked dX of CoreNet.3).  With MXL_GEMM_RELU_BWD on the large-tile path (M and
 * N multiples of 256) the sums are formed from the epilogue's fp32 values inside the GEMM; otherwise by mxl_colsum_bf16 afterwards. */
int mxl_gemm_bf16_colsum(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                         int transA, int transB, int flags, float alpha, const float* bias, const void* aux, int ldaux,
                         float drop_p, unsigned long long seed, unsigned site, float* colsum, void* stream);

/* C = A . B^T (bf16, the K-contiguous form, no epilogue flags) and, in the same call,
 *     delta[(b * (N / 64) + h) * T + t] = sum_{e < 64} C[m][64 h + e] * O[m][64 h + e]      m = b * T + t
 * -- the attention backward's delta (rows = tokens, heads of 64: C = d attn_vec out of the o_net input gradient, O = attn_vec;
 * upstream RelPartialLearnableMultiHeadAttn has no such tensor, it is the softmax backward's row term sum_j P dP) formed from the
 * bf16 values the GEMM stores, instead of a pass over both matrices.  Only on the four-wave large-tile kernel (M, N multiples of
 * 256, K of 64, M a multiple of T, 16-byte aligned rows): MXL_EUNSUPPORTED otherwise, with C written and delta untouched -- the
 * caller then lets mxl_relattn_bwd_fused compute delta itself. */
int mxl_gemm_bf16_headdot(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, const void* O, int ldo,
                          int T, float* delta, void* stream);

/* same kernel, grid.y = batch: operand element offsets (by / bdiv) * s?1 + (by % bdiv) * s?2 */
int mxl_gemm_bf16_batched(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                          int transA, int transB, int flags, float alpha, int ksplits, int batch, int bdiv,
                          long long sA1, long long sA2, long long sB1, long long sB2, long long sC1, long long sC2,
                          void* stream);

/* skinny-M (M <= 64) weight-streaming form for the decode step: C[M,N] = A[M,K] . W[N,K]^T (+bias)(relu);
 * flags: MXL_GEMM_OUT_F32 | MXL_GEMM_BIAS | MXL_GEMM_RELU.  Deterministic (fixed-order in-workgroup split-K). */
int mxl_gemm_skinny_bf16(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc, int flags,
                         const float* bias, void* stream);
/* Narrow-N / long-K decode linears (the FFN output projection: N = d, K = 4d): K is sliced over workgroups as well; slice s
 * writes its fp32 partial product to slabs[s] ((64, N) each, rows < M written) and mxl_ln_residual_fwd_partial finishes the
 * linear: y = LayerNorm(res + bf16(sum_s slabs[s] + bias)) -- reduction, bias and the post-LN residual in one launch.
 * (Two launches with a kernel boundary between them: a single-kernel reduction needs device-scope fences that cost more, on
 * eight L2 domains, than the slicing saves.) */
int mxl_gemm_skinny_partial(const void* A, const void* W, float* slabs, int M, int N, int K, int lda, int ldw, int KS,
                            void* stream);
int mxl_ln_residual_fwd_partial(const float* slabs, int KS, long long slab_stride, const float* bias, const void* res,
                                const float* gamma, const float* beta, void* y, int N, int d, float eps, void* stream);
/* The qkv projection of one decode step with the cache append in its epilogue (one launch instead of mxl_gemm_skinny_bf16 +
 * mxl_kv_append): qkv (B, 3d) = x (B, d) . Wqkv (3d, d)^T; the k / v thirds also go into the head-major rings
 * (B, H, Mring, dh) at slot *t_dev % Mring and q + r_r_bias into qr_out (B, d) -- HF's `cat([mems, h])` + qkv_net on the one
 * new row.  B <= 64. */
int mxl_decode_qkv(const void* x, const void* Wqkv, void* qkv, void* kcache, void* vcache, const int* t_dev,
                   const float* r_r_bias, void* qr_out, int B, int d, int dh, int Mring, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Relative-position banded attention (K4).  Replaces RelPartialLearnableMultiHeadAttn.forward between qkv_net and
 * o_net in upstream modeling_transfo_xl.py (AC/BD einsums, _rel_shift, same_length mask, softmax, P.V), as called
 * through musicnlp/models/transformer_xl.py:163-171.
 *   q   : (B, T, H, dh) bf16, element (b,i,h,e) at q + b*q_bs + i*q_rs + h*dh + e   (strides in elements)
 *   k,v : (B, Kc, H, dh) bf16 with row r = key position p - (T - Kc); T <= Kc <= M + T; positions below are the
 *         zero mems of upstream init_mems (k = v = 0) and are synthesised, not read
 *   rd  : (M, H, dh) bf16, rd[d] = r_net(pos_emb(min(d, clamp_len))), d = query position - key position
 *   r_w_bias, r_r_bias : (H, dh) f32;  out : (B, T, H, dh) bf16;  lse : (B, H, T) f32 (natural log) or NULL
 *   dh in {16, 32, 64};  scale = 1/sqrt(dh)
 * ---------------------------------------------------------------------------------------------------------- */
int mxl_relattn_fwd(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                    const float* r_r_bias, void* out, float* lse, int B, int T, int H, int dh, int M, int Kc,
                    long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs, long long o_bs, int o_rs,
                    float scale, void* stream);

/* Backward of mxl_relattn_fwd (autograd of the upstream attention core).  out/dout share the (o_bs, o_rs) layout.
 *   delta : (B,H,T) f32 scratch (written);  dq (B,T,H,dh) / dk, dv (B,Kc,H,dh) bf16 with their own strides (written);
 *   dg    : (B,H,T,M) bf16, un-skewed score gradient dG[b,h,i,d] = dSr[i, i-d] (written; may be NULL).  The caller
 *           contracts it with (q + r_r_bias) to get d rd:  d rd[d,h,:] = sum_{b,i} dg[b,h,i,d] * (q + r_r_bias)[b,i,h,:]
 *   d_r_w_bias, d_r_r_bias : (H,dh) f32, accumulated (+=).   M % 8 == 0. */
int mxl_relattn_bwd(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                    const float* r_r_bias, const void* out, const void* dout, const float* lse, float* delta,
                    void* dq, void* dk, void* dv, void* dg, float* d_r_w_bias, float* d_r_r_bias /* may be NULL: see mxl_relattn_drd */, int B, int T,
                    int H, int dh, int M, int Kc, long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs,
                    long long o_bs, int o_rs, long long dq_bs, int dq_rs, long long dkv_bs, int dkv_rs,
                    float scale, void* stream);

/* The contraction the caller owes after mxl_relattn_bwd, as one HBM-streaming kernel (dh = 64, T % 32 == 0, M % 8 == 0;
 * MXL_EUNSUPPORTED otherwise -- use mxl_gemm_bf16_batched):
 *   d_rd[delta, h*dh + e] += sum_{b,i} dg[b,h,i,delta] * qr[b,i,h,e]      qr = (q + r_r_bias) in bf16, strides (qr_bs, qr_rs)
 * Optionally also the r_r_bias gradient (rd / d_r_r_bias non-NULL):
 *   d_r_r_bias[h*dh + e] += sum_delta colsum_{b,i}(dg)[h, delta] * rd[delta, h*dh + e]
 * and, when d_r_w_bias_fix is given, the same amount is subtracted there.  This pairs with mxl_relattn_bwd called with
 * d_r_r_bias == NULL: its query-owner kernel then runs in the faster 8-wave form, which has room for one dq accumulator only and
 * therefore leaves the SUM d(r_w_bias) + d(r_r_bias) in d_r_w_bias; after mxl_relattn_drd both hold their own gradient. */
int mxl_relattn_drd(const void* dg, const void* qr, float* d_rd, int B, int T, int H, int dh, int M, long long qr_bs,
                    int qr_rs, int drd_ld, const void* rd, int rd_rs, float* d_r_r_bias, float* d_r_w_bias_fix, void* stream);

/* The pair that keeps phantom distances out of HBM (dh = 64, M % 256 == 0, T % 32 == 0).  With fresh zero memories -- the
 * reference's training: HF Trainer never carries mems, so TransfoXLModel.init_mems supplies zeros every step -- the key positions
 * before the first stored one have k = v = 0 and their score gradient depends on the distance alone:
 *     dG[b,h,i,d] = -scale * delta[b,h,i] * exp(scale * (q + r_r_bias)[b,i,h,:] . rd[d,h,:] - lse[b,h,i]).
 * mxl_relattn_bwd_sparse_dg = mxl_relattn_bwd with d_r_r_bias = NULL that leaves every (32 queries x 256 distances) block of dg
 * lying entirely on such distances UNWRITTEN; mxl_relattn_drd_recompute = mxl_relattn_drd that rebuilds exactly those blocks on
 * MFMA from qr, rd, lse and delta (the buffers mxl_relattn_bwd read / wrote) instead of streaming them.  Half of dg in mode R. */
int mxl_relattn_bwd_sparse_dg(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                              const float* r_r_bias, const void* out, const void* dout, const float* lse, float* delta, void* dq,
                              void* dk, void* dv, void* dg, float* d_r_w_bias, int B, int T, int H, int dh, int M, int Kc,
                              long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs, long long o_bs, int o_rs,
                              long long dq_bs, int dq_rs, long long dkv_bs, int dkv_rs, float scale, void* stream);
/* The same pair with the forward's help: mxl_relattn_fwd_phantom also writes, for the distance blocks whose score gradient
 * mxl_relattn_bwd_sparse_dg does not store,  oph[b,i,h,:] = sum_d 2^(G'[i,d] - mph[b,h,i]) * rd[d,h,:]  (G' = the positional score in
 * log2 units; oph (B,T,H*dh) bf16 with out's strides, mph (B,H,T) f32).  Those blocks' contribution to dq is then the elementwise
 *   -scale * delta_i * 2^(mph_i - lse_i * log2 e) * oph_i
 * and mxl_relattn_bwd_sparse_dg_oph does not walk them (M % 256 == 0, T % 32 == 0).  mxl_relattn_drd_recompute is unchanged. */
int mxl_relattn_fwd_phantom(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                            const float* r_r_bias, void* out, float* lse, void* oph, float* mph, int B, int T, int H,
                            int dh, int M, int Kc, long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs,
                            long long o_bs, int o_rs, float scale, void* stream);
int mxl_relattn_bwd_sparse_dg_oph(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                                  const float* r_r_bias, const void* out, const void* dout, const float* lse,
                                  float* delta, void* dq, void* dk, void* dv, void* dg, float* d_r_w_bias,
                                  const void* oph, const float* mph, int B, int T, int H, int dh, int M, int Kc,
                                  long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs, long long o_bs, int o_rs,
                                  long long dq_bs, int dq_rs, long long dkv_bs, int dkv_rs, float scale, void* stream);
int mxl_relattn_drd_recompute(const void* dg, const void* qr, float* d_rd, int B, int T, int H, int dh, int M, long long qr_bs,
                              int qr_rs, int drd_ld, const void* rd, int rd_rs, float* d_r_r_bias, float* d_r_w_bias_fix,
                              const float* lse, const float* delta, float scale, int Kc, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Round 4: the attention backward as ONE pass over the score cells (dh = 64, T % 32 == 0, M % 32 == 0 [round 6; M % 256 until then:
 * the reference's `small` preset has mem_len 128], Kc % 32 == 0; MXL_EUNSUPPORTED otherwise -- use mxl_relattn_bwd + mxl_relattn_drd).  Same gradients as that pair, no dg tensor:
 *   a workgroup owns 256 keys of one (sequence, head): dk, dv written once; d_rd accumulated on chip per 32-distance block and
 *   added with float atomics (d_rd (M, drd_ld) f32, +=); the partial dq of every (query tile, key block) pair goes to a
 *   slab in `ws` (mxl_relattn_bwd_fused_ws_bytes bytes, opaque) and a finishing kernel sums a query's slabs in fp32 and rounds dq
 *   to bf16 once.  The slabs are bf16 in the shipped build: each (query, key block) partial is rounded to bf16 before that sum,
 *   at most ceil(M / 256) + 1 roundings per dq element (fp32 slabs: build with -DMXL_SLAB_BF16=0; the 12-layer gradient errors are the
 *   same to four digits either way).
 *   d_r_w_bias, d_r_r_bias (H, 64) f32 are accumulated (+=), each with its own gradient (no fix-up pass).
 * Zero memories (Kc < M + T; musicnlp/models/transformer_xl.py:163-171 calls the model without mems, so upstream init_mems
 * supplies zeros): the key positions below the first stored one have k = v = 0 and exist only as distances.  Their part of dq
 * is  -scale * delta_i * 2^(mph_i - lse_i * log2 e) * oph_i  with oph / mph from mxl_relattn_fwd_phantom2(..., oph_all = 1)
 * (required then), and their part of d_rd is owed by the caller: mxl_relattn_drd_phantom.  With Kc == M + T oph / mph are unused.
 * `delta` (B,H,T) f32 scratch is written -- unless bit 1 of `defer_finish` says the caller has filled it already
 * (delta[b][h][i] = sum_e dout . out: mxl_gemm_bf16_headdot forms it inside the GEMM that produces dout).  dq_rs, dq_bs multiples of 8.
 * defer_finish bit 0: the slab sum is left to the caller (mxl_relattn_dq_finish, same ws / oph / mph / lse / delta / dq / d_r_r_bias
 * arguments: with oph it also adds the phantom cells' part of d_r_r_bias, the column sums of their dq term; d_r_r_bias may be NULL).
 * dq and d_r_r_bias are complete only after it. */
size_t mxl_relattn_bwd_fused_ws_bytes(int B, int T, int H, int dh, int M);
int mxl_relattn_bwd_fused(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                          const float* r_r_bias, const void* out, const void* dout, const float* lse, float* delta,
                          void* dq, void* dk, void* dv, float* d_rd, int drd_ld, float* d_r_w_bias, float* d_r_r_bias,
                          const void* oph, const float* mph, void* ws, int B, int T, int H, int dh, int M, int Kc,
                          long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs, long long o_bs, int o_rs,
                          long long dq_bs, int dq_rs, long long dkv_bs, int dkv_rs, float scale, int defer_finish, void* stream);
int mxl_relattn_dq_finish(const void* ws, const void* oph, const float* mph, const float* lse, const float* delta, void* dq,
                          float* d_r_r_bias, int B, int T, int H, int dh, int M, int Kc, long long o_bs, int o_rs, long long dq_bs,
                          int dq_rs, float scale, void* stream);
/* mxl_relattn_fwd_phantom with a choice of which phantom cells enter oph: oph_all = 0 is mxl_relattn_fwd_phantom (the
 * all-phantom 256-distance blocks, for mxl_relattn_bwd_sparse_dg_oph); oph_all = 1 sums over EVERY key position below the first
 * stored key tile (for mxl_relattn_bwd_fused; needs (T - Kc) % 64 == 0).  ph_ws (or NULL; oph_all = 1, dh = 64 only): the forward
 * also fills the per-tile records of mxl_relattn_drd_phantom (mxl_relattn_drd_phantom_ws_bytes bytes), sparing the backward
 * mxl_relattn_drd_phantom_prep. */
int mxl_relattn_fwd_phantom2(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                             const float* r_r_bias, void* out, float* lse, void* oph, float* mph, int oph_all, void* ph_ws, int B,
                             int T, int H, int dh, int M, int Kc, long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs,
                             long long o_bs, int o_rs, float scale, void* stream);
/* d_rd[delta, h*64 + e] += sum over the PHANTOM cells (key position i - delta below T - Kc) of dG[b,h,i,delta] * qr[b,i,h,e],
 * qr = q + r_r_bias, with dG[i,delta] = -scale * delta_i * exp(scale * qr_i . rd[delta] - lse_i) rebuilt on MFMA (two products and
 * one exponential per cell, nothing streamed) -- cell by cell, so that together with mxl_relattn_bwd_fused every (query, distance)
 * pair is counted once.  (T - Kc) % 64 == 0, T % 32 == 0, M % 32 == 0, M <= 8192, dh == 64.
 * `ws` (mxl_relattn_drd_phantom_ws_bytes bytes, 16-byte aligned) holds one record per (sequence, head, 32-query tile): the tile's
 * bf16((q + r_r_bias) * scale * log2 e) rows in the kernel's LDS image order and -lse * log2 e of its queries -- written by the
 * forward (mxl_relattn_fwd_phantom2(..., ph_ws)) or by mxl_relattn_drd_phantom_prep; `delta` (B,H,T) f32 is the array
 * mxl_relattn_bwd_fused fills.  Those cells' part of d r_r_bias is added by mxl_relattn_dq_finish. */
size_t mxl_relattn_drd_phantom_ws_bytes(int B, int T, int H);
int mxl_relattn_drd_phantom_prep(const void* q, long long q_bs, int q_rs, const float* r_r_bias, const float* lse, void* ws, int B, int T,
                                 int H, int dh, float scale, void* stream);
int mxl_relattn_drd_phantom(const void* ws, const float* delta, float* d_rd, int B, int T, int H, int dh, int M, int drd_ld,
                            const void* rd, int rd_rs, int Kc, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * HBM-bound layer pieces.
 * ---------------------------------------------------------------------------------------------------------- */
/* upstream PositionalEmbedding + drop(pos_emb): out[dist][0:d/2]=sin, [d/2:d]=cos of min(dist,clamp)*inv_freq; (M,d) bf16 */
int mxl_sinusoid_table(void* out, int M, int d, int clamp_len, float drop_p, unsigned long long seed, unsigned site,
                       void* stream);
/* upstream AdaptiveEmbedding (div_val=1) + drop: out[n] = drop(E[ids[n]] * scale); ids int64, E (V,d) bf16 */
int mxl_embed_fwd(const void* ids, const void* E, void* out, int N, int d, int V, float scale, float drop_p,
                  unsigned long long seed, unsigned site, void* stream);
/* its backward: dE[ids[n]] += keep * scale * (dout[n] + dout2[n])  (f32 atomics; dout2 may be NULL) */
int mxl_embed_bwd(const void* ids, const void* dout, const void* dout2, float* dE, int N, int d, int V, float scale,
                  float drop_p, unsigned long long seed, unsigned site, void* stream);
/* y = keep * x / (1-p): the final drop(core_out) of TransfoXLModel.forward (and its backward with x := dy); n % 8 == 0 */
int mxl_dropout_bf16(const void* x, void* y, long long n, float drop_p, unsigned long long seed, unsigned site,
                     void* stream);
/* y = LayerNorm(res + drop(x)) * gamma + beta  (post-LN of dec_attn / pos_ff); z (pre-norm sum), mean, rstd saved
 * for the backward (any of z/mean/rstd/res may be NULL).  x,res,y,z (N,d) bf16; d % 8 == 0, d <= 2048 */
int mxl_ln_residual_fwd(const void* x, const void* res, const float* gamma, const float* beta, void* y, void* z,
                        float* mean, float* rstd, int N, int d, float eps, float drop_p, unsigned long long seed,
                        unsigned site, void* stream);
/* backward of the above for upstream grad dy (+ dy2 if non-NULL): dres = dz, dx = keep*dz/(1-p), dgamma/dbeta += */
int mxl_ln_residual_bwd(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd,
                        const float* gamma, void* dres, void* dx, float* dgamma, float* dbeta, int N, int d,
                        float drop_p, unsigned long long seed, unsigned site, void* stream);
/* same, and dxsum[c] += sum_rows dx[row][c] (the stored bf16 values; d <= 1024): x is the output of a biased linear layer
 * (pos_ff.CoreNet.3), so this IS that layer's bias gradient -- one pass over dx less than mxl_colsum_bf16 afterwards */
int mxl_ln_residual_bwd_colsum(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd,
                               const float* gamma, void* dres, void* dx, float* dgamma, float* dbeta, float* dxsum, int N,
                               int d, float drop_p, unsigned long long seed, unsigned site, void* stream);
/* same, with an extra gradient stream added to the residual output: dres = dz + dadd (Reformer's y1 = x1 + f(x2) chains) */
int mxl_ln_residual_bwd_add(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd,
                            const float* gamma, const void* dadd, void* dres, float* dgamma, float* dbeta, int N, int d,
                            void* stream);
/* same, and in the same pass dx = dropout(dres) (mask of (seed, site), element index row * d + column: the mask mxl_dropout_bf16
 * applies to a compact (N, d) matrix) and, if dxsum != NULL, dxsum[c] += sum_rows dx[row][c] -- what mxl_dropout_bf16 /
 * mxl_dropout_colsum_bf16 over dres would produce, bit for bit, without the extra pass (the Reformer backward's
 * y = x + dropout(f(.)) chains of HF ReformerLayer / _ReversibleFunction, modeling_reformer.py:1535-1757 as cited in SURVEY.md;
 * here with stored activations instead of the reversible recompute).  dx may alias dy; d <= 1024 */
int mxl_ln_residual_bwd_add_drop(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd,
                                 const float* gamma, const void* dadd, void* dres, void* dx, float* dxsum, float* dgamma,
                                 float* dbeta, int N, int d, float drop_p, unsigned long long seed, unsigned site, void* stream);
/* out[b][t][:] = bf16(x[b][t][:] + bias[:]) with x strided (x_bs, x_rs elements), out compact (B,T,n) */
int mxl_add_rowbias_bf16(const void* x, long long x_bs, int x_rs, const float* bias, void* out, int B, int T, int n,
                         void* stream);
/* out[n] += sum_m X[m][n]  (bias gradients), X (M,N) bf16 with leading dimension ld */
int mxl_colsum_bf16(const void* X, float* out, int M, int N, int ld, void* stream);
/* Y = dropout(X) with the (seed, site) mask over the flat element index (mxl_dropout_bf16) and out[n] += sum_m Y[m][n] in the same
 * pass (the Reformer's feed_forward.output.dense bias gradient: the regenerated forward mask and the column sums); N % 8 == 0 */
int mxl_dropout_colsum_bf16(const void* X, void* Y, float* out, int M, int N, float drop_p, unsigned long long seed,
                            unsigned site, void* stream);
/* Y[m][n] = X[m][n] - mean_m X[m][n]  (bf16 in / out, fp32 arithmetic; N % 8 == 0).  The positional table enters the r_net
 * weight gradient dW_r = sum_d dRd[d]^T phi[d] centred over the distance axis: sum_d dRd[d] = 0 exactly (every softmax row's
 * score gradients sum to zero and every query sees exactly mem_len distances), so the constant part of phi multiplies nothing
 * but the bf16 rounding noise of the score gradients -- which would otherwise dominate the low-frequency columns. */
int mxl_center_columns_bf16(const void* X, void* Y, int M, int N, void* stream);
/* upstream TransfoXLModel._update_mems: out[b] = cat(mem[b], hid[b])[-M:], all (B, len, d) bf16, out != mem */
int mxl_mem_update(const void* mem, const void* hid, void* out, int B, int M, int T, int d, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Projected adaptive log-softmax head (K7) = upstream ProjectedAdaptiveLogSoftmax (div_val=1) as used at
 * musicnlp/models/transformer_xl.py:185,193,198-200.  logits (B*T, ldl) f32 = hidden . [E ; cluster_weight]^T + bias,
 * columns [0,V) tokens, [V, V+ncl) clusters.  cutoffs_host: ncl ints on the HOST (read at enqueue time).
 * ---------------------------------------------------------------------------------------------------------- */
/* transformer_xl.py:176-182 all-ignored guard on label row 0 (in place, device) */
int mxl_label_guard(void* labels_row0, int T, long long eos, void* stream);
/* nll (B, T-1) f32 with the shift inside; lse (B*T, 2) f32 scratch kept for backward;
 * acc2[0] += sum(nll), acc2[1] += count(nll != 0)  (caller zeroes acc2) */
int mxl_adaptive_nll_fwd(const float* logits, int ldl, const void* labels, float* nll, float* lse, float* acc2, int B,
                         int T, int V, int ncl, const int* cutoffs_host, void* stream);
/* dlogits (B*T, ldd) bf16 of loss = sum(nll[nll != 0]) / count, times grad_scale; pad columns zeroed */
int mxl_adaptive_nll_bwd(const float* logits, int ldl, const void* labels, const float* nll, const float* lse,
                         const float* acc2, void* dlogits, int ldd, int B, int T, int V, int ncl,
                         const int* cutoffs_host, float grad_scale, void* stream);
/* the same gradient as a two-term bf16 sum: dlogits_hi = bf16(d), dlogits_lo = bf16(d - dlogits_hi).  The consumers (the head's
 * input-gradient and weight-gradient GEMMs, the bias column sum) run once per term and accumulate: a confident prediction of a
 * token that is not the label has d near 1 / count, where one bf16 term is too coarse for sums over tokens that cancel. */
int mxl_adaptive_nll_bwd_split(const float* logits, int ldl, const void* labels, const float* nll, const float* lse,
                               const float* acc2, void* dlogits_hi, void* dlogits_lo, int ldd, int B, int T, int V, int ncl,
                               const int* cutoffs_host, float grad_scale, void* stream);
/* labels=None branch: out (N, ldo) f32 full log-probabilities over the V tokens */
int mxl_adaptive_logprob(const float* logits, int ldl, float* out, int ldo, int N, int V, int ncl,
                         const int* cutoffs_host, void* stream);

/* ---- the same head for LARGE vocabularies (sub-word tokenizers, musicnlp/trainer/wordpiece_tokenizer.py:349-452: V up to 262144
 * with cutoffs [20000, 40000, 200000], musicnlp/models/transformer_xl.py:53-66), in upstream's own structure
 * (ProjectedAdaptiveLogSoftmax.forward with labels): the head softmax (c1 shortlist columns + one column per tail cluster) for
 * every token, the tail softmax of cluster i only for the tokens whose label lies in it (upstream's `mask_i.nonzero()` /
 * `index_select`), the caller walking tokens in chunks -- a (tokens x V) tensor never exists.  Token row = b*T + t; the label of
 * row (b, t) is labels[b][t+1] (shift inside, as above); rows with t = T-1 or an ignored label belong to no cluster. */
/* perm (ncl+2, B*T) int32: row g lists the token rows of group g in increasing order (g = 0 shortlist, 1..ncl tail clusters,
 * ncl+1 ignored); counts[g] their number; tgt_head[row] = the row's target column in the packed head logits [0,c1) + [c1, c1+ncl)
 * (-1: ignored); tgt_tail[row] = label - cutoff_i for a tail token, else -1. */
int mxl_cluster_bucket(const void* labels, int B, int T, int V, int ncl, const int* cutoffs_host, int* perm, int* counts,
                       int* tgt_head, int* tgt_tail, void* stream);
/* dst[j][:] = src[idx[j]][:] (bf16 rows of d elements, d % 8 == 0) and its adjoint dst[idx[j]][:] += src[j][:] (idx unique) */
int mxl_gather_rows_bf16(const void* src, int ld_src, const int* idx, void* dst, int n, int d, void* stream);
int mxl_scatter_add_rows_bf16(const void* src, const int* idx, void* dst, int ld_dst, int n, int d, void* stream);
/* logits (n_rows, ld) f32, row j belongs to token rows_idx[j] (or row0 + j when rows_idx is NULL):
 * lse_out[token] = logsumexp(logits[j][0:ncols]),  pick_out[token] = logits[j][tgt[token]] (0 when tgt[token] < 0) */
int mxl_rows_lse_pick(const float* logits, long long ld, int ncols, int n_rows, const int* rows_idx, int row0, const int* tgt,
                      float* lse_out, float* pick_out, void* stream);
/* nll[b][t] = nll_tok[b*T+t] = (head_lse - head_pick) + (tail_lse - tail_pick if a tail token), 0 for ignored rows;
 * acc2[0] += sum, acc2[1] += count(nll != 0) (caller zeroes acc2) -- the reduction of transformer_xl.py:198-200 */
int mxl_bucket_nll_finish(const float* head_lse, const float* head_pick, const float* tail_lse, const float* tail_pick,
                          const int* tgt_head, const int* tgt_tail, float* nll, float* nll_tok, float* acc2, int B, int T,
                          void* stream);
/* out (n_rows, ldo) bf16 = (softmax(logits[j][0:ncols]) - onehot(tgt[token])) * grad_scale / max(acc2[1], 1), pad columns and
 * the rows of ignored / zero-loss tokens zeroed; out_lo (optional): the bf16 remainder (see mxl_adaptive_nll_bwd_split) */
int mxl_rows_softmax_grad(const float* logits, long long ld, int ncols, int n_rows, const int* rows_idx, int row0, const int* tgt,
                          const float* lse, const float* nll_tok, const float* acc2, float grad_scale, void* out_hi, void* out_lo,
                          int ldo, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimiser (K18): torch.optim.AdamW + clip_grad_norm_ as configured at musicnlp/trainer/train.py:165-190.
 * ---------------------------------------------------------------------------------------------------------- */
int mxl_sumsq_f32(const float* x, long long n, float* out_accum, void* stream);
/* p,g,m,v f32 [n]; w16 bf16 working copy (or NULL).  g_eff = g * grad_scale * min(1, max_norm/(sqrt(*sumsq)*grad_scale+1e-6));
 * decoupled weight decay on elements [0, n_decay) only; step >= 1 */
int mxl_adamw_step(float* p, const float* g, float* m, float* v, void* w16, long long n, long long n_decay, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, const float* sumsq,
                   float max_norm, float grad_scale, void* stream);
int mxl_cast_f32_bf16(const float* x, void* y, long long n, void* stream);
/* y = (float)x.  With mxl_cast_f32_bf16 the two ends of the bf16 gradient exchange: a bucket of the flat fp32 gradient buffer
 * is narrowed into a bf16 staging buffer, all-reduced over RCCL at half the bytes (186 MB instead of 372 MB per step at
 * 12L/768d), and widened back in place. */
int mxl_cast_bf16_f32(const void* x, float* y, long long n, void* stream);
/* dst[b][c][r] = src[b][r][c] (bf16; element strides between batch items).  Keeps [in][out] copies of the Linear weights so
 * that the input gradient dX = dY W (autograd of F.linear) runs in the K-contiguous GEMM form. */
int mxl_transpose_bf16(const void* src, void* dst, int rows, int cols, int ld_src, int ld_dst, int batch,
                       long long src_bstride, long long dst_bstride, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Autoregressive decode (A4/A10): replaces the per-token path of HF GenerationMixin.greedy_search / sample driving
 * MyTransfoXLLMHeadModel.prepare_inputs_for_generation (musicnlp/models/transformer_xl.py:223-241) and the upstream
 * cat(mems, h) -> qkv_net recomputation.  All step state is on the device (t_dev = position of the token being fed,
 * rng_ctr), so a step is capture-safe and replays as a hipGraph.
 *   ids     : (B, ld_ids) int64 token buffer; step reads ids[b][t], sampler writes ids[b][t+1]
 *   k/vcache: head-major (B, H, M, dh) bf16 rings of projected K/V rows, slot = position mod M, zero-initialised
 *             (= upstream zero mems)
 * ---------------------------------------------------------------------------------------------------------- */
int mxl_decode_embed(const void* ids, int ld_ids, const int* t_dev, const void* E, void* out, int B, int d, int V,
                     float scale, void* stream);
/* cache[b][t mod M] <- k, v of the (B, 3d) qkv rows of the current token */
int mxl_kv_append(const void* qkv, void* kcache, void* vcache, const int* t_dev, int B, int M, int d, int dh,
                  const float* r_r_bias, void* qr_out, void* stream);   /* qr_out (B, d) bf16 = q + r_r_bias, or NULL */
/* after the prompt forward: cache slots <- K/V rows of the last min(T, M) positions of a (B, T, 3d) qkv buffer */
int mxl_kv_fill(const void* qkv, void* kcache, void* vcache, int B, int T, int M, int d, int dh, void* stream);
/* positional term of a decode step for the whole batch: bd[b][h][r] = sum_e qr[b][h*dh + e] * rd[r][h*dh + e], r < M.
 * qr (B <= 64, ld_qr) bf16 = q + r_r_bias (mxl_decode_qkv / mxl_kv_append), rd (M, ld_rd) bf16 = r_net(pos_emb) of the layer
 * (rows = distances), bd (B, H, M) f32.  dh = 64.  Replaces the `BD = einsum("ibnd,jnd->ijbn", rr_head_q, r_head_k)` of
 * RelPartialLearnableMultiHeadAttn.forward (transformers 4.25.1 modeling_transfo_xl.py) for a single query position. */
int mxl_decode_bd(const void* qr, const void* rd, float* bd, int B, int H, int dh, int M, int ld_qr, int ld_rd, void* stream);
/* single-query relative attention over the ring, distances 0..M-1.  bd (B, H, M) f32 = (q + r_r_bias) . rd[dist], the
 * positional term for the whole batch (one mxl_gemm_bf16_batched per layer over heads); out (B, H*dh) bf16 */
int mxl_relattn_decode(const void* qkv, const void* kcache, const void* vcache, const float* bd, const float* r_w_bias,
                       void* out, const int* t_dev, int B, int H, int dh, int M, float scale, void* stream);
/* The same with every (sequence, head) ring cut into `pieces` (1..8) workgroups -- for launches whose B * H workgroups would
 * not fill the CUs evenly (a CU streams HBM at a fixed rate however many workgroups it holds: 384 rings on 256 CUs take the
 * time of two).  Each piece runs the softmax over its slots with its own reference; the piece that finishes last merges the
 * (o, max, sum) partials in piece order, so the result is reproducible.  The probabilities are rounded to bf16 relative to the
 * piece's own maximum: outputs differ from mxl_relattn_decode's in the last bf16 bit at most.  ws: mxl_relattn_decode_split_ws_bytes
 * bytes; arrived: B * H ints, zero before the first call (the kernel leaves them zero).  pieces = 1 is mxl_relattn_decode. */
size_t mxl_relattn_decode_split_ws_bytes(int B, int H, int dh, int pieces);
int mxl_relattn_decode_split(const void* qkv, const void* kcache, const void* vcache, const float* bd, const float* r_w_bias,
                             void* out, const int* t_dev, int B, int H, int dh, int M, float scale, int pieces, float* ws,
                             int* arrived, void* stream);
/* next token from log-probs (B, ldl): repetition penalty over the ids already in the row (positions 0..*t_dev; HF's
 * RepetitionPenaltyLogitsProcessor, 1.0 = off), then greedy argmax (do_sample=0) or temperature -> top-k -> top-p ->
 * typical-p (HF's TypicalLogitsWarper, 1.0 = off) -> renormalise -> multinomial (do_sample=1), as HF's logits warpers with
 * renormalize_logits=True -- the `sample` strategy's accepted keys at musicnlp/trainer/eval.py:277-326.
 * V <= 2048.  out_probs (B, V) f32 optional: the renormalised distribution actually sampled from (test hook). */
int mxl_sample(const float* logprobs, int ldl, int V, void* ids, int ld_ids, const int* t_dev,
               unsigned long long* rng_ctr, unsigned long long seed, int B, int do_sample, int top_k, float top_p,
               float temperature, float repetition_penalty, float typical_p, float* out_probs, void* stream);
/* The same recipe for V > 2048 (the sub-word vocabularies of musicnlp/trainer/wordpiece_tokenizer.py:349-452, whose sizes pick
 * the cutoff ladder of musicnlp/models/transformer_xl.py:53-66): no sort -- every warper and the multinomial draw are bisections
 * over a 32-bit order key with integer (fixed-point) conditional sums, so the result is independent of timing.  scratch: B * V * 8
 * bytes.  Any V >= 1 is accepted (mxl_sample is faster below 2049). */
int mxl_sample_large(const float* logprobs, int ldl, int V, void* ids, int ld_ids, const int* t_dev,
                     const unsigned long long* rng_ctr, unsigned long long seed, int B, int do_sample, int top_k, float top_p,
                     float temperature, float repetition_penalty, float typical_p, float* out_probs, float* scratch,
                     void* stream);
/* t_dev += 1; rng_ctr += 1 */
int mxl_decode_advance(int* t_dev, unsigned long long* rng_ctr, void* stream);
/* Round 6: mxl_sample + mxl_decode_embed of the sampled token + mxl_decode_advance in one launch -- the tail of one decode step and
 * the head of the next.  The row's workgroup writes ids[b][t + 1] and emb_out[b] = E[token] * scale (bf16, (B, d)); the workgroup
 * that finishes last advances *t_dev and *rng_ctr (counter: one int, zero before the first call, left zero).  `scores` (B, ldl) may
 * be log-probabilities or, when repetition_penalty == 1, the head's raw logits: argmax, top-k / top-p / typical-p and the
 * renormalised draw do not change under the per-row shift that separates the two.  V <= 2048. */
int mxl_sample_step(const float* scores, int ldl, int V, void* ids, int ld_ids, int* t_dev, unsigned long long* rng_ctr,
                    unsigned long long seed, int B, int do_sample, int top_k, float top_p, float temperature,
                    float repetition_penalty, float typical_p, const void* E, void* emb_out, int d, float scale, int* counter,
                    void* stream);
/* Contrastive search (the reference's 'contrastive' strategy, musicnlp/trainer/eval.py:296-302, over the mems patch of
 * musicnlp/models/transformer_xl.py:229-234; HF 4.25.1 GenerationMixin.contrastive_search with `_ranking_fast`):
 *   score[b*K + k] = (1 - alpha) * probs[b*K + k] - alpha * max_{s < S} cos(hid[b*K + k], ctx[b][s]);  sel[b] = argmax_k score
 * ctx (B, ., d) bf16 = last-layer hidden states of the S context positions (batch stride ctx_bs elements), ctx_inv_norm (B, .) f32
 * their reciprocal norms (mxl_row_inv_norm_bf16; batch stride inv_bs), hid (B*K, d) bf16 the candidates' hidden states,
 * probs (B*K) f32 their top-k probabilities; sel int64 (B). */
int mxl_row_inv_norm_bf16(const void* x, long long ld, int n, int d, float* out, void* stream);
int mxl_contrastive_select(const void* ctx, long long ctx_bs, const float* ctx_inv_norm, int inv_bs, int S, const void* hid,
                           const float* probs, float alpha, int B, int K, int d, float* score, void* sel, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Reformer path (A6-A8): replaces HuggingFace modeling_reformer.py as reached through
 * musicnlp/models/reformer.py:114-127 (AxialPositionEmbeddings, LSHSelfAttention._hash_vectors / _stable_argsort /
 * _attend / ReverseSort / hash-round merge, LocalSelfAttention).  Chunk length 64, one chunk look-back, causal decoder.
 * ---------------------------------------------------------------------------------------------------------- */
/* out[b,t] = drop(E[ids]) + cat(W0[t / A1], W1[t % A1]) with HF's 2-D position dropout; E (V,d) bf16, W0 (A0,d0) f32, W1 (A1,d-d0) f32 */
int mxl_axial_embed_fwd(const void* ids, const void* E, const float* W0, const float* W1, void* out, int B, int T, int d,
                        int V, int A0, int A1, int d0, float drop_p, unsigned long long seed, unsigned site_emb,
                        unsigned site_pos, void* stream);
int mxl_axial_embed_bwd(const void* ids, const void* dout, const void* dout2, float* dE, float* dW0, float* dW1, int B, int T,
                        int d, int V, int A0, int A1, int d0, float drop_p, unsigned long long seed, unsigned site_emb,
                        unsigned site_pos, void* stream);
/* buckets[b,h,r*T+t] = r*NB + combined argmax([xR;-xR]) per factor; qk (B,T,H,dh) bf16 strided (bs, rs);
 * rotations (H, dh, n_h, sum(factors)/2) f32 supplied by the caller; factors_host: nfac ints on the host */
int mxl_lsh_hash(const void* qk, long long bs, int rs, const float* rotations, int* buckets, int B, int T, int H, int dh,
                 int n_h, int nfac, const int* factors_host, void* stream);
/* stable sort of the S = n_h*T slots of each of the BH rows by bucket: sorted_idx (slot -> element), sorted_pos = idx % T */
int mxl_lsh_sort(const int* buckets, int* sorted_idx, int* sorted_pos, int BH, int S, int T, int n_buckets_total, void* stream);
/* chunked attention: local (lsh=0, sorted_pos NULL, n_h=1; separate q/k/v) or LSH (lsh=1; k == q == shared qk).
 * out rows (b, round, pos) x (H*dh) bf16; lse (B, n_h, H, T) f32; probs dropout drop_p.
 * T > 64: T % 64 == 0 (chunks of 64 with one look-back chunk).  T <= 64: the single-chunk case -- HF's plain causal attention
 * over the T tokens (no look-back, sorted_pos ignored, n_h must be 1; for lsh the shared-QK key normalisation and the
 * self mask still apply). */
int mxl_chunk_attn_fwd(const void* q, const void* k, const void* v, const int* sorted_pos, void* out, float* lse, int B, int T,
                       int H, int dh, int n_h, int lsh, long long bs, int rs, float drop_p, unsigned long long seed,
                       unsigned site, void* stream);
/* backward: dq, dk, dv (B, n_h, T, H*dh) f32 -- one (T, H*dh) slab per hash round; every element is written exactly once (plain
 * stores, no pre-zeroing) and the rounds are summed by the consumer (mxl_lsh_keynorm_bwd_rounds).  (Until round 6 the rounds of
 * n_h > 1 met in one (B, T, H*dh) buffer through a float atomic per element: 2.9 ms per call at the reference's logged
 * Reformer-base shape against 0.45 ms now.)  n_h == 1: each of the three may instead go out as bf16 through dq16 / dk16 / dv16 (rows
 * of ld16 elements; NULL = use the f32 destination) -- e.g. straight into the (N, 3d) operand of the projection-gradient GEMMs.
 * For lsh, dk is w.r.t. the normalised key (see keynorm_bwd); dout has out's layout; dlse (B,n_h,H,T) f32 or NULL */
int mxl_chunk_attn_bwd(const void* q, const void* k, const void* v, const int* sorted_pos, const void* out, const float* lse,
                       const void* dout, const float* dlse, float* dq, float* dk, float* dv, void* dq16, void* dk16, void* dv16,
                       int ld16, int B, int T, int H, int dh, int n_h, int lsh, long long bs, int rs, float drop_p,
                       unsigned long long seed, unsigned site, void* stream);
/* dqk (B*T rows of ld_dqk elements, H*dh used) bf16 = dq + chain rule of k = qk * rsqrt(mean(qk^2)+1e-6)/sqrt(dh) applied to
 * dk_eff */
int mxl_lsh_keynorm_bwd(const void* qk, long long bs, int rs, const float* dq, const float* dk_eff, void* dqk, int ld_dqk, int B,
                        int T, int H, int dh, void* stream);
/* the same over per-round slabs: dq, dk_eff (and dv) (B, n_h, T, H*dh) f32 are summed over the rounds, in round order, on the way
 * in; dv's sum (the rounds' value gradients) leaves as bf16 rows of ld_dv elements through dv16 (both NULL: no dv) */
int mxl_lsh_keynorm_bwd_rounds(const void* qk, long long bs, int rs, const float* dq, const float* dk_eff, const float* dv, void* dqk,
                               int ld_dqk, void* dv16, int ld_dv, int B, int T, int H, int dh, int n_h, void* stream);
/* hash-round merge: out = sum_r softmax_r(lse) * out_r, and its backward (dout_r, dlse) */
int mxl_lsh_combine(const void* out_r, const float* lse, void* out, int B, int T, int H, int dh, int n_h, void* stream);
int mxl_lsh_combine_bwd(const void* out_r, const float* lse, const void* out, const void* dout, void* dout_r, float* dlse,
                        int B, int T, int H, int dh, int n_h, void* stream);

/* ---- Reformer incremental (cached) decoding: the single-token step of HF's `use_cache` path, i.e. what
 * `model.generate(...)` at musicnlp/trainer/eval.py:333 runs (ReformerDynamicCache HF515:65-148; LSHSelfAttention.forward with
 * past_buckets_states HF515:466-516, 946-1050; LocalSelfAttention HF515:1136-1169, 1327-1329).  The caches hold, per layer,
 * the PROJECTIONS of every position so far ((B, Tmax, H*dh) bf16: k and v for local layers, shared qk and v for LSH layers --
 * HF caches the LayerNorm'ed hidden states and re-projects what it gathers: the same numbers) and, per LSH layer, the offset
 * bucket ids (B*H, n_h, Tmax) int32. */
/* x[b] = E[ids[b, t]] + cat(W0[t / A1], W1[t % A1]): ReformerEmbeddings in eval with start_idx_pos_encodings = t */
int mxl_rf_decode_embed(const void* ids, int ld_ids, int t, const void* E, const float* W0, const float* W1, void* out, int B,
                        int d, int V, int A1, int d0, void* stream);
/* buckets (rows, n_h*T) as written by mxl_lsh_hash (r*NB + b) -> r*(NB+1) + (t < T_real ? b : NB): the padded prefill of HF's
 * eval mode sends pad positions to ONE extra bucket and widens the per-round offsets (HF515:746-756) */
int mxl_lsh_fix_buckets(int* buckets, int rows, int n_h, int T, int T_real, int NB, void* stream);
/* append the new token's bucket ids: raw (rows, n_h) = r*NB + b from mxl_lsh_hash on the one query; cache (rows, n_h, Tmax);
 * *bkmax = maximum id in the cache.  Offsets widen to NB + 1 when *bkmax > n_h*NB - 1 (HF515:961-970); *bkmax is updated. */
int mxl_rf_query_bucket(const int* raw, int* cache, int* bkmax, int rows, int n_h, int NB, int Tmax, int t, void* stream);
/* one query per (sequence, head, round) at position t over <= 128 cached positions.  sorted == NULL: the contiguous range
 * [start, start + count) -- a local layer (lsh = 0: keys / sqrt(dh), no mask) or an LSH layer before its first hashing (lsh = 1:
 * shared-QK key normalisation, self mask -1e5 on position t).  sorted (B*H*n_h, n = t + 1) = mxl_lsh_sort of (cached ; new)
 * bucket ids per round: the 64-slot chunk holding the new token and the chunk before it, slots modulo n (HF515:1032-1050);
 * rounds merged by softmax of their logsumexp (HF515:636-655).  out (B, H*dh) bf16. */
int mxl_rf_decode_attn(const void* q, int ldq, const void* kcache, const void* vcache, const int* sorted, void* out, int B, int H,
                       int dh, int n_h, int Tmax, int n, int t, int start, int count, int lsh, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Evaluation metrics: replaces `preprocess_logits_for_metrics` + ComputeMetrics / IkrMetric on gathered
 * logits (musicnlp/trainer/train.py:265-284, musicnlp/trainer/metrics.py:45-117).
 * ---------------------------------------------------------------------------------------------------------- */
/* ids[n] = argmax_v logits[n][v] (first maximum); logits (N, ld) f32, ids int64 */
int mxl_argmax_rows(const float* logits, int ld, void* ids_out, int N, int V, void* stream);
/* out14[b] = { pitch-class histogram[12] of predicted pitch tokens at non-ignored positions, #correct next tokens, #non-ignored
 * next-token positions }.  preds int64 (B, T-1 if clm_pred_shifted else T), labels int64 (B, T), id2pc int8[V]: pitch class
 * 0..11 of a pitch token id, -1 for every other id (rests and the rare-pitch token included). */
int mxl_eval_counts(const void* preds, int ld_preds, const void* labels, int ld_labels, const signed char* id2pc, int V,
                    int* out14, int B, int T, int clm_pred_shifted, void* stream);

/* the same over a SUB-WORD vocabulary (musicnlp/trainer/wordpiece_tokenizer.py:349-452, pair_merge_tokenizer.py:200-289), whose ids
 * stand for several base tokens: id2hist uint8 (V, 12) = number of pitches of each class an id expands to (the reference's
 * per-id `_id2pchs_exc` lists, wordpiece_tokenizer.py:372-379, reduced to pitch classes) */
int mxl_eval_counts_multi(const void* preds, int ld_preds, const void* labels, int ld_labels, const unsigned char* id2hist, int V,
                          int* out14, int B, int T, int clm_pred_shifted, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Input pipeline: the pad / truncate / label contract of `tokenizer(toks, padding='max_length',
 * truncation=True)` (musicnlp/preprocess/dataset.py:361) + DataCollatorForLanguageModeling(mlm=False) (train.py:360),
 * applied on the device to one packed run of pre-tokenised ids.
 *   tokens  : elem_bytes = 2 (uint16) or 4 (int32) token ids, the B sequences back to back
 *   offsets : int32[B+1] element offsets into `tokens` (sequence b = [offsets[b], offsets[b+1]); may be empty or longer
 *             than max_length: truncated)
 *   ids_out, labels_out : (B, max_length) int64; labels may be NULL.  labels = ids with every pad_id -> -100.
 *   remap (n_tables, v_src) int32 + row_table int32[B] (or both NULL): token v of row b becomes remap[row_table[b]][v]
 *             (row_table[b] < 0: unchanged) -- the reference's step -> degree PitchShift (musicnlp/preprocess/transform.py:
 *             154-237) is one such table per key.
 * ---------------------------------------------------------------------------------------------------------- */
int mxl_pack_clm_batch(const void* tokens, int elem_bytes, const int* offsets, void* ids_out, void* labels_out, int B,
                       int max_length, long long pad_id, const int* remap, const int* row_table, int v_src, void* stream);

/* Post-decode / prompt id ops: out[b] = position of the last (which < 0) or which-th (0-based) occurrence
 * of `token` in row b of ids (B, T) int64, -1 if absent.  With token = the start-of-bar id this is the cut point of
 * MusicGenerator._truncate_last_bar (musicnlp/trainer/eval.py:178-185, which = -1) and of truncate_first_n_bar
 * (eval.py:187-198, which = n_bar). */
int mxl_find_token(const void* ids, int ld_ids, int B, int T, long long token, int which, int* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Measurement hook (no reference counterpart).  While enabled, the launch functions of the attention path bracket
 * EACH of their kernels with a hipEvent pair on the launch stream, so that bench.py can report per-kernel durations measured
 * live inside its timed region.  Enqueue-time cost: two hipEventRecord per kernel; NOT capture-safe (leave it off while a
 * hipGraph is being captured).  Kernel ids: */
#define MXL_KT_RELATTN_FWD   0   /* relattn_fwd_kernel                                  */
#define MXL_KT_RELATTN_DELTA 1   /* relattn_bwd_delta_kernel                            */
#define MXL_KT_RELATTN_DQ    2   /* relattn_bwd_dq8_kernel / relattn_bwd_dq_kernel      */
#define MXL_KT_RELATTN_DKV   3   /* relattn_bwd_dkv_kernel                              */
#define MXL_KT_RELATTN_DRD   4   /* relattn_drd_kernel                                  */
#define MXL_KT_ROWBIAS       5   /* add_rowbias (the q + r_r_bias operand of the dRd contraction) */
#define MXL_KT_RELATTN_FUSED 6   /* relattn_bwd_fused_kernel (mxl_relattn_bwd_fused)    */
#define MXL_KT_RELATTN_DQFIN 7   /* relattn_dq_finish_kernel (mxl_relattn_bwd_fused)    */
#define MXL_KT_CHUNK_FWD     8   /* chunk_attn_fwd_kernel (mxl_chunk_attn_fwd, T > one chunk window)              */
#define MXL_KT_CHUNK_BWD_Q   9   /* chunk_attn_bwd_q_kernel (mxl_chunk_attn_bwd)                                   */
#define MXL_KT_CHUNK_BWD_KV  10  /* chunk_attn_bwd_kv_kernel (mxl_chunk_attn_bwd)                                  */
#define MXL_KT_COUNT         11
int mxl_ktime_enable(int on);
/* waits for every recorded event, adds each kernel's elapsed milliseconds into ms_sum[id] and its launch count into
 * launches[id] (HOST arrays of n >= MXL_KT_COUNT entries, overwritten), and forgets the recorded events */
int mxl_ktime_collect(float* ms_sum, int* launches, int n);

#ifdef __cplusplus
}
#endif
#endif /* MUSICXL_H */
