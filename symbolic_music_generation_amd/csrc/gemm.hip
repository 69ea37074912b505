// bf16 MFMA GEMM for the TransfoXL / Reformer linear layers (K2, K3, K5, K6, K7 of SURVEY.md 2.3).
//
//   C[M,N] (+)= alpha * op(A)[M,K] * op(B)[K,N]      fp32 accumulate on v_mfma_f32_16x16x32_bf16
//
// Operand storage (row-major):
//   transA = 0 : A is [M][K] (K contiguous)      transA = 1 : A is [K][M] (M contiguous)
//   transB = 0 : B is [N][K] (K contiguous, i.e. a torch Linear weight)   transB = 1 : B is [K][N]
// so   y  = x W^T      -> (0,0)     replaces F.linear in upstream qkv_net/o_net/r_net/CoreNet/out_layers
//      dx = dy W       -> (0,1)
//      dW = dy^T x     -> (1,1)     (contraction over tokens; split-K + fp32 atomics)
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 fragments of 16x16.
// global -> registers -> LDS (XOR-swizzled images), LDS double-buffered, one barrier per K-tile.
// K-contiguous operands are read with ds_read_b128, K-strided ones with ds_read_b64_tr_b16.
// The MFMA is issued as (B-fragment, A-fragment) so each lane owns 4 consecutive n of one m (8-byte stores).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "musicxl_internal.h"

namespace {

constexpr int BM = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

struct GemmP {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K, lda, ldb, ldc;
    int ksplit;            // K elements per grid.z slice (multiple of BK)
    const float* bias;     // [N] or null
    const bf16_t* aux;     // relu-backward mask source [M][ldaux] or null
    int ldaux;
    float alpha;
    int flags;
    unsigned long long seed; unsigned site; unsigned thresh; float drop_scale;
    int tiles_m, tiles_n;
    // batching: grid.y = batch index by; operand offsets (elements) = (by / bdiv) * s?1 + (by % bdiv) * s?2
    int bdiv;
    long long sA1, sA2, sB1, sB2, sC1, sC2;
    float* colsum;         // [N] f32 or null: += column sums of the epilogue's values (gemm_nt256_kernel, EPI & GEMM_COLSUM_BIT)
    // GEMM_HEADDOT_BIT (gemm_nt256w4_kernel): hd_delta[(b * (N / 64) + h) * hd_T + t] = sum_{e < 64} C[m][64 h + e] * hd_o[m][64 h + e],  m = b * hd_T + t
    const bf16_t* hd_o; int hd_ldo, hd_T; float* hd_delta;
};
constexpr int GEMM_COLSUM_BIT = 0x80;      // internal epilogue bit (not a public flag): see mxl_gemm_bf16_colsum
constexpr int GEMM_HEADDOT_BIT = 0x400;    // internal epilogue bit: see mxl_gemm_bf16_headdot
// MXL_GEMM_SAVE_RELU_MASK (0x100) / MXL_GEMM_RELU_BWD_BITS (0x200): the relu(+dropout) mask as 128 bits per lane and tile, in the
// large-tile kernel's own accumulator layout -- written by the forward epilogue, read back by ONE 16-byte load per lane and tile in
// the backward one (the bf16 activations as mask cost two dependent loads per row block: 16 serialized round trips per tile).

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

__device__ __forceinline__ int swzT(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

template <bool T, int R_>
__device__ __forceinline__ void g2r(const bf16_t* __restrict__ X, int ld, int R, int kend, int r0, int k0, u32x4 (&v)[R_ / 32]) {
    const int t = threadIdx.x;
    constexpr int RC = R_ / 8;   // 16-byte chunks per k-row of a transposed tile
#pragma unroll
    for (int i = 0; i < R_ / 32; i++) {
        const int c = t + i * 256;
        u32x4 z = {0u, 0u, 0u, 0u};
        if (!T) {
            const int row = c >> 3, kc = c & 7;
            const int gr = r0 + row, gk = k0 + kc * 8;
            const bool ok = (gr < R) && (gk < kend);
            v[i] = ok ? *reinterpret_cast<const u32x4*>(X + (size_t)gr * ld + gk) : z;
        } else {
            const int kr = c / RC, rc = c % RC;
            const int gk = k0 + kr, gr = r0 + rc * 8;
            const bool ok = (gk < kend) && (gr < R);
            v[i] = ok ? *reinterpret_cast<const u32x4*>(X + (size_t)gk * ld + gr) : z;
        }
    }
}

template <bool T, int R_>
__device__ __forceinline__ void r2s(char* base, const u32x4 (&v)[R_ / 32]) {
    const int t = threadIdx.x;
    constexpr int RC = R_ / 8;
#pragma unroll
    for (int i = 0; i < R_ / 32; i++) {
        const int c = t + i * 256;
        int off;
        if (!T) {
            const int row = c >> 3, kc = c & 7;
            off = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
        } else {
            const int kr = c / RC, rc = c % RC;
            off = kr * (R_ * 2) + (((rc >> 1) ^ (swzT(kr) & (R_ / 16 - 1))) << 5) + ((rc & 1) << 4);
        }
        *reinterpret_cast<u32x4*>(base + off) = v[i];
    }
}

// fragment for 16 rows [rb, rb+16) of the tile, k-step ks (32 k's): lane l holds row rb+(l&15), k = 8*(l>>4)+j
template <bool T, int R_>
__device__ __forceinline__ bf16x8 ldfrag(const char* base, int rb, int ks) {
    const int l = threadIdx.x & 63;
    const int g = l >> 4, li = l & 15;
    if (!T) {
        const int row = rb + li;
        const int chunk = ks * 4 + g;
        return *reinterpret_cast<const bf16x8*>(base + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
    } else {
        const int q = li >> 2, pp = li & 3;
        const int krow = ks * 32 + 8 * g + q;
        const int col = rb + 4 * pp;
        const int off = krow * (R_ * 2) + (((col >> 4) ^ (swzT(krow) & (R_ / 16 - 1))) << 5) + ((col & 15) << 1);
        const lds_bf16x4* p0 = (const lds_bf16x4*)(base + off);
        const lds_bf16x4* p1 = (const lds_bf16x4*)(base + off + 4 * (R_ * 2));
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)p0);
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)p1);
        bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return r;
    }
}

// One 1x4 output quad (row m, columns n..n+3, n < N): alpha, bias, relu, dropout, relu-backward mask, aux add ...
// `auxq`: the aux values of columns n..n+3 (4 bf16 in two dwords) if the caller has loaded them already
// `biasq`: bias[n..n+3] if the caller holds them in registers
__device__ __forceinline__ void epilogue_vals(const GemmP& p, const int flags, const int m, const int n, const f32x4 a4_, float (&v)[4],
                                              const uint32_t* auxq = nullptr, const float* biasq = nullptr) {
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = a4_[r] * p.alpha;
            const bool full = (n + 3 < p.N);
            if ((flags & MXL_GEMM_BIAS) && biasq) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] += biasq[r];
            } else if (flags & MXL_GEMM_BIAS) {
                if (full) {
                    const float b0 = p.bias[n], b1 = p.bias[n + 1], b2 = p.bias[n + 2], b3 = p.bias[n + 3];
                    v[0] += b0; v[1] += b1; v[2] += b2; v[3] += b3;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) if (n + r < p.N) v[r] += p.bias[n + r];
                }
            }
            if (flags & MXL_GEMM_RELU) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
            }
            if (flags & MXL_GEMM_DROPOUT) {
                // 32-bit element index (host check: M * N <= 2^32).  WITHOUT relu: the same decisions as dropout_keep's 64-bit form,
                // which the stand-alone dropout / LayerNorm kernels use when they regenerate a mask.  WITH relu: a mask of its own
                // (below; include/musicxl.h, MXL_GEMM_DROPOUT) that nothing regenerates
                // dropout_keep32(seed, site, i0 + r) with the index spread of the quad's four elements formed from ONE multiply:
                // (i0 + r) * C = i0 * C + r * C (mod 2^32) -- the same masks, 9 quarter-rate integer multiplies per quad instead of 12
                const uint32_t i0 = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
                const uint32_t mix = mxl_hash32((uint32_t)p.seed ^ (p.site * 0x9E3779B9U)) + (uint32_t)(p.seed >> 32);
                const uint32_t h0 = i0 * 0x9E3779B1U;
                if (flags & MXL_GEMM_RELU) {
                    // relu + dropout = the FFN's hidden activations: nobody regenerates this mask (the backward reads it from the saved
                    // bits, or from the zeros of the activations themselves), so it need not be dropout_keep's -- one hash per PAIR of
                    // elements of the quad, a 16-bit decision each (drop probability quantised to 2^-16): 5 quarter-rate multiplies
                    // per quad instead of 9
                    const uint32_t t16 = p.thresh >> 16;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const uint32_t h = mxl_hash32((h0 + (uint32_t)(2 * k) * 0x9E3779B1U) ^ mix);
                        v[2 * k] = ((h & 0xffffu) >= t16) ? v[2 * k] * p.drop_scale : 0.f;
                        v[2 * k + 1] = ((h >> 16) >= t16) ? v[2 * k + 1] * p.drop_scale : 0.f;
                    }
                } else
#pragma unroll
                for (int r = 0; r < 4; r++)
                    v[r] = (mxl_hash32((h0 + (uint32_t)r * 0x9E3779B1U) ^ mix) >= p.thresh) ? v[r] * p.drop_scale : 0.f;
            }
            if ((flags & MXL_GEMM_RELU_BWD) && auxq) {
                v[0] = bf2f((bf16_t)(auxq[0] & 0xffffu)) > 0.f ? v[0] : 0.f;
                v[1] = bf2f((bf16_t)(auxq[0] >> 16)) > 0.f ? v[1] : 0.f;
                v[2] = bf2f((bf16_t)(auxq[1] & 0xffffu)) > 0.f ? v[2] : 0.f;
                v[3] = bf2f((bf16_t)(auxq[1] >> 16)) > 0.f ? v[3] : 0.f;
            } else if (flags & MXL_GEMM_RELU_BWD) {
                if (full && ((p.ldaux & 3) == 0)) {
                    const bf16x4 a4 = *reinterpret_cast<const bf16x4*>(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = bf2f((bf16_t)a4[r]) > 0.f ? v[r] : 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (n + r < p.N) {
                            const float a = bf2f(p.aux[(size_t)m * p.ldaux + n + r]);
                            v[r] = a > 0.f ? v[r] : 0.f;
                        }
                }
            }
            if ((flags & MXL_GEMM_ADD_AUX) && auxq) {
                v[0] += bf2f((bf16_t)(auxq[0] & 0xffffu)); v[1] += bf2f((bf16_t)(auxq[0] >> 16));
                v[2] += bf2f((bf16_t)(auxq[1] & 0xffffu)); v[3] += bf2f((bf16_t)(auxq[1] >> 16));
            } else if (flags & MXL_GEMM_ADD_AUX) {
                if (full && ((p.ldaux & 3) == 0)) {
                    const bf16x4 a4 = *reinterpret_cast<const bf16x4*>(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] += bf2f((bf16_t)a4[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (n + r < p.N) v[r] += bf2f(p.aux[(size_t)m * p.ldaux + n + r]);
                }
            }
}
// ... and its store
__device__ __forceinline__ void epilogue_store(const GemmP& p, const int flags, const int m, const int n, const float (&v)[4]) {
            const bool full = (n + 3 < p.N);
            if (flags & MXL_GEMM_OUT_F32_ATOMIC) {
                float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
#pragma unroll
                for (int r = 0; r < 4; r++) if (n + r < p.N) atomicAdd(c + r, v[r]);
            } else if (flags & MXL_GEMM_OUT_F32) {
                float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
                if (full && ((p.ldc & 3) == 0)) {
                    *reinterpret_cast<f32x4*>(c) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) if (n + r < p.N) c[r] = v[r];
                }
            } else {
                bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n;
                if (full && ((p.ldc & 3) == 0)) {
                    u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                    if (flags & (1 << 30)) __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(c));
                    else *reinterpret_cast<u32x2*>(c) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) if (n + r < p.N) c[r] = f2bf(v[r]);
                }
            }
}
__device__ __forceinline__ void epilogue_quad(const GemmP& p, const int flags, const int m, const int n, const f32x4 a4_) {
    float v[4];
    epilogue_vals(p, flags, m, n, a4_, v);
    epilogue_store(p, flags, m, n, v);
}
// Two quads of one row in neighbouring 16-column blocks (columns n and n + 16, both inside the matrix), bf16 output with 16-byte
// aligned rows: lanes l and l ^ 16 trade one packed quad each (v_permlane16_swap) so that every lane holds 8 consecutive columns
// and the pair leaves as ONE 16-byte store per lane -- a wave instruction then writes 16 rows x 64 contiguous bytes instead of
// 16 x 32.  The store path of a CU handles ~30 ns per wave-instruction made of 32-byte row segments but ~11 ns per instruction
// made of 64-byte segments, and half as many instructions (scripts/ubench/stores.hip: a 256 x 256 bf16 tile leaves a CU in 8.1 us
// the first way, 1.4 us the second -- with no other CU active).
// `axv` (have_ax): the lane's 16 bytes of aux at the store address (pair_aux_load below), loaded by the caller ahead of time; `bq0` / `bq1`:
// the bias quads of the two blocks, in registers
__device__ __forceinline__ u32x4 pair_aux_load(const GemmP& p, const int m, const int n, const int l) {
    const int acol = ((l >> 4) & 1) ? n + 16 - 4 : n;
    return *reinterpret_cast<const u32x4*>(p.aux + (size_t)m * p.ldaux + acol);
}
__device__ __forceinline__ void epilogue_pair_bf16(const GemmP& p, const int flags, const int m, const int n, const int l,
                                                   const f32x4 q0, const f32x4 q1, float* cs0 = nullptr, float* cs1 = nullptr,
                                                   const bool have_ax = false, const u32x4 axv = u32x4{0u, 0u, 0u, 0u},
                                                   const float* bq0 = nullptr, const float* bq1 = nullptr,
                                                   const int mask_mode = 0, uint32_t* mbits = nullptr, const int mshift = 0) {
    float v0[4], v1[4];
    if (mask_mode == 2) {        // relu-backward from the saved bits: bits mshift .. mshift + 7 of *mbits belong to this pair
        const uint32_t b = *mbits >> mshift;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v0[r] = (b >> r) & 1u ? q0[r] * p.alpha : 0.f;
            v1[r] = (b >> (4 + r)) & 1u ? q1[r] * p.alpha : 0.f;
        }
    } else
    if ((flags & (MXL_GEMM_RELU_BWD | MXL_GEMM_ADD_AUX)) && (p.ldaux & 7) == 0 && (reinterpret_cast<uintptr_t>(p.aux) & 15) == 0) {
        // the aux operand the same way round: one 16-byte load per lane at the address the store below uses (8 consecutive
        // columns of one block), then lanes l and l ^ 16 trade halves so that each holds the aux of its own two quads
        const u32x4 ax = have_ax ? axv : pair_aux_load(p, m, n, l);
        const auto s0 = __builtin_amdgcn_permlane16_swap(ax[0], ax[2], false, false);    // (lo, hi): odd rows' lo <-> even rows' hi
        const auto s1 = __builtin_amdgcn_permlane16_swap(ax[1], ax[3], false, false);
        const uint32_t aq0[2] = {s0[0], s1[0]}, aq1[2] = {s0[1], s1[1]};                  // block 0 quad, block 1 quad of this lane
        epilogue_vals(p, flags, m, n, q0, v0, aq0, bq0);
        epilogue_vals(p, flags, m, n + 16, q1, v1, aq1, bq1);
    } else {
        epilogue_vals(p, flags, m, n, q0, v0, nullptr, bq0);
        epilogue_vals(p, flags, m, n + 16, q1, v1, nullptr, bq1);
    }
    if (mask_mode == 1) {        // save: which of the eight outputs are positive (after relu and dropout)
        uint32_t b = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) b |= (v0[r] > 0.f ? 1u << r : 0u) | (v1[r] > 0.f ? 16u << r : 0u);
        *mbits |= b << mshift;
    }
    if (cs0) {
#pragma unroll
        for (int r = 0; r < 4; r++) { cs0[r] += v0[r]; cs1[r] += v1[r]; }
    }
    unsigned a0 = pack2bf(v0[0], v0[1]), a1 = pack2bf(v0[2], v0[3]), b0 = pack2bf(v1[0], v1[1]), b1 = pack2bf(v1[2], v1[3]);
    // even 16-lane rows keep block 0 and receive the neighbour's block-0 quad; odd rows keep block 1 and receive the neighbour's
    const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    const int col = ((l >> 4) & 1) ? n + 16 - 4 : n;
    bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + col;
    *reinterpret_cast<u32x4*>(c) = u32x4{r0[0], r1[0], r0[1], r1[1]};
}

// The same for two quads of one 16-column block in neighbouring row blocks (rows m and m + 16): even 16-lane rows end up with 8
// consecutive columns of row m, odd ones with 8 consecutive columns of row m + 16 (32-byte row segments, but 16 bytes per lane
// and half the instructions).
__device__ __forceinline__ void epilogue_rowpair_bf16(const GemmP& p, const int flags, const int m, const int n, const int l,
                                                      const f32x4 q0, const f32x4 q1) {
    float v0[4], v1[4];
    epilogue_vals(p, flags, m, n, q0, v0);
    epilogue_vals(p, flags, m + 16, n, q1, v1);
    unsigned a0 = pack2bf(v0[0], v0[1]), a1 = pack2bf(v0[2], v0[3]), b0 = pack2bf(v1[0], v1[1]), b1 = pack2bf(v1[2], v1[3]);
    const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    const bool odd = (l >> 4) & 1;
    bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (size_t)(odd ? m + 16 : m) * p.ldc + (odd ? n - 4 : n);
    *reinterpret_cast<u32x4*>(c) = u32x4{r0[0], r1[0], r0[1], r1[1]};
}

// SWAP (the split-K weight-gradient form: fp32 atomics, no other epilogue work): the MFMAs are issued (A, B) instead of (B, A), so a
// lane owns one COLUMN and 4 consecutive rows of a fragment, and each of the 4 atomic instructions of a fragment covers 4 rows x
// 64 contiguous bytes.  In the (B, A) layout an atomic instruction is 16 rows x 4 separate dwords: measured on 128 x 128 fp32
// tiles from 512 colliding workgroups (scripts/ubench/atomic_tiles.hip) that form sustains 0.32 TB/s chip-wide, this one 1.35.
template <bool AT, bool BT, int BN_, bool SWAP = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmP p) {
    constexpr int BN = BN_;              // 128, or 64 for skinny-N problems (the per-head dRd contraction: N = d_head)
    constexpr int NF = BN / 32;          // 16-wide n-fragments per wave
    constexpr int B_BYTES = BN * BK * 2;
    constexpr int STAGE = TILE_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware remap: blocks b and b+8 share an XCD (L2); give each XCD a contiguous run of tiles so
    // neighbouring tiles (same A row panel) hit the same L2.  Bijective for any grid size.
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    if (gridDim.y > 1) {
        const long long b1 = blockIdx.y / p.bdiv, b2 = blockIdx.y % p.bdiv;
        p.A += b1 * p.sA1 + b2 * p.sA2;
        p.B += b1 * p.sB1 + b2 * p.sB2;
        const long long co = b1 * p.sC1 + b2 * p.sC2;
        if (p.flags & (MXL_GEMM_OUT_F32 | MXL_GEMM_OUT_F32_ATOMIC)) p.C = reinterpret_cast<float*>(p.C) + co;
        else p.C = reinterpret_cast<bf16_t*>(p.C) + co;
    }
    const int tm = bid / p.tiles_n, tn = bid % p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * p.ksplit;
    const int kend = min(p.K, kbeg + p.ksplit);
    const int nk = (kend - kbeg + BK - 1) / BK;

    const int wid = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int wr = wid >> 1, wc = wid & 1;

    f32x4 acc[4][NF];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NF; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[4], rb[BN / 32];
    g2r<AT, 128>(p.A, p.lda, p.M, kend, m0, kbeg, ra);
    g2r<BT, BN>(p.B, p.ldb, p.N, kend, n0, kbeg, rb);
    r2s<AT, 128>(smem, ra);
    r2s<BT, BN>(smem + TILE_BYTES, rb);
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; kt++) {
        const bool more = (kt + 1 < nk);
        if (more) {
            g2r<AT, 128>(p.A, p.lda, p.M, kend, m0, kbeg + (kt + 1) * BK, ra);
            g2r<BT, BN>(p.B, p.ldb, p.N, kend, n0, kbeg + (kt + 1) * BK, rb);
        }
        const char* sa = smem + cur * STAGE;
        const char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 fa[4], fb[NF];
#pragma unroll
            for (int i = 0; i < 4; i++) fa[i] = ldfrag<AT, 128>(sa, wr * 64 + i * 16, ks);
#pragma unroll
            for (int j = 0; j < NF; j++) fb[j] = ldfrag<BT, BN>(sb, wc * (BN / 2) + j * 16, ks);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < NF; j++)
                    acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, fa[i]),
                                                                               __builtin_bit_cast(mfma_bf16x8, fb[j]), acc[i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, fb[j]),
                                                                               __builtin_bit_cast(mfma_bf16x8, fa[i]), acc[i][j], 0, 0, 0);
        }
        if (more) {
            char* da = smem + (cur ^ 1) * STAGE;
            r2s<AT, 128>(da, ra);
            r2s<BT, BN>(da + TILE_BYTES, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    if (SWAP) {   // acc[i][j][r]: m = m0 + wr*64 + i*16 + (l>>4)*4 + r, n = n0 + wc*(BN/2) + j*16 + (l&15)
        float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < NF; j++) {
                const int n = n0 + wc * (BN / 2) + j * 16 + (l & 15);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int m = m0 + wr * 64 + i * 16 + (l >> 4) * 4 + r;
                    if (m < p.M && n < p.N) atomicAdd(C + (size_t)m * p.ldc + n, acc[i][j][r] * p.alpha);
                }
            }
        return;
    }
    // epilogue.  acc[i][j][r]: m = m0 + wr*64 + i*16 + (l&15), n = n0 + wc*(BN/2) + j*16 + (l>>4)*4 + r
    const int flags = p.flags;
    const int mrow_l = l & 15, nq = (l >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + wr * 64 + i * 16 + mrow_l;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NF; j++) {
            const int n = n0 + wc * (BN / 2) + j * 16 + nq;
            if (n >= p.N) continue;
            epilogue_quad(p, flags, m, n, acc[i][j]);
        }
    }
}

// =====================================================================================================================
// Large-tile persistent kernel for the K-contiguous ("NT") form y = x W^T with M >= 256: the forward linears and, through
// the transposed weight copies the engine keeps, the dX products.
//
//   tile 256 x BN x 32 (BN = 256 or 192), 512 threads = 8 waves (2 x 4), wave tile 128 x BN/4 = 8 x NFN fragments: 12 (11)
//   KB of fragment reads per 32 (24) MFMAs -- the 128^2 kernel above reads 16 KB per 32 and sits on the LDS-bandwidth roof.
//   HBM/L2 -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write pass) into a FOUR-stage ring of 32 KB
//   stages; the XOR swizzle of the LDS image is applied on the per-lane SOURCE address (the LDS side of the DMA is
//   lane-linear).  Loads run three K-steps ahead and stay in flight across the one raw s_barrier per K-step: counted
//   s_waitcnt vmcnt(4), never 0 inside a tile.  Fragments are register double-buffered: the ds_reads of step g+1 are
//   issued before the MFMAs of step g.
//   One workgroup per CU walks tiles bid, bid + G, ...; the K-step sequence is flattened across tiles, so the first
//   stages of the next tile are already in flight during a tile's epilogue.
//   Measured (scripts/perf_gemm_table.py, perf_gemm_probe.py, C3 shapes, random operands): 600-880 TF/s at K = 768..3072,
//   1.1 PF/s at K = 8192; the MFMA-only loop tops out at 1.42 PF/s and the DMA + fragment-read loop alone at 1.09 PF/s
//   (L2 -> LDS traffic), so the remaining gap to hipBLASLt (0.8-1.2 / 1.55 PF/s) is the per-tile epilogue (all CUs store
//   at once: ~10 us per round at the HBM write rate, not hidden) and operand traffic per flop.  A two-workgroup-per-CU
//   256 x 128 variant and an LDS-transposed full-line epilogue were tried and measured no better.
//
//   step g:   glds(g+3) -> stage (g+3)&3     (last read by the fragment loads of step g-1, issued in step g-2)
//             ds_read fragments(g+1)         (landed: waited at the end of step g-1, published by that barrier)
//             32 MFMAs on fragments(g)
//             s_waitcnt vmcnt(4)  [step g+2 landed; step g+3 may still fly]      s_barrier
//   last step of a tile: vmcnt(0) instead (everything issued has landed), barrier, epilogue; the first step after an
//   epilogue needs no vmcnt wait (its requirement was met by that drain) -- so the epilogue's stores, which share the
//   counter, are not waited for until one step later.
// =====================================================================================================================
constexpr int G2_OP_BYTES = 256 * 32 * 2;      // one operand, one stage: [256 rows][32 k] bf16 = 16 KiB
constexpr int G2_STAGE = 2 * G2_OP_BYTES;
constexpr int G2_SMEM = 4 * G2_STAGE;          // 128 KiB

// Slot swizzle of the [rows][32 k] bf16 images (64-byte rows, four 16-byte slots): row r keeps k-chunk c in slot c ^ f(r/4 mod 4)
// with f = {0, 2, 3, 1}.  ds_read_b128 is serviced in four 16-lane groups -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
// same + 32 -- so with fragment lanes = (row l & 15, k-group l >> 4) one group holds rows 0-3 and 12-15 of one k-group and rows
// 4-11 of its neighbour g ^ 1: the four rows that share r mod 4 (same 64-byte quarter of the 256-byte bank row) must land in four
// different slots, i.e. {f0, f1 ^ 1, f2 ^ 1, f3} distinct.  The identity (c ^ (r/4 mod 4)) gives {0, 0, 3, 3}: a 2-way conflict on
// every fragment read, 8 LDS cycles instead of 4 (measured: +2-3 % on the C3 shapes).
__device__ __forceinline__ int g2_swz(int rb) { return ((((rb ^ (rb >> 1)) & 1) << 1) | (rb >> 1)) & 3; }

typedef __attribute__((address_space(1))) const void* g2_gptr;
typedef __attribute__((address_space(3))) void* g2_lptr;

// EPI >= 0: the epilogue flags as a compile-time constant (the common combinations; EPI == 0 also means alpha == 1).  With the
// flags only known at run time the general epilogue's pointers, dropout constants and 64-bit indices all stay live next to the 128
// accumulators and spill (71 values at BN = 256).
#ifndef NT_ABL
#define NT_ABL 0                // timing ablations (garbage results): 1 no DMA in the loop, 2 no fragment reads, 4 no barrier
#endif
template <int NFN, int EPI = -1>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(GemmP p) {
    constexpr int BN = 64 * NFN;
    constexpr bool PF_ACROSS = !(EPI >= 0 && (EPI & (MXL_GEMM_DROPOUT | 0x100)));     // fragment prefetch across tile boundaries (not beside the register-hungry dropout / mask-saving epilogues)
    constexpr int WN = 4, WM = 2;                           // wave grid: 2 x 4 waves of 128 x BN/4
    constexpr int FM = 16 / WM, FN = BN / 16 / WN;          // 16 x 16 fragments per wave: 8 rows x NFN columns
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int nwg = p.tiles_m * p.tiles_n;
    const int G = gridDim.x;
    int bid = blockIdx.x;
    {   // XCD-aware remap of the launch slots: slots b, b+8, ... (one XCD) walk neighbouring tiles (same A row panels)
        const int q = G >> 3, r = G & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nk = p.K >> 5;                                  // even (host check)
    const int my_tiles = (nwg - bid + G - 1) / G;
    const int S = my_tiles * nk;                              // flattened K-steps of this workgroup

    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int wr = wid / WN, wc = wid % WN;

    // ---- issue side.  DMA instruction i of wave w fills LDS rows (8i + w) * 16 .. +15 of an operand image (16 rows x 64 B,
    // lane -> row l >> 2, 16-byte slot l & 3); the slot holds k-chunk (l & 3) ^ g2_swz((row >> 2) & 3) = (l & 3) ^ g2_swz(l >> 4).
    // With BN = 192 the B image has 12 row groups: waves 4..7 issue three DMAs per step instead of four, and count their
    // waits accordingly (`per_step`, wave-uniform).
    // LDS-DMA as buffer_load ... lds, not global_load_lds: hipcc's s_waitcnt pass books a global_load_lds as a FLAT access that
    // touches both memory and LDS ("pending flat"), and while one is in flight EVERY wait it places for a register loaded from
    // memory is s_waitcnt vmcnt(0).  The epilogue's bias / aux loads then each drained the whole queue -- the next tile's
    // prefetched stages and, because stores count in vmcnt, every store issued so far: 258 (bias) / 358 (aux) full drains per tile
    // in the ISA.  The MUBUF form is booked as an ordinary vector-memory access and those waits come out counted.
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, -1, 0x00020000);
    unsigned ga[2], gb[2];          // element offsets of this lane's source rows (+ k-chunk): 32 bits (host check)
    int gi = 0, i_t = 0, i_tile = bid;
    auto set_ptrs = [&](int tile) {
        const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
        const int kc = ((l & 3) ^ g2_swz(l >> 4)) * 8;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = (8 * i + wid) * 16 + (l >> 2);
            ga[i] = (unsigned)min(tm * 256 + row, p.M - 1) * (unsigned)p.lda + kc;   // rows past the edge re-read the last row
            gb[i] = (unsigned)min(tn * BN + row, p.N - 1) * (unsigned)p.ldb + kc;
        }
    };
    set_ptrs(i_tile);
    auto issue_next = [&]() {
        char* st = smem + (gi & 3) * G2_STAGE;
        const int k0 = i_t << 5;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (g2_lptr)(st + (8 * i + wid) * 1024), 16, (int)((ga[i] + k0) * 2u), 0, 0, 0);
            if (NFN == 4 || i == 0 || wid < 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (g2_lptr)(st + G2_OP_BYTES + (8 * i + wid) * 1024), 16,
                                                         (int)((gb[i] + k0) * 2u), 0, 0, 0);
        }
        gi++;
        if (++i_t == nk) {
            i_t = 0;
            i_tile += G;
            if (i_tile < nwg) set_ptrs(i_tile);
        }
    };
    // ---- fragment addresses: row = base + (l & 15), k-group g = l >> 4 sits in slot g ^ ((row >> 2) & 3)
    const int fsw = ((l >> 4) ^ g2_swz((l >> 2) & 3)) << 4;
    const int a_off = (wr * (FM * 16) + (l & 15)) * 64 + fsw;
    const int b_off = G2_OP_BYTES + (wc * (FN * 16) + (l & 15)) * 64 + fsw;
    auto frags = [&](int g, bf16x8 (&fa)[FM], bf16x8 (&fb)[FN]) {
        const char* st = smem + (g & 3) * G2_STAGE;
#pragma unroll
        for (int j = 0; j < FN; j++) fb[j] = *reinterpret_cast<const bf16x8*>(st + b_off + j * 1024);
#pragma unroll
        for (int i = 0; i < FM; i++) fa[i] = *reinterpret_cast<const bf16x8*>(st + a_off + i * 1024);
    };

    f32x4 acc[FM][FN];
    // prologue: three steps in flight, steps 0 and 1 landed before the first barrier
    issue_next();
    if (S > 1) issue_next();
    if (S > 2) issue_next();
    const bool three = (NFN == 3) && wid >= 4;          // this wave's DMAs per step: 3 instead of 4
    if (S > 2) { if (three) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    bf16x8 fa0[FM], fb0[FN], fa1[FM], fb1[FN];
    frags(0, fa0, fb0);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): nothing outstanding at loop entry, the loop body needs no wait before its MFMAs

    int g = 0;                              // flattened step being computed
    auto step = [&](bool first_of_later_tile, bool last_of_tile, bf16x8 (&fa)[FM], bf16x8 (&fb)[FN], bf16x8 (&na)[FM],
                    bf16x8 (&nb)[FN]) {
        const bool issued = gi < S;
        // The two waves of a SIMD (w and w + 4) take the step's two halves in opposite order: waves 0-3 issue their DMAs and read the
        // next fragments, then compute; waves 4-7 compute first (their fragments were read at the end of the previous step) and issue
        // and read afterwards -- one wave's LDS-DMA issue (60-185 cycles a piece) and ds_reads run beside the other's MFMAs instead
        // of both waves doing the same thing at the same time.  ffn1 forward 694 -> 650 us, ffn2 dX 699 -> 647, the K = 768 shapes
        // -2 ... -7 %, K >= 2304 -1 % (profiles/r04_gemm_wave_phases.txt).
        const bool late = wid >= 4;
        if (!late) {
            if (issued && !(NT_ABL & 1)) issue_next();
            // (with a bias epilogue not across a tile boundary: the 48 fragment registers are what the preloaded bias values live in;
            // the next tile's first fragments are then read after the epilogue)
            if (g + 1 < S && !(last_of_tile && !PF_ACROSS) && !(NT_ABL & 2)) frags(g + 1, na, nb);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < FM; i++)
#pragma unroll
            for (int j = 0; j < FN; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, fb[j]),
                                                                    __builtin_bit_cast(mfma_bf16x8, fa[i]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (late) {
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < S && !(last_of_tile && !PF_ACROSS) && !(NT_ABL & 2)) frags(g + 1, na, nb);
            if (issued && !(NT_ABL & 1)) issue_next();
        }
        // (the fragment reads of step g+1 need not finish before this barrier: their stage is not re-filled before the
        // barrier of step g+1, and the MFMAs of step g+1 wait for them anyway)
        if (last_of_tile || !issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (!first_of_later_tile) {
            if (three) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        if (!(NT_ABL & 4)) __builtin_amdgcn_s_barrier();
        if (NT_ABL & 1) gi++;
        g++;
    };

    // compile-time BIAS variants (launched with alpha == 1 only): the bias is the accumulators' START VALUE -- its loads have the
    // whole K loop to arrive and the epilogue carries neither the 16 bias registers nor their waits (loaded in the epilogue, every
    // bias load was the youngest memory operation at its wait: a drain of all the stores before it, 258 per tile in the ISA)
    constexpr bool BIAS_INIT = EPI >= 0 && (EPI & MXL_GEMM_BIAS);
    const int flags = EPI >= 0 ? (BIAS_INIT ? EPI & ~MXL_GEMM_BIAS : EPI) : p.flags;
    if (EPI == 0 || BIAS_INIT) p.alpha = 1.f;
#pragma unroll 1
    for (int tile = bid; tile < nwg; tile += G) {
        if (BIAS_INIT) {
            const int nb0 = (tile % p.tiles_n) * BN + wc * (FN * 16) + (l >> 4) * 4;
#pragma unroll
            for (int j = 0; j < FN; j++) {
                f32x4 b4;
                const int n = nb0 + j * 16;
                if (n + 3 < p.N) b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
                else
#pragma unroll
                    for (int r = 0; r < 4; r++) b4[r] = n + r < p.N ? p.bias[n + r] : 0.f;
#pragma unroll
                for (int i = 0; i < FM; i++) acc[i][j] = b4;
            }
        } else {
#pragma unroll
            for (int i = 0; i < FM; i++)
#pragma unroll
                for (int j = 0; j < FN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int t = 0; t < nk; t += 2) {
            step(t == 0 && tile != bid, false, fa0, fb0, fa1, fb1);
            step(false, t + 2 >= nk, fa1, fb1, fa0, fb0);
        }
        // epilogue.  acc[i][j][r]: m = m0 + wr*FM*16 + i*16 + (l&15), n = n0 + wc*FN*16 + j*16 + (l>>4)*4 + r
        const int m0 = (tile / p.tiles_n) * 256, n0 = (tile % p.tiles_n) * BN;
        const bool pairs = !(flags & MXL_GEMM_OUT_F32) && n0 + BN <= p.N && m0 + 256 <= p.M && (p.ldc & 7) == 0 &&
                           (reinterpret_cast<uintptr_t>(p.C) & 15) == 0;                 // workgroup-uniform: interior tile
        if (pairs) {
            // 16-byte stores: column blocks (0,1), (2,3) of a row block pair up; with three column blocks the third one pairs up
            // across row blocks (i, i+1)
            constexpr bool CS = EPI >= 0 && (EPI & GEMM_COLSUM_BIT) && !(FN & 1);      // (launched only when every tile is interior)
            float cs[CS ? FN : 1][4];
            if (CS) {
#pragma unroll
                for (int j = 0; j < FN; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) cs[j][r] = 0.f;
            }
            // the aux rows one row block ahead, from the third row block on (the first two are loaded where they are used: with all
            // 128 accumulators still live there is no room for the 8 extra registers, and the spills' own waits cost more than the
            // drains they replace: 832 -> 1089 us at the C3 shape)
            // -- and even so hipcc spills (46-124 scratch operations in the aux instantiations).  Off: the code stays for a compiler that
            // places it.
            constexpr bool HA = false && EPI >= 0 && (EPI & (MXL_GEMM_RELU_BWD | MXL_GEMM_ADD_AUX)) && !(FN & 1);
            constexpr int MM = (EPI >= 0 && (EPI & 0x100)) ? 1 : (EPI >= 0 && (EPI & 0x200)) ? 2 : 0;      // relu-mask bits: save / use
            uint32_t mb[4] = {0u, 0u, 0u, 0u};
            if (MM == 2) {
                const u32x4 t4 = *(reinterpret_cast<const u32x4*>(p.aux) + ((size_t)tile * 8 + wid) * 64 + l);
                mb[0] = t4[0]; mb[1] = t4[1]; mb[2] = t4[2]; mb[3] = t4[3];
            }
            const bool aux_ok = HA && (p.ldaux & 7) == 0 && (reinterpret_cast<uintptr_t>(p.aux) & 15) == 0;
            u32x4 axn[HA ? FN / 2 : 1];
#pragma unroll
            for (int i = 0; i < FM; i++) {
                const int m = m0 + wr * (FM * 16) + i * 16 + (l & 15);
                u32x4 axc[HA ? FN / 2 : 1];
                if (aux_ok) {
#pragma unroll
                    for (int j = 0; j + 1 < FN; j += 2) {
                        axc[j / 2] = i >= 2 ? axn[j / 2] : pair_aux_load(p, m, n0 + wc * (FN * 16) + j * 16 + (l >> 4) * 4, l);
                        if (i >= 1 && i + 1 < FM) axn[j / 2] = pair_aux_load(p, m + 16, n0 + wc * (FN * 16) + j * 16 + (l >> 4) * 4, l);
                    }
                }
#pragma unroll
                for (int j = 0; j + 1 < FN; j += 2)
                    epilogue_pair_bf16(p, flags & ~(GEMM_COLSUM_BIT | 0x300), m, n0 + wc * (FN * 16) + j * 16 + (l >> 4) * 4, l, acc[i][j], acc[i][j + 1],
                                       CS ? cs[CS ? j : 0] : nullptr, CS ? cs[CS ? j + 1 : 0] : nullptr,
                                       aux_ok, axc[HA ? j / 2 : 0], nullptr, nullptr,
                                       MM, &mb[(i * FN / 2 + j / 2) / 4], 8 * ((i * FN / 2 + j / 2) & 3));
                if ((FN & 1) && !(i & 1))
                    epilogue_rowpair_bf16(p, flags, m, n0 + wc * (FN * 16) + (FN - 1) * 16 + (l >> 4) * 4, l, acc[i][FN - 1],
                                          acc[(i + 1) % FM][FN - 1]);
                if (MM == 1) __builtin_amdgcn_sched_barrier(0);      // row blocks in order: the dropout epilogue is at the register limit
            }
            if (MM == 1)       // (address formed here, not held across the row blocks: the epilogue is at the register limit)
                *(reinterpret_cast<u32x4*>(const_cast<bf16_t*>(p.aux)) + ((size_t)tile * 8 + wid) * 64 + l) = u32x4{mb[0], mb[1], mb[2], mb[3]};
            if (CS) {
                // the lane's 4 x FN column sums over its 8 row blocks -> over the 16 lanes (rows) that share the columns -> one
                // atomic per column from each of the two wave rows
#pragma unroll
                for (int j = 0; j < FN; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        // all-reduce over the 16-lane row by rotations (v_add_f32 with a DPP row_ror operand: no LDS traffic)
                        float v = cs[j][r];
                        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
                        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
                        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
                        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
                        cs[j][r] = v;
                    }
                if ((l & 15) == 0) {
#pragma unroll
                    for (int j = 0; j < FN; j++)
#pragma unroll
                        for (int r = 0; r < 4; r++) atomicAdd(p.colsum + n0 + wc * (FN * 16) + j * 16 + (l >> 4) * 4 + r, cs[j][r]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < FM; i++) {
                const int m = m0 + wr * (FM * 16) + i * 16 + (l & 15);
                if (m >= p.M) continue;
#pragma unroll
                for (int j = 0; j < FN; j++) {
                    const int n = n0 + wc * (FN * 16) + j * 16 + (l >> 4) * 4;
                    if (n >= p.N) continue;
                    epilogue_quad(p, flags, m, n, acc[i][j]);
                }
            }
        }
        if (!PF_ACROSS && g < S) {          // first fragments of the next tile (their stage landed with the last step's drain)
            frags(g, fa0, fb0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
    }
}


// =====================================================================================================================
// The same machine with FOUR waves of 128 x 128 (round 6): a wave's fragment reads per 32-deep K-step are (128 + 128) rows x 64 B =
// 16 KB for 64 MFMAs instead of (128 + 64) x 64 B = 12 KB for 32 -- 64 KB per workgroup and step instead of 96, against a matrix
// pipe that needs 1024 cycles for the step and an LDS that moves 128 B per cycle (96 KB of reads + the 32 KB the DMA writes = the
// same 1024: the eight-wave kernel sits on both roofs at once).
// One wave per SIMD, 256 accumulator registers: they live in a[0:255] for the whole kernel because every MFMA is an asm statement
// with a "+a" operand -- hipcc's own allocation of this loop moved accumulators between the two register files ~100 times per
// K-step and spilled; with the constraint it has nothing to decide.  Fragment reads are asm too (two
// register sets, the next step's reads spread over the second half of this step's MFMAs), so their wait is this kernel's own:
// lgkmcnt(0) at the end of a step.
//   step g:   MFMAs 0..31 of step g with the 8 LDS-DMA pieces of step g+3 between them (stage (g+3)&3: last read in step g-2)
//             s_waitcnt vmcnt(16) [g+1 has landed; g+2, g+3 may fly]    s_barrier
//             MFMAs 32..63 with the 16 fragment reads of step g+1 between them        s_waitcnt lgkmcnt(0)
//   What the loop costs (ablation builds, K = 8192, TF/s: profiles/r06_gemm_w4_notes.txt): 1349 as it stands; without the pieces
//   1894, without the fragment reads 1517, without the barrier 1386, the MFMAs alone 2219.  A piece costs its SIMD ~60 cycles of
//   issue wherever it is placed and whoever issues it (the eight-wave kernel, 1234: without its pieces 1619 -- the partner wave
//   does not fill the hole), because the vector-memory path takes 16 B per cycle and SIMD: a 256 x 256 x 32 step is 8 KB per SIMD
//   = 512 cycles beside 1024 cycles of MFMA.  Plain loads into registers are no cheaper (1097 against 1227 on one box).
//   last step of a tile: + vmcnt(0) (this wave's pieces of g+2, g+3 landed; the next mid-step barrier publishes them), epilogue;
//   the two steps after an epilogue place no vmcnt wait (what they need was drained), so the epilogue's stores -- which share
//   the counter -- are first waited for two and a half steps later.
// =====================================================================================================================
// Accumulator (i, j) is pinned: a[4 (8 i + j) : + 3] in every asm statement that touches it (the constraint strings spell the physical
// registers: csrc/gemm_w4_gen.inc, generated by scripts/gen_gemm_w4.py).  With "+a" alone the loop compiled clean, but hipcc then spilled
// ~80 accumulators to scratch in front of the epilogue and reloaded them into v[] from there -- every reload a vmcnt(0) between the
// tile's stores; with the registers named, the epilogue reads them with v_accvgpr_read and nothing moves.
#define W4S_(x) #x
#define W4S(x) W4S_(x)
#define W4_AREG(lo, hi) "{a[" W4S(lo) ":" W4S(hi) "]}"
#define W4_MFMA(i, j, lo, hi)                                                                                                     \
    do {                                                                                                                          \
        if (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=" W4_AREG(lo, hi)(acc[i][j]) : "v"(fb[j]), "v"(fa[i])); \
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+" W4_AREG(lo, hi)(acc[i][j]) : "v"(fb[j]), "v"(fa[i]));     \
    } while (0)
#define W4_ACC_READ(i, j, r0, r1, r2, r3, lo, hi, q_)                                                                             \
    do {                                                                                                                          \
        float x0_, x1_, x2_, x3_;                                                                                                 \
        asm volatile("v_accvgpr_read_b32 %0, a" W4S(r0) "\n\tv_accvgpr_read_b32 %1, a" W4S(r1) "\n\tv_accvgpr_read_b32 %2, a" W4S(r2)  \
                     "\n\tv_accvgpr_read_b32 %3, a" W4S(r3)                                                                       \
                     : "=&v"(x0_), "=&v"(x1_), "=&v"(x2_), "=&v"(x3_) : W4_AREG(lo, hi)(acc[i][j]));                               \
        q_ = f32x4{x0_, x1_, x2_, x3_};                                                                                           \
    } while (0)
#include "gemm_w4_gen.inc"
template <int OFF>
__device__ __forceinline__ void w4_read(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
// fragment read N of a step's sixteen: the eight B fragments first (the next step starts with row 0 against all of them), then A
template <int N>
__device__ __forceinline__ void w4_rd(bf16x8 (&na)[8], bf16x8 (&nb)[8], uint32_t aa, uint32_t bb) {
#if defined(W4_ABL) && (W4_ABL & 2)
    return;
#endif
    if (N < 8) w4_read<(N & 7) * 1024>(nb[N & 7], bb);
    else w4_read<(N & 7) * 1024>(na[N & 7], aa);
}
template <int EPI = -1>
__global__ __launch_bounds__(256, 1) void gemm_nt256w4_kernel(GemmP p) {
    constexpr int BN = 256, FM = 8, FN = 8;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int nwg = p.tiles_m * p.tiles_n;
    const int G = gridDim.x;
    int bid = blockIdx.x;
    {   // XCD-aware remap of the launch slots, as in gemm_nt256_kernel
        const int q = G >> 3, r = G & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nk = p.K >> 5;                                  // even (host check)
    const int my_tiles = (nwg - bid + G - 1) / G;
    const int S = my_tiles * nk;                              // flattened K-steps of this workgroup
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int wr = wid >> 1, wc = wid & 1;

    // ---- issue side: DMA piece i (0..3) of wave w fills LDS rows (4 i + w) * 16 .. + 15 of an operand image (same image as the
    // eight-wave kernel: lane -> row l >> 2, 16-byte slot l & 3 holding k-chunk (l & 3) ^ g2_swz(l >> 4))
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, -1, 0x00020000);
    // (named scalars, not arrays, and macros, not lambdas: with the pieces behind branches hipcc kept ga[] / gb[] and the buffer
    // descriptors in scratch memory and put a waterfall loop around every piece)
    unsigned ga0, ga1, ga2, ga3, gb0, gb1, gb2, gb3;      // byte offsets of this lane's source rows (+ k-chunk): 32 bits (host check)
    int gi = 0, i_t = 0, i_tile = bid;
    const int kc2 = ((l & 3) ^ g2_swz(l >> 4)) * 16;
#define W4_SET_PTRS(tile_)                                                                                                        \
    do {                                                                                                                          \
        const int tm_ = (tile_) / p.tiles_n, tn_ = (tile_) % p.tiles_n;                                                           \
        const int r0_ = wid * 16 + (l >> 2);                                                                                      \
        ga0 = (unsigned)min(tm_ * 256 + r0_, p.M - 1) * (unsigned)(p.lda * 2) + kc2;      /* rows past the edge re-read the last row */ \
        ga1 = (unsigned)min(tm_ * 256 + r0_ + 64, p.M - 1) * (unsigned)(p.lda * 2) + kc2;                                          \
        ga2 = (unsigned)min(tm_ * 256 + r0_ + 128, p.M - 1) * (unsigned)(p.lda * 2) + kc2;                                         \
        ga3 = (unsigned)min(tm_ * 256 + r0_ + 192, p.M - 1) * (unsigned)(p.lda * 2) + kc2;                                         \
        gb0 = (unsigned)min(tn_ * BN + r0_, p.N - 1) * (unsigned)(p.ldb * 2) + kc2;                                                \
        gb1 = (unsigned)min(tn_ * BN + r0_ + 64, p.N - 1) * (unsigned)(p.ldb * 2) + kc2;                                           \
        gb2 = (unsigned)min(tn_ * BN + r0_ + 128, p.N - 1) * (unsigned)(p.ldb * 2) + kc2;                                          \
        gb3 = (unsigned)min(tn_ * BN + r0_ + 192, p.N - 1) * (unsigned)(p.ldb * 2) + kc2;                                          \
    } while (0)
    W4_SET_PTRS(i_tile);
    // piece k of the step being issued (k < 4 operand A, else B): LDS rows (4 (k & 3) + w) * 16 .. of the operand image of stage gi & 3
#if defined(W4_ABL) && (W4_ABL & 8)      // timing ablation: plain loads into registers instead of LDS-DMA (nothing reaches LDS)
    u32x4 dmy[8];
#define W4_DMA_A(i_, g_) dmy[i_] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)(g_), i_t << 6, 0))
#define W4_DMA_B(i_, g_) dmy[4 + (i_)] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)(g_), i_t << 6, 0))
#else
#define W4_DMA_A(i_, g_) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (g2_lptr)(smem + (gi & 3) * G2_STAGE + (4 * (i_) + wid) * 1024), 16, (int)(g_), i_t << 6, 0, 0)
#define W4_DMA_B(i_, g_) __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (g2_lptr)(smem + (gi & 3) * G2_STAGE + G2_OP_BYTES + (4 * (i_) + wid) * 1024), 16, (int)(g_), i_t << 6, 0, 0)
#endif
#define W4_DMA_0 W4_DMA_A(0, ga0)
#define W4_DMA_1 W4_DMA_A(1, ga1)
#define W4_DMA_2 W4_DMA_A(2, ga2)
#define W4_DMA_3 W4_DMA_A(3, ga3)
#define W4_DMA_4 W4_DMA_B(0, gb0)
#define W4_DMA_5 W4_DMA_B(1, gb1)
#define W4_DMA_6 W4_DMA_B(2, gb2)
#define W4_DMA_7 W4_DMA_B(3, gb3)
#define W4_ADVANCE_STEP()        /* the bookkeeping of a step's eight pieces (outside the branches that hold the pieces) */          \
    do {                                                                                                                          \
        gi++;                                                                                                                     \
        if (++i_t == nk) {                                                                                                        \
            i_t = 0;                                                                                                              \
            i_tile += G;                                                                                                          \
            if (i_tile < nwg) W4_SET_PTRS(i_tile);                                                                                \
        }                                                                                                                         \
    } while (0)
#define W4_ISSUE_STEP() do { W4_DMA_0; W4_DMA_1; W4_DMA_2; W4_DMA_3; W4_DMA_4; W4_DMA_5; W4_DMA_6; W4_DMA_7; W4_ADVANCE_STEP(); } while (0)
#ifndef W4_ABL
#define W4_ABL 0                // timing ablations (garbage results): 1 no DMA pieces in the loop, 2 no fragment reads, 4 no barrier
#endif
    // ---- fragment addresses: row = base + (l & 15), k-group l >> 4 in slot (l >> 4) ^ g2_swz((row >> 2) & 3); fragment i at + i * 1024
    const int fsw = ((l >> 4) ^ g2_swz((l >> 2) & 3)) << 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_off = lds0 + (wr * 128 + (l & 15)) * 64 + fsw;
    const uint32_t b_off = lds0 + G2_OP_BYTES + (wc * 128 + (l & 15)) * 64 + fsw;

    f32x4 acc[FM][FN];
    bf16x8 fa0[FM], fb0[FN], fa1[FM], fb1[FN];
    W4_ISSUE_STEP();
    W4_ISSUE_STEP();
    W4_ISSUE_STEP();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int g = 0;                  // flattened step being computed
    int since_drain = 2;        // steps since the last vmcnt(0) of this wave (the prologue's wait covers steps 0, 1 only: treat as old)
    // FIRST: first step of a tile (C = 0).  `last`: last step of a tile.
    auto step = [&](auto first_, const bool last, bf16x8 (&fa)[FM], bf16x8 (&fb)[FN], bf16x8 (&na)[FM], bf16x8 (&nb)[FN]) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_)::value;
        // (no condition on "is there a step g + 3 / g + 1": past the end of the workgroup's sequence the pieces re-read the last
        // tile's rows into stages nobody reads any more and the fragment reads fetch what is there -- branches between the MFMAs
        // cost more than three steps' worth of dummy traffic; the kernel drains its DMAs before it ends)
        const uint32_t aa = a_off + ((g + 1) & 3) * G2_STAGE, bb = b_off + ((g + 1) & 3) * G2_STAGE;
#define W4_PIECE(k_) do { if (!(W4_ABL & 1)) { W4_DMA_##k_; } } while (0)
        W4_HALF0
        if (since_drain >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // in flight may stay: the pieces of steps g + 2 and g + 3
        if (!(W4_ABL & 4)) __builtin_amdgcn_s_barrier();
        W4_HALF1
#undef W4_PIECE
        W4_ADVANCE_STEP();
#if defined(W4_ABL) && (W4_ABL & 8)
#pragma unroll
        for (int q_ = 0; q_ < 8; q_++) asm volatile("" :: "v"(dmy[q_]));
#if (W4_ABL & 16)      // ... and written to LDS from there (some stage: timing only)
#pragma unroll
        for (int q_ = 0; q_ < 8; q_++) *reinterpret_cast<u32x4*>(smem + (gi & 3) * G2_STAGE + (q_ * 4 + wid) * 1024 + l * 16) = dmy[q_];
#endif
#endif
        if (last) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); since_drain = 0; }
        else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); since_drain++; }
        g++;
    };
    auto first_frags = [&]() __attribute__((always_inline)) {      // fragments of step g (its stage has landed and been published) -> set 0
        const uint32_t aa = a_off + (g & 3) * G2_STAGE, bb = b_off + (g & 3) * G2_STAGE;
        w4_rd<0>(fa0, fb0, aa, bb); w4_rd<1>(fa0, fb0, aa, bb); w4_rd<2>(fa0, fb0, aa, bb); w4_rd<3>(fa0, fb0, aa, bb);
        w4_rd<4>(fa0, fb0, aa, bb); w4_rd<5>(fa0, fb0, aa, bb); w4_rd<6>(fa0, fb0, aa, bb); w4_rd<7>(fa0, fb0, aa, bb);
        w4_rd<8>(fa0, fb0, aa, bb); w4_rd<9>(fa0, fb0, aa, bb); w4_rd<10>(fa0, fb0, aa, bb); w4_rd<11>(fa0, fb0, aa, bb);
        w4_rd<12>(fa0, fb0, aa, bb); w4_rd<13>(fa0, fb0, aa, bb); w4_rd<14>(fa0, fb0, aa, bb); w4_rd<15>(fa0, fb0, aa, bb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    first_frags();

    constexpr bool BIAS_INIT = EPI >= 0 && (EPI & MXL_GEMM_BIAS);
    const int flags = EPI >= 0 ? ((BIAS_INIT ? EPI & ~MXL_GEMM_BIAS : EPI) & ~GEMM_HEADDOT_BIT) : p.flags;
    if (EPI == 0 || BIAS_INIT || EPI == GEMM_HEADDOT_BIT) p.alpha = 1.f;
#pragma unroll 1
    for (int tile = bid; tile < nwg; tile += G) {
        if (BIAS_INIT) {
            const int nb0 = (tile % p.tiles_n) * BN + wc * (FN * 16) + (l >> 4) * 4;
#pragma unroll
            for (int j = 0; j < FN; j++) {
                f32x4 b4;
                const int n = nb0 + j * 16;
                if (n + 3 < p.N) b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
                else
#pragma unroll
                    for (int r = 0; r < 4; r++) b4[r] = n + r < p.N ? p.bias[n + r] : 0.f;
#pragma unroll
                for (int i = 0; i < FM; i++) acc[i][j] = b4;
            }
        }
#pragma unroll 1
        for (int t = 0; t < nk; t += 2) {
            if (t == 0 && !BIAS_INIT) step(std::true_type{}, false, fa0, fb0, fa1, fb1);
            else step(std::false_type{}, false, fa0, fb0, fa1, fb1);
            step(std::false_type{}, t + 2 >= nk, fa1, fb1, fa0, fb0);
        }
        // epilogue.  acc[i][j][r]: m = m0 + wr*128 + i*16 + (l&15), n = n0 + wc*128 + j*16 + (l>>4)*4 + r.  Every tile is interior and
        // the output bf16 with 16-byte aligned rows (host check: other problems take the eight-wave kernel) -- with the edge path
        // compiled in, hipcc spilled 156 accumulators to scratch in front of the epilogue
        const int m0 = (tile / p.tiles_n) * 256, n0 = (tile % p.tiles_n) * BN;
        const int mrow = m0 + wr * (FM * 16) + (l & 15);
        constexpr bool HD = EPI >= 0 && (EPI & GEMM_HEADDOT_BIT);
        if (!HD) {
#define W4_EPAIR(i, j, a0, a1, a2, a3, ah, b0, b1, b2, b3, bh, m_)                                                                \
            {                                                                                                                     \
                f32x4 q0_, q1_;                                                                                                   \
                W4_ACC_READ(i, j, a0, a1, a2, a3, a0, ah, q0_);                                                                   \
                W4_ACC_READ(i, j + 1, b0, b1, b2, b3, b0, bh, q1_);                                                               \
                epilogue_pair_bf16(p, flags, m_, n0 + wc * 128 + (j) * 16 + (l >> 4) * 4, l, q0_, q1_);                            \
            }
            W4_EPILOGUE_ROW_0(mrow) W4_EPILOGUE_ROW_1(mrow + 16) W4_EPILOGUE_ROW_2(mrow + 32) W4_EPILOGUE_ROW_3(mrow + 48)
            W4_EPILOGUE_ROW_4(mrow + 64) W4_EPILOGUE_ROW_5(mrow + 80) W4_EPILOGUE_ROW_6(mrow + 96) W4_EPILOGUE_ROW_7(mrow + 112)
#undef W4_EPAIR
        } else {
            // plain bf16 store + the per-(row, 64-column head) dot product of the stored values with a second matrix (mxl_gemm_bf16_headdot:
            // the attention backward's delta = sum_e dO . O out of the GEMM that produces dO, instead of a pass over dO and O).  Behind the
            // permlane swap a lane holds 8 consecutive columns of its row: the same 16 bytes of the second matrix, loaded a row block ahead;
            // the two pairs of a head add up in the lane, then over the four 16-lane groups.
            const int odd = (l >> 4) & 1;
            const int cbase = n0 + wc * 128 + (l >> 4) * 4 + (odd ? 12 : 0);        // pair j: + 16 j
            u32x4 ox[2][4];
#define W4_HD_LOAD(set_, m_)                                                                                                       \
            _Pragma("unroll") for (int jj = 0; jj < 4; jj++)                                                                      \
                ox[set_][jj] = *reinterpret_cast<const u32x4*>(p.hd_o + (size_t)(m_) * p.hd_ldo + cbase + 32 * jj);
            W4_HD_LOAD(0, mrow)
            float hs0 = 0.f, hs1 = 0.f;
#define W4_EPAIR(i, j, a0, a1, a2, a3, ah, b0, b1, b2, b3, bh, m_)                                                                \
            {                                                                                                                     \
                f32x4 q0_, q1_;                                                                                                   \
                W4_ACC_READ(i, j, a0, a1, a2, a3, a0, ah, q0_);                                                                   \
                W4_ACC_READ(i, j + 1, b0, b1, b2, b3, b0, bh, q1_);                                                               \
                const unsigned pa0 = pack2bf(q0_[0], q0_[1]), pa1 = pack2bf(q0_[2], q0_[3]);                                       \
                const unsigned pb0 = pack2bf(q1_[0], q1_[1]), pb1 = pack2bf(q1_[2], q1_[3]);                                       \
                const auto r0 = __builtin_amdgcn_permlane16_swap(pa0, pb0, false, false);                                         \
                const auto r1 = __builtin_amdgcn_permlane16_swap(pa1, pb1, false, false);                                         \
                const u32x4 ov = u32x4{r0[0], r1[0], r0[1], r1[1]};                                                               \
                *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)(m_) * p.ldc + cbase + 16 * (j)) = ov;          \
                const u32x4 xv = ox[(i) & 1][(j) >> 1];                                                                           \
                float t_ = 0.f;                                                                                                   \
                _Pragma("unroll") for (int k_ = 0; k_ < 4; k_++)                                                                  \
                    t_ += __builtin_bit_cast(float, ov[k_] << 16) * __builtin_bit_cast(float, xv[k_] << 16) +                      \
                          __builtin_bit_cast(float, ov[k_] & 0xffff0000u) * __builtin_bit_cast(float, xv[k_] & 0xffff0000u);       \
                if ((j) < 4) hs0 += t_; else hs1 += t_;                                                                           \
            }
            // after a row block: the head sums over the four lane groups, written by group 0 (16 consecutive rows)
#define W4_HD_ROW(i, ROW_)                                                                                                         \
            if ((i) < 7) { W4_HD_LOAD(((i) + 1) & 1, mrow + 16 * ((i) + 1)) }                                                      \
            ROW_(mrow + 16 * (i))                                                                                                 \
            {                                                                                                                     \
                hs0 += __shfl_xor(hs0, 16, 64); hs0 += __shfl_xor(hs0, 32, 64);                                                    \
                hs1 += __shfl_xor(hs1, 16, 64); hs1 += __shfl_xor(hs1, 32, 64);                                                    \
                if ((l >> 4) == 0) {                                                                                              \
                    const int mm_ = mrow + 16 * (i), b_ = mm_ / p.hd_T, tt_ = mm_ - b_ * p.hd_T, h_ = (n0 + wc * 128) >> 6;        \
                    float* dp_ = p.hd_delta + ((size_t)b_ * (p.N >> 6) + h_) * p.hd_T + tt_;                                       \
                    dp_[0] = hs0; dp_[p.hd_T] = hs1;                                                                              \
                }                                                                                                                 \
                hs0 = 0.f; hs1 = 0.f;                                                                                             \
            }
            W4_HD_ROW(0, W4_EPILOGUE_ROW_0) W4_HD_ROW(1, W4_EPILOGUE_ROW_1) W4_HD_ROW(2, W4_EPILOGUE_ROW_2) W4_HD_ROW(3, W4_EPILOGUE_ROW_3)
            W4_HD_ROW(4, W4_EPILOGUE_ROW_4) W4_HD_ROW(5, W4_EPILOGUE_ROW_5) W4_HD_ROW(6, W4_EPILOGUE_ROW_6) W4_HD_ROW(7, W4_EPILOGUE_ROW_7)
#undef W4_EPAIR
#undef W4_HD_ROW
#undef W4_HD_LOAD
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the dummy pieces behind the last step: no LDS-DMA in flight when the workgroup ends
#undef W4_SET_PTRS
#undef W4_DMA_A
#undef W4_DMA_B
#undef W4_ADVANCE_STEP
#undef W4_ISSUE_STEP
}

// =====================================================================================================================
// Large-tile kernel for the weight gradients dW[m][n] += sum_k A[k][m] B[k][n] (A = dY (tokens x out), B = X (tokens x in): both
// operands K-strided), split-K over workgroups, fp32 atomics.  Same machine as gemm_nt256_kernel -- 256 x 256 tile, 8 waves of
// 128 x 64, four-stage global_load_lds ring three K-steps ahead, one raw barrier per 32-deep K-step, counted vmcnt -- with the
// operand images the other way round: a stage holds [32 k][256 m] (512-byte k-rows, the operands' own row order: every DMA
// instruction moves two whole k-rows), and fragments come out through ds_read_b64_tr_b16 (two per 16 x 32 fragment).  The
// 32-byte granules of a k-row are XOR-swizzled by (k & 3) | ((k >> 3) & 1) << 2 -- on the SOURCE side of the DMA -- so that the
// eight k-rows a half-wave's transposed read touches fall into eight different 32-byte bank groups.
// Why now: at 131072 tokens per launch (per-GPU batch 64) a K-slice is ~18 k tokens long, so the 256 KB of fp32 atomics a
// workgroup ends with (64 MB per GEMM at ~1.3 TB/s = 50 us) are a few percent of its loop, while the 128 x 128 kernel re-reads
// the token-long panels twice as often (2.6 GB of HBM traffic per launch against 1 GB of operands, profiles/r02_c3_pmc_traffic).
// =====================================================================================================================
__global__ __launch_bounds__(512) void gemm_tt256_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int wr = wid >> 2, wc = wid & 3;                      // 2 x 4 waves of 128 x 64
    const int ntile = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {   // launch slots b, b + 8, ... share an XCD: give each XCD a contiguous run of (slice, tile) pairs, i.e. mostly ONE K-slice,
        // whose workgroups stream the same token rows of dY and X at the same time (L2 hits instead of HBM reads)
        const int G = gridDim.x, q = G >> 3, r = G & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int slice = bid / ntile, tile = bid - slice * ntile;
    const int m0 = (tile / p.tiles_n) * 256, n0 = (tile % p.tiles_n) * 256;
    const int kbeg = slice * p.ksplit, kend = min(p.K, kbeg + p.ksplit);
    const int S = (kend - kbeg) >> 5;                          // K-steps of this workgroup (host: K-slices are multiples of 32)
    if (S <= 0) return;

    // ---- issue side: DMA instruction i of wave w fills k-rows 2 (8 i + w), 2 (8 i + w) + 1 of an operand image; lane -> k-row
    // (l >> 5), 16-byte chunk l & 31, which holds source chunk ((c >> 1) ^ swz(krow)) << 1 | (c & 1)
    unsigned ga[2], gb[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int krow = 2 * (8 * i + wid) + (l >> 5), c = l & 31;
        const int sw = (krow & 3) | (((krow >> 3) & 1) << 2);
        const int cs = (((c >> 1) ^ sw) << 1) | (c & 1);
        ga[i] = (unsigned)(kbeg + krow) * (unsigned)p.lda + (unsigned)(m0 + cs * 8);
        gb[i] = (unsigned)(kbeg + krow) * (unsigned)p.ldb + (unsigned)(n0 + cs * 8);
    }
    int gi = 0;
    const unsigned stepA = 32u * (unsigned)p.lda, stepB = 32u * (unsigned)p.ldb;
    auto issue_next = [&]() {
        char* st = smem + (gi & 3) * G2_STAGE;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            __builtin_amdgcn_global_load_lds((g2_gptr)(p.A + ga[i]), (g2_lptr)(st + (8 * i + wid) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((g2_gptr)(p.B + gb[i]), (g2_lptr)(st + G2_OP_BYTES + (8 * i + wid) * 1024), 16, 0, 0);
            ga[i] += stepA; gb[i] += stepB;
        }
        gi++;
    };
    // ---- fragment addresses (bytes inside an operand image): lane -> k-row 8 (l >> 4) + ((l & 15) >> 2), columns 4 (l & 3) ..
    // of the fragment's 16; granule G of the row sits at (G ^ sw) * 32
    const int fq = (l & 15) >> 2, fg = l >> 4;
    const int fsw = fq | ((fg & 1) << 2);
    const int fbase = (8 * fg + fq) * 512 + 8 * (l & 3);
    int offA[8], offB[4];
#pragma unroll
    for (int i = 0; i < 8; i++) offA[i] = fbase + (((wr * 8 + i) ^ fsw) << 5);
#pragma unroll
    for (int j = 0; j < 4; j++) offB[j] = G2_OP_BYTES + fbase + (((wc * 4 + j) ^ fsw) << 5);
    // The transposed reads are inline asm: through the builtin hipcc puts s_waitcnt vmcnt(0) in front of the first read of every
    // step (it cannot tell the read from the LDS-DMA writes in flight), which drains the ring each step (measured: 760 TFLOP/s on
    // the ffn shapes with the builtin).  A fragment = two 8-byte reads whose results the MFMA takes as one 128-bit operand; the
    // waits are this kernel's own: lgkmcnt(0) + sched_barrier once per step, before the barrier that precedes the fragments' use.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    auto tr_frag = [&](uint32_t addr) -> bf16x8 {
        u64x2 v;
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048"
                     : "=&v"(v.x), "=&v"(v.y) : "v"(addr) : "memory");
        return __builtin_bit_cast(bf16x8, v);
    };
    auto frags = [&](int g, bf16x8 (&fa)[8], bf16x8 (&fb)[4]) {
        const uint32_t st = lds0 + (g & 3) * G2_STAGE;
#pragma unroll
        for (int j = 0; j < 4; j++) fb[j] = tr_frag(st + offB[j]);
#pragma unroll
        for (int i = 0; i < 8; i++) fa[i] = tr_frag(st + offA[i]);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue_next();
    if (S > 1) issue_next();
    if (S > 2) issue_next();
    if (S > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 fa0[8], fb0[4], fa1[8], fb1[4];
    frags(0, fa0, fb0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    int g = 0;
    auto step = [&](bf16x8 (&fa)[8], bf16x8 (&fb)[4], bf16x8 (&na)[8], bf16x8 (&nb)[4]) {
        const bool issued = gi < S;
        const bool late = wid >= 4;       // the SIMD's second wave computes first, loads afterwards (as in gemm_nt256_kernel): -5 ... -8 %
        if (!late) {
            if (issued) issue_next();
            if (g + 1 < S) frags(g + 1, na, nb);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)      // (A, B) order: a lane owns one column n and four consecutive rows m (atomics: 4 rows x 64 B)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, fa[i]),
                                                                    __builtin_bit_cast(mfma_bf16x8, fb[j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (late) {
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < S) frags(g + 1, na, nb);
            if (issued) issue_next();
        }
        // the next step's fragments have landed (asm loads: hipcc does not count them) and so has the stage after it
        if (!issued) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        g++;
    };
#pragma unroll 1
    for (int t = 0; t + 1 < S; t += 2) {
        step(fa0, fb0, fa1, fb1);
        step(fa1, fb1, fa0, fb0);
    }
    if (S & 1) step(fa0, fb0, fa1, fb1);

    // acc[i][j][r]: m = m0 + wr*128 + i*16 + (l>>4)*4 + r, n = n0 + wc*64 + j*16 + (l&15)
    float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + wc * 64 + j * 16 + (l & 15);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + wr * 128 + i * 16 + (l >> 4) * 4 + r;
                atomicAdd(C + (size_t)m * p.ldc + n, acc[i][j][r] * p.alpha);
            }
        }
}

}  // namespace

// Compute units the persistent GEMM grids (gemm_nt256_kernel, gemm_tt256_kernel: one 512-thread workgroup holding most of a CU's LDS
// for the whole launch) may occupy: all of them, minus what the caller set aside with mxl_set_reserved_cus -- under data parallelism
// RCCL's reduction kernels run on another stream beside the backward, and a grid that holds every CU for its whole duration leaves
// them nowhere to start until it ends.
static int g_reserved_cus = 0;
static int gemm_n_cu() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0, cus = 0;
        n_cu = (hipGetDevice(&dev) == hipSuccess &&
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus : 256;
    }
    const int left = n_cu - g_reserved_cus;
    return left >= 8 ? left : 8;
}
// which large-tile kernel the last K-contiguous product went to: 0 none of them, 1 eight waves x 256-wide tiles, 2 eight waves x 192-wide,
// 3 four waves (tests: the shapes they mean to put on a kernel are on it)
static int g_last_nt_kernel = 0;
extern "C" int mxl_gemm_last_nt_kernel() { return g_last_nt_kernel; }
extern "C" int mxl_set_reserved_cus(int k) {
    MXL_CHECK_ARG(k >= 0 && k <= 128);
    g_reserved_cus = k;
    return MXL_OK;
}
// BN = 256 or 192 for the large-tile NT kernel: whichever wastes less of the chip (rounds of one tile per CU x tile width)
static bool nt256_use192(int M, int N) {
    const int n_cu = gemm_n_cu();
    const int tm = (M + 255) / 256;
    const int t256 = tm * ((N + 255) / 256), t192 = tm * ((N + 191) / 192);
    const long long c256 = (long long)((t256 + n_cu - 1) / n_cu) * 256, c192 = (long long)((t192 + n_cu - 1) / n_cu) * 192;
    return c192 < c256;
}

extern "C" size_t mxl_gemm_relu_mask_bytes(int M, int N) {
    if (M <= 0 || N <= 0 || (M % 256) != 0 || (N % 256) != 0 || nt256_use192(M, N)) return 0;
    return (size_t)(M / 256) * (size_t)(N / 256) * 8 * 64 * 16;       // 128 bits per lane, wave and tile = M * N / 8 bytes
}

static int gemm_launch(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                       int transA, int transB, int flags, float alpha, const float* bias,
                       const void* aux, int ldaux, int ksplits,
                       float drop_p, unsigned long long seed, unsigned site, void* stream,
                       int batch, int bdiv, long long sA1, long long sA2, long long sB1, long long sB2,
                       long long sC1, long long sC2, float* colsum = nullptr, bool* colsum_fused = nullptr,
                       const void* hd_o = nullptr, int hd_ldo = 0, int hd_T = 0, float* hd_delta = nullptr, bool* hd_fused = nullptr) {
    MXL_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0);
    g_last_nt_kernel = 0;
    MXL_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0);
    MXL_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0);
    // K-contiguous operands are fetched in 8-element chunks along K: K must be chunk-exact.
    if (!transA || !transB) MXL_CHECK_ARG((K % 8) == 0);
    MXL_CHECK_ARG(transA ? lda >= ((M + 7) & ~7) : lda >= K);
    MXL_CHECK_ARG(transB ? ldb >= ((N + 7) & ~7) : ldb >= K);
    MXL_CHECK_ARG(ldc >= N);
    if (flags & MXL_GEMM_BIAS) MXL_CHECK_ARG(bias != nullptr);
    if (flags & (MXL_GEMM_SAVE_RELU_MASK | MXL_GEMM_RELU_BWD_BITS)) {
        MXL_CHECK_ARG(aux != nullptr && ((uintptr_t)aux & 15) == 0 && !(flags & (MXL_GEMM_RELU_BWD | MXL_GEMM_ADD_AUX)));
        MXL_CHECK_ARG(!((flags & MXL_GEMM_SAVE_RELU_MASK) && (flags & MXL_GEMM_RELU_BWD_BITS)));
        if (mxl_gemm_relu_mask_bytes(M, N) == 0 || transA || transB || batch != 1 || ksplits > 1 || (K % 64) != 0 || (ldc & 7) != 0 ||
            ((uintptr_t)C & 15) != 0 || (flags & (MXL_GEMM_OUT_F32 | MXL_GEMM_OUT_F32_ATOMIC)) || getenv("MXL_GEMM_NO256") ||
            (long long)M * lda >= (1ll << 31) || (long long)N * ldb >= (1ll << 31))
            return MXL_EUNSUPPORTED;
    }
    if (flags & (MXL_GEMM_RELU_BWD | MXL_GEMM_ADD_AUX)) MXL_CHECK_ARG(aux != nullptr && ldaux >= N);
    MXL_CHECK_ARG(!((flags & MXL_GEMM_RELU_BWD) && (flags & MXL_GEMM_ADD_AUX)));
    if (ksplits < 1) ksplits = 1;
    if (ksplits > 1) {
        MXL_CHECK_ARG(flags & MXL_GEMM_OUT_F32_ATOMIC);
        MXL_CHECK_ARG(!(flags & (MXL_GEMM_BIAS | MXL_GEMM_RELU | MXL_GEMM_DROPOUT | MXL_GEMM_RELU_BWD | MXL_GEMM_ADD_AUX)));
    }
    GemmP p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    int ktiles = (K + BK - 1) / BK;
    int per = (ktiles + ksplits - 1) / ksplits;
    ksplits = (ktiles + per - 1) / per;
    p.ksplit = per * BK;
    p.bias = bias; p.aux = (const bf16_t*)aux; p.ldaux = ldaux; p.alpha = alpha; p.flags = flags;
    p.colsum = colsum;
    p.hd_o = (const bf16_t*)hd_o; p.hd_ldo = hd_ldo; p.hd_T = hd_T; p.hd_delta = hd_delta;
    p.seed = seed; p.site = site; p.thresh = dropout_thresh(drop_p);
    { static const bool nt = getenv("MXL_GEMM_NT") != nullptr; if (nt) p.flags |= (1 << 30); }
    p.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    if ((flags & MXL_GEMM_DROPOUT) && drop_p <= 0.f) p.flags &= ~MXL_GEMM_DROPOUT;
    // the dropout mask is indexed by the 32-bit element index (kernels that regenerate it use the same index for < 2^32 elements)
    if (p.flags & MXL_GEMM_DROPOUT) MXL_CHECK_ARG((unsigned long long)M * (unsigned long long)N <= 0xffffffffull);
    const int BN = (N <= 64) ? 64 : 128;
    p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
    MXL_CHECK_ARG(batch >= 1 && bdiv >= 1);
    if (batch > 1) {
        MXL_CHECK_ARG((sA1 % 8) == 0 && (sA2 % 8) == 0 && (sB1 % 8) == 0 && (sB2 % 8) == 0);
        MXL_CHECK_ARG(!(flags & (MXL_GEMM_RELU_BWD)));
    }
    p.bdiv = bdiv; p.sA1 = sA1; p.sA2 = sA2; p.sB1 = sB1; p.sB2 = sB2; p.sC1 = sC1; p.sC2 = sC2;
    hipStream_t s = (hipStream_t)stream;
    if (!transA && !transB && batch == 1 && ksplits == 1 && (K % 64) == 0 && M >= 256 && N >= 192 &&
        (long long)M * lda < (1ll << 31) && (long long)N * ldb < (1ll << 31) &&
        !(flags & MXL_GEMM_OUT_F32_ATOMIC) && !getenv("MXL_GEMM_NO256")) {
        const int n_cu = gemm_n_cu();
        const int tm = (M + 255) / 256;
        const bool use192 = nt256_use192(M, N);
        p.tiles_m = tm; p.tiles_n = use192 ? (N + 191) / 192 : (N + 255) / 256;
        const int ntile = p.tiles_m * p.tiles_n;
        dim3 grid(ntile < n_cu ? ntile : n_cu);
        // compile-time epilogues for the combinations the engines use; everything else takes the run-time form
#define MXL_NT256_LAUNCH(EPI_)                                                                                                   \
    do {                                                                                                                         \
        static bool attr_e = false;                                                                                              \
        if (!attr_e) {                                                                                                           \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256_kernel<4, EPI_>),                         \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, G2_SMEM);                             \
            if (e != hipSuccess) return (int)e;                                                                                  \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256_kernel<3, EPI_>),                                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, G2_SMEM);                                        \
            if (e != hipSuccess) return (int)e;                                                                                  \
            attr_e = true;                                                                                                       \
        }                                                                                                                        \
        if (use192) hipLaunchKernelGGL((gemm_nt256_kernel<3, EPI_>), grid, dim3(512), G2_SMEM, s, p);                       \
        else hipLaunchKernelGGL((gemm_nt256_kernel<4, EPI_>), grid, dim3(512), G2_SMEM, s, p);                                     \
        g_last_nt_kernel = use192 ? 2 : 1;                                                                                      \
    } while (0)
        // four waves of 128 x 128 (gemm_nt256w4_kernel) where the tile width is 256
        const char* w4_env = getenv("MXL_GEMM_W4");                           // (read per call: the test compares the two kernels)
        const bool w4 = !(w4_env && w4_env[0] == '0');
#define MXL_NT256W4_LAUNCH(EPI_)                                                                                                 \
    do {                                                                                                                         \
        static bool attr_w = false;                                                                                              \
        if (!attr_w) {                                                                                                           \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256w4_kernel<EPI_>),                          \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, G2_SMEM);                             \
            if (e != hipSuccess) return (int)e;                                                                                  \
            attr_w = true;                                                                                                       \
        }                                                                                                                        \
        hipLaunchKernelGGL((gemm_nt256w4_kernel<EPI_>), grid, dim3(256), G2_SMEM, s, p);                                          \
        g_last_nt_kernel = 3;                                                                                                    \
    } while (0)
        const bool w4_ok = w4 && !use192 && (M % 256) == 0 && (N % 256) == 0 && (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 && p.alpha == 1.f;
        if (w4_ok && p.flags == 0 && hd_delta && (N % 128) == 0 && hd_T > 0 && (M % hd_T) == 0 && (hd_ldo & 7) == 0 && ((uintptr_t)hd_o & 15) == 0) {
            MXL_NT256W4_LAUNCH(GEMM_HEADDOT_BIT);
            *hd_fused = true;
        }
        else if (w4_ok && p.flags == 0) MXL_NT256W4_LAUNCH(0);
        else if (w4_ok && p.flags == MXL_GEMM_BIAS) MXL_NT256W4_LAUNCH(MXL_GEMM_BIAS);
        else if (p.flags == 0 && p.alpha == 1.f) MXL_NT256_LAUNCH(0);
        else if (p.alpha != 1.f && (p.flags & MXL_GEMM_BIAS) && !(p.flags & (MXL_GEMM_SAVE_RELU_MASK | MXL_GEMM_RELU_BWD_BITS))) MXL_NT256_LAUNCH(-1);
        else if (p.alpha != 1.f && (p.flags & MXL_GEMM_SAVE_RELU_MASK)) return MXL_EUNSUPPORTED;
        else if (p.flags == MXL_GEMM_BIAS) MXL_NT256_LAUNCH(MXL_GEMM_BIAS);
        else if (p.flags == (MXL_GEMM_BIAS | MXL_GEMM_OUT_F32)) MXL_NT256_LAUNCH(MXL_GEMM_BIAS | MXL_GEMM_OUT_F32);     // the heads' fp32 logits
        else if (p.flags == (MXL_GEMM_BIAS | MXL_GEMM_RELU)) MXL_NT256_LAUNCH(MXL_GEMM_BIAS | MXL_GEMM_RELU);
        else if (p.flags == (MXL_GEMM_BIAS | MXL_GEMM_RELU | MXL_GEMM_DROPOUT)) MXL_NT256_LAUNCH(MXL_GEMM_BIAS | MXL_GEMM_RELU | MXL_GEMM_DROPOUT);
        else if (p.flags == (MXL_GEMM_BIAS | MXL_GEMM_RELU | MXL_GEMM_DROPOUT | MXL_GEMM_SAVE_RELU_MASK))
            MXL_NT256_LAUNCH(MXL_GEMM_BIAS | MXL_GEMM_RELU | MXL_GEMM_DROPOUT | MXL_GEMM_SAVE_RELU_MASK);
        else if (p.flags == (MXL_GEMM_BIAS | MXL_GEMM_RELU | MXL_GEMM_SAVE_RELU_MASK))
            MXL_NT256_LAUNCH(MXL_GEMM_BIAS | MXL_GEMM_RELU | MXL_GEMM_SAVE_RELU_MASK);
        else if (p.flags == MXL_GEMM_RELU_BWD_BITS && colsum) {
            MXL_NT256_LAUNCH(MXL_GEMM_RELU_BWD_BITS | GEMM_COLSUM_BIT);
            *colsum_fused = true;
        }
        else if (p.flags == MXL_GEMM_RELU_BWD_BITS) MXL_NT256_LAUNCH(MXL_GEMM_RELU_BWD_BITS);
        else if (p.flags & (MXL_GEMM_SAVE_RELU_MASK | MXL_GEMM_RELU_BWD_BITS)) return MXL_EUNSUPPORTED;
        else if (p.flags == MXL_GEMM_RELU_BWD && colsum && !use192 && (M % 256) == 0 && (N % 256) == 0 && (ldc & 7) == 0 &&
                 ((uintptr_t)C & 15) == 0) {
            // every tile interior (the paired-store epilogue): the column sums of the output ride along (mxl_gemm_bf16_colsum)
            MXL_NT256_LAUNCH(MXL_GEMM_RELU_BWD | GEMM_COLSUM_BIT);
            *colsum_fused = true;
        }
        else if (p.flags == MXL_GEMM_RELU_BWD) MXL_NT256_LAUNCH(MXL_GEMM_RELU_BWD);
        else if (p.flags == MXL_GEMM_ADD_AUX) MXL_NT256_LAUNCH(MXL_GEMM_ADD_AUX);                    // Reformer residual epilogues
        else if (p.flags == (MXL_GEMM_ADD_AUX | MXL_GEMM_DROPOUT)) MXL_NT256_LAUNCH(MXL_GEMM_ADD_AUX | MXL_GEMM_DROPOUT);
        else if (p.flags == (MXL_GEMM_BIAS | MXL_GEMM_ADD_AUX)) MXL_NT256_LAUNCH(MXL_GEMM_BIAS | MXL_GEMM_ADD_AUX);
        else if (p.flags == (MXL_GEMM_BIAS | MXL_GEMM_ADD_AUX | MXL_GEMM_DROPOUT))
            MXL_NT256_LAUNCH(MXL_GEMM_BIAS | MXL_GEMM_ADD_AUX | MXL_GEMM_DROPOUT);
        else MXL_NT256_LAUNCH(-1);
#undef MXL_NT256_LAUNCH
#undef MXL_NT256W4_LAUNCH
        MXL_LAUNCH_CHECK();
        return MXL_OK;
    }
    // weight gradients at training sizes: the 256 x 256 DMA-fed form (it picks its own K-slicing: one workgroup per CU)
    if (transA && transB && batch == 1 && p.flags == MXL_GEMM_OUT_F32_ATOMIC && (M % 256) == 0 && (N % 256) == 0 && (K % 32) == 0 &&
        K >= 8192 && (long long)K * lda < (1ll << 31) && (long long)K * ldb < (1ll << 31) && ((uintptr_t)C % 16) == 0 &&
        !getenv("MXL_GEMM_NO_TT256")) {
        static bool attr_tt = false;
        if (!attr_tt) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tt256_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, G2_SMEM);
            if (e != hipSuccess) return (int)e;
            attr_tt = true;
        }
        const int n_cu_tt = gemm_n_cu();
        p.tiles_m = M / 256; p.tiles_n = N / 256;
        const int ntile = p.tiles_m * p.tiles_n;
        int ks = n_cu_tt / ntile;
        if (ks < 1) ks = 1;
        const int steps = K / 32;
        if (ks > steps / 64) ks = steps / 64 > 0 ? steps / 64 : 1;          // at least 64 K-steps per slice
        const int per_steps = (steps + ks - 1) / ks;
        ks = (steps + per_steps - 1) / per_steps;
        p.ksplit = per_steps * 32;
        hipLaunchKernelGGL(gemm_tt256_kernel, dim3(ntile * ks), dim3(512), G2_SMEM, s, p);
        MXL_LAUNCH_CHECK();
        return MXL_OK;
    }
    dim3 grid(p.tiles_m * p.tiles_n, batch, ksplits), block(256);
#define MXL_GEMM_LAUNCH(AT_, BT_)                                                                                          \
    do {                                                                                                                   \
        if (BN == 64) hipLaunchKernelGGL((gemm_bf16_kernel<AT_, BT_, 64>), grid, block, 2 * (TILE_BYTES + 64 * BK * 2), s, p); \
        else hipLaunchKernelGGL((gemm_bf16_kernel<AT_, BT_, 128>), grid, block, 4 * TILE_BYTES, s, p);                      \
    } while (0)
    const int epi = MXL_GEMM_BIAS | MXL_GEMM_RELU | MXL_GEMM_DROPOUT | MXL_GEMM_RELU_BWD | MXL_GEMM_ADD_AUX | MXL_GEMM_OUT_F32;
    if (transA && transB && (p.flags & MXL_GEMM_OUT_F32_ATOMIC) && !(p.flags & epi)) {     // weight gradients, dRd fallback
        if (BN == 64) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, 64, true>), grid, block, 2 * (TILE_BYTES + 64 * BK * 2), s, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<true, true, 128, true>), grid, block, 4 * TILE_BYTES, s, p);
    } else
    if (!transA && !transB) MXL_GEMM_LAUNCH(false, false);
    else if (!transA && transB) MXL_GEMM_LAUNCH(false, true);
    else if (transA && !transB) MXL_GEMM_LAUNCH(true, false);
    else MXL_GEMM_LAUNCH(true, true);
#undef MXL_GEMM_LAUNCH
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                             int transA, int transB, int flags, float alpha, const float* bias,
                             const void* aux, int ldaux, int ksplits,
                             float drop_p, unsigned long long seed, unsigned site, void* stream) {
    return gemm_launch(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, flags, alpha, bias, aux, ldaux, ksplits, drop_p,
                       seed, site, stream, 1, 1, 0, 0, 0, 0, 0, 0);
}

extern "C" int mxl_gemm_bf16_colsum(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                    int transA, int transB, int flags, float alpha, const float* bias, const void* aux, int ldaux,
                                    float drop_p, unsigned long long seed, unsigned site, float* colsum, void* stream) {
    MXL_CHECK_ARG(colsum && !(flags & (MXL_GEMM_OUT_F32 | MXL_GEMM_OUT_F32_ATOMIC)));
    bool fused = false;
    const int rc = gemm_launch(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, flags, alpha, bias, aux, ldaux, 1, drop_p, seed, site,
                               stream, 1, 1, 0, 0, 0, 0, 0, 0, colsum, &fused);
    if (rc != MXL_OK || fused) return rc;
    return mxl_colsum_bf16(C, colsum, M, N, ldc, stream);
}

extern "C" int mxl_gemm_bf16_headdot(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, const void* O,
                                     int ldo, int T, float* delta, void* stream) {
    MXL_CHECK_ARG(O && delta && T > 0 && ldo >= N);
    bool fused = false;
    const int rc = gemm_launch(A, B, C, M, N, K, lda, ldb, ldc, 0, 0, 0, 1.f, nullptr, nullptr, 0, 1, 0.f, 0, 0, stream, 1, 1, 0, 0, 0, 0, 0, 0,
                               nullptr, nullptr, O, ldo, T, delta, &fused);
    if (rc != MXL_OK) return rc;
    return fused ? MXL_OK : MXL_EUNSUPPORTED;         // (C is written either way; delta only when the four-wave kernel took the problem)
}

extern "C" int mxl_gemm_bf16_batched(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb,
                                     int ldc, int transA, int transB, int flags, float alpha, int ksplits, int batch,
                                     int bdiv, long long sA1, long long sA2, long long sB1, long long sB2, long long sC1,
                                     long long sC2, void* stream) {
    return gemm_launch(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, flags, alpha, nullptr, nullptr, 0, ksplits, 0.f, 0,
                       0, stream, batch, bdiv, sA1, sA2, sB1, sB2, sC1, sC2);
}
