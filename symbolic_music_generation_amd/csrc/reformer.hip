// Reformer path: axial position embeddings, LSH hashing, stable bucket sort, chunked
// attention (local and LSH share one kernel), hash-round combine -- forward and backward.
// Restates HuggingFace modeling_reformer.py as reached from musicnlp/models/reformer.py:114-127 ("HF515:" = line numbers
// of the transformers 5.15 copy used to pin the oracle).
#include "common.h"
#include "musicxl_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// =====================================================================================================================
// embeddings: out[b,t,:] = drop(E[ids[b,t]]) + cat(W0[t / A1], W1[t % A1]) with 2-D dropout over the (b, t % A1) columns
// (HF515:222-256, 324-354)
// =====================================================================================================================
__global__ void axial_embed_fwd_kernel(const long long* ids, const bf16_t* E, const float* W0, const float* W1, bf16_t* out,
                                       int B, int T, int d, int V, int A1, int d0, unsigned thresh, float dscale,
                                       unsigned long long seed, unsigned site_emb, unsigned site_pos) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * T * d) return;
    const int k = (int)(gid % d);
    const long long n = gid / d;
    const int t = (int)(n % T), b = (int)(n / T);
    long long id = ids[n];
    if (id < 0 || id >= V) id = 0;
    float e = bf2f(E[(size_t)id * d + k]);
    float pos = k < d0 ? W0[(size_t)(t / A1) * d0 + k] : W1[(size_t)(t % A1) * (d - d0) + (k - d0)];
    if (thresh) {
        e = dropout_keep(seed, site_emb, (uint64_t)gid, thresh) ? e * dscale : 0.f;
        pos = dropout_keep(seed, site_pos, (uint64_t)b * A1 + (t % A1), thresh) ? pos * dscale : 0.f;
    }
    out[gid] = f2bf(e + pos);
}

__global__ void axial_embed_bwd_kernel(const long long* ids, const bf16_t* dout, const bf16_t* dout2, float* dE, float* dW0,
                                       float* dW1, int B, int T, int d, int V, int A1, int d0, unsigned thresh, float dscale,
                                       unsigned long long seed, unsigned site_emb, unsigned site_pos) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * T * d) return;
    const int k = (int)(gid % d);
    const long long n = gid / d;
    const int t = (int)(n % T), b = (int)(n / T);
    float g = bf2f(dout[gid]);
    if (dout2) g += bf2f(dout2[gid]);
    float ge = g, gp = g;
    if (thresh) {
        ge = dropout_keep(seed, site_emb, (uint64_t)gid, thresh) ? g * dscale : 0.f;
        gp = dropout_keep(seed, site_pos, (uint64_t)b * A1 + (t % A1), thresh) ? g * dscale : 0.f;
    }
    const long long id = ids[n];
    if (id >= 0 && id < V) atomicAdd(dE + (size_t)id * d + k, ge);
    if (k < d0) atomicAdd(dW0 + (size_t)(t / A1) * d0 + k, gp);
    else atomicAdd(dW1 + (size_t)(t % A1) * (d - d0) + (k - d0), gp);
}

// The position tables' gradients without atomics in the loop (round 6).  The element-wise kernel above spends 0.48 ms of the C4 step on
// 134 M global float atomics -- 537 MB of added bytes at the chip's ~1.2 TB/s atomic rate -- and half of them go to the two
// position tables, whose row is a function of t alone: a thread here OWNS a (row, 8 columns) cell of a 32-column slab and walks
// that row's tokens (W0 row r: t = r A1 .. r A1 + A1 - 1; W1 row c: t = c, c + A1, ...) through registers, eight 16-byte loads in
// flight, with one atomic per cell at the end (the workgroups of a slab split the sequences).  Masks: the forward's one decision per
// (sequence, t % A1) column.  (Accumulating the word table's slabs in LDS the same way was tried first: ds_add_f32 ran at about one
// lane per three cycles and CU, 365 us for the table against 235 us of global atomics -- profiles/r06_experiments_not_kept.txt.)
__global__ __launch_bounds__(256) void axial_embed_bwd_pos_kernel(const bf16_t* dout, const bf16_t* dout2, float* dW0, float* dW1, int B, int T,
                                                                   int d, int A0, int A1, int d0, unsigned thresh, float dscale,
                                                                   unsigned long long seed, unsigned site_pos) {
    const int c0 = blockIdx.x * 32;
    const bool w0 = c0 < d0;
    const int rows = w0 ? A0 : A1, cnt = w0 ? A1 : A0;
    const int q = threadIdx.x & 3;
    for (int r = threadIdx.x >> 2; r < rows; r += 64) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b = blockIdx.y; b < B; b += gridDim.y) {
            if (!w0 && thresh && !dropout_keep(seed, site_pos, (uint64_t)b * A1 + r, thresh)) continue;     // W1 row r IS the column t % A1
            const bf16_t* src = dout + ((size_t)b * T) * d + c0 + 8 * q;
            const bf16_t* src2 = dout2 ? dout2 + ((size_t)b * T) * d + c0 + 8 * q : nullptr;
            for (int k0 = 0; k0 < cnt; k0 += 8) {
                bf16x8 a[8], a2[8];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + u;
                    const int t = w0 ? r * A1 + k : r + A1 * k;
                    ok[u] = k < cnt && t < T;
                    const size_t o = (size_t)(ok[u] ? t : 0) * d;
                    a[u] = *reinterpret_cast<const bf16x8*>(src + o);
                    if (src2) a2[u] = *reinterpret_cast<const bf16x8*>(src2 + o);
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (!ok[u]) continue;
                    if (w0 && thresh && !dropout_keep(seed, site_pos, (uint64_t)b * A1 + (k0 + u), thresh)) continue;   // (t % A1 = k for a W0 row)
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[j] += bf2f((bf16_t)a[u][j]) + (src2 ? bf2f((bf16_t)a2[u][j]) : 0.f);
                }
            }
        }
        float* dst = w0 ? dW0 + (size_t)r * d0 + c0 + 8 * q : dW1 + (size_t)r * (d - d0) + (c0 - d0) + 8 * q;
#pragma unroll
        for (int j = 0; j < 8; j++) if (acc[j] != 0.f) atomicAdd(dst + j, thresh ? acc[j] * dscale : acc[j]);
    }
}
// ... with it, the element-wise kernel's word-table part alone
__global__ void axial_embed_bwd_word_kernel(const long long* ids, const bf16_t* dout, const bf16_t* dout2, float* dE, long long n, int d, int V,
                                            unsigned thresh, float dscale, unsigned long long seed, unsigned site_emb) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    const long long id = ids[gid / d];
    if (id < 0 || id >= V) return;
    float g = bf2f(dout[gid]);
    if (dout2) g += bf2f(dout2[gid]);
    if (thresh) g = dropout_keep(seed, site_emb, (uint64_t)gid, thresh) ? g * dscale : 0.f;
    atomicAdd(dE + (size_t)id * d + (int)(gid % d), g);
}

// =====================================================================================================================
// LSH hashing (HF515:698-770): buckets[b,h,r*T+t] = r*NB + combine_f argmax([x R_f ; -x R_f])
// rotations (H, dh, n_h, R2) f32 given explicitly (HF draws them from the global RNG inside the layer)
// =====================================================================================================================
constexpr int MAX_R2 = 64;
struct HashGeom { int nfac; int fac[4]; };

template <int RP>   // accumulators per thread: R2 rounded up to 16 / 32 / 64 (the C4 shape, 256 buckets = 16 x 16, has R2 = 16)
__global__ __launch_bounds__(256) void lsh_hash_kernel(const bf16_t* qk, long long bs, int rs, const float* rot, int* buckets,
                                                       int B, int T, int H, int dh, int n_h, int R2, int NB, HashGeom g) {
    extern __shared__ float srot[];  // [dh][R2] for this (h, round)
    const int h = blockIdx.y % H, r = blockIdx.y / H, b = blockIdx.z;
    for (int i = threadIdx.x; i < dh * R2; i += 256) {
        const int e = i / R2, c = i % R2;
        srot[i] = rot[(((size_t)h * dh + e) * n_h + r) * R2 + c];
    }
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float acc[RP];
#pragma unroll
    for (int c = 0; c < RP; c++) acc[c] = 0.f;
    const bf16_t* x = qk + (size_t)b * bs + (size_t)t * rs + h * dh;
    // the row is read in 16-byte pieces (2-byte loads at a row pitch fetch one cache line per element); the FMA order over e
    // is unchanged, so bucket ids do not move
    for (int e8 = 0; e8 < dh; e8 += 8) {
        const bf16x8 xq = *reinterpret_cast<const bf16x8*>(x + e8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float xv = bf2f((bf16_t)xq[j]);
            const float* rr = srot + (e8 + j) * R2;
#pragma unroll
            for (int c = 0; c < RP; c++)
                if (c < R2) acc[c] = fmaf(xv, rr[c], acc[c]);
        }
    }
    int bucket = 0, cur = 0, prod = 1;
    for (int f = 0; f < g.nfac; f++) {
        const int half = g.fac[f] / 2;
        // argmax over [v_0..v_{half-1}, -v_0..-v_{half-1}], first maximum wins (torch.argmax)
        float best = -INFINITY;
        int arg = 0;
#pragma unroll
        for (int c = 0; c < RP; c++)
            if (c >= cur && c < cur + half) { const float v = acc[c]; if (v > best) { best = v; arg = c - cur; } }
#pragma unroll
        for (int c = 0; c < RP; c++)
            if (c >= cur && c < cur + half) { const float v = -acc[c]; if (v > best) { best = v; arg = half + c - cur; } }
        bucket += prod * arg;
        prod *= g.fac[f];
        cur += half;
    }
    buckets[((size_t)b * H + h) * n_h * T + (size_t)r * T + t] = r * NB + bucket;
}

// MFMA form for dh = 32 / 64: a wave hashes 16 tokens with v_mfma_f32_16x16x32_bf16.  x is bf16 already; the fp32 rotations enter
// as three bf16 terms (hi + mid + lo = 24 mantissa bits), so the products are exact to fp32 and only the summation order differs
// from the scalar kernel (a flipped near-tie in ~1e-4 of the tokens, as between any two fp32 orders).  The token rows are read as
// MFMA fragments -- 16 rows x 64 contiguous bytes per load instead of 64 rows x 16 -- and the 1024 FMAs + broadcast LDS reads per
// token of the scalar form are six MFMAs per 16 tokens.
typedef __attribute__((ext_vector_type(8))) __bf16 hash_bf16x8;
template <int NBLK>   // 16-column blocks of rotations: R2 <= 16 * NBLK
__global__ __launch_bounds__(256) void lsh_hash_mfma_kernel(const bf16_t* qk, long long bs, int rs, const float* rot, int* buckets,
                                                            int B, int T, int H, int dh, int n_h, int R2, int NB, HashGeom g) {
    constexpr int TG = 8;                                         // 16-token groups per wave: the rotation fragments are built once
    __shared__ float sacc[4][16][16 * NBLK + 1];
    const int h = blockIdx.y % H, r = blockIdx.y / H, b = blockIdx.z;
    const int wid = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, kg = l >> 4;
    const int KS = dh / 32;                                       // 1 or 2 K-steps
    // rotation fragments: row n = nb*16 + li of R^T, k = 32 ks + 8 kg .. + 7, as three bf16 terms
    bf16x8 rf[2][NBLK][3];
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int nb = 0; nb < NBLK; nb++) {
            const int n = nb * 16 + li;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int e = 32 * ks + 8 * kg + j;
                const float v = (ks < KS && n < R2) ? rot[(((size_t)h * dh + e) * n_h + r) * R2 + n] : 0.f;
                const bf16_t h0 = f2bf(v);
                const float r1 = v - bf2f(h0);
                const bf16_t h1 = f2bf(r1);
                const bf16_t h2 = f2bf(r1 - bf2f(h1));
                rf[ks][nb][0][j] = (short)h0; rf[ks][nb][1][j] = (short)h1; rf[ks][nb][2][j] = (short)h2;
            }
        }
#pragma unroll 1
    for (int grp = 0; grp < TG; grp++) {
        const int t0 = ((blockIdx.x * 4 + wid) * TG + grp) * 16;
        if (t0 >= T) break;                                       // wave-uniform
        f32x4 acc[NBLK];
#pragma unroll
        for (int nb = 0; nb < NBLK; nb++) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int tok = min(t0 + li, T - 1);
        const bf16_t* x = qk + (size_t)b * bs + (size_t)tok * rs + h * dh;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            if (ks < KS) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(x + 32 * ks + 8 * kg);
#pragma unroll
                for (int nb = 0; nb < NBLK; nb++)
#pragma unroll
                    for (int sp = 2; sp >= 0; sp--)               // smallest term first
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hash_bf16x8, rf[ks][nb][sp]),
                                                                          __builtin_bit_cast(hash_bf16x8, xf), acc[nb], 0, 0, 0);
            }
        }
        // D[n][token]: lane = token li (column), rows n = nb*16 + 4*kg + j -> one LDS row per token
#pragma unroll
        for (int nb = 0; nb < NBLK; nb++)
#pragma unroll
            for (int j = 0; j < 4; j++) sacc[wid][li][nb * 16 + 4 * kg + j] = acc[nb][j];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (kg == 0 && t0 + li < T) {
            const float* a = sacc[wid][li];
            int bucket = 0, cur = 0, prod = 1;
            for (int f = 0; f < g.nfac; f++) {
                const int half = g.fac[f] / 2;
                float best = -INFINITY;
                int arg = 0;
                for (int c = 0; c < half; c++) { const float v = a[cur + c]; if (v > best) { best = v; arg = c; } }
                for (int c = 0; c < half; c++) { const float v = -a[cur + c]; if (v > best) { best = v; arg = half + c; } }
                bucket += prod * arg;
                prod *= g.fac[f];
                cur += half;
            }
            buckets[((size_t)b * H + h) * n_h * T + (size_t)r * T + t0 + li] = r * NB + bucket;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();                          // the row is consumed before the next group overwrites it
    }
}

// =====================================================================================================================
// stable counting sort of S = n_h*T slots by bucket per (b,h) (HF515:151-157, 772-789: argsort of S*bucket + index).
// one wave per (b,h): histogram -> scan -> in-order multisplit with ballot ranking.  Outputs sidx (slot -> element index
// in [0,S)) and spos = sidx % T.
// =====================================================================================================================
__global__ __launch_bounds__(64) void lsh_sort_kernel(const int* buckets, int* sidx, int* spos, int S, int T, int NBT) {
    extern __shared__ int cnt[];  // [NBT]
    const int lane = threadIdx.x;
    const int* bk = buckets + (size_t)blockIdx.x * S;
    int* so = sidx + (size_t)blockIdx.x * S;
    int* sp = spos + (size_t)blockIdx.x * S;
    for (int i = lane; i < NBT; i += 64) cnt[i] = 0;
    __syncthreads();
    for (int i = lane; i < S; i += 64) atomicAdd(&cnt[bk[i]], 1);
    __syncthreads();
    // exclusive scan over NBT counters (wave-serial over 64-wide strips)
    int carry = 0;
    for (int base = 0; base < NBT; base += 64) {
        const int i = base + lane;
        int v = i < NBT ? cnt[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
        if (i < NBT) cnt[i] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    __syncthreads();
    int nbits = 1;
    while ((1 << nbits) < NBT) nbits++;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int base = 0; base < S; base += 64) {
        const int i = base + lane;
        const bool ok = i < S;
        const int b = ok ? bk[i] : -1;
        unsigned long long peers = __ballot(ok);
        for (int bit = 0; bit < nbits; bit++) {
            const bool on = (b >> bit) & 1;
            const unsigned long long bal = __ballot(on && ok);
            peers &= on ? bal : ~bal;
        }
        const int rank = __popcll(peers & lt);
        int off = 0;
        if (ok) off = cnt[b];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // all lanes have read cnt[] before any leader updates it (single wave)
        if (ok) {
            const int dst = off + rank;
            so[dst] = i;
            sp[dst] = i % T;
            if (rank == 0) cnt[b] = off + __popcll(peers);
        }
        __syncthreads();
    }
}

// The same sort with W waves per (b,h): wave w owns the contiguous segment [w * seg, (w+1) * seg) of the slots (seg a multiple of
// 64), histograms it into its own counters, and after one exclusive scan over (bucket-major, wave-minor) runs the in-order
// multisplit on its segment -- the result is the same stable permutation, W times as many slots in flight.
template <int W>
__global__ __launch_bounds__(64 * W) void lsh_sort_mw_kernel(const int* buckets, int* sidx, int* spos, int S, int T, int NBT) {
    extern __shared__ int cnt[];  // [W][NBT] per-wave counters, then per-wave write offsets
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int* bk = buckets + (size_t)blockIdx.x * S;
    int* so = sidx + (size_t)blockIdx.x * S;
    int* sp = spos + (size_t)blockIdx.x * S;
    const int seg = ((S + W - 1) / W + 63) & ~63;
    const int lo = w * seg, hi = min(S, lo + seg);
    int* mine = cnt + w * NBT;
    for (int i = threadIdx.x; i < W * NBT; i += 64 * W) cnt[i] = 0;
    __syncthreads();
    for (int i = lo + lane; i < hi; i += 64) atomicAdd(&mine[bk[i]], 1);
    __syncthreads();
    // exclusive scan in (bucket, wave) order by wave 0: offset[w][b] = sum over buckets < b of all waves + waves < w of bucket b
    if (w == 0) {
        int carry = 0;
        for (int base = 0; base < NBT; base += 64) {
            const int b = base + lane;
            int tot = 0;
            if (b < NBT)
                for (int ww = 0; ww < W; ww++) tot += cnt[ww * NBT + b];
            int incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
            if (b < NBT) {
                int run = carry + incl - tot;
                for (int ww = 0; ww < W; ww++) { const int c = cnt[ww * NBT + b]; cnt[ww * NBT + b] = run; run += c; }
            }
            carry += __shfl(incl, 63, 64);
        }
    }
    __syncthreads();
    int nbits = 1;
    while ((1 << nbits) < NBT) nbits++;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int base = lo; base < lo + seg; base += 64) {          // same trip count for every wave (no block barrier inside)
        const int i = base + lane;
        const bool ok = i < hi;
        const int b = ok ? bk[i] : -1;
        unsigned long long peers = __ballot(ok);
        for (int bit = 0; bit < nbits; bit++) {
            const bool on = (b >> bit) & 1;
            const unsigned long long bal = __ballot(on && ok);
            peers &= on ? bal : ~bal;
        }
        const int rank = __popcll(peers & lt);
        int off = 0;
        if (ok) off = mine[b];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // all lanes have read the wave's counters before any leader updates one (one wave)
        if (ok) {
            const int dst = off + rank;
            so[dst] = i;
            sp[dst] = i % T;
            if (rank == 0) mine[b] = off + __popcll(peers);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // the update is in LDS before the next strip of this wave reads it
    }
}

// =====================================================================================================================
// chunked attention, forward.  Slots s = 0..S-1 (sorted order for LSH, identity for local), chunk = 64 slots; queries
// of chunk c attend the 128 keys of chunks (c-1 mod NC, c) (HF515:362-383).  dots = f_k * (q . x_k) with
//   local: f = 1/sqrt(dh)                  LSH: x = shared qk, f_k = rsqrt(mean(x_k^2) + 1e-6) / sqrt(dh)   (HF515:1052-1066)
// masks on ORIGINAL positions: causal (q_pos >= k_pos) else -1e9; LSH self mask (q_pos == k_pos) -> -1e5 afterwards.
// lane = query ("swapped" S^T = K Q^T), single pass over the 128 keys.
// =====================================================================================================================
constexpr int SC_MAXT = 64;      // sequences up to one chunk take the single-chunk kernels

// Dropout on the attention probabilities of the chunk kernels.  A cell is (query slot s = (b * H + h) * S + qslot, key index kw in
// that slot's 128-key window).  ONE lowbias32 round serves a 2 x 2 block of cells -- slots (s, s ^ 1) x keys (kw, kw ^ 1): two 32-bit
// words (the hash, and a multiply-xorshift of it), the word picked by the key's parity, the 16-bit half by the slot's, against a
// 16-bit threshold (p to 1 / 65536).  The forward and the query-owner backward hold one query and two neighbouring keys per lane,
// the key-owner backward one key and two neighbouring queries: each then hashes once per two cells instead of once per cell (the
// per-cell hash on a 64-bit index was half of the ~30 VALU operations per cell these kernels spend next to 0.5 MFMA).
struct ChunkDrop {
    uint32_t key;       // block-row key: hash input of block (s >> 1, 0); block column c adds c * 0x9E3779B1
    uint32_t t16;
};
__device__ __forceinline__ ChunkDrop chunk_drop_row(unsigned long long seed, unsigned site, uint64_t slot, unsigned thresh) {
    const uint32_t mix = mxl_hash32((uint32_t)seed ^ (site * 0x9E3779B9U)) + (uint32_t)(seed >> 32);
    const uint64_t blk = (slot >> 1) * 64;
    ChunkDrop d;
    d.key = (((uint32_t)blk * 0x9E3779B1U) ^ ((uint32_t)(blk >> 32) * 0x85EBCA77U)) + mix;
    d.t16 = thresh >> 16;
    return d;
}
// the two words of block column kw >> 1
__device__ __forceinline__ void chunk_drop_words(const ChunkDrop& d, int kw, uint32_t& w0, uint32_t& w1) {
    w0 = mxl_hash32(d.key + (uint32_t)(kw >> 1) * 0x9E3779B1U);
    w1 = (w0 * 0x85EBCA77U) ^ (w0 >> 13);
}
__device__ __forceinline__ bool chunk_drop_keep(const ChunkDrop& d, uint32_t w0, uint32_t w1, int kw, int slot_parity) {
    const uint32_t w = (kw & 1) ? w1 : w0;
    return ((slot_parity ? (w >> 16) : (w & 0xffffu)) >= d.t16);
}

struct ChunkP {
    const bf16_t *q, *k, *v;
    const int* spos;            // (B,H,S) or null (identity)
    bf16_t* out;                // rows (b, round, pos) x d
    float* lse;                 // (B, n_h, H, T)
    const bf16_t* dout;         // backward: same layout as out
    const float* dlse;          // (B, n_h, H, T) or null
    float *dq, *dk, *dv;        // f32 (B, T, d) accumulators (atomicAdd)
    bf16_t *dq16, *dk16, *dv16; // n_h == 1 only: bf16 destinations (row stride ld16) that replace the f32 one when non-null
    int ld16;
    long long bs; int rs;       // q/k/v strides (elements)
    int B, T, H, S, n_h, lsh;
    float scale;
    unsigned thresh; float dscale; unsigned long long seed; unsigned site;
};

#ifndef CHUNK_BWD_WPE
#define CHUNK_BWD_WPE 3                    // waves per SIMD the backward chunk kernels are compiled for: these kernels are bound by the latency
                                           // of their gathers (sorted position -> K / V row), so a third workgroup per CU is worth more than the
                                           // five registers the key-owner kernel spills for it: 312 -> 269 us and 411 -> 369 us at C4 (same-box A/B,
                                           // profiles/r06_rf_ab1_waves_per_simd.log)
#endif
template <int DH> struct GeoC {
    static constexpr int KS = DH / 16, EB = (DH + 31) / 32, ROWB = DH * 2, CH = DH / 8;
    static constexpr int ROWS = 192;
    static constexpr int T_BYTES = ROWS * ROWB;
    static constexpr int SMEM = 2 * T_BYTES + ROWS * 8;   // K, V images + kpos[192] + kfac[192]
    __device__ static __forceinline__ int koff(int row, int ch) {
        if (DH == 64) return row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4);
        return row * ROWB + (ch << 4);
    }
    __device__ static __forceinline__ int eoff(int row, int e) { return koff(row, e >> 3) + ((e & 7) << 1); }
};

// stage the 3 key/value chunks (cprev, c0, c0+1) of this workgroup into LDS; returns nothing, all threads participate
template <int DH>
__device__ __forceinline__ void stage_kv(const ChunkP& p, char* sK, char* sV, int* sPos, float* sFac, int b, int h, int c0, int NC) {
    using G = GeoC<DH>;
    const int tid = threadIdx.x;
    const int* sp = p.spos ? p.spos + ((size_t)b * p.H + h) * p.S : nullptr;
    // three batches -- every sorted position, then every K / V row, then the LDS stores -- instead of one row at a time: a row's
    // K / V address depends on its sorted position, so row by row the staging was 2 x NIT dependent memory round trips per
    // workgroup (12 at dh = 64), which is what these gather-bound kernels spent their time on
    constexpr int NIT = (G::ROWS * G::CH + 255) / 256;
    int posv[NIT];
    bool okv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int c = tid + it * 256;
        const int row = c / G::CH;
        int chunk = c0 - 1 + (row >> 6);
        if (chunk < 0) chunk += NC;
        okv[it] = (c < G::ROWS * G::CH) && chunk < NC;
        const int slot = chunk * 64 + (row & 63);
        posv[it] = okv[it] ? (sp ? sp[slot] : slot) : 0;
    }
    u32x4 kvv[NIT], vvv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int ch = (tid + it * 256) % G::CH;
        const bf16_t* kr = p.k + (size_t)b * p.bs + (size_t)posv[it] * p.rs + h * DH + ch * 8;
        const bf16_t* vr = p.v + (size_t)b * p.bs + (size_t)posv[it] * p.rs + h * DH + ch * 8;
        kvv[it] = *reinterpret_cast<const u32x4*>(kr);          // (row 0 of the sequence for invalid rows: zeroed below)
        vvv[it] = *reinterpret_cast<const u32x4*>(vr);
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int c = tid + it * 256;
        if (c < G::ROWS * G::CH) {
            const int row = c / G::CH, ch = c % G::CH;
            const u32x4 z = {0u, 0u, 0u, 0u};
            const u32x4 kv = okv[it] ? kvv[it] : z, vv = okv[it] ? vvv[it] : z;
            *reinterpret_cast<u32x4*>(sK + G::koff(row, ch)) = kv;
            *reinterpret_cast<u32x4*>(sV + G::koff(row, ch)) = vv;
            // per-row key factor: the CH lanes holding one row are consecutive
            float ss = 0.f;
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&kv);
#pragma unroll
            for (int j = 0; j < 8; j++) { const float x = bf2f(e[j]); ss += x * x; }
            for (int o = 1; o < G::CH; o <<= 1) ss += __shfl_xor(ss, o, 64);
            if (ch == 0) {
                sPos[row] = okv[it] ? posv[it] : 0x7fffffff;   // invalid rows: position +inf -> masked by causality
                sFac[row] = p.lsh ? rsqrtf(ss / (float)DH + 1e-6f) * p.scale : p.scale;
            }
        }
    }
}

template <int DH>
__global__ __launch_bounds__(256, 1) void chunk_attn_fwd_kernel(ChunkP p) {
    using G = GeoC<DH>;
    constexpr int KS = G::KS, EB = G::EB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = sK + G::T_BYTES;
    int* sPos = reinterpret_cast<int*>(sV + G::T_BYTES);
    float* sFac = reinterpret_cast<float*>(sPos + G::ROWS);
    const int tid = threadIdx.x, wid = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
    int bx_, h, b;
    xcd_block(bx_, h, b);     // the chunk workgroups of one (head, sequence) share an XCD's L2 (neighbours share a K/V chunk)
    const int NC = p.S / 64;
    const int c0 = bx_ * 2;
    // the wave's own query row is requested BEFORE the key / value staging (its sorted position, then the row: two dependent round
    // trips that used to start only after the staging barrier)
    const int chunk = c0 + (wid >> 1);
    const int qslot = min(chunk, NC - 1) * 64 + 32 * (wid & 1) + r;
    const int* sp = p.spos ? p.spos + ((size_t)b * p.H + h) * p.S : nullptr;
    const int qpos = sp ? sp[qslot] : qslot;
    const int round = qslot / p.T;
    bf16x8 qf[KS];
    {
        const bf16_t* qp = p.q + (size_t)b * p.bs + (size_t)qpos * p.rs + h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks + 8 * hh);
    }
    stage_kv<DH>(p, sK, sV, sPos, sFac, b, h, c0, NC);
    __syncthreads();
    if (chunk >= NC) return;
    const int kb0 = 64 * (wid >> 1);                 // first LDS row of this chunk's 128 keys
    f32x16 s[4];
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
#pragma unroll
        for (int j = 0; j < 16; j++) s[kb][j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(sK + G::koff(kb0 + 32 * kb + r, 2 * ks + hh));
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                            __builtin_bit_cast(mfma_bf16x8, qf[ks]), s[kb], 0, 0, 0);
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; kb++)
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int row = kb0 + 32 * kb + (j & 3) + 8 * (j >> 2) + 4 * hh;
            const int kp = sPos[row];
            float v = s[kb][j] * sFac[row];
            v = (qpos >= kp) ? v : -1e9f;
            if (p.lsh) v = (qpos != kp) ? v : -1e5f;
            s[kb][j] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; kb++)
#pragma unroll
        for (int j = 0; j < 16; j++) { const float e = __expf(s[kb][j] - mx); s[kb][j] = e; sum += e; }
    sum += __shfl_xor(sum, 32, 64);
    // the unnormalised exponentials (<= 1) go into the P V product; 1 / sum and the dropout scale are applied to O once, at the end
    const float inv = (p.thresh ? p.dscale : 1.f) / sum;
    const size_t orow = ((size_t)b * p.n_h + round) * p.T + qpos;
    if (hh == 0 && p.lse) p.lse[(((size_t)b * p.n_h + round) * p.H + h) * p.T + qpos] = mx + __logf(sum);
    // probabilities (+ dropout), P^T fragments straight from the accumulators
    f32x16 o[EB];
#pragma unroll
    for (int e = 0; e < EB; e++)
#pragma unroll
        for (int j = 0; j < 16; j++) o[e][j] = 0.f;
    const int gq = l >> 4, li = l & 15, q4 = li >> 2, pp = li & 3;
    const ChunkDrop dr = chunk_drop_row(p.seed, p.site, ((uint64_t)b * p.H + h) * p.S + qslot, p.thresh);
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
#pragma unroll
        for (int st = 0; st < 2; st++) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                float p0 = s[kb][8 * st + j], p1 = s[kb][8 * st + j + 1];
                if (p.thresh) {
                    const int k0 = 32 * kb + ((8 * st + j) & 3) + 8 * ((8 * st + j) >> 2) + 4 * hh;      // even; k0 + 1 is its block partner
                    uint32_t w0, w1;
                    chunk_drop_words(dr, k0, w0, w1);
                    p0 = chunk_drop_keep(dr, w0, w1, k0, qslot & 1) ? p0 : 0.f;
                    p1 = chunk_drop_keep(dr, w0, w1, k0 + 1, qslot & 1) ? p1 : 0.f;
                }
                const uint32_t w = pack2bf(p0, p1);
                pf[j] = (short)(w & 0xffff); pf[j + 1] = (short)(w >> 16);
            }
#pragma unroll
            for (int e = 0; e < EB; e++) {
                const int key = kb0 + 32 * kb + 16 * st + 4 * hh + q4;
                const int ecol = 32 * e + 16 * (gq & 1) + 4 * pp;
                bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ecol < DH) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sV + G::eoff(key, ecol)));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sV + G::eoff(key + 8, ecol)));
                    a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
                o[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                               __builtin_bit_cast(mfma_bf16x8, pf), o[e], 0, 0, 0);
            }
        }
    }
    bf16_t* op = p.out + orow * (size_t)(p.H * DH) + h * DH;
#pragma unroll
    for (int e = 0; e < EB; e++)
#pragma unroll
        for (int grp = 0; grp < 4; grp++) {
            const int e0 = 32 * e + 8 * grp + 4 * hh;
            if (e0 < DH) {
                u32x2 w = {pack2bf(o[e][4 * grp] * inv, o[e][4 * grp + 1] * inv), pack2bf(o[e][4 * grp + 2] * inv, o[e][4 * grp + 3] * inv)};
                *reinterpret_cast<u32x2*>(op + e0) = w;
            }
        }
}

// =====================================================================================================================
// chunked attention, backward.  Part A (lane = query): dq.  Part B (lane = key, key-owner over chunk kc with the queries
// of chunks kc and kc+1 mod NC): dk', dv.  All three accumulate with f32 atomics into (B, T, d) buffers (a position occurs
// n_h times; for n_h = 1 every element receives exactly one add).  dk' is the gradient w.r.t. the *effective* key
// k' = f_k * x_k for LSH (the normalisation chain is applied by lsh_keynorm_bwd), and w.r.t. k for local attention.
// With dropout: O = (keep*P/(1-p)) V;  dS = P * (keep*dPd/(1-p) - delta) + dlse * P,  delta = rowsum(dO * O).
// =====================================================================================================================
template <int DH>
__global__ __launch_bounds__(256, CHUNK_BWD_WPE) void chunk_attn_bwd_q_kernel(ChunkP p) {
    using G = GeoC<DH>;
    constexpr int KS = G::KS, EB = G::EB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = sK + G::T_BYTES;
    int* sPos = reinterpret_cast<int*>(sV + G::T_BYTES);
    float* sFac = reinterpret_cast<float*>(sPos + G::ROWS);
    const int tid = threadIdx.x, wid = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
    int bx_, h, b;
    xcd_block(bx_, h, b);     // the chunk workgroups of one (head, sequence) share an XCD's L2 (neighbours share a K/V chunk)
    const int NC = p.S / 64;
    const int c0 = bx_ * 2;
    const int chunk = c0 + (wid >> 1);
    const int kb0 = 64 * (wid >> 1);
    const int qslot = min(chunk, NC - 1) * 64 + 32 * (wid & 1) + r;      // (requested before the staging: see the forward kernel)
    const int* sp = p.spos ? p.spos + ((size_t)b * p.H + h) * p.S : nullptr;
    const int qpos = sp ? sp[qslot] : qslot;
    const int round = qslot / p.T;
    const int d = p.H * DH;
    const size_t orow = ((size_t)b * p.n_h + round) * p.T + qpos;
    bf16x8 qf[KS], dof[KS];
    float dlt = 0.f;
    {
        const bf16_t* qp = p.q + (size_t)b * p.bs + (size_t)qpos * p.rs + h * DH;
        const bf16_t* dop = p.dout + orow * d + h * DH;
        const bf16_t* op = p.out + orow * d + h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks + 8 * hh);
            dof[ks] = *reinterpret_cast<const bf16x8*>(dop + 16 * ks + 8 * hh);
            const bf16x8 ov = *reinterpret_cast<const bf16x8*>(op + 16 * ks + 8 * hh);
#pragma unroll
            for (int j = 0; j < 8; j++) dlt += bf2f((bf16_t)dof[ks][j]) * bf2f((bf16_t)ov[j]);
        }
        dlt += __shfl_xor(dlt, 32, 64);
    }
    const size_t sidx_ = (((size_t)b * p.n_h + round) * p.H + h) * p.T + qpos;
    const float lse = p.lse[sidx_];
    const float dl = p.dlse ? p.dlse[sidx_] : 0.f;
    stage_kv<DH>(p, sK, sV, sPos, sFac, b, h, c0, NC);
    __syncthreads();
    if (chunk >= NC) return;
    f32x16 s[4], dp[4];
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
#pragma unroll
        for (int j = 0; j < 16; j++) { s[kb][j] = 0.f; dp[kb][j] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(sK + G::koff(kb0 + 32 * kb + r, 2 * ks + hh));
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                            __builtin_bit_cast(mfma_bf16x8, qf[ks]), s[kb], 0, 0, 0);
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(sV + G::koff(kb0 + 32 * kb + r, 2 * ks + hh));
            dp[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, av),
                                                             __builtin_bit_cast(mfma_bf16x8, dof[ks]), dp[kb], 0, 0, 0);
        }
    }
    const ChunkDrop dr = chunk_drop_row(p.seed, p.site, ((uint64_t)b * p.H + h) * p.S + qslot, p.thresh);
#pragma unroll
    for (int kb = 0; kb < 4; kb++)
#pragma unroll
        for (int j2 = 0; j2 < 16; j2 += 2) {
            uint32_t w0 = 0u, w1 = 0u;       // registers j2, j2 + 1 are neighbouring keys: one block, one hash
            if (p.thresh) chunk_drop_words(dr, 32 * kb + (j2 & 3) + 8 * (j2 >> 2) + 4 * hh, w0, w1);
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const int j = j2 + jj;
                const int kk = 32 * kb + (j & 3) + 8 * (j >> 2) + 4 * hh;
                const int row = kb0 + kk;
                const int kp = sPos[row];
                const float f = sFac[row];
                float v = s[kb][j] * f;
                v = (qpos >= kp) ? v : -1e9f;
                if (p.lsh) v = (qpos != kp) ? v : -1e5f;
                const float pr = __expf(v - lse);
                float g = dp[kb][j];
                if (p.thresh) g = chunk_drop_keep(dr, w0, w1, kk, qslot & 1) ? g * p.dscale : 0.f;
                // gradient w.r.t. the raw dot (q . x_k): dS * f_k ; masked entries have pr == 0 (or constant score -> no grad)
                const bool live = (qpos >= kp) && !(p.lsh && qpos == kp);
                s[kb][j] = live ? (pr * (g - dlt) + dl * pr) * f : 0.f;
            }
        }
    // dq^T[e, q] = sum_k X^T[e, k] * dSf^T[k, q]
    f32x16 aq[EB];
#pragma unroll
    for (int e = 0; e < EB; e++)
#pragma unroll
        for (int j = 0; j < 16; j++) aq[e][j] = 0.f;
    const int gq = l >> 4, li = l & 15, q4 = li >> 2, pp = li & 3;
#pragma unroll
    for (int kb = 0; kb < 4; kb++)
#pragma unroll
        for (int st = 0; st < 2; st++) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const uint32_t w = pack2bf(s[kb][8 * st + j], s[kb][8 * st + j + 1]);
                pf[j] = (short)(w & 0xffff); pf[j + 1] = (short)(w >> 16);
            }
#pragma unroll
            for (int e = 0; e < EB; e++) {
                const int key = kb0 + 32 * kb + 16 * st + 4 * hh + q4;
                const int ecol = 32 * e + 16 * (gq & 1) + 4 * pp;
                bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ecol < DH) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sK + G::eoff(key, ecol)));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sK + G::eoff(key + 8, ecol)));
                    a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
                aq[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                __builtin_bit_cast(mfma_bf16x8, pf), aq[e], 0, 0, 0);
            }
        }
    // the gradient row of (round, position): with several hash rounds every round has its own (T, d) slab -- each element is
    // written exactly once, with plain stores, and the rounds are summed by mxl_lsh_keynorm_bwd_rounds (until round 6 the rounds
    // met in ONE (T, d) buffer through a float atomic per element: 16 atomic instructions of 4-byte pieces per lane)
    float* dqp = p.dq + orow * d + h * DH;
    if (p.dq16) {       // n_h == 1, bf16 straight into the projection-gradient operand
        bf16_t* o16 = p.dq16 + ((size_t)b * p.T + qpos) * p.ld16 + h * DH;
#pragma unroll
        for (int e = 0; e < EB; e++)
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                const int e0 = 32 * e + 8 * grp + 4 * hh;
                if (e0 < DH)
                    *reinterpret_cast<u32x2*>(o16 + e0) = u32x2{pack2bf(aq[e][4 * grp], aq[e][4 * grp + 1]), pack2bf(aq[e][4 * grp + 2], aq[e][4 * grp + 3])};
            }
    } else {            // every (round, position, head) occurs once: plain 16-byte stores, no atomics, no pre-zeroing needed
#pragma unroll
        for (int e = 0; e < EB; e++)
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                const int e0 = 32 * e + 8 * grp + 4 * hh;
                if (e0 < DH) *reinterpret_cast<f32x4*>(dqp + e0) = f32x4{aq[e][4 * grp], aq[e][4 * grp + 1], aq[e][4 * grp + 2], aq[e][4 * grp + 3]};
            }
    }
}

// key-owner: workgroup = key chunk kc; wave w owns 32 keys of it for one of the two query chunks: w = 2*qsel + khalf
template <int DH>
__global__ __launch_bounds__(256, CHUNK_BWD_WPE) void chunk_attn_bwd_kv_kernel(ChunkP p) {
    using G = GeoC<DH>;
    constexpr int KS = G::KS, EB = G::EB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: Q image [128][DH], dO image [128][DH], qpos[128], lse[128], delta[128], dlse[128]
    char* sQ = smem;
    char* sDO = sQ + 128 * G::ROWB;
    int* sQpos = reinterpret_cast<int*>(sDO + 128 * G::ROWB);
    float* sLse = reinterpret_cast<float*>(sQpos + 128);
    float* sDl = sLse + 128;
    float* sDlse = sDl + 128;
    const int tid = threadIdx.x, wid = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
    int bx_, h, b;
    xcd_block(bx_, h, b);     // the chunk workgroups of one (head, sequence) share an XCD's L2 (neighbours share a K/V chunk)
    const int NC = p.S / 64;
    const int kc = bx_;
    const int d = p.H * DH;
    const int* sp = p.spos ? p.spos + ((size_t)b * p.H + h) * p.S : nullptr;
    // the wave's own key rows first (sorted position, then K / V), then the 128 queries in three batches -- positions, rows, stores
    // -- instead of row by row (each row was two to three dependent round trips: see stage_kv)
    const int qsel = wid >> 1;                 // 0: queries of chunk kc (keys are their "current" chunk), 1: chunk kc+1
    const int kslot = kc * 64 + 32 * (wid & 1) + r;
    const int kpos = sp ? sp[kslot] : kslot;
    bf16x8 kf[KS], vf[KS];
    {
        const bf16_t* kp = p.k + (size_t)b * p.bs + (size_t)kpos * p.rs + h * DH;
        const bf16_t* vp = p.v + (size_t)b * p.bs + (size_t)kpos * p.rs + h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(kp + 16 * ks + 8 * hh);
            vf[ks] = *reinterpret_cast<const bf16x8*>(vp + 16 * ks + 8 * hh);
        }
    }
    // stage the 128 queries: rows 0-63 = chunk kc, rows 64-127 = chunk kc+1 (mod NC)
    constexpr int NIT = (128 * G::CH + 255) / 256;
    int posq[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int c = min(tid + it * 256, 128 * G::CH - 1);
        const int row = c / G::CH;
        int chunk = kc + (row >> 6);
        if (chunk >= NC) chunk -= NC;
        const int slot = chunk * 64 + (row & 63);
        posq[it] = sp ? sp[slot] : slot;
    }
    u32x4 qv_[NIT], dv_[NIT], ov_[NIT];
    float lse_[NIT], dlse_[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int c = min(tid + it * 256, 128 * G::CH - 1);
        const int row = c / G::CH, ch = c % G::CH;
        int chunk = kc + (row >> 6);
        if (chunk >= NC) chunk -= NC;
        const int slot = chunk * 64 + (row & 63);
        const int round = slot / p.T;
        const size_t orow = ((size_t)b * p.n_h + round) * p.T + posq[it];
        qv_[it] = *reinterpret_cast<const u32x4*>(p.q + (size_t)b * p.bs + (size_t)posq[it] * p.rs + h * DH + ch * 8);
        dv_[it] = *reinterpret_cast<const u32x4*>(p.dout + orow * d + h * DH + ch * 8);
        ov_[it] = *reinterpret_cast<const u32x4*>(p.out + orow * d + h * DH + ch * 8);
        const size_t si = (((size_t)b * p.n_h + round) * p.H + h) * p.T + posq[it];
        lse_[it] = p.lse[si];
        dlse_[it] = p.dlse ? p.dlse[si] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int c = tid + it * 256;
        if (c < 128 * G::CH) {
            const int row = c / G::CH, ch = c % G::CH;
            *reinterpret_cast<u32x4*>(sQ + G::koff(row, ch)) = qv_[it];
            *reinterpret_cast<u32x4*>(sDO + G::koff(row, ch)) = dv_[it];
            float dl = 0.f;
            const bf16_t* a = reinterpret_cast<const bf16_t*>(&dv_[it]);
            const bf16_t* o = reinterpret_cast<const bf16_t*>(&ov_[it]);
#pragma unroll
            for (int j = 0; j < 8; j++) dl += bf2f(a[j]) * bf2f(o[j]);
            for (int o2 = 1; o2 < G::CH; o2 <<= 1) dl += __shfl_xor(dl, o2, 64);
            if (ch == 0) {
                sQpos[row] = posq[it];
                sLse[row] = lse_[it];
                sDl[row] = dl;
                sDlse[row] = dlse_[it];
            }
        }
    }
    __syncthreads();
    // with a single chunk (NC == 1) the "previous" chunk is the chunk itself: HF concatenates it twice; keep both
    float ss = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int j = 0; j < 8; j++) { const float x = bf2f((bf16_t)kf[ks][j]); ss += x * x; }
    ss += __shfl_xor(ss, 32, 64);
    const float fac = p.lsh ? rsqrtf(ss / (float)DH + 1e-6f) * p.scale : p.scale;
    f32x16 ak[EB], av[EB];
#pragma unroll
    for (int e = 0; e < EB; e++)
#pragma unroll
        for (int j = 0; j < 16; j++) { ak[e][j] = 0.f; av[e][j] = 0.f; }
    const int gq = l >> 4, li = l & 15, q4 = li >> 2, pp = li & 3;
    // index of this key inside the 128-key window of the query chunk (for the dropout counter): chunk kc is the
    // "current" chunk for queries of kc (window index 64 + x) and the "previous" chunk for queries of kc+1 (index x)
    const int kwin = (qsel == 0 ? 64 : 0) + 32 * (wid & 1) + r;
#pragma unroll 1
    for (int qb = 0; qb < 2; qb++) {           // two 32-query blocks of the selected query chunk
        const int q0 = 64 * qsel + 32 * qb;    // LDS row of the first query
        f32x16 s, dp;
#pragma unroll
        for (int j = 0; j < 16; j++) { s[j] = 0.f; dp[j] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(sQ + G::koff(q0 + r, 2 * ks + hh));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                        __builtin_bit_cast(mfma_bf16x8, kf[ks]), s, 0, 0, 0);
            const bf16x8 ad = *reinterpret_cast<const bf16x8*>(sDO + G::koff(q0 + r, 2 * ks + hh));
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, ad),
                                                         __builtin_bit_cast(mfma_bf16x8, vf[ks]), dp, 0, 0, 0);
        }
        f32x16 pr;
#pragma unroll
        for (int j2 = 0; j2 < 16; j2 += 2) {
            int chunkq = kc + qsel;
            if (chunkq >= NC) chunkq -= NC;
            // registers j2, j2 + 1 are neighbouring query slots (even, odd): one block row, one hash
            uint32_t w0 = 0u, w1 = 0u;
            ChunkDrop dr;
            dr.key = 0u; dr.t16 = 0u;
            if (p.thresh) {
                const int qs0 = chunkq * 64 + 32 * qb + (j2 & 3) + 8 * (j2 >> 2) + 4 * hh;
                dr = chunk_drop_row(p.seed, p.site, ((uint64_t)b * p.H + h) * p.S + qs0, p.thresh);
                chunk_drop_words(dr, kwin, w0, w1);
            }
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const int j = j2 + jj;
                const int ii = (j & 3) + 8 * (j >> 2) + 4 * hh;
                const int qrow = q0 + ii;
                const int qp = sQpos[qrow];
                const int qslot = chunkq * 64 + 32 * qb + ii;
                float v = s[j] * fac;
                v = (qp >= kpos) ? v : -1e9f;
                if (p.lsh) v = (qp != kpos) ? v : -1e5f;
                const float pv = __expf(v - sLse[qrow]);
                float g = dp[j];
                float pd = pv;
                if (p.thresh) {
                    const bool keep = chunk_drop_keep(dr, w0, w1, kwin, qslot & 1);
                    g = keep ? g * p.dscale : 0.f;
                    pd = keep ? pv * p.dscale : 0.f;
                }
                const bool live = (qp >= kpos) && !(p.lsh && qp == kpos);
                const float ds = live ? (pv * (g - sDl[qrow]) + sDlse[qrow] * pv) : 0.f;
                pr[j] = pd;
                s[j] = p.lsh ? ds : ds * fac;      // local: d k = dS * scale * q ; LSH: d k' = dS * q (chain applied later)
            }
        }
#pragma unroll
        for (int st = 0; st < 2; st++) {
            bf16x8 pf, df;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const uint32_t w = pack2bf(pr[8 * st + j], pr[8 * st + j + 1]);
                pf[j] = (short)(w & 0xffff); pf[j + 1] = (short)(w >> 16);
                const uint32_t w2 = pack2bf(s[8 * st + j], s[8 * st + j + 1]);
                df[j] = (short)(w2 & 0xffff); df[j + 1] = (short)(w2 >> 16);
            }
#pragma unroll
            for (int e = 0; e < EB; e++) {
                const int qrow = q0 + 16 * st + 4 * hh + q4;
                const int ecol = 32 * e + 16 * (gq & 1) + 4 * pp;
                bf16x8 a1 = {0, 0, 0, 0, 0, 0, 0, 0}, a2 = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ecol < DH) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sDO + G::eoff(qrow, ecol)));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sDO + G::eoff(qrow + 8, ecol)));
                    a1 = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const bf16x4 lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sQ + G::eoff(qrow, ecol)));
                    const bf16x4 hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sQ + G::eoff(qrow + 8, ecol)));
                    a2 = bf16x8{lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
                }
                av[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a1),
                                                                __builtin_bit_cast(mfma_bf16x8, pf), av[e], 0, 0, 0);
                ak[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a2),
                                                                __builtin_bit_cast(mfma_bf16x8, df), ak[e], 0, 0, 0);
            }
        }
    }
    // the two query chunks that see this key live in different waves of the workgroup (qsel): combine them through LDS so
    // each (round, key, e) is written once, with plain stores
    __syncthreads();
    constexpr int RS = DH + 4;                        // padded row: spreads the 32 key rows of a wave over the banks
    float* red = reinterpret_cast<float*>(smem);     // [2 kinds][64 keys][RS] f32 (re-uses the whole query staging area)
    const int krow = 32 * (wid & 1) + r;
    if (qsel == 1) {
#pragma unroll
        for (int e = 0; e < EB; e++)
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int ee = 32 * e + (j & 3) + 8 * (j >> 2) + 4 * hh;
                if (ee < DH) { red[krow * RS + ee] = ak[e][j]; red[64 * RS + krow * RS + ee] = av[e][j]; }
            }
    }
    __syncthreads();
    if (qsel == 0) {
        const size_t krow_g = ((size_t)b * p.n_h + kslot / p.T) * p.T + kpos;      // (round, position) row: see the query-owner kernel
        float* dkp = p.dk + krow_g * d + h * DH;
        float* dvp = p.dv + krow_g * d + h * DH;
#pragma unroll
        for (int e = 0; e < EB; e++)
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                const int e0 = 32 * e + 8 * grp + 4 * hh;
                if (e0 < DH) {
                    f32x4 vk, vv;
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        vk[t] = ak[e][4 * grp + t] + red[krow * RS + e0 + t];
                        vv[t] = av[e][4 * grp + t] + red[64 * RS + krow * RS + e0 + t];
                    }
                    const size_t o16 = ((size_t)b * p.T + kpos) * p.ld16 + h * DH + e0;
                    if (p.dk16) *reinterpret_cast<u32x2*>(p.dk16 + o16) = u32x2{pack2bf(vk[0], vk[1]), pack2bf(vk[2], vk[3])};
                    else *reinterpret_cast<f32x4*>(dkp + e0) = vk;
                    if (p.dv16) *reinterpret_cast<u32x2*>(p.dv16 + o16) = u32x2{pack2bf(vv[0], vv[1]), pack2bf(vv[2], vv[3])};
                    else *reinterpret_cast<f32x4*>(dvp + e0) = vv;
                }
            }
    }
}

// =====================================================================================================================
// LSH glue: key-normalisation chain, hash-round combine (HF515:636-655) and its backward
// =====================================================================================================================
// dqk[n][h][:] = dq + f * dk' - x * (sum_e dk'_e x_e) * (m + eps)^(-3/2) / dh^(3/2),  m = mean(x^2),  f = (m+eps)^(-1/2)/sqrt(dh)
// n_h > 1: dq, dk' (and dv) are per-round slabs (B, n_h, T, d), summed here in round order; dv's sum leaves as bf16 through dv16
__global__ void lsh_keynorm_bwd_kernel(const bf16_t* qk, long long bs, int rs, const float* dq, const float* dkp, bf16_t* dqk,
                                       int ld_dqk, int B, int T, int H, int dh, int n_h, const float* dv, bf16_t* dv16, int ld_dv) {
    // dh/8 consecutive lanes own one (b, t, h) vector: 16-byte bf16 loads, 2 x 16-byte fp32 loads per operand
    const int lpv = dh >> 3;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long vec = gid / lpv;
    const int c = (int)(gid % lpv);
    const bool ok = vec < (long long)B * T * H;
    const long long v_ = ok ? vec : 0;
    const int h = (int)(v_ % H);
    const long long n = v_ / H;
    const int t = (int)(n % T), b = (int)(n / T);
    const bf16x8 xv = *reinterpret_cast<const bf16x8*>(qk + (size_t)b * bs + (size_t)t * rs + h * dh + c * 8);
    const size_t rstride = (size_t)T * H * dh;                                                   // one round's slab
    const size_t o = ((size_t)b * n_h * T + t) * (size_t)(H * dh) + (size_t)h * dh + c * 8;      // round 0
    f32x4 k0 = *reinterpret_cast<const f32x4*>(dkp + o), k1 = *reinterpret_cast<const f32x4*>(dkp + o + 4);
    f32x4 q0 = *reinterpret_cast<const f32x4*>(dq + o), q1 = *reinterpret_cast<const f32x4*>(dq + o + 4);
    for (int r = 1; r < n_h; r++) {
        k0 += *reinterpret_cast<const f32x4*>(dkp + o + r * rstride); k1 += *reinterpret_cast<const f32x4*>(dkp + o + r * rstride + 4);
        q0 += *reinterpret_cast<const f32x4*>(dq + o + r * rstride); q1 += *reinterpret_cast<const f32x4*>(dq + o + r * rstride + 4);
    }
    if (dv16 && ok) {
        f32x4 v0 = *reinterpret_cast<const f32x4*>(dv + o), v1 = *reinterpret_cast<const f32x4*>(dv + o + 4);
        for (int r = 1; r < n_h; r++) {
            v0 += *reinterpret_cast<const f32x4*>(dv + o + r * rstride); v1 += *reinterpret_cast<const f32x4*>(dv + o + r * rstride + 4);
        }
        const u32x4 wv = {pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3]), pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3])};
        *reinterpret_cast<u32x4*>(dv16 + (size_t)n * ld_dv + (size_t)h * dh + c * 8) = wv;
    }
    float x[8], dk[8] = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
    float dqv[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
    float m = 0.f, dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) { x[j] = bf2f((bf16_t)xv[j]); m += x[j] * x[j]; dot += dk[j] * x[j]; }
    for (int off = 1; off < lpv; off <<= 1) { m += __shfl_xor(m, off, 64); dot += __shfl_xor(dot, off, 64); }
    m = m / (float)dh + 1e-6f;
    const float rs_ = rsqrtf(m);
    const float f = rs_ * rsqrtf((float)dh);
    const float cc = dot * rs_ * rs_ * rs_ / ((float)dh * sqrtf((float)dh));
    if (ok) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = dqv[j] + f * dk[j] - x[j] * cc;
        u32x4 w = {pack2bf(r[0], r[1]), pack2bf(r[2], r[3]), pack2bf(r[4], r[5]), pack2bf(r[6], r[7])};
        *reinterpret_cast<u32x4*>(dqk + (size_t)n * ld_dqk + (size_t)h * dh + c * 8) = w;
    }
}

// out[b,t,h,:] = sum_r w_r out_r[b,r,t,h,:],  w = softmax_r(lse[b,r,h,t]); a thread owns 8 consecutive elements of one (b, t, h)
// vector (16-byte loads and stores; the round weights once per thread instead of once per element)
__global__ void lsh_combine_kernel(const bf16_t* out_r, const float* lse, bf16_t* out, int B, int T, int H, int dh, int n_h) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = H * dh, cpr = d >> 3;
    if (gid >= (long long)B * T * cpr) return;
    const int c = (int)(gid % cpr), h = (c * 8) / dh;
    const long long n = gid / cpr;
    const int t = (int)(n % T), b = (int)(n / T);
    float mx = -INFINITY;
    for (int r = 0; r < n_h; r++) mx = fmaxf(mx, lse[(((size_t)b * n_h + r) * H + h) * T + t]);
    float den = 0.f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    for (int r = 0; r < n_h; r++) {
        const float w = __expf(lse[(((size_t)b * n_h + r) * H + h) * T + t] - mx);
        den += w;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(out_r + (((size_t)b * n_h + r) * T + t) * d + c * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] += w * bf2f((bf16_t)v[j]);
    }
    const u32x4 wv = {pack2bf(acc[0] / den, acc[1] / den), pack2bf(acc[2] / den, acc[3] / den), pack2bf(acc[4] / den, acc[5] / den),
                      pack2bf(acc[6] / den, acc[7] / den)};
    *reinterpret_cast<u32x4*>(out + (size_t)n * d + c * 8) = wv;
}

// dout_r = w_r * dout ;  dlse_r = w_r * sum_e dout_e (out_r,e - out_e).  dh / 8 consecutive lanes own one (b, t, h) vector: 16-byte
// loads and stores, the dot product by a lane reduction.  (Until round 6: one thread per vector walking its dh elements with 2-byte
// accesses, 128 bytes apart from its neighbour's -- 6.7 ms per call at the reference's logged shape, a quarter of that step.)
__global__ void lsh_combine_bwd_kernel(const bf16_t* out_r, const float* lse, const bf16_t* out, const bf16_t* dout,
                                       bf16_t* dout_r, float* dlse, int B, int T, int H, int dh, int n_h) {
    const int lpv = dh >> 3;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long vec = gid / lpv;
    const int c = (int)(gid % lpv);
    const bool ok = vec < (long long)B * T * H;
    const long long v_ = ok ? vec : 0;
    const int h = (int)(v_ % H);
    const long long n = v_ / H;
    const int t = (int)(n % T), b = (int)(n / T);
    const int d = H * dh;
    float mx = -INFINITY;
    for (int r = 0; r < n_h; r++) mx = fmaxf(mx, lse[(((size_t)b * n_h + r) * H + h) * T + t]);
    float den = 0.f;
    for (int r = 0; r < n_h; r++) den += __expf(lse[(((size_t)b * n_h + r) * H + h) * T + t] - mx);
    const size_t o = (size_t)n * d + (size_t)h * dh + c * 8;
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(dout + o), ov = *reinterpret_cast<const bf16x8*>(out + o);
    float g[8], oo[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { g[j] = bf2f((bf16_t)gv[j]); oo[j] = bf2f((bf16_t)ov[j]); }
    for (int r = 0; r < n_h; r++) {
        const size_t si = (((size_t)b * n_h + r) * H + h) * T + t;
        const float w = __expf(lse[si] - mx) / den;
        const size_t orr = (((size_t)b * n_h + r) * T + t) * d + (size_t)h * dh + c * 8;
        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(out_r + orr);
        float dot = 0.f, wg[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { dot += g[j] * (bf2f((bf16_t)rv[j]) - oo[j]); wg[j] = w * g[j]; }
        for (int off = 1; off < lpv; off <<= 1) dot += __shfl_xor(dot, off, 64);
        if (ok) {
            const u32x4 wv = {pack2bf(wg[0], wg[1]), pack2bf(wg[2], wg[3]), pack2bf(wg[4], wg[5]), pack2bf(wg[6], wg[7])};
            *reinterpret_cast<u32x4*>(dout_r + orr) = wv;
            if (c == 0) dlse[si] = w * dot;
        }
    }
}

template <int DH>
int launch_chunk(const ChunkP& p, int mode, hipStream_t s) {
    using G = GeoC<DH>;
    static bool attr = false;
    const int kv_smem = 2 * 128 * G::ROWB + 128 * 16;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_attn_fwd_kernel<DH>), hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_attn_bwd_q_kernel<DH>), hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const int NC = p.S / 64;
    if (mode == 0) {
        mxl_kt::Scope kt(MXL_KT_CHUNK_FWD, s);
        hipLaunchKernelGGL((chunk_attn_fwd_kernel<DH>), dim3((NC + 1) / 2, p.H, p.B), dim3(256), G::SMEM, s, p);
    } else {
        {
            mxl_kt::Scope kt(MXL_KT_CHUNK_BWD_Q, s);
            hipLaunchKernelGGL((chunk_attn_bwd_q_kernel<DH>), dim3((NC + 1) / 2, p.H, p.B), dim3(256), G::SMEM, s, p);
        }
        mxl_kt::Scope kt(MXL_KT_CHUNK_BWD_KV, s);
        hipLaunchKernelGGL((chunk_attn_bwd_kv_kernel<DH>), dim3(NC, p.H, p.B), dim3(256), kv_smem, s, p);
    }
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

int fill_chunk(ChunkP& p, const void* q, const void* k, const void* v, const int* spos, void* out, float* lse, int B, int T,
               int H, int dh, int n_h, int lsh, long long bs, int rs, float drop_p, unsigned long long seed, unsigned site) {
    if (!(q && k && v && out && lse)) return MXL_EINVAL;
    if (!(B > 0 && T > 0 && H > 0 && n_h >= 1)) return MXL_EINVAL;
    if (T <= SC_MAXT) { if (n_h != 1) return MXL_EINVAL; }     // single chunk: one plain attention, no hash rounds
    else if ((T % 64) != 0) return MXL_EINVAL;
    if ((rs % 8) || (bs % 8)) return MXL_EINVAL;
    if (n_h > 1 && !lsh) return MXL_EINVAL;
    if (lsh && !spos && T > SC_MAXT) return MXL_EINVAL;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.spos = spos; p.out = (bf16_t*)out; p.lse = lse;
    p.dout = nullptr; p.dlse = nullptr; p.dq = p.dk = p.dv = nullptr;
    p.dq16 = p.dk16 = p.dv16 = nullptr; p.ld16 = 0;
    p.bs = bs; p.rs = rs; p.B = B; p.T = T; p.H = H; p.S = n_h * T; p.n_h = n_h; p.lsh = lsh;
    p.scale = 1.f / sqrtf((float)dh);
    p.thresh = dropout_thresh(drop_p); p.dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; p.seed = seed; p.site = site;
    return MXL_OK;
}

// =====================================================================================================================
// Single-chunk path (T <= chunk length): HF does plain causal attention over the T tokens -- no hashing, no sort, no look-back
// chunk (HF515:547-549, 1244-1262 `do_standard_self_attention`; local layers the same, HF515:1424-1437).  Sizes are tiny
// (the reference's `debug` preset, max_length 64): one workgroup per (b, h), everything in LDS as fp32, no MFMA.
//   local: s_ij = q_i . k_j * scale                     LSH: s_ij = qk_i . (qk_j * rsqrt(mean(qk_j^2) + 1e-6) * scale), self -> -1e5
//   causal -1e9, softmax, dropout on the probabilities, out_i = sum_j p_ij v_j
// Backward writes dq, dk (LSH: w.r.t. the normalised key, like the chunked kernels), dv in fp32 (plain stores).
// =====================================================================================================================
template <int DH, bool BWD>
__global__ __launch_bounds__(256) void single_attn_kernel(ChunkP p) {
    constexpr int LD = DH + 1;
    extern __shared__ float sm[];
    float* sq = sm;                       // [64][LD]
    float* sk = sq + SC_MAXT * LD;        // [64][LD]  (already multiplied by its per-key factor)
    float* sv = sk + SC_MAXT * LD;        // [64][LD]
    float* sp = sv + SC_MAXT * LD;        // [64][65]  probabilities (pre-dropout)
    float* sdo = sp + SC_MAXT * 65;       // [64][LD]  (backward) dO
    float* sds = sdo + SC_MAXT * LD;      // [64][65]  (backward) dS
    const int h = blockIdx.x, b = blockIdx.y, T = p.T, d = p.H * DH, tid = threadIdx.x;
    for (int i = tid; i < T * DH; i += 256) {
        const int t = i / DH, e = i % DH;
        const size_t off = (size_t)b * p.bs + (size_t)t * p.rs + h * DH + e;
        sq[t * LD + e] = bf2f(p.q[off]);
        sk[t * LD + e] = bf2f(p.k[off]);
        sv[t * LD + e] = bf2f(p.v[off]);
        if (BWD) sdo[t * LD + e] = bf2f(p.dout[((size_t)b * T + t) * d + h * DH + e]);
    }
    __syncthreads();
    if (tid < T) {                         // per-key factor folded into the key rows
        float f = p.scale;
        if (p.lsh) {
            float ss = 0.f;
            for (int e = 0; e < DH; e++) ss += sk[tid * LD + e] * sk[tid * LD + e];
            f = rsqrtf(ss / (float)DH + 1e-6f) * p.scale;
        }
        for (int e = 0; e < DH; e++) sk[tid * LD + e] *= f;
    }
    __syncthreads();
    for (int idx = tid; idx < T * T; idx += 256) {
        const int i = idx / T, j = idx % T;
        float a = 0.f;
        for (int e = 0; e < DH; e++) a += sq[i * LD + e] * sk[j * LD + e];
        a = (i >= j) ? a : -1e9f;
        if (p.lsh) a = (i != j) ? a : -1e5f;
        sp[i * 65 + j] = a;
    }
    __syncthreads();
    if (tid < T) {
        float m = -INFINITY, z = 0.f;
        for (int j = 0; j < T; j++) m = fmaxf(m, sp[tid * 65 + j]);
        for (int j = 0; j < T; j++) z += __expf(sp[tid * 65 + j] - m);
        const float l = m + __logf(z);
        for (int j = 0; j < T; j++) sp[tid * 65 + j] = __expf(sp[tid * 65 + j] - l);
        if (!BWD && p.lse) p.lse[((size_t)b * p.H + h) * T + tid] = l;
    }
    __syncthreads();
    auto keep = [&](int i, int j) -> float {
        if (!p.thresh) return 1.f;
        return dropout_keep(p.seed, p.site, (((uint64_t)b * p.H + h) * T + i) * 64 + j, p.thresh) ? p.dscale : 0.f;
    };
    if (!BWD) {
        for (int idx = tid; idx < T * DH; idx += 256) {
            const int i = idx / DH, e = idx % DH;
            float a = 0.f;
            for (int j = 0; j <= i; j++) a += sp[i * 65 + j] * keep(i, j) * sv[j * LD + e];
            p.out[((size_t)b * T + i) * d + h * DH + e] = f2bf(a);
        }
        return;
    }
    // g_ij = keep * dO_i . v_j ;  dS_ij = p_ij (g_ij - sum_j' p_ij' g_ij')
    for (int idx = tid; idx < T * T; idx += 256) {
        const int i = idx / T, j = idx % T;
        float a = 0.f;
        for (int e = 0; e < DH; e++) a += sdo[i * LD + e] * sv[j * LD + e];
        sds[i * 65 + j] = a * keep(i, j);
    }
    __syncthreads();
    if (tid < T) {
        float dl = 0.f;
        for (int j = 0; j < T; j++) dl += sp[tid * 65 + j] * sds[tid * 65 + j];
        for (int j = 0; j < T; j++) {
            const bool live = (tid >= j) && !(p.lsh && tid == j);
            sds[tid * 65 + j] = live ? sp[tid * 65 + j] * (sds[tid * 65 + j] - dl) : 0.f;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < T * DH; idx += 256) {
        const int t = idx / DH, e = idx % DH;
        float aq = 0.f, ak = 0.f, av = 0.f;
        for (int j = 0; j < T; j++) {
            aq += sds[t * 65 + j] * sk[j * LD + e];                       // dq_t  = sum_j dS_tj k'_j
            ak += sds[j * 65 + t] * sq[j * LD + e];                       // dk'_t = sum_i dS_it q_i
            av += sp[j * 65 + t] * keep(j, t) * sdo[j * LD + e];          // dv_t  = sum_i pd_it dO_i
        }
        const size_t o = ((size_t)b * T + t) * d + h * DH + e;
        const size_t o16 = ((size_t)b * T + t) * p.ld16 + h * DH + e;
        ak = p.lsh ? ak : ak * p.scale;           // local: the factor was folded into k', undo to get d k
        if (p.dq16) p.dq16[o16] = f2bf(aq); else p.dq[o] = aq;
        if (p.dk16) p.dk16[o16] = f2bf(ak); else p.dk[o] = ak;
        if (p.dv16) p.dv16[o16] = f2bf(av); else p.dv[o] = av;
    }
}

template <int DH>
int launch_single(const ChunkP& p, int bwd, hipStream_t s) {
    const size_t smem = (size_t)(4 * SC_MAXT * (DH + 1) + 2 * SC_MAXT * 65) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&single_attn_kernel<DH, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&single_attn_kernel<DH, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (bwd) hipLaunchKernelGGL((single_attn_kernel<DH, true>), dim3(p.H, p.B), dim3(256), smem, s, p);
    else hipLaunchKernelGGL((single_attn_kernel<DH, false>), dim3(p.H, p.B), dim3(256), smem, s, p);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

}  // namespace

extern "C" int mxl_axial_embed_fwd(const void* ids, const void* E, const float* W0, const float* W1, void* out, int B, int T,
                                   int d, int V, int A0, int A1, int d0, float drop_p, unsigned long long seed,
                                   unsigned site_emb, unsigned site_pos, void* stream) {
    MXL_CHECK_ARG(ids && E && W0 && W1 && out && B > 0 && T > 0 && d > 0 && d0 > 0 && d0 < d && T <= A0 * A1);
    const long long n = (long long)B * T * d;
    hipLaunchKernelGGL(axial_embed_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)ids, (const bf16_t*)E, W0, W1, (bf16_t*)out, B, T, d, V, A1, d0, dropout_thresh(drop_p),
                       drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, site_emb, site_pos);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_axial_embed_bwd(const void* ids, const void* dout, const void* dout2, float* dE, float* dW0, float* dW1,
                                   int B, int T, int d, int V, int A0, int A1, int d0, float drop_p, unsigned long long seed,
                                   unsigned site_emb, unsigned site_pos, void* stream) {
    MXL_CHECK_ARG(ids && dout && dE && dW0 && dW1 && B > 0 && T > 0 && d0 > 0 && d0 < d && T <= A0 * A1);
    const long long n = (long long)B * T * d;
    // position tables by row-owning threads (32-column slabs that do not straddle d0), the word table by the element-wise atomics
    const bool split_off = getenv("MXL_AXIAL_BWD_GLOBAL") != nullptr;          // (read per call: the test switches forms)
    if (!split_off && (d % 32) == 0 && (d0 % 32) == 0 && (long long)B * T >= 4096 && ((uintptr_t)dout % 16) == 0 &&
        (!dout2 || ((uintptr_t)dout2 % 16) == 0)) {
        const unsigned thresh = dropout_thresh(drop_p);
        const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
        hipLaunchKernelGGL(axial_embed_bwd_word_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const long long*)ids,
                           (const bf16_t*)dout, (const bf16_t*)dout2, dE, n, d, V, thresh, dscale, seed, site_emb);
        hipLaunchKernelGGL(axial_embed_bwd_pos_kernel, dim3(d / 32, B < 16 ? B : 16), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout,
                           (const bf16_t*)dout2, dW0, dW1, B, T, d, A0, A1, d0, thresh, dscale, seed, site_pos);
        MXL_LAUNCH_CHECK();
        return MXL_OK;
    }
    hipLaunchKernelGGL(axial_embed_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)ids, (const bf16_t*)dout, (const bf16_t*)dout2, dE, dW0, dW1, B, T, d, V, A1, d0,
                       dropout_thresh(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, site_emb, site_pos);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_lsh_hash(const void* qk, long long bs, int rs, const float* rotations, int* buckets, int B, int T, int H,
                            int dh, int n_h, int nfac, const int* factors_host, void* stream) {
    MXL_CHECK_ARG(qk && rotations && buckets && factors_host && B > 0 && T > 0 && H > 0 && n_h >= 1 && nfac >= 1 && nfac <= 4);
    HashGeom g;
    g.nfac = nfac;
    int R2 = 0, NB = 1;
    for (int i = 0; i < 4; i++) g.fac[i] = 2;
    for (int i = 0; i < nfac; i++) {
        MXL_CHECK_ARG(factors_host[i] >= 2 && (factors_host[i] % 2) == 0);
        g.fac[i] = factors_host[i]; R2 += factors_host[i] / 2; NB *= factors_host[i];
    }
    MXL_CHECK_ARG(R2 <= MAX_R2 && (size_t)dh * R2 * 4 <= 48 * 1024);
    if ((dh == 32 || dh == 64) && (rs % 8) == 0 && (bs % 8) == 0 && ((uintptr_t)qk % 16) == 0) {
        const auto kfm = R2 <= 16 ? lsh_hash_mfma_kernel<1> : R2 <= 32 ? lsh_hash_mfma_kernel<2> : lsh_hash_mfma_kernel<4>;
        hipLaunchKernelGGL(kfm, dim3((T + 511) / 512, H * n_h, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qk, bs, rs,
                           rotations, buckets, B, T, H, dh, n_h, R2, NB, g);
        MXL_LAUNCH_CHECK();
        return MXL_OK;
    }
    const auto kfn = R2 <= 16 ? lsh_hash_kernel<16> : R2 <= 32 ? lsh_hash_kernel<32> : lsh_hash_kernel<64>;
    hipLaunchKernelGGL(kfn, dim3((T + 255) / 256, H * n_h, B), dim3(256), (size_t)dh * R2 * 4, (hipStream_t)stream,
                       (const bf16_t*)qk, bs, rs, rotations, buckets, B, T, H, dh, n_h, R2, NB, g);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_lsh_sort(const int* buckets, int* sorted_idx, int* sorted_pos, int BH, int S, int T, int n_buckets_total,
                            void* stream) {
    MXL_CHECK_ARG(buckets && sorted_idx && sorted_pos && BH > 0 && S > 0 && T > 0 && n_buckets_total > 0 && n_buckets_total <= 8192);
    // long rows: 8 waves per row (the single-wave form leaves the chip to B*H waves); their counters: 8 * NBT ints of LDS
    if (S >= 2048 && n_buckets_total <= 1024)
        hipLaunchKernelGGL(lsh_sort_mw_kernel<8>, dim3(BH), dim3(512), (size_t)8 * n_buckets_total * 4, (hipStream_t)stream, buckets,
                           sorted_idx, sorted_pos, S, T, n_buckets_total);
    else
    hipLaunchKernelGGL(lsh_sort_kernel, dim3(BH), dim3(64), (size_t)n_buckets_total * 4, (hipStream_t)stream, buckets, sorted_idx,
                       sorted_pos, S, T, n_buckets_total);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_chunk_attn_fwd(const void* q, const void* k, const void* v, const int* sorted_pos, void* out, float* lse,
                                  int B, int T, int H, int dh, int n_h, int lsh, long long bs, int rs, float drop_p,
                                  unsigned long long seed, unsigned site, void* stream) {
    ChunkP p;
    int rc = fill_chunk(p, q, k, v, sorted_pos, out, lse, B, T, H, dh, n_h, lsh, bs, rs, drop_p, seed, site);
    if (rc) return rc;
    if (T <= SC_MAXT) {
        switch (dh) {
            case 16: return launch_single<16>(p, 0, (hipStream_t)stream);
            case 32: return launch_single<32>(p, 0, (hipStream_t)stream);
            case 64: return launch_single<64>(p, 0, (hipStream_t)stream);
            default: return MXL_EUNSUPPORTED;
        }
    }
    switch (dh) {
        case 16: return launch_chunk<16>(p, 0, (hipStream_t)stream);
        case 32: return launch_chunk<32>(p, 0, (hipStream_t)stream);
        case 64: return launch_chunk<64>(p, 0, (hipStream_t)stream);
        default: return MXL_EUNSUPPORTED;
    }
}

extern "C" int mxl_chunk_attn_bwd(const void* q, const void* k, const void* v, const int* sorted_pos, const void* out,
                                  const float* lse, const void* dout, const float* dlse, float* dq, float* dk, float* dv,
                                  void* dq16, void* dk16, void* dv16, int ld16, int B,
                                  int T, int H, int dh, int n_h, int lsh, long long bs, int rs, float drop_p,
                                  unsigned long long seed, unsigned site, void* stream) {
    ChunkP p;
    int rc = fill_chunk(p, q, k, v, sorted_pos, (void*)out, (float*)lse, B, T, H, dh, n_h, lsh, bs, rs, drop_p, seed, site);
    if (rc) return rc;
    MXL_CHECK_ARG(dout && (dq || dq16) && (dk || dk16) && (dv || dv16));
    if (dq16 || dk16 || dv16) MXL_CHECK_ARG(n_h == 1 && ld16 >= H * dh && (ld16 % 4) == 0);
    p.dout = (const bf16_t*)dout; p.dlse = dlse; p.dq = dq; p.dk = dk; p.dv = dv;
    p.dq16 = (bf16_t*)dq16; p.dk16 = (bf16_t*)dk16; p.dv16 = (bf16_t*)dv16; p.ld16 = ld16;
    if (T <= SC_MAXT) {
        switch (dh) {
            case 16: return launch_single<16>(p, 1, (hipStream_t)stream);
            case 32: return launch_single<32>(p, 1, (hipStream_t)stream);
            case 64: return launch_single<64>(p, 1, (hipStream_t)stream);
            default: return MXL_EUNSUPPORTED;
        }
    }
    switch (dh) {
        case 16: return launch_chunk<16>(p, 1, (hipStream_t)stream);
        case 32: return launch_chunk<32>(p, 1, (hipStream_t)stream);
        case 64: return launch_chunk<64>(p, 1, (hipStream_t)stream);
        default: return MXL_EUNSUPPORTED;
    }
}

extern "C" int mxl_lsh_keynorm_bwd(const void* qk, long long bs, int rs, const float* dq, const float* dk_eff, void* dqk,
                                   int ld_dqk, int B, int T, int H, int dh, void* stream) {
    MXL_CHECK_ARG(qk && dq && dk_eff && dqk && B > 0 && T > 0 && H > 0 && dh > 0 && ld_dqk >= H * dh && (ld_dqk % 8) == 0);
    MXL_CHECK_ARG((dh % 8) == 0 && (64 % (dh / 8)) == 0);
    const long long n = (long long)B * T * H * (dh / 8);
    hipLaunchKernelGGL(lsh_keynorm_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)qk, bs, rs, dq, dk_eff, (bf16_t*)dqk, ld_dqk, B, T, H, dh, 1, nullptr, nullptr, 0);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_lsh_keynorm_bwd_rounds(const void* qk, long long bs, int rs, const float* dq, const float* dk_eff, const float* dv,
                                          void* dqk, int ld_dqk, void* dv16, int ld_dv, int B, int T, int H, int dh, int n_h,
                                          void* stream) {
    MXL_CHECK_ARG(qk && dq && dk_eff && dqk && B > 0 && T > 0 && H > 0 && dh > 0 && n_h >= 1 && ld_dqk >= H * dh && (ld_dqk % 8) == 0);
    MXL_CHECK_ARG((dh % 8) == 0 && (64 % (dh / 8)) == 0);
    MXL_CHECK_ARG((dv == nullptr) == (dv16 == nullptr));
    if (dv16) MXL_CHECK_ARG(ld_dv >= H * dh && (ld_dv % 8) == 0 && ((uintptr_t)dv16 % 16) == 0);
    const long long n = (long long)B * T * H * (dh / 8);
    hipLaunchKernelGGL(lsh_keynorm_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)qk, bs, rs, dq, dk_eff, (bf16_t*)dqk, ld_dqk, B, T, H, dh, n_h, dv, (bf16_t*)dv16, ld_dv);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_lsh_combine(const void* out_r, const float* lse, void* out, int B, int T, int H, int dh, int n_h, void* stream) {
    MXL_CHECK_ARG(out_r && lse && out && B > 0 && T > 0 && n_h >= 1 && dh > 0 && (dh % 8) == 0);
    const long long n = (long long)B * T * H * (dh / 8);
    hipLaunchKernelGGL(lsh_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)out_r, lse, (bf16_t*)out, B, T, H, dh, n_h);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_lsh_combine_bwd(const void* out_r, const float* lse, const void* out, const void* dout, void* dout_r,
                                   float* dlse, int B, int T, int H, int dh, int n_h, void* stream) {
    MXL_CHECK_ARG(out_r && lse && out && dout && dout_r && dlse && B > 0 && T > 0 && n_h >= 1);
    MXL_CHECK_ARG(dh > 0 && (dh % 8) == 0 && (64 % (dh / 8)) == 0);
    const long long n = (long long)B * T * H * (dh / 8);
    hipLaunchKernelGGL(lsh_combine_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)out_r, lse, (const bf16_t*)out, (const bf16_t*)dout, (bf16_t*)dout_r, dlse, B, T, H, dh, n_h);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}
