// HBM-bound kernels of the TransfoXL path: sinusoid table (K3 input), embedding (K1), residual+LayerNorm (K5/K6
// epilogue), column sums (bias grads), memory append (K8).  All bf16 traffic is 16-byte vectorised; one wave owns a row.
#include "common.h"
#include "musicxl_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// sinusoid table: out[dist][0:d/2] = sin(p*inv_freq), out[dist][d/2:d] = cos(p*inv_freq), p = min(dist, clamp)
// (upstream PositionalEmbedding; [sin | cos] halves), followed by drop(pos_emb).
// ---------------------------------------------------------------------------------------------------------
__global__ void sinusoid_kernel(bf16_t* out, int M, int d, int clamp_len, unsigned thresh, float scale,
                                unsigned long long seed, unsigned site) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = d >> 1;
    if (idx >= M * half) return;
    const int dist = idx / half, k = idx % half;
    const float p = (float)((clamp_len > 0 && dist > clamp_len) ? clamp_len : dist);
    // inv_freq_k = 1 / 10000^(2k/d)   (torch: 1 / (10000 ** (arange(0, d, 2) / d)))
    const float inv_freq = 1.0f / powf(10000.0f, (float)(2 * k) / (float)d);
    const float a = p * inv_freq;
    float s = sinf(a), c = cosf(a);
    if (thresh) {
        const uint64_t i0 = (uint64_t)dist * d + k, i1 = i0 + half;
        s = dropout_keep(seed, site, i0, thresh) ? s * scale : 0.f;
        c = dropout_keep(seed, site, i1, thresh) ? c * scale : 0.f;
    }
    out[(size_t)dist * d + k] = f2bf(s);
    out[(size_t)dist * d + half + k] = f2bf(c);
}

// ---------------------------------------------------------------------------------------------------------
// embedding: out[n][:] = drop(E[ids[n]][:] * scale)      (AdaptiveEmbedding, div_val = 1)
// ---------------------------------------------------------------------------------------------------------
__global__ void embed_fwd_kernel(const long long* ids, const bf16_t* E, bf16_t* out, int N, int d, int V, float scale,
                                 unsigned thresh, float dscale, unsigned long long seed, unsigned site) {
    const int chunks = d >> 3;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)N * chunks) return;
    const int n = (int)(gid / chunks), c = (int)(gid % chunks);
    long long id = ids[n];
    if (id < 0 || id >= V) id = 0;  // defensive: never index out of the table
    const bf16x8 e = *reinterpret_cast<const bf16x8*>(E + (size_t)id * d + c * 8);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        v[j] = bf2f((bf16_t)e[j]) * scale;
        if (thresh) v[j] = dropout_keep(seed, site, (uint64_t)n * d + c * 8 + j, thresh) ? v[j] * dscale : 0.f;
    }
    u32x4 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
    *reinterpret_cast<u32x4*>(out + (size_t)n * d + c * 8) = o;
}

// dE[ids[n]][:] += dout[n][:] * scale * keep   (fp32 atomics; rows are 4*d contiguous bytes per wave-instruction)
__global__ void embed_bwd_kernel(const long long* ids, const bf16_t* dout, const bf16_t* dout2, float* dE, int N, int d, int V, float scale,
                                 unsigned thresh, float dscale, unsigned long long seed, unsigned site) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)N * d) return;
    const int n = (int)(gid / d), k = (int)(gid % d);
    const long long id = ids[n];
    if (id < 0 || id >= V) return;
    float g = bf2f(dout[(size_t)n * d + k]);
    if (dout2) g += bf2f(dout2[(size_t)n * d + k]);
    g *= scale;
    if (thresh) g = dropout_keep(seed, site, (uint64_t)n * d + k, thresh) ? g * dscale : 0.f;
    atomicAdd(dE + (size_t)id * d + k, g);
}

// ---------------------------------------------------------------------------------------------------------
// z = res + drop(x);  y = LayerNorm(z) * gamma + beta        (post-LN of dec_attn / pos_ff)
// one wave per row; row kept in registers (d <= 2048).  Saves z (bf16), mean, rstd for the backward.
// ---------------------------------------------------------------------------------------------------------
constexpr int LN_MAXCH = 4;  // chunks of 8 per lane -> d <= 64*8*4 = 2048

template <int NCH>   // 8-element chunks per lane: 1 (d <= 512), 2 (d <= 1024) or NCH -- sizes the register arrays
__global__ __launch_bounds__(256) void ln_res_fwd_kernel(const bf16_t* x, const bf16_t* res, const float* gamma,
                                                         const float* beta, bf16_t* y, bf16_t* z, float* mean,
                                                         float* rstd, int N, int d, float eps, unsigned thresh,
                                                         float dscale, unsigned long long seed, unsigned site) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const int chunks = d >> 3;
    float v[NCH][8];
    // gamma / beta requested with the row, not after the two reductions (and after the mean / rstd stores, which the compiler
    // cannot move them across): at decode sizes (64 rows) the launch is one dependent chain and this removes a round trip from it
    f32x4 gq[NCH][2], bq[NCH][2];
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = lane + i * 64;
        if (c < chunks) {
            gq[i][0] = *reinterpret_cast<const f32x4*>(gamma + c * 8); gq[i][1] = *reinterpret_cast<const f32x4*>(gamma + c * 8 + 4);
            bq[i][0] = *reinterpret_cast<const f32x4*>(beta + c * 8); bq[i][1] = *reinterpret_cast<const f32x4*>(beta + c * 8 + 4);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = lane + i * 64;
        if (c < chunks) {
            const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + (size_t)row * d + c * 8);
            bf16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (res) rv = *reinterpret_cast<const bf16x8*>(res + (size_t)row * d + c * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float a = bf2f((bf16_t)xv[j]);
                if (thresh) a = dropout_keep32(seed, site, (uint32_t)row * d + c * 8 + j, thresh) ? a * dscale : 0.f;
                a += bf2f((bf16_t)rv[j]);
                // z is stored in bf16; normalise the *stored* value so forward and backward agree
                a = bf2f(f2bf(a));
                v[i][j] = a;
                s += a;
            }
        }
    }
    s = wave_sum(s);
    const float mu = s / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = lane + i * 64;
        if (c < chunks) {
#pragma unroll
            for (int j = 0; j < 8; j++) { const float t = v[i][j] - mu; q += t * t; }
        }
    }
    q = wave_sum(q);
    const float rs = rsqrtf(q / (float)d + eps);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = lane + i * 64;
        if (c < chunks) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (v[i][j] - mu) * rs * gq[i][j >> 2][j & 3] + bq[i][j >> 2][j & 3];
            u32x4 ov = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
            *reinterpret_cast<u32x4*>(y + (size_t)row * d + c * 8) = ov;
            if (z) {
                u32x4 zv = {pack2bf(v[i][0], v[i][1]), pack2bf(v[i][2], v[i][3]), pack2bf(v[i][4], v[i][5]),
                            pack2bf(v[i][6], v[i][7])};
                *reinterpret_cast<u32x4*>(z + (size_t)row * d + c * 8) = zv;
            }
        }
    }
}

// y = LayerNorm(res + bf16(sum_s slab_s + bias)): the consumer of mxl_gemm_skinny_partial -- the K-slice reduction, the bias and
// the post-LN residual of a decode-step linear in one launch (inference: no dropout, nothing saved for a backward)
template <int NCH>
__global__ __launch_bounds__(256) void ln_res_partial_fwd_kernel(const float* slabs, int KS, long long slab_stride, const float* bias,
                                                                 const bf16_t* res, const float* gamma, const float* beta, bf16_t* y,
                                                                 int N, int d, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const int chunks = d >> 3;
    float v[NCH][8];
    f32x4 gq[NCH][2], bq[NCH][2];          // requested with the row (see ln_res_fwd_kernel)
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = lane + i * 64;
        if (c < chunks) {
            gq[i][0] = *reinterpret_cast<const f32x4*>(gamma + c * 8); gq[i][1] = *reinterpret_cast<const f32x4*>(gamma + c * 8 + 4);
            bq[i][0] = *reinterpret_cast<const f32x4*>(beta + c * 8); bq[i][1] = *reinterpret_cast<const f32x4*>(beta + c * 8 + 4);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = lane + i * 64;
        if (c < chunks) {
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = 0.f;
            for (int sl = 0; sl < KS; sl++) {
                const float* pp = slabs + (size_t)sl * slab_stride + (size_t)row * d + c * 8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(pp), hi = *reinterpret_cast<const f32x4*>(pp + 4);
#pragma unroll
                for (int j = 0; j < 4; j++) { a[j] += lo[j]; a[4 + j] += hi[j]; }
            }
            const bf16x8 rv = *reinterpret_cast<const bf16x8*>(res + (size_t)row * d + c * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float t = a[j] + (bias ? bias[c * 8 + j] : 0.f);
                t = bf2f(f2bf(t)) + bf2f((bf16_t)rv[j]);          // the linear's output is a bf16 tensor in the unfused path
                t = bf2f(f2bf(t));
                v[i][j] = t;
                s += t;
            }
        }
    }
    s = wave_sum(s);
    const float mu = s / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = lane + i * 64;
        if (c < chunks) {
#pragma unroll
            for (int j = 0; j < 8; j++) { const float t = v[i][j] - mu; q += t * t; }
        }
    }
    q = wave_sum(q);
    const float rs = rsqrtf(q / (float)d + eps);
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = lane + i * 64;
        if (c < chunks) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (v[i][j] - mu) * rs * gq[i][j >> 2][j & 3] + bq[i][j >> 2][j & 3];
            u32x4 ov = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
            *reinterpret_cast<u32x4*>(y + (size_t)row * d + c * 8) = ov;
        }
    }
}

// backward: dz = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat));  dres = dz (+ dres_in);  dx = keep*dscale*dz
// dgamma += sum_rows dy*xhat, dbeta += sum_rows dy : per-block partials through LDS, then one fp32 atomic per column.
#ifndef LNB_ROWS_
#define LNB_ROWS_ 64
#endif
constexpr int LNB_ROWS = LNB_ROWS_;  // rows per block (8 waves x 8 rows): 2d atomics per block, 4096 waves for the 32768-row C3 matrices
constexpr int LNB_THREADS = 512;

template <int NCH, bool CS = false>
__global__ __launch_bounds__(LNB_THREADS) __attribute__((amdgpu_waves_per_eu(NCH <= 2 ? 4 : 1, 8))) void ln_res_bwd_kernel(const bf16_t* dy, const bf16_t* dy2, const bf16_t* z,
                                                         const float* mean, const float* rstd, const float* gamma,
                                                         bf16_t* dres, bf16_t* dx, float* dgamma, float* dbeta, int N,
                                                         int d, unsigned thresh, float dscale, unsigned long long seed,
                                                         unsigned site, const bf16_t* dadd, float* dxsum) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // per-wave slabs [8 waves][2][d] of dgamma / dbeta partials (plain 16-byte writes, then a column-wise sum): LDS atomics from
    // eight waves onto the same 2d addresses cost as much as several rows of work.  (NCH > 2: d up to 2048 would need 128 KB of
    // slabs -- those keep the shared [2][d] accumulators and LDS atomics.)
    constexpr bool SLABS = NCH <= 2;
    float* sg = reinterpret_cast<float*>(smem_raw);  // [d] dgamma partial (slab 0 when SLABS)
    float* sb = sg + d;                              // [d] dbeta partial
    if (!SLABS) {
        for (int i = threadIdx.x; i < 2 * d; i += LNB_THREADS) sg[i] = 0.f;
        __syncthreads();
    }
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunks = d >> 3;
    float ag[NCH][8], ab[NCH][8];
    float ac[CS ? NCH : 1][8];             // column sums of dx as stored (CS: dxsum != null, SLABS form only): the bias gradient of the
                                           // linear layer whose output gradient dx is -- saves a pass over dx (mxl_colsum_bf16)
#pragma unroll
    for (int i = 0; i < NCH; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) { ag[i][j] = 0.f; ab[i][j] = 0.f; if (CS) ac[i][j] = 0.f; }
    for (int rr = 0; rr < LNB_ROWS / (LNB_THREADS / 64); rr++) {
        const int row = blockIdx.x * LNB_ROWS + rr * (LNB_THREADS / 64) + wid;
        if (row >= N) break;
        const float mu = mean[row], rs = rstd[row];
        float g[NCH][8], xh[NCH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = lane + i * 64;
            if (c < chunks) {
                const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dy + (size_t)row * d + c * 8);
                bf16x8 dv2 = {0, 0, 0, 0, 0, 0, 0, 0};
                if (dy2) dv2 = *reinterpret_cast<const bf16x8*>(dy2 + (size_t)row * d + c * 8);
                const bf16x8 zv = *reinterpret_cast<const bf16x8*>(z + (size_t)row * d + c * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float dyv = bf2f((bf16_t)dv[j]) + bf2f((bf16_t)dv2[j]);
                    const float xhat = (bf2f((bf16_t)zv[j]) - mu) * rs;
                    const float gg = dyv * gamma[c * 8 + j];
                    g[i][j] = gg; xh[i][j] = xhat;
                    s1 += gg; s2 += gg * xhat;
                    ag[i][j] += dyv * xhat; ab[i][j] += dyv;
                }
            }
        }
        s1 = wave_sum(s1) / (float)d;
        s2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = lane + i * 64;
            if (c < chunks) {
                float o[8], ox[8];
                const bool drop_sum = dadd && dres && dx;       // dx = dropout(dres as stored): the mask goes on the SUM (mxl_ln_residual_bwd_add_drop)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    o[j] = rs * (g[i][j] - s1 - xh[i][j] * s2);
                    ox[j] = o[j];
                    if (thresh && !drop_sum) ox[j] = dropout_keep32(seed, site, (uint32_t)row * d + c * 8 + j, thresh) ? o[j] * dscale : 0.f;
                }
                if (dres) {
                    if (dadd) {
                        const bf16x8 av = *reinterpret_cast<const bf16x8*>(dadd + (size_t)row * d + c * 8);
#pragma unroll
                        for (int j = 0; j < 8; j++) o[j] += bf2f((bf16_t)av[j]);
                    }
                    u32x4 ov = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
                    *reinterpret_cast<u32x4*>(dres + (size_t)row * d + c * 8) = ov;
                    if (drop_sum) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {        // from the rounded values, as a separate dropout pass over dres would read them
                            const float lo = __uint_as_float(ov[j] << 16), hi = __uint_as_float(ov[j] & 0xffff0000u);
                            ox[2 * j] = !thresh ? lo : dropout_keep32(seed, site, (uint32_t)row * d + c * 8 + 2 * j, thresh) ? lo * dscale : 0.f;
                            ox[2 * j + 1] = !thresh ? hi : dropout_keep32(seed, site, (uint32_t)row * d + c * 8 + 2 * j + 1, thresh) ? hi * dscale : 0.f;
                        }
                    }
                }
                if (dx) {
                    u32x4 ov = {pack2bf(ox[0], ox[1]), pack2bf(ox[2], ox[3]), pack2bf(ox[4], ox[5]), pack2bf(ox[6], ox[7])};
                    *reinterpret_cast<u32x4*>(dx + (size_t)row * d + c * 8) = ov;
                    if (CS) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {       // the rounded values, as a separate column sum over dx would see them
                            ac[i][2 * j] += __uint_as_float(ov[j] << 16);
                            ac[i][2 * j + 1] += __uint_as_float(ov[j] & 0xffff0000u);
                        }
                    }
                }
            }
        }
    }
    if (SLABS) {
        float* mg = sg + (size_t)wid * 2 * d;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = lane + i * 64;
            if (c < chunks) {
                *reinterpret_cast<f32x4*>(mg + c * 8) = f32x4{ag[i][0], ag[i][1], ag[i][2], ag[i][3]};
                *reinterpret_cast<f32x4*>(mg + c * 8 + 4) = f32x4{ag[i][4], ag[i][5], ag[i][6], ag[i][7]};
                *reinterpret_cast<f32x4*>(mg + d + c * 8) = f32x4{ab[i][0], ab[i][1], ab[i][2], ab[i][3]};
                *reinterpret_cast<f32x4*>(mg + d + c * 8 + 4) = f32x4{ab[i][4], ab[i][5], ab[i][6], ab[i][7]};
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * d; i += LNB_THREADS) {       // i < d: dgamma column i, else dbeta column i - d
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < LNB_THREADS / 64; w++) t += sg[(size_t)w * 2 * d + i];
            atomicAdd((i < d ? dgamma : dbeta - d) + i, t);
        }
        if (CS) {              // third accumulator through the same slabs
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int c = lane + i * 64;
                if (c < chunks) {
                    *reinterpret_cast<f32x4*>(mg + c * 8) = f32x4{ac[i][0], ac[i][1], ac[i][2], ac[i][3]};
                    *reinterpret_cast<f32x4*>(mg + c * 8 + 4) = f32x4{ac[i][4], ac[i][5], ac[i][6], ac[i][7]};
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < d; i += LNB_THREADS) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < LNB_THREADS / 64; w++) t += sg[(size_t)w * 2 * d + i];
                atomicAdd(dxsum + i, t);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = lane + i * 64;
        if (c < chunks) {
#pragma unroll
            for (int j = 0; j < 8; j++) { atomicAdd(&sg[c * 8 + j], ag[i][j]); atomicAdd(&sb[c * 8 + j], ab[i][j]); }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < d; i += LNB_THREADS) {
        atomicAdd(dgamma + i, sg[i]);
        atomicAdd(dbeta + i, sb[i]);
    }
}

// out[n] += sum_m X[m][n]   (bias gradients: CoreNet.0/3 bias, crit bias).  bf16 in, fp32 atomic out.
// block = 32 column-threads (8 columns each, 16-byte loads) x 8 row-lanes; rows_per_block rows per block.
// DROP: X is first passed through the dropout mask of (seed, site) (element index r * N + c: ld == N) and written to Y; the sums
// are over the stored Y (mxl_dropout_colsum_bf16: the Reformer's FFN-output bias gradient, one pass instead of two)
template <bool DROP>
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* X, float* out, int M, int N, int ld, int rows_per_block,
                                                     bf16_t* Y, unsigned thresh, float dscale, unsigned long long seed, unsigned site) {
    __shared__ float red[8][256];
    const int ct = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = blockIdx.x * 256 + ct * 8;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    if (col + 7 < N && (ld & 7) == 0) {
        int r = r0 + rl;
        auto drop_store = [&](bf16x8 v, int row) -> bf16x8 {     // DROP: mask, scale, store; returns the stored values
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; j++)
                o[j] = dropout_keep32(seed, site, (uint32_t)row * (uint32_t)N + (uint32_t)(col + j), thresh) ? bf2f((bf16_t)v[j]) * dscale : 0.f;
            const u32x4 w = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
            *reinterpret_cast<u32x4*>(Y + (size_t)row * ld + col) = w;
            return __builtin_bit_cast(bf16x8, w);
        };
        for (; r + 24 < r1; r += 32) {            // four independent 16-byte loads in flight per thread
            bf16x8 v0 = *reinterpret_cast<const bf16x8*>(X + (size_t)r * ld + col);
            bf16x8 v1 = *reinterpret_cast<const bf16x8*>(X + (size_t)(r + 8) * ld + col);
            bf16x8 v2 = *reinterpret_cast<const bf16x8*>(X + (size_t)(r + 16) * ld + col);
            bf16x8 v3 = *reinterpret_cast<const bf16x8*>(X + (size_t)(r + 24) * ld + col);
            if (DROP) { v0 = drop_store(v0, r); v1 = drop_store(v1, r + 8); v2 = drop_store(v2, r + 16); v3 = drop_store(v3, r + 24); }
#pragma unroll
            for (int j = 0; j < 8; j++)
                acc[j] += (bf2f((bf16_t)v0[j]) + bf2f((bf16_t)v1[j])) + (bf2f((bf16_t)v2[j]) + bf2f((bf16_t)v3[j]));
        }
        for (; r < r1; r += 8) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(X + (size_t)r * ld + col);
            if (DROP) v = drop_store(v, r);
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] += bf2f((bf16_t)v[j]);
        }
    } else if (col < N) {
        for (int r = r0 + rl; r < r1; r += 8)
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (col + j < N) acc[j] += bf2f(X[(size_t)r * ld + col + j]);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) red[rl][ct * 8 + j] = acc[j];
    __syncthreads();
    const int c = threadIdx.x;
    if (blockIdx.x * 256 + c < N) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) s += red[k][c];
        atomicAdd(out + blockIdx.x * 256 + c, s);
    }
}

// new_mem[b] = cat(mem[b], hid[b])[-M:]   (TransfoXLModel._update_mems; batch-major (B, len, d))
__global__ void mem_update_kernel(const bf16_t* mem, const bf16_t* hid, bf16_t* out, int B, int M, int T, int d) {
    const int chunks = d >> 3;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * M * chunks) return;
    const int c = (int)(gid % chunks);
    const int r = (int)((gid / chunks) % M);
    const int b = (int)(gid / ((long long)chunks * M));
    const int src = r + T;  // index into cat(mem, hid) of length M+T, keeping the last M
    u32x4 v;
    if (src < M) v = *reinterpret_cast<const u32x4*>(mem + ((size_t)b * M + src) * d + c * 8);
    else v = *reinterpret_cast<const u32x4*>(hid + ((size_t)b * T + (src - M)) * d + c * 8);
    *reinterpret_cast<u32x4*>(out + ((size_t)b * M + r) * d + c * 8) = v;
}

// out[b][t][:] = bf16(x[b][t][:] + bias[:])  (materialises q + r_r_bias for the dRd contraction)
__global__ void add_rowbias_kernel(const bf16_t* x, long long x_bs, int x_rs, const float* bias, bf16_t* out, int B, int T, int n) {
    const int chunks = n >> 3;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * T * chunks) return;
    const int c = (int)(gid % chunks);
    const long long row = gid / chunks;
    const int b = (int)(row / T), t = (int)(row % T);
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + b * x_bs + (long long)t * x_rs + c * 8);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = bf2f((bf16_t)v[j]) + bias[c * 8 + j];
    u32x4 w = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
    *reinterpret_cast<u32x4*>(out + row * n + c * 8) = w;
}

// y = keep * x / (1 - p)   (final `drop(core_out)` of TransfoXLModel.forward; its own backward with x := dy)
__global__ void dropout_kernel(const bf16_t* x, bf16_t* y, long long n8, unsigned thresh, float dscale,
                               unsigned long long seed, unsigned site) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++)   // n <= 2^32 (host check): the 32-bit index form makes the same decisions as dropout_keep
            o[j] = dropout_keep32(seed, site, (uint32_t)i * 8u + j, thresh) ? bf2f((bf16_t)v[j]) * dscale : 0.f;
        u32x4 w = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
        *reinterpret_cast<u32x4*>(y + i * 8) = w;
    }
}

// dst[b][c][r] = src[b][r][c]: 64 x 64 bf16 tiles through LDS (row pitch 65 keeps both phases conflict-free).  Used to keep
// [in][out] copies of the linear weights beside the [out][in] ones, so that dX = dY W runs in the K-contiguous GEMM form.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* src, bf16_t* dst, int rows, int cols, int ld_src,
                                                             int ld_dst, long long sb, long long db) {
    __shared__ bf16_t t[64][65];
    src += (size_t)blockIdx.z * sb;
    dst += (size_t)blockIdx.z * db;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4)
        if (r0 + i < rows && c0 + tx < cols) t[i][tx] = src[(size_t)(r0 + i) * ld_src + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 64; i += 4)
        if (c0 + i < cols && r0 + tx < rows) dst[(size_t)(c0 + i) * ld_dst + r0 + tx] = t[tx][i];
}

// Y = X - column means: one workgroup per 8 columns (16-byte row pieces), 256 threads stride the rows, fp32 sums through LDS
__global__ __launch_bounds__(256) void center_columns_kernel(const bf16_t* X, bf16_t* Y, int M, int N) {
    __shared__ float part[256][8];
    const int c0 = blockIdx.x * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int m = threadIdx.x; m < M; m += 256) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(X + (size_t)m * N + c0);
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] += bf2f((bf16_t)v[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) part[threadIdx.x][j] = s[j];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int j = 0; j < 8; j++) part[threadIdx.x][j] += part[threadIdx.x + o][j];
        __syncthreads();
    }
    float mean[8];
#pragma unroll
    for (int j = 0; j < 8; j++) mean[j] = part[0][j] / (float)M;
    for (int m = threadIdx.x; m < M; m += 256) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(X + (size_t)m * N + c0);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = bf2f((bf16_t)v[j]) - mean[j];
        const u32x4 w = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
        *reinterpret_cast<u32x4*>(Y + (size_t)m * N + c0) = w;
    }
}

}  // namespace

extern "C" int mxl_center_columns_bf16(const void* X, void* Y, int M, int N, void* stream) {
    MXL_CHECK_ARG(X && Y && M > 0 && N > 0 && (N % 8) == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0);
    hipLaunchKernelGGL(center_columns_kernel, dim3(N / 8), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X, (bf16_t*)Y, M, N);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_dropout_bf16(const void* x, void* y, long long n, float drop_p, unsigned long long seed, unsigned site,
                                void* stream) {
    MXL_CHECK_ARG(x && y && n > 0 && (n % 8) == 0 && drop_p > 0.f && drop_p < 1.f && n <= 0x100000000ll);
    long long blocks = (n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)y, n / 8, dropout_thresh(drop_p), 1.f / (1.f - drop_p), seed, site);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_add_rowbias_bf16(const void* x, long long x_bs, int x_rs, const float* bias, void* out, int B, int T,
                                    int n, void* stream) {
    MXL_CHECK_ARG(x && bias && out && B > 0 && T > 0 && n > 0 && (n % 8) == 0 && (x_rs % 8) == 0 && (x_bs % 8) == 0);
    const long long tot = (long long)B * T * (n / 8);
    {
        mxl_kt::Scope kt(MXL_KT_ROWBIAS, (hipStream_t)stream);
        hipLaunchKernelGGL(add_rowbias_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, x_bs, x_rs, bias, (bf16_t*)out, B, T, n);
    }
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_sinusoid_table(void* out, int M, int d, int clamp_len, float drop_p, unsigned long long seed,
                                  unsigned site, void* stream) {
    MXL_CHECK_ARG(out && M > 0 && d > 0 && (d % 2) == 0);
    const int n = M * (d / 2);
    hipLaunchKernelGGL(sinusoid_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (bf16_t*)out, M, d,
                       clamp_len, dropout_thresh(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, site);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_embed_fwd(const void* ids, const void* E, void* out, int N, int d, int V, float scale, float drop_p,
                             unsigned long long seed, unsigned site, void* stream) {
    MXL_CHECK_ARG(ids && E && out && N > 0 && d > 0 && (d % 8) == 0 && V > 0);
    const long long n = (long long)N * (d / 8);
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)ids, (const bf16_t*)E, (bf16_t*)out, N, d, V, scale, dropout_thresh(drop_p),
                       drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, site);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_embed_bwd(const void* ids, const void* dout, const void* dout2, float* dE, int N, int d, int V,
                             float scale, float drop_p, unsigned long long seed, unsigned site, void* stream) {
    MXL_CHECK_ARG(ids && dout && dE && N > 0 && d > 0 && V > 0);
    const long long n = (long long)N * d;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)ids, (const bf16_t*)dout, (const bf16_t*)dout2, dE, N, d, V, scale, dropout_thresh(drop_p),
                       drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, site);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_ln_residual_fwd(const void* x, const void* res, const float* gamma, const float* beta, void* y,
                                   void* z, float* mean, float* rstd, int N, int d, float eps, float drop_p,
                                   unsigned long long seed, unsigned site, void* stream) {
    MXL_CHECK_ARG(x && gamma && beta && y && N > 0 && d > 0 && (d % 8) == 0 && d <= 64 * 8 * LN_MAXCH);
    MXL_CHECK_ARG(((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0);      // read as 16-byte vectors
    MXL_CHECK_ARG(drop_p <= 0.f || (unsigned long long)N * d <= 0xffffffffull);      // the dropout mask is indexed in 32 bits
    const auto kfn = d <= 512 ? ln_res_fwd_kernel<1> : d <= 1024 ? ln_res_fwd_kernel<2> : ln_res_fwd_kernel<LN_MAXCH>;
    hipLaunchKernelGGL(kfn, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)res, gamma, beta, (bf16_t*)y, (bf16_t*)z, mean, rstd, N, d, eps,
                       dropout_thresh(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, site);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

static int ln_residual_bwd_launch(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd,
                                  const float* gamma, void* dres, void* dx, float* dgamma, float* dbeta, int N, int d,
                                  float drop_p, unsigned long long seed, unsigned site, float* dxsum, void* stream) {
    MXL_CHECK_ARG(dy && z && mean && rstd && gamma && dgamma && dbeta && N > 0 && (d % 8) == 0 && d <= 64 * 8 * LN_MAXCH);
    MXL_CHECK_ARG(!dxsum || (dx && d <= 1024));
    MXL_CHECK_ARG(drop_p <= 0.f || (unsigned long long)N * d <= 0xffffffffull);
    const auto kfn = dxsum ? (d <= 512 ? ln_res_bwd_kernel<1, true> : ln_res_bwd_kernel<2, true>)
                           : (d <= 512 ? ln_res_bwd_kernel<1> : d <= 1024 ? ln_res_bwd_kernel<2> : ln_res_bwd_kernel<LN_MAXCH>);
    hipLaunchKernelGGL(kfn, dim3((N + LNB_ROWS - 1) / LNB_ROWS), dim3(LNB_THREADS), (d <= 1024 ? LNB_THREADS / 64 : 1) * 2 * d * sizeof(float),
                       (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)dy2, (const bf16_t*)z, mean, rstd, gamma,
                       (bf16_t*)dres, (bf16_t*)dx, dgamma, dbeta, N, d, dropout_thresh(drop_p),
                       drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, seed, site, (const bf16_t*)nullptr, dxsum);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_ln_residual_bwd(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd,
                                   const float* gamma, void* dres, void* dx, float* dgamma, float* dbeta, int N, int d,
                                   float drop_p, unsigned long long seed, unsigned site, void* stream) {
    return ln_residual_bwd_launch(dy, dy2, z, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, drop_p, seed, site, nullptr, stream);
}

extern "C" int mxl_ln_residual_bwd_colsum(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd,
                                          const float* gamma, void* dres, void* dx, float* dgamma, float* dbeta, float* dxsum,
                                          int N, int d, float drop_p, unsigned long long seed, unsigned site, void* stream) {
    MXL_CHECK_ARG(dxsum);
    return ln_residual_bwd_launch(dy, dy2, z, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, drop_p, seed, site, dxsum, stream);
}

extern "C" int mxl_ln_residual_bwd_add(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd,
                                       const float* gamma, const void* dadd, void* dres, float* dgamma, float* dbeta, int N,
                                       int d, void* stream) {
    MXL_CHECK_ARG(dy && z && mean && rstd && gamma && dres && dgamma && dbeta && N > 0 && (d % 8) == 0 && d <= 64 * 8 * LN_MAXCH);
    const auto kfn = d <= 512 ? ln_res_bwd_kernel<1> : d <= 1024 ? ln_res_bwd_kernel<2> : ln_res_bwd_kernel<LN_MAXCH>;
    hipLaunchKernelGGL(kfn, dim3((N + LNB_ROWS - 1) / LNB_ROWS), dim3(LNB_THREADS), (d <= 1024 ? LNB_THREADS / 64 : 1) * 2 * d * sizeof(float),
                       (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)dy2, (const bf16_t*)z, mean, rstd, gamma,
                       (bf16_t*)dres, (bf16_t*)nullptr, dgamma, dbeta, N, d, 0u, 1.f, 0ull, 0u, (const bf16_t*)dadd, (float*)nullptr);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

// dres = LayerNorm-backward(dy + dy2) + dadd and, in the same pass, dx = dropout(dres) with the mask of (seed, site) (the element
// index is row * d + column, as in mxl_dropout_bf16) and, with dxsum, dxsum[c] += column sums of dx: what mxl_ln_residual_bwd_add
// followed by mxl_dropout_bf16 / mxl_dropout_colsum_bf16 over dres produce, bit for bit.  dx may alias dy (a wave reads its whole
// row before it writes it).
extern "C" int mxl_ln_residual_bwd_add_drop(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd,
                                            const float* gamma, const void* dadd, void* dres, void* dx, float* dxsum, float* dgamma,
                                            float* dbeta, int N, int d, float drop_p, unsigned long long seed, unsigned site,
                                            void* stream) {
    MXL_CHECK_ARG(dy && z && mean && rstd && gamma && dadd && dres && dx && dgamma && dbeta && N > 0 && (d % 8) == 0 && d <= 1024);
    MXL_CHECK_ARG(dres != dx && dres != dy && drop_p > 0.f && drop_p < 1.f && (unsigned long long)N * d <= 0xffffffffull);
    const auto kfn = dxsum ? (d <= 512 ? ln_res_bwd_kernel<1, true> : ln_res_bwd_kernel<2, true>)
                           : (d <= 512 ? ln_res_bwd_kernel<1> : ln_res_bwd_kernel<2>);
    hipLaunchKernelGGL(kfn, dim3((N + LNB_ROWS - 1) / LNB_ROWS), dim3(LNB_THREADS), (LNB_THREADS / 64) * 2 * d * sizeof(float),
                       (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)dy2, (const bf16_t*)z, mean, rstd, gamma,
                       (bf16_t*)dres, (bf16_t*)dx, dgamma, dbeta, N, d, dropout_thresh(drop_p), 1.f / (1.f - drop_p), seed, site,
                       (const bf16_t*)dadd, dxsum);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_ln_residual_fwd_partial(const float* slabs, int KS, long long slab_stride, const float* bias, const void* res,
                                           const float* gamma, const float* beta, void* y, int N, int d, float eps, void* stream) {
    MXL_CHECK_ARG(slabs && res && gamma && beta && y && KS >= 1 && N > 0 && d > 0 && (d % 8) == 0 && d <= 64 * 8 * LN_MAXCH);
    MXL_CHECK_ARG(slab_stride >= (long long)N * d && ((uintptr_t)slabs % 16) == 0 && (slab_stride % 4) == 0);
    MXL_CHECK_ARG(((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0);
    const auto kfn = d <= 512 ? ln_res_partial_fwd_kernel<1> : d <= 1024 ? ln_res_partial_fwd_kernel<2> : ln_res_partial_fwd_kernel<LN_MAXCH>;
    hipLaunchKernelGGL(kfn, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, slabs, KS, slab_stride, bias, (const bf16_t*)res,
                       gamma, beta, (bf16_t*)y, N, d, eps);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_colsum_bf16(const void* X, float* out, int M, int N, int ld, void* stream) {
    MXL_CHECK_ARG(X && out && M > 0 && N > 0 && ld >= N);
    const int rpb = 128;      // 768 blocks for a 32768 x 768 matrix: three per CU, four loads in flight per thread
    hipLaunchKernelGGL(colsum_kernel<false>, dim3((N + 255) / 256, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)X, out, M, N, ld, rpb, (bf16_t*)nullptr, 0u, 1.f, 0ull, 0u);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_dropout_colsum_bf16(const void* X, void* Y, float* out, int M, int N, float drop_p, unsigned long long seed,
                                       unsigned site, void* stream) {
    MXL_CHECK_ARG(X && Y && out && M > 0 && N > 0 && (N % 8) == 0 && drop_p > 0.f && drop_p < 1.f);
    MXL_CHECK_ARG((unsigned long long)M * N <= 0xffffffffull && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0);
    const int rpb = 128;
    hipLaunchKernelGGL(colsum_kernel<true>, dim3((N + 255) / 256, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)X, out, M, N, N, rpb, (bf16_t*)Y, dropout_thresh(drop_p), 1.f / (1.f - drop_p), seed, site);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_mem_update(const void* mem, const void* hid, void* out, int B, int M, int T, int d, void* stream) {
    MXL_CHECK_ARG(mem && hid && out && B > 0 && M > 0 && T > 0 && (d % 8) == 0 && out != mem);
    const long long n = (long long)B * M * (d / 8);
    hipLaunchKernelGGL(mem_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)mem, (const bf16_t*)hid, (bf16_t*)out, B, M, T, d);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_transpose_bf16(const void* src, void* dst, int rows, int cols, int ld_src, int ld_dst, int batch,
                                  long long src_bstride, long long dst_bstride, void* stream) {
    MXL_CHECK_ARG(src && dst && src != dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows && batch >= 1);
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dst, rows, cols, ld_src, ld_dst, src_bstride, dst_bstride);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}
