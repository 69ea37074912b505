// Reformer incremental (cached) decoding: the single-token step of HF's `use_cache` path (ReformerDynamicCache, HF515:65-148;
// LSHSelfAttention.forward with `past_buckets_states`, HF515:466-516 and 946-1050; LocalSelfAttention, HF515:1136-1169) -- what
// `model.generate(...)` at musicnlp/trainer/eval.py:333 runs for a Reformer.  HF caches the LayerNorm'ed hidden states and
// re-projects the ones it gathers; here the projections of every position (local: k, v; LSH: shared qk, v) are cached instead
// -- the same numbers, computed once.
//
//   mxl_rf_decode_embed   x[b] = E[ids[b, t]] + cat(W0[t / A1], W1[t % A1])                         (HF515:311-354 in eval)
//   mxl_lsh_fix_buckets   padded prefill: pads -> the extra bucket, per-round offsets r * (NB + 1)     (HF515:746-756)
//   mxl_rf_query_bucket   bucket ids of the new token appended to the cache; offsets widen to NB + 1 when the cache already
//                         holds a bucket id above n_h * NB - 1 (HF515:961-970 `increase_num_buckets`)
//   mxl_rf_decode_attn    one query per (sequence, head, hash round) over <= 128 cached positions: a contiguous range (local
//                         layers, and LSH layers before their first hashing) or the two 64-slot chunks around the new token in
//                         the bucket-sorted order (LSH); shared-QK key normalisation, self mask -1e5, rounds merged by their
//                         logsumexp weights.  HBM-bound: 128 rows x 2 x dh x 2 B per (sequence, head, round).
#include "common.h"
#include "musicxl_internal.h"

namespace {

__global__ __launch_bounds__(256) void rf_decode_embed_kernel(const long long* ids, int ld_ids, int t, const bf16_t* E,
                                                              const float* W0, const float* W1, bf16_t* out, int B, int d,
                                                              int V, int A1, int d0) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * d) return;
    const int b = idx / d, e = idx % d;
    long long tok = ids[(size_t)b * ld_ids + t];
    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
    const float pos = e < d0 ? W0[(size_t)(t / A1) * d0 + e] : W1[(size_t)(t % A1) * (d - d0) + (e - d0)];
    out[idx] = f2bf(bf2f(E[(size_t)tok * d + e]) + pos);
}

// buckets (rows, n_h * T): entry (row, r, t) holds r * NB + b from mxl_lsh_hash.  Rewritten to r * (NB + 1) + (t < T_real ? b : NB).
__global__ __launch_bounds__(256) void lsh_fix_buckets_kernel(int* buckets, long long n, int T, int T_real, int n_h, int NB) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int within = (int)(i % ((long long)n_h * T));
    const int r = within / T, t = within % T;
    const int b = buckets[i] - r * NB;
    buckets[i] = r * (NB + 1) + (t < T_real ? b : NB);
}

// raw (B*H, n_h) = r * NB + b for the new token; cache (B*H, n_h, Tmax); bkmax: running maximum of the cached ids (one int).
// ONE workgroup: every thread reads the maximum of the PAST buckets before anyone adds the new ones (HF decides the widening
// from `past_buckets.max()`).
__global__ __launch_bounds__(1024) void rf_query_bucket_kernel(const int* raw, int* cache, int* bkmax, int rows, int n_h, int NB,
                                                               int Tmax, int t) {
    const int i = threadIdx.x;
    const int past_max = *bkmax;
    __syncthreads();
    if (i >= rows * n_h) return;
    const int r = i % n_h;
    const int inc = past_max > n_h * NB - 1 ? 1 : 0;
    const int v = raw[i] - r * NB + r * (NB + inc);
    cache[(size_t)i * Tmax + t] = v;
    atomicMax(bkmax, v);
}

struct DecAttnP {
    const bf16_t *q, *kc, *vc;
    const int* sorted;        // (B*H*n_h, n) bucket-sorted positions, or NULL: contiguous range
    bf16_t* out;
    int B, H, n_h, Tmax, n, t, start, count, lsh, ldq;
};

// block = (head, sequence), one wave per hash round.  Phase A: lane = key slot (two per lane): score; phase B: lane = (key
// subgroup, channel): out = sum_k p_k v_k.
template <int DH>
__global__ __launch_bounds__(256) void rf_decode_attn_kernel(DecAttnP p) {
    __shared__ float s_p[4][128];
    __shared__ int s_pos[4][128];
    __shared__ float s_out[4][DH];
    __shared__ float s_lse[4];
    const int h = blockIdx.x, b = blockIdx.y;
    const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int d = p.H * DH;
    const bf16_t* qp = p.q + (size_t)b * p.ldq + h * DH;
    float qf[DH];
#pragma unroll
    for (int c = 0; c < DH / 8; c++) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(qp + c * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) qf[c * 8 + j] = bf2f((bf16_t)v[j]);
    }
    int count = p.count;
    int my_pos[2];
    if (p.sorted) {
        // the new token (position t = n - 1) sits at sorted slot `rank`; window = its 64-slot chunk and the one before, modulo n
        const int* so = p.sorted + ((size_t)(b * p.H + h) * p.n_h + r) * p.n;
        int found = -1;
        for (int i = lane; i < p.n; i += 64)
            if (so[i] == p.t) found = i;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) found = max(found, __shfl_xor(found, o, 64));
        const int start = (found / 64 - 1) * 64;
        count = 128;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            int slot = (start + lane + 64 * k) % p.n;
            if (slot < 0) slot += p.n;
            my_pos[k] = so[slot];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 2; k++) my_pos[k] = p.start + lane + 64 * k;
    }
    const float inv_sqrt_dh = rsqrtf((float)DH);
    float sc[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int slot = lane + 64 * k;
        sc[k] = -INFINITY;
        if (slot < count) {
            const bf16_t* kp = p.kc + ((size_t)b * p.Tmax + my_pos[k]) * d + h * DH;
            float dot = 0.f, ss = 0.f;
#pragma unroll
            for (int c = 0; c < DH / 8; c++) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(kp + c * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float kv = bf2f((bf16_t)v[j]);
                    dot += qf[c * 8 + j] * kv;
                    ss += kv * kv;
                }
            }
            if (p.lsh) {
                dot *= rsqrtf(ss / (float)DH + 1e-6f) * inv_sqrt_dh;       // key = qk / sqrt(mean(qk^2) + eps) / sqrt(dh)
                if (my_pos[k] == p.t) dot = -1e5f;                          // a token attends to itself only as a last resort
            } else {
                dot *= inv_sqrt_dh;                                         // HF scales the local keys by 1 / sqrt(dh)
            }
            sc[k] = dot;
        }
        s_pos[r][slot] = slot < count ? my_pos[k] : 0;
    }
    float m = fmaxf(sc[0], sc[1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float e0 = sc[0] == -INFINITY ? 0.f : __expf(sc[0] - m), e1 = sc[1] == -INFINITY ? 0.f : __expf(sc[1] - m);
    float l = e0 + e1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
    s_p[r][lane] = e0 / l;
    s_p[r][lane + 64] = e1 / l;
    if (lane == 0) s_lse[r] = m + __logf(l);
    __syncthreads();
    // phase B
    constexpr int G = 64 / DH;                      // key subgroups per wave
    const int e = lane % DH, g = lane / DH;
    float acc = 0.f;
    for (int k = g; k < count; k += G) {
        const bf16_t* vp = p.vc + ((size_t)b * p.Tmax + s_pos[r][k]) * d + h * DH;
        acc += s_p[r][k] * bf2f(vp[e]);
    }
#pragma unroll
    for (int o = DH; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (g == 0) s_out[r][e] = acc;
    __syncthreads();
    if (r == 0 && lane < DH) {
        float o = s_out[0][lane];
        if (p.n_h > 1) {                              // out = sum_r softmax_r(lse_r) out_r   (HF515:636-655)
            float mx = s_lse[0];
            for (int rr = 1; rr < p.n_h; rr++) mx = fmaxf(mx, s_lse[rr]);
            float den = 0.f, num = 0.f;
            for (int rr = 0; rr < p.n_h; rr++) {
                const float w = __expf(s_lse[rr] - mx);
                den += w;
                num += w * s_out[rr][lane];
            }
            o = num / den;
        }
        p.out[(size_t)b * d + h * DH + lane] = f2bf(o);
    }
}

}  // namespace

extern "C" int mxl_rf_decode_embed(const void* ids, int ld_ids, int t, const void* E, const float* W0, const float* W1, void* out,
                                   int B, int d, int V, int A1, int d0, void* stream) {
    MXL_CHECK_ARG(ids && E && W0 && W1 && out && B > 0 && d > 0 && t >= 0 && A1 > 0 && d0 > 0 && d0 < d);
    hipLaunchKernelGGL(rf_decode_embed_kernel, dim3((B * d + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)ids, ld_ids, t, (const bf16_t*)E, W0, W1, (bf16_t*)out, B, d, V, A1, d0);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_lsh_fix_buckets(int* buckets, int rows, int n_h, int T, int T_real, int NB, void* stream) {
    MXL_CHECK_ARG(buckets && rows > 0 && n_h > 0 && T > 0 && T_real > 0 && T_real <= T && NB > 0);
    const long long n = (long long)rows * n_h * T;
    hipLaunchKernelGGL(lsh_fix_buckets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, buckets, n, T,
                       T_real, n_h, NB);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_rf_query_bucket(const int* raw, int* cache, int* bkmax, int rows, int n_h, int NB, int Tmax, int t,
                                   void* stream) {
    MXL_CHECK_ARG(raw && cache && bkmax && rows > 0 && n_h > 0 && rows * n_h <= 1024 && NB > 0 && t >= 0 && t < Tmax);
    hipLaunchKernelGGL(rf_query_bucket_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, raw, cache, bkmax, rows, n_h, NB, Tmax, t);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_rf_decode_attn(const void* q, int ldq, const void* kcache, const void* vcache, const int* sorted, void* out,
                                  int B, int H, int dh, int n_h, int Tmax, int n, int t, int start, int count, int lsh,
                                  void* stream) {
    MXL_CHECK_ARG(q && kcache && vcache && out && B > 0 && H > 0 && n_h >= 1 && n_h <= 4 && Tmax > 0 && t >= 0 && t < Tmax);
    MXL_CHECK_ARG((ldq % 8) == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)kcache % 16) == 0);
    if (sorted) MXL_CHECK_ARG(n == t + 1 && n >= 64 && lsh);        // a sorted window exists once 64 positions are bucketed
    else MXL_CHECK_ARG(n_h == 1 && start >= 0 && count >= 1 && count <= 128 && start + count <= Tmax);
    DecAttnP p;
    p.q = (const bf16_t*)q; p.kc = (const bf16_t*)kcache; p.vc = (const bf16_t*)vcache; p.sorted = sorted; p.out = (bf16_t*)out;
    p.B = B; p.H = H; p.n_h = n_h; p.Tmax = Tmax; p.n = n; p.t = t; p.start = start; p.count = count; p.lsh = lsh; p.ldq = ldq;
    dim3 grid(H, B), block(64 * n_h);
    hipStream_t s = (hipStream_t)stream;
    switch (dh) {
        case 16: hipLaunchKernelGGL(rf_decode_attn_kernel<16>, grid, block, 0, s, p); break;
        case 32: hipLaunchKernelGGL(rf_decode_attn_kernel<32>, grid, block, 0, s, p); break;
        case 64: hipLaunchKernelGGL(rf_decode_attn_kernel<64>, grid, block, 0, s, p); break;
        default: return MXL_EUNSUPPORTED;
    }
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}
