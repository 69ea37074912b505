// Sampler for vocabularies beyond the LDS sort of decode.hip (V > 2048: the WordPiece / pair-merge vocabularies of
// musicnlp/trainer/wordpiece_tokenizer.py:349-452 with the cutoff ladder of musicnlp/models/transformer_xl.py:53-66).
// Same recipe and order as sample_kernel -- HF GenerationMixin.sample / greedy_search: repetition penalty -> temperature ->
// top-k -> top-p -> typical-p -> renormalise -> multinomial (musicnlp/trainer/eval.py:277-333 builds these arguments) --
// without sorting the row.  Every warper is a statement of the form "token i stays iff the weight of the tokens ORDERED BEFORE it
// is below a target" (top-k: order by score, weight 1, target k; top-p: order by score, weight p, target top_p; typical-p: order
// by |-log p - H| ascending, weight p, target typical_p), and so is the multinomial draw (order by index, weight p, target
// u * total: the token where the running sum crosses is the sample).  One routine serves all four: a bisection over the 32-bit
// order key for the largest key value whose "weight at or above it" still reaches the target -- 32 passes over the row, each a
// conditional sum with no atomics and a fixed reduction order (weights are 31-bit fixed point, sums 64-bit integers: the result
// does not depend on timing, so eager steps and hipGraph replays agree) -- and ties at that key value are resolved by index with
// the same routine.  One 1024-thread workgroup per row; the row's scores and weights live in a caller-provided scratch
// (8 bytes per vocabulary entry) that stays in L2.  ~30 us per selection at V = 32768: generation at these vocabularies streams
// V * d * 2 bytes of head weights per step anyway.
#include "common.h"
#include "musicxl_internal.h"

namespace {

constexpr int SL_NT = 1024;
typedef unsigned long long u64;

__device__ __forceinline__ uint32_t ord_f32(float x) {       // monotone: larger float -> larger uint (no NaNs here)
    const uint32_t b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

struct Red {
    u64 part[SL_NT / 64];
    float fpart[SL_NT / 64];
    int ipart[SL_NT / 64];
};

// block-wide sums / max in a fixed order (lane tree, then the 16 wave partials in wave order): deterministic
__device__ __forceinline__ u64 block_sum_u64(u64 v, Red& r) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, o, 64), hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
        v += ((u64)hi << 32) | lo;
    }
    __syncthreads();                 // the previous use of the partials is over
    if ((threadIdx.x & 63) == 0) r.part[threadIdx.x >> 6] = v;
    __syncthreads();
    u64 s = 0;
#pragma unroll
    for (int w = 0; w < SL_NT / 64; w++) s += r.part[w];
    return s;
}
__device__ __forceinline__ float block_sum_f32(float v, Red& r) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) r.fpart[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < SL_NT / 64; w++) s += r.fpart[w];
    return s;
}
__device__ __forceinline__ float block_max_f32(float v, Red& r) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) r.fpart[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = -INFINITY;
#pragma unroll
    for (int w = 0; w < SL_NT / 64; w++) s = fmaxf(s, r.fpart[w]);
    return s;
}

// The selection.  key(i) -> 32-bit order key (larger = earlier in the order), wt(i) -> weight (0 = not a candidate).
// Returns the crossing element: with the candidates ordered by (key descending, index ascending) it is the first one whose
// inclusive running weight reaches `target` (1 <= target <= total weight).  `ustar` / `istar` describe the kept prefix:
// element i is at or before the crossing iff key(i) > ustar || (key(i) == ustar && i <= istar).
template <class KeyF, class WtF>
__device__ __forceinline__ void select_cross(int V, u64 target, KeyF key, WtF wt, Red& r, uint32_t& ustar, int& istar) {
    const int tid = threadIdx.x;
    // largest u with W{key >= u} >= target
    uint32_t lo = 0u, hi = 0xFFFFFFFFu;
    while (lo < hi) {
        const uint32_t mid = lo + (uint32_t)(((u64)hi - lo + 1) >> 1);      // upper middle: lo < mid <= hi
        u64 acc = 0;
        for (int i = tid; i < V; i += SL_NT) acc += (key(i) >= mid) ? wt(i) : 0;
        if (block_sum_u64(acc, r) >= target) lo = mid; else hi = mid - 1;
    }
    ustar = lo;
    u64 acc = 0;
    for (int i = tid; i < V; i += SL_NT) acc += (key(i) > lo) ? wt(i) : 0;
    const u64 before = block_sum_u64(acc, r);
    // among the candidates with key == ustar, in index order: smallest index I with W{key == ustar, i <= I} >= target - before
    const u64 need = target - before;
    int ilo = 0, ihi = V - 1;
    while (ilo < ihi) {
        const int mid = ilo + ((ihi - ilo) >> 1);                            // lower middle
        u64 a2 = 0;
        for (int i = tid; i <= mid; i += SL_NT) a2 += (key(i) == lo) ? wt(i) : 0;
        if (block_sum_u64(a2, r) >= need) ihi = mid; else ilo = mid + 1;
    }
    istar = ilo;
}

__global__ __launch_bounds__(SL_NT) void sample_large_kernel(const float* logp, int ldl, int V, long long* ids, int ld_ids,
                                                             const int* t_dev, const unsigned long long* rng_ctr,
                                                             unsigned long long seed, int do_sample, int top_k, float top_p,
                                                             float temperature, float repetition_penalty, float typical_p,
                                                             float* out_probs, float* scratch) {
    __shared__ Red r;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logp + (size_t)b * ldl;
    float* K = scratch + (size_t)b * 2 * V;                               // scores (after penalty and temperature); -inf = dropped
    uint32_t* W = reinterpret_cast<uint32_t*>(K + V);                     // fixed-point weights exp(K - max) * 2^31; 0 = dropped
    const float invt = 1.f / temperature;
    const int t = *t_dev;
    for (int i = tid; i < V; i += SL_NT) K[i] = row[i] * invt;
    __syncthreads();
    if (repetition_penalty != 1.f) {       // HF RepetitionPenaltyLogitsProcessor (see sample_kernel): duplicates store the same value
        const long long* hist = ids + (size_t)b * ld_ids;
        for (int j = tid; j <= t; j += SL_NT) {
            const long long tok = hist[j];
            if (tok >= 0 && tok < V) {
                const float v = row[tok];
                K[tok] = (v < 0.f ? v * repetition_penalty : v / repetition_penalty) * invt;
            }
        }
        __syncthreads();
    }
    long long* dst = ids + (size_t)b * ld_ids + t + 1;
    float m;
    {
        float mx = -INFINITY;
        for (int i = tid; i < V; i += SL_NT) mx = fmaxf(mx, K[i]);
        m = block_max_f32(mx, r);
    }
    if (!do_sample) {                      // argmax, ties -> lowest index (torch.argmax)
        int best = 0x7fffffff;
        for (int i = tid; i < V; i += SL_NT)
            if (K[i] == m) { best = i; break; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
        __syncthreads();
        if ((tid & 63) == 0) r.ipart[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            int bb = r.ipart[0];
            for (int w = 1; w < SL_NT / 64; w++) bb = min(bb, r.ipart[w]);
            *dst = bb == 0x7fffffff ? 0 : bb;
        }
        return;
    }
    auto kkey = [&](int i) { return ord_f32(K[i]); };
    auto drop_after = [&](uint32_t ustar, int istar, auto key) {          // everything ordered after the crossing element goes
        for (int i = tid; i < V; i += SL_NT) {
            const uint32_t u = key(i);
            if (!(u > ustar || (u == ustar && i <= istar))) K[i] = -INFINITY;
        }
        __syncthreads();
    };
    auto weights = [&]() -> u64 {                                            // W from K; returns the total
        u64 acc = 0;
        for (int i = tid; i < V; i += SL_NT) {
            const float k = K[i];
            // a surviving score keeps a weight of at least 1 unit (2^-31 of the largest): it can still be drawn, never for free
            const uint32_t w = k == -INFINITY ? 0u : max(1u, (uint32_t)(__expf(k - m) * 2147483648.f));
            W[i] = w;
            acc += w;
        }
        const u64 tot = block_sum_u64(acc, r);
        return tot;                                                          // (block_sum's barriers also publish W)
    };
    // ---- top-k: order by score, weight 1 per surviving score, target k
    if (top_k > 0 && top_k < V) {
        uint32_t us; int is;
        select_cross(V, (u64)top_k, kkey, [&](int i) { return (u64)(K[i] != -INFINITY); }, r, us, is);
        drop_after(us, is, kkey);
    }
    // ---- top-p (HF TopPLogitsWarper: token stays iff the probability mass of the tokens ordered before it is below top_p)
    if (top_p > 0.f && top_p < 1.f) {
        const u64 tot = weights();
        const double x = (double)top_p * (double)tot;
        u64 target = (u64)x;
        if ((double)target < x) target++;                                   // ceil: (integer sum < x) <=> (integer sum < ceil(x))
        target = target < 1 ? 1 : (target > tot ? tot : target);
        uint32_t us; int is;
        select_cross(V, target, kkey, [&](int i) { return (u64)W[i]; }, r, us, is);
        drop_after(us, is, kkey);
    }
    // ---- typical-p (HF TypicalLogitsWarper): order by |-log p - H| ascending over the surviving support
    if (typical_p > 0.f && typical_p < 1.f) {
        const u64 tot = weights();
        float zs = 0.f;
        for (int i = tid; i < V; i += SL_NT) zs += K[i] == -INFINITY ? 0.f : __expf(K[i] - m);
        const float logz = __logf(block_sum_f32(zs, r));
        float hs = 0.f;
        for (int i = tid; i < V; i += SL_NT) {
            const float k = K[i];
            if (k != -INFINITY) {
                const float nl = k - m - logz, p = __expf(nl);
                if (p > 0.f) hs -= p * nl;
            }
        }
        const float ent = block_sum_f32(hs, r);
        auto dkey = [&](int i) {                                            // smaller deviation -> larger key; dropped -> 0
            const float k = K[i];
            return k == -INFINITY ? 0u : ~__float_as_uint(fabsf(-(k - m - logz) - ent));
        };
        const double x = (double)typical_p * (double)tot;
        u64 target = (u64)x;
        if ((double)target < x) target++;
        target = target < 1 ? 1 : (target > tot ? tot : target);
        uint32_t us; int is;
        select_cross(V, target, dkey, [&](int i) { return (u64)W[i]; }, r, us, is);
        // (dkey reads K: decide first, then drop)
        for (int i = tid; i < V; i += SL_NT) {
            const uint32_t u = dkey(i);
            W[i] = (u > us || (u == us && i <= is)) ? 1u : 0u;
        }
        __syncthreads();
        for (int i = tid; i < V; i += SL_NT)
            if (!W[i]) K[i] = -INFINITY;
        __syncthreads();
    }
    // ---- renormalise and draw: order by index, weight p, target u * total
    const u64 tot = weights();
    const unsigned long long ctr = *rng_ctr;
    const uint32_t h1 = mxl_hash32((uint32_t)(ctr * 0x9E3779B97F4A7C15ULL >> 32) ^ mxl_hash32((uint32_t)b + 0x85ebca6bU * (uint32_t)seed));
    const uint32_t h2 = mxl_hash32(h1 + (uint32_t)ctr + (uint32_t)(seed >> 32));
    const double u01 = (double)(h2 >> 8) * (1.0 / 16777216.0);            // the same uniform as sample_kernel
    u64 target = (u64)(u01 * (double)tot) + 1;
    target = target > tot ? tot : target;
    uint32_t us; int is;
    select_cross(V, target, [&](int i) { return ~(uint32_t)i; }, [&](int i) { return (u64)W[i]; }, r, us, is);
    if (tid == 0) *dst = (long long)(~us);                                  // keys are unique: the crossing element is index ~ustar
    if (out_probs) {
        float zs = 0.f;
        for (int i = tid; i < V; i += SL_NT) zs += K[i] == -INFINITY ? 0.f : __expf(K[i] - m);
        const float z = block_sum_f32(zs, r);
        for (int i = tid; i < V; i += SL_NT) out_probs[(size_t)b * V + i] = K[i] == -INFINITY ? 0.f : __expf(K[i] - m) / z;
    }
}

}  // namespace

extern "C" int mxl_sample_large(const float* logprobs, int ldl, int V, void* ids, int ld_ids, const int* t_dev,
                                const unsigned long long* rng_ctr, unsigned long long seed, int B, int do_sample, int top_k,
                                float top_p, float temperature, float repetition_penalty, float typical_p, float* out_probs,
                                float* scratch, void* stream) {
    MXL_CHECK_ARG(logprobs && ids && t_dev && rng_ctr && scratch && B > 0 && V > 0 && V <= (1 << 30) && temperature > 0.f);
    MXL_CHECK_ARG(repetition_penalty > 0.f && typical_p > 0.f && ldl >= V);
    hipLaunchKernelGGL(sample_large_kernel, dim3(B), dim3(SL_NT), 0, (hipStream_t)stream, logprobs, ldl, V, (long long*)ids,
                       ld_ids, t_dev, rng_ctr, seed, do_sample, top_k, top_p, temperature, repetition_penalty, typical_p,
                       out_probs, scratch);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}
