// Projected adaptive log-softmax head (K7): row kernels over the fp32 logits produced by one GEMM against
// [E ; cluster_weight] (N_tok x (V + n_clusters), ld = ldl).  Restates upstream ProjectedAdaptiveLogSoftmax
// (div_val = 1) as called at musicnlp/models/transformer_xl.py:185,193:
//   cutoffs = []      : logprob = log_softmax(logit)
//   cutoffs = [c1..]  : head = cols [0,c1) + cluster cols [V, V+ncl);  cluster i>=1: cols [c_i, c_{i+1})
//                       logprob[j in cluster i] = head_logprob[cluster col i-1] + tail_logprob_i[j]
// with labels: shift inside (hidden[:, :-1] vs labels[:, 1:]); ignored (-100) labels give 0.
// One wave per token row.
#include "common.h"
#include "musicxl_internal.h"

namespace {

struct HeadGeom {
    int V, ncl;       // vocabulary, number of tail clusters (0..3)
    int cut[5];       // cut[0]=0, cut[1]=c1 (== V when ncl == 0), ..., cut[ncl+1] = V
};

__device__ __forceinline__ float wave_lse_range(const float* row, int lo, int hi, int lane, float extra_max,
                                                const float* extra, int n_extra) {
    // logsumexp over row[lo:hi] (+ n_extra values at extra[]), wave-cooperative.  Ranges of up to 64 * 32 columns (every
    // vocabulary of the reference) are read ONCE into registers, all loads in flight together; longer ones take two passes.
    constexpr int RMAX = 32;
    float m = -INFINITY;
    if (hi - lo <= 64 * RMAX) {
        float buf[RMAX];
#pragma unroll
        for (int k = 0; k < RMAX; k++) {
            const int j = lo + lane + 64 * k;
            buf[k] = (j < hi) ? row[j] : -INFINITY;
        }
#pragma unroll
        for (int k = 0; k < RMAX; k++) m = fmaxf(m, buf[k]);
        if (lane < n_extra) m = fmaxf(m, extra[lane]);
        m = wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < RMAX; k++) s += __expf(buf[k] - m);          // exp(-inf) = 0 for the padding
        if (lane < n_extra) s += __expf(extra[lane] - m);
        s = wave_sum(s);
        return m + __logf(s);
    }
    for (int j = lo + lane; j < hi; j += 64) m = fmaxf(m, row[j]);
    if (lane < n_extra) m = fmaxf(m, extra[lane]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lo + lane; j < hi; j += 64) s += __expf(row[j] - m);
    if (lane < n_extra) s += __expf(extra[lane] - m);
    s = wave_sum(s);
    return m + __logf(s);
}

// transformer_xl.py:176-182: if every label of row 0 (from position 1 on) is -100, set labels[0,1] = eos.
__global__ void label_guard_kernel(long long* labels, int T, long long eos) {
    __shared__ int any_valid;
    if (threadIdx.x == 0) any_valid = 0;
    __syncthreads();
    for (int t = 1 + threadIdx.x; t < T; t += blockDim.x)
        if (labels[t] != -100) any_valid = 1;
    __syncthreads();
    if (threadIdx.x == 0 && !any_valid && T > 1) labels[1] = eos;
}

// nll[b][t] for t in [0, T-1); acc[0] += sum(nll), acc[1] += count(nll != 0).  lse_out[row][0] = head lse,
// lse_out[row][1] = tail lse of the label's cluster (if any).
constexpr int NLL_ROWS_PER_WAVE = 4;    // 16 token rows per block (one atomic pair per block); more rows per wave leave too few waves to hide the per-row latency chain
__global__ __launch_bounds__(256) void nll_fwd_kernel(const float* logits, int ldl, const long long* labels, float* nll,
                                                      float* lse_out, float* acc, int B, int T, HeadGeom g) {
    __shared__ float part[8];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float tsum = 0.f, tcnt = 0.f;
    for (int rr = 0; rr < NLL_ROWS_PER_WAVE; rr++) {
        const int row = (blockIdx.x * 4 + wid) * NLL_ROWS_PER_WAVE + rr;
        if (row >= B * T) break;
        const int b = row / T, t = row % T;
        if (t == T - 1) continue;  // hidden[:, :-1]
        const long long lab = labels[(size_t)b * T + t + 1];
        float* out = nll + (size_t)b * (T - 1) + t;
        if (lab < 0 || lab >= g.V) {  // -100 (or any out-of-vocab id): no cluster matches -> stays 0
            if (lane == 0) { *out = 0.f; lse_out[2 * (size_t)row] = 0.f; lse_out[2 * (size_t)row + 1] = 0.f; }
            continue;
        }
        const float* lr = logits + (size_t)row * ldl;
        const float head_lse = wave_lse_range(lr, 0, g.cut[1], lane, 0.f, lr + g.V, g.ncl);
        int ci = 0;
        for (int i = 1; i <= g.ncl; i++) if (lab >= g.cut[i]) ci = i;
        float v, tail_lse = 0.f;
        if (ci == 0) {
            v = head_lse - lr[lab];
        } else {
            tail_lse = wave_lse_range(lr, g.cut[ci], g.cut[ci + 1], lane, 0.f, nullptr, 0);
            v = (head_lse - lr[g.V + ci - 1]) + (tail_lse - lr[lab]);
        }
        if (lane == 0) {
            *out = v;
            lse_out[2 * (size_t)row] = head_lse;
            lse_out[2 * (size_t)row + 1] = tail_lse;
        }
        tsum += v;
        if (v != 0.f) tcnt += 1.f;
    }
    if (lane == 0) { part[wid] = tsum; part[4 + wid] = tcnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc, (part[0] + part[1]) + (part[2] + part[3]));
        atomicAdd(acc + 1, (part[4] + part[5]) + (part[6] + part[7]));
    }
}

// dlogits (bf16, ld = ldd, pad columns zeroed) for loss = sum(nll[nll != 0]) / count  (transformer_xl.py:200).
// dlo (optional, same shape): the bf16 remainder d - bf16(d).  A probability near 1 on a token that is NOT the label (an
// untrained tied-embedding model predicts its own input) is a logit gradient near 1/count whose bf16 spacing, 2^-8 relative,
// is as large as the quantity the column sums of dlogits are made of (sum_t p_t[v] - count_t[label = v]: two nearly equal
// numbers); with the remainder carried as a second bf16 term the GEMMs and column sums that consume dlogits see it to 2^-16.
__global__ __launch_bounds__(256) void nll_bwd_kernel(const float* logits, int ldl, const long long* labels,
                                                      const float* nll, const float* lse_in, const float* acc,
                                                      bf16_t* dlogits, bf16_t* dlo, int ldd, int B, int T, HeadGeom g,
                                                      float gscale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * T) return;
    const int lane = threadIdx.x & 63;
    const int b = row / T, t = row % T;
    bf16_t* dr = dlogits + (size_t)row * ldd;
    bf16_t* dl = dlo ? dlo + (size_t)row * ldd : nullptr;
    long long lab = -100;
    float v = 0.f;
    if (t < T - 1) { lab = labels[(size_t)b * T + t + 1]; v = nll[(size_t)b * (T - 1) + t]; }
    if (lab < 0 || lab >= g.V || v == 0.f) {
        for (int j = lane; j < ldd; j += 64) {
            dr[j] = 0;
            if (dl) dl[j] = 0;
        }
        return;
    }
    const float cnt = fmaxf(acc[1], 1.f);
    const float gs = gscale / cnt;
    const float* lr = logits + (size_t)row * ldl;
    const float head_lse = lse_in[2 * (size_t)row], tail_lse = lse_in[2 * (size_t)row + 1];
    int ci = 0;
    for (int i = 1; i <= g.ncl; i++) if (lab >= g.cut[i]) ci = i;
    const int ncols = g.V + g.ncl;
    for (int j = lane; j < ldd; j += 64) {
        float d = 0.f;
        if (j < g.cut[1]) {
            d = __expf(lr[j] - head_lse) - ((ci == 0 && j == lab) ? 1.f : 0.f);
        } else if (j >= g.V && j < ncols) {
            d = __expf(lr[j] - head_lse) - ((ci > 0 && j == g.V + ci - 1) ? 1.f : 0.f);
        } else if (ci > 0 && j >= g.cut[ci] && j < g.cut[ci + 1]) {
            d = __expf(lr[j] - tail_lse) - (j == lab ? 1.f : 0.f);
        }
        const bf16_t hi = f2bf(d * gs);
        dr[j] = hi;
        if (dl) dl[j] = f2bf(d * gs - bf2f(hi));
    }
}

// full log-probabilities (labels = None branch): out[row][0:V]
__global__ __launch_bounds__(256) void logprob_full_kernel(const float* logits, int ldl, float* out, int ldo, int N, HeadGeom g) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const float* lr = logits + (size_t)row * ldl;
    float* o = out + (size_t)row * ldo;
    const float head_lse = wave_lse_range(lr, 0, g.cut[1], lane, 0.f, lr + g.V, g.ncl);
    for (int j = lane; j < g.cut[1]; j += 64) o[j] = lr[j] - head_lse;
    for (int ci = 1; ci <= g.ncl; ci++) {
        const float tail_lse = wave_lse_range(lr, g.cut[ci], g.cut[ci + 1], lane, 0.f, nullptr, 0);
        const float base = lr[g.V + ci - 1] - head_lse;
        for (int j = g.cut[ci] + lane; j < g.cut[ci + 1]; j += 64) o[j] = base + (lr[j] - tail_lse);
    }
}

int make_geom(HeadGeom& g, int V, int ncl, const int* cutoffs) {
    if (V <= 0 || ncl < 0 || ncl > 3) return MXL_EINVAL;
    if (ncl > 0 && !cutoffs) return MXL_EINVAL;
    g.V = V; g.ncl = ncl; g.cut[0] = 0;
    for (int i = 0; i < ncl; i++) {
        g.cut[i + 1] = cutoffs[i];
        if (cutoffs[i] <= g.cut[i] || cutoffs[i] >= V) return MXL_EINVAL;
    }
    g.cut[ncl + 1] = V;
    for (int i = ncl + 2; i < 5; i++) g.cut[i] = V;
    return MXL_OK;
}

}  // namespace

extern "C" int mxl_label_guard(void* labels_row0, int T, long long eos, void* stream) {
    MXL_CHECK_ARG(labels_row0 && T > 0);
    hipLaunchKernelGGL(label_guard_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (long long*)labels_row0, T, eos);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_adaptive_nll_fwd(const float* logits, int ldl, const void* labels, float* nll, float* lse,
                                    float* acc2, int B, int T, int V, int ncl, const int* cutoffs_host, void* stream) {
    MXL_CHECK_ARG(logits && labels && nll && lse && acc2 && B > 0 && T > 1);
    HeadGeom g;
    int rc = make_geom(g, V, ncl, cutoffs_host);
    if (rc) return rc;
    MXL_CHECK_ARG(ldl >= V + ncl);
    hipLaunchKernelGGL(nll_fwd_kernel, dim3((B * T + 4 * NLL_ROWS_PER_WAVE - 1) / (4 * NLL_ROWS_PER_WAVE)), dim3(256), 0, (hipStream_t)stream, logits, ldl,
                       (const long long*)labels, nll, lse, acc2, B, T, g);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_adaptive_nll_bwd(const float* logits, int ldl, const void* labels, const float* nll, const float* lse,
                                    const float* acc2, void* dlogits, int ldd, int B, int T, int V, int ncl,
                                    const int* cutoffs_host, float grad_scale, void* stream) {
    MXL_CHECK_ARG(logits && labels && nll && lse && acc2 && dlogits && B > 0 && T > 1);
    HeadGeom g;
    int rc = make_geom(g, V, ncl, cutoffs_host);
    if (rc) return rc;
    MXL_CHECK_ARG(ldl >= V + ncl && ldd >= V + ncl);
    hipLaunchKernelGGL(nll_bwd_kernel, dim3((B * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ldl,
                       (const long long*)labels, nll, lse, acc2, (bf16_t*)dlogits, (bf16_t*)nullptr, ldd, B, T, g, grad_scale);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_adaptive_nll_bwd_split(const float* logits, int ldl, const void* labels, const float* nll, const float* lse,
                                          const float* acc2, void* dlogits_hi, void* dlogits_lo, int ldd, int B, int T, int V,
                                          int ncl, const int* cutoffs_host, float grad_scale, void* stream) {
    MXL_CHECK_ARG(logits && labels && nll && lse && acc2 && dlogits_hi && dlogits_lo && B > 0 && T > 1);
    HeadGeom g;
    int rc = make_geom(g, V, ncl, cutoffs_host);
    if (rc) return rc;
    MXL_CHECK_ARG(ldl >= V + ncl && ldd >= V + ncl);
    hipLaunchKernelGGL(nll_bwd_kernel, dim3((B * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ldl,
                       (const long long*)labels, nll, lse, acc2, (bf16_t*)dlogits_hi, (bf16_t*)dlogits_lo, ldd, B, T, g,
                       grad_scale);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_adaptive_logprob(const float* logits, int ldl, float* out, int ldo, int N, int V, int ncl,
                                    const int* cutoffs_host, void* stream) {
    MXL_CHECK_ARG(logits && out && N > 0);
    HeadGeom g;
    int rc = make_geom(g, V, ncl, cutoffs_host);
    if (rc) return rc;
    MXL_CHECK_ARG(ldl >= V + ncl && ldo >= V);
    hipLaunchKernelGGL(logprob_full_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ldl, out, ldo,
                       N, g);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}
