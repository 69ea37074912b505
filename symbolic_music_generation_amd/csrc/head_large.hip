// Projected adaptive log-softmax for LARGE vocabularies (sub-word tokenizers: musicnlp/trainer/wordpiece_tokenizer.py:349-452
// reach V = 262144, for which musicnlp/models/transformer_xl.py:53-66 sets cutoffs [20000, 40000, 200000]; >= 32768 -> [10000];
// >= 16384 -> [5000]).  head.hip works on ONE (tokens x (V + clusters)) logit matrix, which at these sizes is hundreds of GB.
// Here the head follows upstream's own structure (ProjectedAdaptiveLogSoftmax.forward with labels, transformers 4.25.1
// modeling_transfo_xl_utilities.py): every token takes the HEAD softmax over the c1 shortlist tokens + one column per tail
// cluster; a token whose label lies in tail cluster i additionally takes the TAIL softmax of that cluster only.  The tokens of a
// cluster are bucketed (upstream: `mask_i.nonzero()` / index_select), their hidden rows gathered, and the tail GEMM runs over
// those rows alone; the host walks tokens in chunks, so no (tokens x V) tensor ever exists.
//
//   mxl_cluster_bucket      labels -> per-cluster token lists (stable), counts, per-token target columns
//   mxl_gather_rows_bf16    dst[j] = src[idx[j]]                 (hidden rows of one cluster)
//   mxl_scatter_add_rows_bf16  dst[idx[j]] += src[j]             (their hidden gradients back; idx unique)
//   mxl_rows_lse_pick       per logit row: logsumexp over the columns and the target column's logit
//   mxl_bucket_nll_finish   nll[b][t] = (head_lse - head_pick) + (tail_lse - tail_pick); loss sums
//   mxl_rows_softmax_grad   d logits (bf16, optionally a two-term sum) = (softmax - onehot(target)) * grad_scale / count
#include "common.h"
#include "musicxl_internal.h"

namespace {

struct BucketGeom {
    int V, ncl;
    int cut[5];      // cut[0] = 0, cut[1] = c1, ..., cut[ncl + 1] = V
};

// one block per group g: 0 = shortlist tokens, 1..ncl = tail clusters, ncl + 1 = ignored rows (label -100 / out of range /
// the last position of a sequence: hidden[:, :-1]).  perm[g][0 .. counts[g]) = token rows b*T + t in increasing order.
__global__ __launch_bounds__(1024) void cluster_bucket_kernel(const long long* labels, int B, int T, BucketGeom g, int* perm,
                                                              int* counts, int* tgt_head, int* tgt_tail) {
    __shared__ int wtot[16];
    __shared__ int wbase[17];
    const int grp = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int N = B * T;
    int* out = perm + (size_t)grp * N;
    int running = 0;
    for (int base = 0; base < N; base += 1024) {
        const int row = base + tid;
        int ci = g.ncl + 1;
        long long lab = -100;
        if (row < N) {
            const int b = row / T, t = row - b * T;
            if (t < T - 1) {
                lab = labels[(size_t)b * T + t + 1];
                if (lab >= 0 && lab < g.V) {
                    ci = 0;
                    for (int i = 1; i <= g.ncl; i++) if (lab >= g.cut[i]) ci = i;
                }
            }
        }
        const bool mine = (row < N) && (ci == grp);
        const unsigned long long bal = __ballot(mine);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wid] = __popcll(bal);
        __syncthreads();
        if (tid == 0) {
            int s = 0;
            for (int w = 0; w < 16; w++) { wbase[w] = s; s += wtot[w]; }
            wbase[16] = s;
        }
        __syncthreads();
        if (mine) {
            out[running + wbase[wid] + before] = row;
            if (grp == g.ncl + 1) { tgt_head[row] = -1; tgt_tail[row] = -1; }
            else if (grp == 0) { tgt_head[row] = (int)lab; tgt_tail[row] = -1; }
            else { tgt_head[row] = g.cut[1] + grp - 1; tgt_tail[row] = (int)lab - g.cut[grp]; }
        }
        running += wbase[16];
        __syncthreads();
    }
    if (tid == 0) counts[grp] = running;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* src, int ld_src, const int* idx, bf16_t* dst, int n, int d) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n) return;
    const bf16_t* s = src + (size_t)idx[j] * ld_src;
    bf16_t* o = dst + (size_t)j * d;
    for (int c = lane * 8; c < d; c += 512) *reinterpret_cast<u32x4*>(o + c) = *reinterpret_cast<const u32x4*>(s + c);
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const bf16_t* src, const int* idx, bf16_t* dst, int ld_dst, int n, int d) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n) return;
    const bf16_t* s = src + (size_t)j * d;
    bf16_t* o = dst + (size_t)idx[j] * ld_dst;
    for (int c = lane * 8; c < d; c += 512) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(s + c);
        bf16x8 b = *reinterpret_cast<const bf16x8*>(o + c);
#pragma unroll
        for (int k = 0; k < 8; k++) b[k] = (short)f2bf(bf2f((bf16_t)b[k]) + bf2f((bf16_t)a[k]));
        *reinterpret_cast<bf16x8*>(o + c) = b;
    }
}

// one wave per logit row: running (max, sum) per lane over the row, merged across the wave at the end -- the row is read once
__global__ __launch_bounds__(256) void rows_lse_pick_kernel(const float* logits, long long ld, int ncols, int n_rows, const int* rows_idx,
                                                            int row0, const int* tgt, float* lse_out, float* pick_out) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n_rows) return;
    const int tok = rows_idx ? rows_idx[j] : row0 + j;
    const float* lr = logits + (size_t)j * ld;
    float m = -INFINITY, s = 0.f;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
    int c0 = 0;
    if (vec) {
        const int n4 = ncols & ~3;
        for (int c = lane * 4; c < n4; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(lr + c);
            const float vm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            if (vm > m) { s *= __expf(m - vm); m = vm; }
            s += __expf(v[0] - m) + __expf(v[1] - m) + __expf(v[2] - m) + __expf(v[3] - m);
        }
        c0 = n4;
    }
    for (int c = c0 + lane; c < ncols; c += 64) {
        const float v = lr[c];
        if (v > m) { s *= __expf(m - v); m = v; }
        s += __expf(v - m);
    }
    const float mw = wave_max(m);
    s = (m == -INFINITY) ? 0.f : s * __expf(m - mw);
    s = wave_sum(s);
    if (lane == 0) {
        lse_out[tok] = mw + __logf(s);
        const int t = tgt[tok];
        pick_out[tok] = (t >= 0 && t < ncols) ? lr[t] : 0.f;
    }
}

__global__ __launch_bounds__(256) void bucket_nll_finish_kernel(const float* hlse, const float* hpick, const float* tlse,
                                                                const float* tpick, const int* tgt_head, const int* tgt_tail,
                                                                float* nll, float* nll_tok, float* acc, int B, int T) {
    __shared__ float part[8];
    const int row = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float v = 0.f;
    const int N = B * T;
    if (row < N) {
        const int b = row / T, t = row - b * T;
        if (tgt_head[row] >= 0) {
            v = hlse[row] - hpick[row];
            if (tgt_tail[row] >= 0) v += tlse[row] - tpick[row];
        }
        nll_tok[row] = v;
        if (t < T - 1) nll[(size_t)b * (T - 1) + t] = v;
    }
    const float s = wave_sum(v), c = wave_sum(v != 0.f ? 1.f : 0.f);
    if (lane == 0) { part[wid] = s; part[4 + wid] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc, (part[0] + part[1]) + (part[2] + part[3]));
        atomicAdd(acc + 1, (part[4] + part[5]) + (part[6] + part[7]));
    }
}

__global__ __launch_bounds__(256) void rows_softmax_grad_kernel(const float* logits, long long ld, int ncols, int n_rows,
                                                                const int* rows_idx, int row0, const int* tgt, const float* lse,
                                                                const float* nll_tok, const float* acc, float gscale, bf16_t* out,
                                                                bf16_t* out_lo, int ldo) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n_rows) return;
    const int tok = rows_idx ? rows_idx[j] : row0 + j;
    bf16_t* o = out + (size_t)j * ldo;
    bf16_t* ol = out_lo ? out_lo + (size_t)j * ldo : nullptr;
    const int t = tgt[tok];
    if (t < 0 || nll_tok[tok] == 0.f) {      // ignored label, or a per-token loss of exactly 0 (dropped by `losses != 0`)
        for (int c = lane; c < ldo; c += 64) { o[c] = 0; if (ol) ol[c] = 0; }
        return;
    }
    const float gs = gscale / fmaxf(acc[1], 1.f);
    const float l = lse[tok];
    const float* lr = logits + (size_t)j * ld;
    for (int c = lane; c < ldo; c += 64) {
        float dv = 0.f;
        if (c < ncols) dv = (__expf(lr[c] - l) - (c == t ? 1.f : 0.f)) * gs;
        const bf16_t hi = f2bf(dv);
        o[c] = hi;
        if (ol) ol[c] = f2bf(dv - bf2f(hi));
    }
}

int make_bgeom(BucketGeom& g, int V, int ncl, const int* cutoffs) {
    if (V <= 0 || ncl < 1 || ncl > 3 || !cutoffs) return MXL_EINVAL;
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

extern "C" int mxl_cluster_bucket(const void* labels, int B, int T, int V, int ncl, const int* cutoffs_host, int* perm, int* counts,
                                  int* tgt_head, int* tgt_tail, void* stream) {
    MXL_CHECK_ARG(labels && perm && counts && tgt_head && tgt_tail && B > 0 && T > 1);
    BucketGeom g;
    const int rc = make_bgeom(g, V, ncl, cutoffs_host);
    if (rc) return rc;
    hipLaunchKernelGGL(cluster_bucket_kernel, dim3(ncl + 2), dim3(1024), 0, (hipStream_t)stream, (const long long*)labels, B, T, g,
                       perm, counts, tgt_head, tgt_tail);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_gather_rows_bf16(const void* src, int ld_src, const int* idx, void* dst, int n, int d, void* stream) {
    MXL_CHECK_ARG(src && idx && dst && n > 0 && d > 0 && (d % 8) == 0 && (ld_src % 8) == 0);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, idx,
                       (bf16_t*)dst, n, d);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_scatter_add_rows_bf16(const void* src, const int* idx, void* dst, int ld_dst, int n, int d, void* stream) {
    MXL_CHECK_ARG(src && idx && dst && n > 0 && d > 0 && (d % 8) == 0 && (ld_dst % 8) == 0);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, idx,
                       (bf16_t*)dst, ld_dst, n, d);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_rows_lse_pick(const float* logits, long long ld, int ncols, int n_rows, const int* rows_idx, int row0,
                                 const int* tgt, float* lse_out, float* pick_out, void* stream) {
    MXL_CHECK_ARG(logits && tgt && lse_out && pick_out && n_rows > 0 && ncols > 0 && ld >= ncols);
    hipLaunchKernelGGL(rows_lse_pick_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, ncols, n_rows,
                       rows_idx, row0, tgt, lse_out, pick_out);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_bucket_nll_finish(const float* head_lse, const float* head_pick, const float* tail_lse, const float* tail_pick,
                                     const int* tgt_head, const int* tgt_tail, float* nll, float* nll_tok, float* acc2, int B, int T,
                                     void* stream) {
    MXL_CHECK_ARG(head_lse && head_pick && tail_lse && tail_pick && tgt_head && tgt_tail && nll && nll_tok && acc2 && B > 0 && T > 1);
    hipLaunchKernelGGL(bucket_nll_finish_kernel, dim3((B * T + 255) / 256), dim3(256), 0, (hipStream_t)stream, head_lse, head_pick,
                       tail_lse, tail_pick, tgt_head, tgt_tail, nll, nll_tok, acc2, B, T);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_rows_softmax_grad(const float* logits, long long ld, int ncols, int n_rows, const int* rows_idx, int row0,
                                     const int* tgt, const float* lse, const float* nll_tok, const float* acc2, float grad_scale,
                                     void* out_hi, void* out_lo, int ldo, void* stream) {
    MXL_CHECK_ARG(logits && tgt && lse && nll_tok && acc2 && out_hi && n_rows > 0 && ncols > 0 && ld >= ncols && ldo >= ncols);
    hipLaunchKernelGGL(rows_softmax_grad_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, ncols, n_rows,
                       rows_idx, row0, tgt, lse, nll_tok, acc2, grad_scale, (bf16_t*)out_hi, (bf16_t*)out_lo, ldo);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}
