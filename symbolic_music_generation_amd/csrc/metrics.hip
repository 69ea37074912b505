// Evaluation metrics on the device: greedy ids from the logits and the per-sequence counts behind next-token
// accuracy and the in-key ratio, so that only (B, T) ids / (B, 14) integers ever leave the GPU (the reference gathers the full
// (B, T, V) logits to the host: musicnlp/util/train/trainer_eval_wrap.py:310-314, then musicnlp/trainer/train.py:265-284 and
// musicnlp/trainer/metrics.py:45-117 on numpy).
#include "common.h"
#include "musicxl_internal.h"

namespace {

// ids[row] = argmax_v x[row][v], first maximum wins (numpy / torch.argmax on ties); one wave per row
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* x, int ld, long long* out, int N, int V) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (size_t)row * ld;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
        const float f = xr[v];
        if (f > best || (f == best && v < arg)) { best = f; arg = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oa = __shfl_xor(arg, o, 64);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) out[row] = (arg == 0x7fffffff) ? 0 : arg;      // all-NaN / -inf row: id 0
}

// One workgroup per sequence.  out[b] = { pitch-class histogram [12] of the predicted pitch tokens at non-ignored positions,
// number of correct next-token predictions, number of non-ignored next-token positions }.
//   labels (B, T);  preds (B, T - 1) when `shifted` (pred[j] answers label[j+1]), else (B, T) (pred[j] sits at position j,
//   answers label[j+1]).   In-key positions (metrics.py:45-52): pred[j] where label'[j] != -100, label' = labels[:, 1:] if shifted.
//   Accuracy positions (train.py:277-283): pairs (pred[j], label[j+1]), j < T - 1, label[j+1] != -100.
__global__ __launch_bounds__(256) void eval_counts_kernel(const long long* preds, int ldp, const long long* labels, int ldl,
                                                          const signed char* id2pc, int V, int* out, int T, int shifted) {
    __shared__ int hist[14];
    if (threadIdx.x < 14) hist[threadIdx.x] = 0;
    __syncthreads();
    const int b = blockIdx.x;
    const long long* pr = preds + (size_t)b * ldp;
    const long long* lb = labels + (size_t)b * ldl;
    int correct = 0, total = 0;
    const int np = shifted ? T - 1 : T;
    for (int j = threadIdx.x; j < np; j += 256) {
        const long long p = pr[j];
        const long long lk = shifted ? lb[j + 1] : lb[j];              // in-key mask
        if (lk != -100 && p >= 0 && p < V) {
            const int pc = id2pc[p];
            if (pc >= 0) atomicAdd(&hist[pc], 1);
        }
        if (j < T - 1) {
            const long long ln = lb[j + 1];
            if (ln != -100) { total++; correct += (p == ln) ? 1 : 0; }
        }
    }
    correct = (int)wave_sum((float)correct);      // exact below 2^24 positions per wave
    total = (int)wave_sum((float)total);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&hist[12], correct); atomicAdd(&hist[13], total); }
    __syncthreads();
    if (threadIdx.x < 14) out[b * 14 + threadIdx.x] = hist[threadIdx.x];
}

// the same for SUB-WORD vocabularies (WordPiece / pair-merge ids stand for several base tokens: the reference's
// `WordPieceMusicTokenizer.ids2pitches` / `PairMergeTokenizer.ids2pitches`, wordpiece_tokenizer.py:450-452, expand every id into the
// pitches of its base tokens): id2hist[v][pc] = how many pitches of class pc id v stands for
__global__ __launch_bounds__(256) void eval_counts_multi_kernel(const long long* preds, int ldp, const long long* labels, int ldl,
                                                                const unsigned char* id2hist, int V, int* out, int T, int shifted) {
    __shared__ int hist[14];
    if (threadIdx.x < 14) hist[threadIdx.x] = 0;
    __syncthreads();
    const int b = blockIdx.x;
    const long long* pr = preds + (size_t)b * ldp;
    const long long* lb = labels + (size_t)b * ldl;
    int correct = 0, total = 0;
    const int np = shifted ? T - 1 : T;
    for (int j = threadIdx.x; j < np; j += 256) {
        const long long p = pr[j];
        const long long lk = shifted ? lb[j + 1] : lb[j];
        if (lk != -100 && p >= 0 && p < V) {
            const unsigned char* h = id2hist + (size_t)p * 12;
#pragma unroll
            for (int pc = 0; pc < 12; pc++) if (h[pc]) atomicAdd(&hist[pc], (int)h[pc]);
        }
        if (j < T - 1) {
            const long long ln = lb[j + 1];
            if (ln != -100) { total++; correct += (p == ln) ? 1 : 0; }
        }
    }
    correct = (int)wave_sum((float)correct);
    total = (int)wave_sum((float)total);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&hist[12], correct); atomicAdd(&hist[13], total); }
    __syncthreads();
    if (threadIdx.x < 14) out[b * 14 + threadIdx.x] = hist[threadIdx.x];
}

}  // namespace

extern "C" int mxl_argmax_rows(const float* logits, int ld, void* ids_out, int N, int V, void* stream) {
    MXL_CHECK_ARG(logits && ids_out && N > 0 && V > 0 && ld >= V);
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, (long long*)ids_out, N, V);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_eval_counts(const void* preds, int ld_preds, const void* labels, int ld_labels, const signed char* id2pc,
                               int V, int* out14, int B, int T, int clm_pred_shifted, void* stream) {
    MXL_CHECK_ARG(preds && labels && id2pc && out14 && B > 0 && T > 1 && V > 0);
    MXL_CHECK_ARG(ld_labels >= T && ld_preds >= (clm_pred_shifted ? T - 1 : T) && T < (1 << 24));
    hipLaunchKernelGGL(eval_counts_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long long*)preds, ld_preds,
                       (const long long*)labels, ld_labels, id2pc, V, out14, T, clm_pred_shifted);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_eval_counts_multi(const void* preds, int ld_preds, const void* labels, int ld_labels, const unsigned char* id2hist,
                                     int V, int* out14, int B, int T, int clm_pred_shifted, void* stream) {
    MXL_CHECK_ARG(preds && labels && id2hist && out14 && B > 0 && T > 1 && V > 0);
    MXL_CHECK_ARG(ld_labels >= T && ld_preds >= (clm_pred_shifted ? T - 1 : T) && T < (1 << 24));
    hipLaunchKernelGGL(eval_counts_multi_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long long*)preds, ld_preds,
                       (const long long*)labels, ld_labels, id2hist, V, out14, T, clm_pred_shifted);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Input pipeline: a batch arrives as ONE packed run of tokens (uint16 / int32, as stored in the pre-tokenised
// file) plus B+1 offsets; this kernel produces what `tokenizer(toks, padding='max_length', truncation=True)`
// (musicnlp/preprocess/dataset.py:361) followed by DataCollatorForLanguageModeling(mlm=False) (train.py:360) hands the model:
//   ids[b][t]    = tok[off[b] + t]  for t < min(len_b, max_length), pad_id beyond
//   labels[b][t] = ids[b][t], with every pad_id replaced by -100
// ---------------------------------------------------------------------------------------------------------------------
namespace {
template <typename TOK>
__global__ __launch_bounds__(256) void pack_clm_kernel(const TOK* tok, const int* off, long long* ids, long long* labels, int B,
                                                       int L, long long pad_id, const int* remap, const int* row_table, int Vsrc) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)B * L) return;
    const int b = (int)(gid / L), t = (int)(gid % L);
    const int o0 = off[b], n = off[b + 1] - o0;
    long long v = pad_id;
    if (t < n) {
        v = (long long)tok[o0 + t];
        // step-pitch -> degree-pitch (and any other vocabulary change) as a table look-up: table row_table[b] of `remap`
        if (remap && row_table[b] >= 0 && v >= 0 && v < Vsrc) v = remap[(size_t)row_table[b] * Vsrc + v];
    }
    ids[gid] = v;
    if (labels) labels[gid] = (v == pad_id) ? -100 : v;
}

// out[b] = index of the last (which < 0) or the which-th (0-based) occurrence of `token` in row b, -1 if there is none; one wave
// per row, 64 positions per ballot
__global__ __launch_bounds__(64) void find_token_kernel(const long long* ids, int ld, int T, long long token, int which, int* out) {
    const int b = blockIdx.x, l = threadIdx.x;
    const long long* row = ids + (size_t)b * ld;
    int seen = 0, res = -1;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + l;
        const bool hit = t < T && row[t] == token;
        const unsigned long long m = __ballot(hit);
        if (which < 0) {
            if (m) res = t0 + 63 - __clzll(m);
        } else {
            const int c = __popcll(m);
            if (res < 0 && seen + c > which) {
                unsigned long long mm = m;
                for (int i = which - seen; i > 0; i--) mm &= mm - 1;      // drop the lower set bits
                res = t0 + __ffsll((long long)mm) - 1;
            }
            seen += c;
        }
    }
    if (l == 0) out[b] = res;
}
}  // namespace

extern "C" int mxl_find_token(const void* ids, int ld_ids, int B, int T, long long token, int which, int* out, void* stream) {
    MXL_CHECK_ARG(ids && out && B > 0 && T > 0 && ld_ids >= T);
    hipLaunchKernelGGL(find_token_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const long long*)ids, ld_ids, T, token, which,
                       out);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_pack_clm_batch(const void* tokens, int elem_bytes, const int* offsets, void* ids_out, void* labels_out, int B,
                                  int max_length, long long pad_id, const int* remap, const int* row_table, int v_src,
                                  void* stream) {
    MXL_CHECK_ARG(tokens && offsets && ids_out && B > 0 && max_length > 0 && (elem_bytes == 2 || elem_bytes == 4));
    MXL_CHECK_ARG(!remap || (row_table && v_src > 0));
    const long long n = (long long)B * max_length;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (elem_bytes == 2)
        hipLaunchKernelGGL(pack_clm_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)tokens,
                           offsets, (long long*)ids_out, (long long*)labels_out, B, max_length, pad_id, remap, row_table, v_src);
    else
        hipLaunchKernelGGL(pack_clm_kernel<int>, grid, dim3(256), 0, (hipStream_t)stream, (const int*)tokens, offsets,
                           (long long*)ids_out, (long long*)labels_out, B, max_length, pad_id, remap, row_table, v_src);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}
