// Autoregressive decode step for Transformer-XL: everything a step needs lives on the device
// (position counter, RNG counter, token buffer), so the whole step is capture-safe and replays as a hipGraph.
//
// Upstream recomputes K/V of all mem_len memory rows every step (cat(mems, h) through qkv_net).  Because qkv_net has no
// bias, caching the *projected* K/V rows is numerically the same computation done once: a ring of M slots per layer,
// slot = position mod M.  A query at position t sees distances d = 0..M-1 (same_length window); slots never written
// are the zero mems of `init_mems` (k = v = 0): they still take softmax mass through the positional term, exactly as
// upstream, because the cache is zero-initialised and BD is evaluated for all M distances.
#include "common.h"
#include "musicxl_internal.h"

namespace {

// h0[b] = E[ids[b][t]] * scale          (t read from device memory)
__global__ void decode_embed_kernel(const long long* ids, int ld_ids, const int* t_dev, const bf16_t* E, bf16_t* out, int B,
                                    int d, int V, float scale) {
    const int chunks = d >> 3;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * chunks) return;
    const int b = gid / chunks, c = gid % chunks;
    long long id = ids[(size_t)b * ld_ids + *t_dev];
    if (id < 0 || id >= V) id = 0;
    const bf16x8 e = *reinterpret_cast<const bf16x8*>(E + (size_t)id * d + c * 8);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = bf2f((bf16_t)e[j]) * scale;
    u32x4 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
    *reinterpret_cast<u32x4*>(out + (size_t)b * d + c * 8) = o;
}

// cache[b][t mod M] = (k, v) of the current token (rows of the (B, 3d) qkv buffer)
// ring layout is head-major (B, H, M, dh): the attention kernel of one (b, h) then streams one contiguous 2*M*dh-byte
// region instead of 128-byte pieces at a d-element stride (DRAM page locality)
// optionally also qr[b][:] = q + r_r_bias (bf16), the operand of the per-step BD product -- one launch instead of two
__global__ void kv_append_kernel(const bf16_t* qkv, bf16_t* kc, bf16_t* vc, const int* t_dev, int B, int M, int d, int dh,
                                 const float* rrb, bf16_t* qr) {
    const int chunks = d >> 3;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * chunks) return;
    const int b = gid / chunks, c = gid % chunks;
    if (qr) {
        const bf16x8 qv = *reinterpret_cast<const bf16x8*>(qkv + (size_t)b * 3 * d + c * 8);
        u32x4 w;
        bf16_t* wp = reinterpret_cast<bf16_t*>(&w);
#pragma unroll
        for (int j = 0; j < 8; j++) wp[j] = f2bf(bf2f((bf16_t)qv[j]) + rrb[c * 8 + j]);
        *reinterpret_cast<u32x4*>(qr + (size_t)b * d + c * 8) = w;
    }
    const int slot = (*t_dev) % M;
    const int h = (c * 8) / dh, e = (c * 8) % dh, H = d / dh;
    const u32x4 k = *reinterpret_cast<const u32x4*>(qkv + (size_t)b * 3 * d + d + c * 8);
    const u32x4 v = *reinterpret_cast<const u32x4*>(qkv + (size_t)b * 3 * d + 2 * d + c * 8);
    const size_t o = (((size_t)b * H + h) * M + slot) * dh + e;
    *reinterpret_cast<u32x4*>(kc + o) = k;
    *reinterpret_cast<u32x4*>(vc + o) = v;
}

// bulk fill after the prompt forward: cache slots (p mod M) <- K/V rows of positions p in [max(0, T-M), T)
__global__ void kv_fill_kernel(const bf16_t* qkv, bf16_t* kc, bf16_t* vc, int B, int T, int M, int d, int dh) {
    const int chunks = d >> 3;
    const int keep = min(T, M);
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * keep * chunks) return;
    const int c = (int)(gid % chunks);
    const int r = (int)((gid / chunks) % keep);
    const int b = (int)(gid / ((long long)chunks * keep));
    const int pos = T - keep + r;
    const int slot = pos % M;
    const bf16_t* row = qkv + ((size_t)b * T + pos) * 3 * d;
    const int h = (c * 8) / dh, e = (c * 8) % dh, H = d / dh;
    const size_t o = (((size_t)b * H + h) * M + slot) * dh + e;
    *reinterpret_cast<u32x4*>(kc + o) = *reinterpret_cast<const u32x4*>(row + d + c * 8);
    *reinterpret_cast<u32x4*>(vc + o) = *reinterpret_cast<const u32x4*>(row + 2 * d + c * 8);
}

// one workgroup per (head, batch row, ring piece); dh = 8 * LPK, LPK lanes share one key row, 64/LPK keys per wave instruction.
// gridDim.z = NS pieces of the ring (round 5).  A CU streams HBM at ~24 GB/s however many workgroups it holds, so a launch of
// B * H workgroups that is not a multiple of the CU count is as slow as its fullest CU: 384 rings (a lane's 32 sequences x 12
// heads) on 256 CUs take the time of two rings (1 MB: 40.6 us) where 768 half-rings take that of 1.5 (profiles/r05_decode_notes.txt).
// With NS > 1 every piece runs the whole kernel on its slots with its own softmax reference and leaves (o, max, sum) in `ws`; the
// piece that arrives last (one counter per (sequence, head), reset by that piece for the next launch) merges them in piece
// order -- a fixed order, so the result does not depend on who is last.  NS = 1 is the single-workgroup form, bit for bit as before.
template <int DH>
__global__ __launch_bounds__(256) void decode_attn_kernel(const bf16_t* qkv, const bf16_t* kc, const bf16_t* vc,
                                                          const float* bd, const float* rwb,
                                                          bf16_t* out, const int* t_dev, int B, int H, int M, float scale,
                                                          float* ws, int* arrived) {
    constexpr int LPK = DH / 8;          // lanes per key row
    constexpr int KPW = 64 / LPK;        // keys per wave per iteration
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);          // [M] scores -> probabilities
    float* red = sc + M;                                 // [4][KPW... ] scratch: 4 * 64 * 8 floats max
    __shared__ float wred[8];
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int c8 = lane % LPK, ksub = lane / LPK;
    const int d = H * DH;
    const int t = *t_dev;
    const int tm = t % M;
    // ring slots that have never been written (t < M - 1: the sequence is shorter than the memory) are HF's zero-initialised
    // mems: k = v = 0, so their score is the positional term alone and they add nothing to the output -- no K / V bytes are
    // read for them
    const int nvalid = t + 1 < M ? t + 1 : M;
    // this piece's written slots [lo, hi) and never-written slots [plo, phi)
    const int NS = gridDim.z, zi = blockIdx.z;
    const int piece = NS > 1 ? ((((nvalid + NS - 1) / NS) + 31) & ~31) : nvalid;
    const int lo = min(zi * piece, nvalid), hi = min(lo + piece, nvalid);
    const int ppiece = (M - nvalid + NS - 1) / NS;
    const int plo = min(nvalid + zi * ppiece, M), phi = min(plo + ppiece, M);

    float qw[8];
    {
        const bf16x8 qv = *reinterpret_cast<const bf16x8*>(qkv + (size_t)b * 3 * d + h * DH + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float q = bf2f((bf16_t)qv[j]);
            // mirror the training kernels: (q + bias) is rounded to bf16 before the contraction
            qw[j] = bf2f(f2bf(q + rwb[h * DH + c8 * 8 + j]));
        }
    }
    // positional term BD[b,h,dist] = (q + r_r_bias) . Rd[dist], precomputed for the whole batch by one batched GEMM per layer
    // (reads the Rd table once per head instead of once per (sequence, head))
    const float* bdrow = bd + ((size_t)b * H + h) * M;
    const bf16_t* kb = kc + ((size_t)b * H + h) * M * DH + c8 * 8;     // head-major ring: rows are DH apart
    const bf16_t* vb = vc + ((size_t)b * H + h) * M * DH + c8 * 8;

    // pass 1: scores.  Explicit batches of U key groups: all U loads of a batch are issued before any is consumed, and batch i + 1
    // is requested before batch i is consumed (two register sets), so each wave keeps 16-32 KiB in flight the whole pass (a plain
    // unroll pragma left one load per iteration on the critical path: ~HBM latency per 8 keys; single batches of 16 drained
    // between batches: 1.31 -> 1.24 ms per full-ring step with the second set and the V prefetch below).
    // The ring rows are read once per step and there are 4.8 GB of them: non-temporal loads, so that they do not push the
    // step's weights, bd and activations out of L2 / MALL (measured over the C5 generation: 51.6 -> 56.2 k tok/s; U 8 -> 16
    // another +1 %).  250 registers: two workgroups per CU; 12 / 16-deep batches held to 168 registers (three per CU) spill and run at
    // half the speed, 20-deep ones spill too (gpurun_out/r04_decode_db3.log).
    constexpr int U = 16;                   // double-buffered batches: batch i + 1 is requested before batch i is consumed
    float mx = -1e30f;
    {
        bf16x8 kA[U], kB[U];
        float bA[U], bB[U];
        auto load_k = [&](int s0, bf16x8 (&kv)[U], float (&bv)[U]) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int s = s0 + u * 4 * KPW + ksub;
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                kv[u] = (s < hi) ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(kb + (size_t)s * DH)) : z;
                int dist = tm - s;
                if (dist < 0) dist += M;
                bv[u] = (s < M && c8 == 0) ? bdrow[dist] : 0.f;
            }
        };
        auto use_k = [&](int s0, const bf16x8 (&kv)[U], const float (&bv)[U]) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int s = s0 + u * 4 * KPW + ksub;
                float acc = bv[u];
#pragma unroll
                for (int j = 0; j < 8; j++) acc += qw[j] * bf2f((bf16_t)kv[u][j]);
#pragma unroll
                for (int o = 1; o < LPK; o <<= 1) acc += __shfl_xor(acc, o, 64);
                acc *= scale;
                if (s < hi) {
                    if (c8 == 0) sc[s] = acc;
                    mx = fmaxf(mx, acc);
                }
            }
        };
        // Only the batches that hold a written slot: the ring slots [nvalid, M) are zero memories, whose score is the positional term
        // alone -- they take ONE multiply each (below) instead of a zero dot product, its lane reduction and the batch bookkeeping.
        // (Until round 5 both passes walked all M slots whatever the sequence length: at 1153 written slots of 2048 -- the mean over
        // the C5 generation -- the kernel took 35.7 us against 40.3 us with a full ring, i.e. it was bound by that loop, not by HBM.)
        constexpr int ST = 4 * KPW * U;
        int s0 = lo + wid * KPW;
        if (s0 < hi) load_k(s0, kA, bA);
        for (; s0 < hi; s0 += 2 * ST) {
            if (s0 + ST < hi) load_k(s0 + ST, kB, bB);
            use_k(s0, kA, bA);
            if (s0 + ST < hi) {
                if (s0 + 2 * ST < hi) load_k(s0 + 2 * ST, kA, bA);
                use_k(s0 + ST, kB, bB);
            }
        }
    }
    for (int sp = plo + tid; sp < phi; sp += 256) {      // the never-written slots: score = scale * BD[distance]
        int dist = tm - sp;
        if (dist < 0) dist += M;
        const float a = bdrow[dist] * scale;
        sc[sp] = a;
        mx = fmaxf(mx, a);
    }
    // the first V batch is requested before the softmax section: its latency runs under the two barriers and the exponentials
    bf16x8 vA[U], vB[U];
    auto load_v = [&](int s0, bf16x8 (&vv)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = s0 + u * 4 * KPW + ksub;
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            vv[u] = (s < hi) ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(vb + (size_t)s * DH)) : z;
        }
    };
    if (lo + wid * KPW < hi) load_v(lo + wid * KPW, vA);
    mx = wave_max(mx);
    if (lane == 0) wred[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    float sum = 0.f;
    for (int s = lo + tid; s < hi; s += 256) {
        const float pv = __expf(sc[s] - mx);
        sc[s] = pv;
        sum += pv;
    }
    for (int s = plo + tid; s < phi; s += 256) sum += __expf(sc[s] - mx);      // (v = 0 there: only the denominator)
    sum = wave_sum(sum);
    if (lane == 0) wred[4 + wid] = sum;
    __syncthreads();
    const float inv = 1.f / (wred[4] + wred[5] + wred[6] + wred[7]);

    // pass 2: o = sum_s p_s v_s   (P rounded to bf16 like the training kernel's MFMA operand); same batched loads
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = 0.f;
    {
        auto use_v = [&](int s0, const bf16x8 (&vv)[U]) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int s = s0 + u * 4 * KPW + ksub;
                const float pv = (s < hi) ? bf2f(f2bf(sc[s])) : 0.f;
#pragma unroll
                for (int j = 0; j < 8; j++) o[j] += pv * bf2f((bf16_t)vv[u][j]);
            }
        };
        constexpr int ST = 4 * KPW * U;
        int s0 = lo + wid * KPW;
        for (; s0 < hi; s0 += 2 * ST) {          // (the never-written slots hold v = 0: nothing to add)
            if (s0 + ST < hi) load_v(s0 + ST, vB);
            use_v(s0, vA);
            if (s0 + ST < hi) {
                if (s0 + 2 * ST < hi) load_v(s0 + 2 * ST, vA);
                use_v(s0 + ST, vB);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int off = LPK; off < 64; off <<= 1) o[j] += __shfl_xor(o[j], off, 64);
    }
    if (ksub == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) red[wid * DH + c8 * 8 + j] = o[j];
    }
    __syncthreads();
    if (NS == 1) {
        if (tid < DH) {
            const float v = (red[tid] + red[DH + tid] + red[2 * DH + tid] + red[3 * DH + tid]) * inv;
            out[(size_t)b * d + h * DH + tid] = f2bf(v);
        }
        return;
    }
    // ---- ring pieces: leave (o unnormalised, max, sum), the last arrival merges.  The pieces of a ring may sit on different XCDs
    // (one L2 each): the partials travel as agent-scope (sc1) stores and loads and the counter as a relaxed agent-scope atomic, the
    // stores waited for (vmcnt) before the workgroup's arrival is counted -- the pattern scripts/ubench/grid_barrier.hip measured;
    // the compiler's release / acquire fences (buffer_wbl2 / buffer_inv of the whole L2) made a 30 us launch take 90.
    float* wsp = ws + ((size_t)b * H + h) * NS * (DH + 2);
    if (tid < DH)
        __hip_atomic_store(wsp + zi * (DH + 2) + tid, red[tid] + red[DH + tid] + red[2 * DH + tid] + red[3 * DH + tid], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
        __hip_atomic_store(wsp + zi * (DH + 2) + DH, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(wsp + zi * (DH + 2) + DH + 1, wred[4] + wred[5] + wred[6] + wred[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int s_last;
    if (tid == 0) s_last = (__hip_atomic_fetch_add(arrived + (size_t)b * H + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NS - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    if (tid < DH) {
        float m = -1e30f;
        for (int k = 0; k < NS; k++) m = fmaxf(m, __hip_atomic_load(wsp + k * (DH + 2) + DH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        float num = 0.f, den = 0.f;
        for (int k = 0; k < NS; k++) {
            const float a = __expf(__hip_atomic_load(wsp + k * (DH + 2) + DH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - m);
            num += a * __hip_atomic_load(wsp + k * (DH + 2) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            den += a * __hip_atomic_load(wsp + k * (DH + 2) + DH + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        out[(size_t)b * d + h * DH + tid] = f2bf(num / den);
    }
    if (tid == 0) __hip_atomic_store(arrived + (size_t)b * H + h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// BD of one decode step (mxl_decode_bd): workgroup = (256 distances, head), wave = 64 distances x all (<= 64) batch rows x dh = 64:
// 16 fragment loads straight from global memory (qr and the head's Rd columns are L2-resident), 32 MFMAs, 16-byte fp32 stores
// along the distance axis.  The batched 128 x 128-tile GEMM this replaces took 9.7 us per layer for 0.2 GFLOP.
typedef __attribute__((ext_vector_type(8))) __bf16 dec_mfma_bf16x8;
__global__ __launch_bounds__(256) void decode_bd_kernel(const bf16_t* qr, const bf16_t* rd, float* bd, int B, int H, int M,
                                                        int ld_qr, int ld_rd) {
    const int h = blockIdx.y, wid = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int r0 = blockIdx.x * 256 + wid * 64;
    if (r0 >= M) return;
    const int li = l & 15, kq = 8 * (l >> 4);
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 fr[4][2], fq[4][2];
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int r = r0 + f * 16 + li, b = f * 16 + li;
            fr[f][ks] = r < M ? *reinterpret_cast<const bf16x8*>(rd + (size_t)r * ld_rd + h * 64 + ks * 32 + kq) : z;
            fq[f][ks] = b < B ? *reinterpret_cast<const bf16x8*>(qr + (size_t)b * ld_qr + h * 64 + ks * 32 + kq) : z;
        }
    // acc[mf][nf][j]: b = mf*16 + (l & 15), r = r0 + nf*16 + 4*(l >> 4) + j
#pragma unroll
    for (int mf = 0; mf < 4; mf++) {
        const int b = mf * 16 + li;
#pragma unroll
        for (int nf = 0; nf < 4; nf++) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dec_mfma_bf16x8, fr[nf][ks]),
                                                              __builtin_bit_cast(dec_mfma_bf16x8, fq[mf][ks]), acc, 0, 0, 0);
            const int r = r0 + nf * 16 + 4 * (l >> 4);
            if (b < B && r < M) {
                float* o = bd + ((size_t)b * H + h) * M + r;
                if (r + 3 < M && (M & 3) == 0) *reinterpret_cast<f32x4*>(o) = acc;
                else
#pragma unroll
                    for (int j = 0; j < 4; j++) if (r + j < M) o[j] = acc[j];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// sampler: HF GenerationMixin.sample / greedy_search on log-probs (B, V): repetition penalty (logits processor, over every
// token already in the row) -> temperature -> top-k -> top-p -> typical-p -> renormalise -> multinomial
// (musicnlp/trainer/eval.py:277-333 builds these arguments).  One workgroup per row; bitonic sort of (value, index) in LDS
// (V <= 2048).  greedy = argmax (ties -> lowest index, like torch.argmax).
// ---------------------------------------------------------------------------------------------------------------
constexpr int SORT_N = 2048;

// the row's next token (returned to every thread of the workgroup)
__device__ __forceinline__ int sample_row(const float* logp, int ldl, int V, const long long* ids, int ld_ids,
                                          const int* t_dev, const unsigned long long* rng_ctr, unsigned long long seed,
                                          int do_sample, int top_k, float top_p, float temperature,
                                          float repetition_penalty, float typical_p, float* out_probs) {
    __shared__ float key[SORT_N];
    __shared__ int idx[SORT_N];
    __shared__ int sh_pick;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logp + (size_t)b * ldl;
    const float invt = 1.f / temperature;
    for (int i = tid; i < SORT_N; i += 256) {
        key[i] = i < V ? row[i] * invt : -INFINITY;
        idx[i] = i;
    }
    __syncthreads();
    if (repetition_penalty != 1.f) {
        // HF RepetitionPenaltyLogitsProcessor: every id present in the row so far (prompt + generated, positions 0..t) has its
        // raw score multiplied (score < 0) or divided (score >= 0) by the penalty, once however often it occurs -- the writes
        // below all store the same value, so duplicates are harmless.
        const int tcur = *t_dev;
        const long long* hist = ids + (size_t)b * ld_ids;
        for (int j = tid; j <= tcur; j += 256) {
            const long long tok = hist[j];
            if (tok >= 0 && tok < V) {
                const float v = row[tok];
                key[tok] = (v < 0.f ? v * repetition_penalty : v / repetition_penalty) * invt;
            }
        }
        __syncthreads();
    }
    // Small supports (greedy, or top-k <= 64) need only the first few entries of the sorted order: take them by repeated
    // workgroup arg-max (same order as the sort: value descending, ties by ascending index) -- 2 barriers per entry instead of
    // the 66 barrier stages of the full sort.
    const int nsel = !do_sample ? 1 : ((top_k > 0 && top_k <= 64 && top_k < V) ? top_k : 0);
    if (nsel > 0) {
        __shared__ float wbest[4];
        __shared__ int warg[4];
        float vals[SORT_N / 256];
#pragma unroll
        for (int e = 0; e < SORT_N / 256; e++) vals[e] = key[tid + 256 * e];
        __syncthreads();                                  // everyone has its copy before key[] is overwritten with the selection
        for (int rnd = 0; rnd < nsel; rnd++) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int e = 0; e < SORT_N / 256; e++) {
                const int gi = tid + 256 * e;
                if (gi < V && (vals[e] > bv || (vals[e] == bv && gi < bi))) { bv = vals[e]; bi = gi; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if ((tid & 63) == 0) { wbest[tid >> 6] = bv; warg[tid >> 6] = bi; }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const float ov = wbest[w];
                const int oi = warg[w];
                if (w == 0 || ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (bi == 0x7fffffff) bi = 0;                 // every remaining value is -inf
            if ((bi & 255) == tid) {
#pragma unroll
                for (int e = 0; e < SORT_N / 256; e++) if (e == (bi >> 8)) vals[e] = -INFINITY;
            }
            if (tid == 0) { key[rnd] = bv; idx[rnd] = bi; }
            __syncthreads();
        }
    } else
    // bitonic sort, descending by key, ties by ascending index
    for (int k = 2; k <= SORT_N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < SORT_N; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = ((i & k) == 0);
                    const float a = key[i], c = key[ixj];
                    const int ia = idx[i], ic = idx[ixj];
                    const bool a_first = (a > c) || (a == c && ia < ic);   // a should come before c in descending order
                    if (up ? !a_first : a_first) {
                        key[i] = c; key[ixj] = a; idx[i] = ic; idx[ixj] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (!do_sample) {
        if (tid == 0) sh_pick = idx[0];
        __syncthreads();
        return sh_pick;
    }
    int keep = (top_k > 0 && top_k < V) ? top_k : V;
    __shared__ int sh_keep;
    __shared__ float sh_sum, sh_max;       // the head of the order can itself be dropped as atypical: keep its value
    // softmax over the kept prefix (max is key[0]); serial prefix over <= V terms by one thread is negligible here
    if (tid == 0) {
        const float m = key[0];
        float s = 0.f;
        for (int i = 0; i < keep; i++) s += __expf(key[i] - m);
        if (top_p > 0.f && top_p < 1.f) {
            // HF TopPLogitsWarper: keep the smallest prefix whose cumulative probability exceeds top_p (>= 1 token)
            float c = 0.f;
            int kk = 0;
            for (int i = 0; i < keep; i++) {
                c += __expf(key[i] - m) / s;
                kk = i + 1;
                if (c >= top_p) break;
            }
            keep = kk;
            s = 0.f;
            for (int i = 0; i < keep; i++) s += __expf(key[i] - m);
        }
        sh_keep = keep;
        sh_sum = s;
        sh_max = m;
    }
    if (typical_p > 0.f && typical_p < 1.f) {
        // HF TypicalLogitsWarper on the surviving support: order the tokens by |-log p - H| ascending and keep the shortest
        // prefix of that order whose mass reaches typical_p, i.e. token i stays iff the mass of the tokens ordered before it
        // is below typical_p.  No second sort: each thread accumulates that "mass before" for its own entries.
        __shared__ float pp[SORT_N], dev[SORT_N];
        __shared__ float red[4];
        __syncthreads();
        keep = sh_keep;
        const float m = sh_max, logs = __logf(sh_sum);
        float part = 0.f;
        for (int i = tid; i < keep; i += 256) {
            const float nl = key[i] - m - logs;
            const float p = __expf(nl);
            pp[i] = p;
            if (p > 0.f) part -= p * nl;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        const float ent = red[0] + red[1] + red[2] + red[3];
        for (int i = tid; i < keep; i += 256) dev[i] = fabsf(-(key[i] - m - logs) - ent);
        __syncthreads();
        bool drop[SORT_N / 256];
#pragma unroll
        for (int e = 0; e < SORT_N / 256; e++) {
            const int i = tid + 256 * e;
            drop[e] = false;
            if (i < keep) {
                const float di = dev[i];
                float before = 0.f;
                for (int j = 0; j < keep; j++) {
                    const float dj = dev[j];
                    before += (dj < di || (dj == di && j < i)) ? pp[j] : 0.f;
                }
                drop[e] = before >= typical_p;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < SORT_N / 256; e++)
            if (drop[e]) key[tid + 256 * e] = -INFINITY;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int i = 0; i < keep; i++) s += __expf(key[i] - m);
            sh_sum = s;
        }
    }
    if (tid == 0) {
        keep = sh_keep;
        const float m = sh_max, s = sh_sum;
        // uniform in [0,1) from a counter-based hash (counter advanced by the advance kernel)
        const unsigned long long ctr = *rng_ctr;
        const uint32_t h1 = mxl_hash32((uint32_t)(ctr * 0x9E3779B97F4A7C15ULL >> 32) ^ mxl_hash32((uint32_t)b + 0x85ebca6bU * (uint32_t)seed));
        const uint32_t h2 = mxl_hash32(h1 + (uint32_t)ctr + (uint32_t)(seed >> 32));
        const float u = (float)(h2 >> 8) * (1.0f / 16777216.0f);
        float c = 0.f;
        int pick = -1, last = 0;
        for (int i = 0; i < keep; i++) {
            const float p = __expf(key[i] - m) / s;
            if (p > 0.f) last = i;
            c += p;
            if (u < c && p > 0.f) { pick = i; break; }
        }
        if (pick < 0) pick = last;
        sh_pick = idx[pick];
        if (out_probs) {   // diagnostic / test hook: renormalised probabilities of the kept support, in vocab order
            for (int i = 0; i < V; i++) out_probs[(size_t)b * V + i] = 0.f;
            for (int i = 0; i < keep; i++) out_probs[(size_t)b * V + idx[i]] = __expf(key[i] - m) / s;
        }
    }
    __syncthreads();
    return sh_pick;
}

__global__ __launch_bounds__(256) void sample_kernel(const float* logp, int ldl, int V, long long* ids, int ld_ids,
                                                     const int* t_dev, unsigned long long* rng_ctr, unsigned long long seed,
                                                     int do_sample, int top_k, float top_p, float temperature,
                                                     float repetition_penalty, float typical_p, float* out_probs) {
    const int tok = sample_row(logp, ldl, V, ids, ld_ids, t_dev, rng_ctr, seed, do_sample, top_k, top_p, temperature,
                               repetition_penalty, typical_p, out_probs);
    if (threadIdx.x == 0) ids[(size_t)blockIdx.x * ld_ids + *t_dev + 1] = tok;
}

// Round 6: the sampler with the two launches that always follow it.  The sampled token's embedding row E[tok] * scale -- the input
// of the NEXT decode step (mxl_decode_embed's arithmetic) -- is written by the row's workgroup, and the workgroup that finishes last
// advances the position and RNG counters (mxl_decode_advance): every other workgroup has read them by then.  `scores` may be the
// head's raw logits instead of log-probabilities when no repetition penalty is in force: every other warper, the argmax and the
// renormalised draw are invariant under the per-row shift log-softmax applies.
__global__ __launch_bounds__(256) void sample_step_kernel(const float* scores, int ldl, int V, long long* ids, int ld_ids,
                                                          int* t_dev, unsigned long long* rng_ctr, unsigned long long seed,
                                                          int do_sample, int top_k, float top_p, float temperature,
                                                          float repetition_penalty, float typical_p, const bf16_t* E, bf16_t* emb_out,
                                                          int d, float scale, int* counter) {
    const int tok = sample_row(scores, ldl, V, ids, ld_ids, t_dev, rng_ctr, seed, do_sample, top_k, top_p, temperature,
                               repetition_penalty, typical_p, nullptr);
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) ids[(size_t)b * ld_ids + *t_dev + 1] = tok;
    const int id = (tok < 0 || tok >= V) ? 0 : tok;
    for (int c = tid; c < (d >> 3); c += 256) {
        const bf16x8 e = *reinterpret_cast<const bf16x8*>(E + (size_t)id * d + c * 8);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = bf2f((bf16_t)e[j]) * scale;
        const u32x4 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
        *reinterpret_cast<u32x4*>(emb_out + (size_t)b * d + c * 8) = o;
    }
    __syncthreads();                            // every thread of the workgroup is past its reads of *t_dev / *rng_ctr
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (int)gridDim.x - 1) {
            __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *t_dev += 1;
            *rng_ctr += 1;
        }
    }
}

__global__ void advance_kernel(int* t_dev, unsigned long long* rng_ctr) {
    *t_dev += 1;
    *rng_ctr += 1;
}

}  // namespace

extern "C" int mxl_decode_embed(const void* ids, int ld_ids, const int* t_dev, const void* E, void* out, int B, int d,
                                int V, float scale, void* stream) {
    MXL_CHECK_ARG(ids && t_dev && E && out && B > 0 && (d % 8) == 0);
    const int n = B * (d / 8);
    hipLaunchKernelGGL(decode_embed_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const long long*)ids,
                       ld_ids, t_dev, (const bf16_t*)E, (bf16_t*)out, B, d, V, scale);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_kv_append(const void* qkv, void* kcache, void* vcache, const int* t_dev, int B, int M, int d, int dh,
                             const float* r_r_bias, void* qr_out, void* stream) {
    MXL_CHECK_ARG(qkv && kcache && vcache && t_dev && B > 0 && M > 0 && (d % 8) == 0 && dh > 0 && (dh % 8) == 0 && (d % dh) == 0);
    MXL_CHECK_ARG(!qr_out || r_r_bias);
    const int n = B * (d / 8);
    hipLaunchKernelGGL(kv_append_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                       (bf16_t*)kcache, (bf16_t*)vcache, t_dev, B, M, d, dh, r_r_bias, (bf16_t*)qr_out);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_kv_fill(const void* qkv, void* kcache, void* vcache, int B, int T, int M, int d, int dh, void* stream) {
    MXL_CHECK_ARG(qkv && kcache && vcache && B > 0 && T > 0 && M > 0 && (d % 8) == 0 && dh > 0 && (dh % 8) == 0 && (d % dh) == 0);
    const long long n = (long long)B * (T < M ? T : M) * (d / 8);
    hipLaunchKernelGGL(kv_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)qkv, (bf16_t*)kcache, (bf16_t*)vcache, B, T, M, d, dh);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_decode_bd(const void* qr, const void* rd, float* bd, int B, int H, int dh, int M, int ld_qr, int ld_rd,
                             void* stream) {
    MXL_CHECK_ARG(qr && rd && bd && B > 0 && B <= 64 && H > 0 && M > 0);
    if (dh != 64) return MXL_EUNSUPPORTED;
    MXL_CHECK_ARG(ld_qr >= H * 64 && ld_rd >= H * 64 && (ld_qr % 8) == 0 && (ld_rd % 8) == 0);
    MXL_CHECK_ARG(((uintptr_t)qr % 16) == 0 && ((uintptr_t)rd % 16) == 0 && ((uintptr_t)bd % 16) == 0);
    hipLaunchKernelGGL(decode_bd_kernel, dim3((M + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qr,
                       (const bf16_t*)rd, bd, B, H, M, ld_qr, ld_rd);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

static int relattn_decode_launch(const void* qkv, const void* kcache, const void* vcache, const float* bd, const float* r_w_bias,
                                 void* out, const int* t_dev, int B, int H, int dh, int M, float scale, int pieces, float* ws,
                                 int* arrived, void* stream);

extern "C" int mxl_relattn_decode(const void* qkv, const void* kcache, const void* vcache, const float* bd,
                                  const float* r_w_bias, void* out, const int* t_dev, int B, int H,
                                  int dh, int M, float scale, void* stream) {
    return relattn_decode_launch(qkv, kcache, vcache, bd, r_w_bias, out, t_dev, B, H, dh, M, scale, 1, nullptr, nullptr, stream);
}

extern "C" size_t mxl_relattn_decode_split_ws_bytes(int B, int H, int dh, int pieces) {
    if (B <= 0 || H <= 0 || dh <= 0 || pieces < 1) return 0;
    return (size_t)B * H * pieces * (dh + 2) * sizeof(float);
}

extern "C" int mxl_relattn_decode_split(const void* qkv, const void* kcache, const void* vcache, const float* bd,
                                        const float* r_w_bias, void* out, const int* t_dev, int B, int H, int dh, int M, float scale,
                                        int pieces, float* ws, int* arrived, void* stream) {
    MXL_CHECK_ARG(pieces >= 1 && pieces <= 8);
    if (pieces > 1) MXL_CHECK_ARG(ws && arrived);
    return relattn_decode_launch(qkv, kcache, vcache, bd, r_w_bias, out, t_dev, B, H, dh, M, scale, pieces, ws, arrived, stream);
}

static int relattn_decode_launch(const void* qkv, const void* kcache, const void* vcache, const float* bd, const float* r_w_bias,
                                 void* out, const int* t_dev, int B, int H, int dh, int M, float scale, int pieces, float* ws,
                                 int* arrived, void* stream) {
    MXL_CHECK_ARG(qkv && kcache && vcache && bd && r_w_bias && out && t_dev && B > 0 && H > 0 && M > 0);
    const size_t shm = (size_t)M * 4 + 4 * 64 * 4;
    MXL_CHECK_ARG(shm <= 64 * 1024);
    dim3 grid(H, B, pieces);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(DH) hipLaunchKernelGGL((decode_attn_kernel<DH>), grid, dim3(256), shm, s, (const bf16_t*)qkv, (const bf16_t*)kcache, \
                                      (const bf16_t*)vcache, bd, r_w_bias, (bf16_t*)out, t_dev, B, H, M, scale, ws, arrived)
    switch (dh) {
        case 16: LAUNCH(16); break;
        case 32: LAUNCH(32); break;
        case 64: LAUNCH(64); break;
        default: return MXL_EUNSUPPORTED;
    }
#undef LAUNCH
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_sample(const float* logprobs, int ldl, int V, void* ids, int ld_ids, const int* t_dev,
                          unsigned long long* rng_ctr, unsigned long long seed, int B, int do_sample, int top_k, float top_p,
                          float temperature, float repetition_penalty, float typical_p, float* out_probs, void* stream) {
    MXL_CHECK_ARG(logprobs && ids && t_dev && rng_ctr && B > 0 && V > 0 && V <= SORT_N && temperature > 0.f);
    MXL_CHECK_ARG(repetition_penalty > 0.f && typical_p > 0.f);
    hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logprobs, ldl, V, (long long*)ids, ld_ids,
                       t_dev, rng_ctr, seed, do_sample, top_k, top_p, temperature, repetition_penalty, typical_p, out_probs);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_sample_step(const float* scores, int ldl, int V, void* ids, int ld_ids, int* t_dev, unsigned long long* rng_ctr,
                               unsigned long long seed, int B, int do_sample, int top_k, float top_p, float temperature,
                               float repetition_penalty, float typical_p, const void* E, void* emb_out, int d, float scale,
                               int* counter, void* stream) {
    MXL_CHECK_ARG(scores && ids && t_dev && rng_ctr && E && emb_out && counter && B > 0 && V > 0 && V <= SORT_N && temperature > 0.f);
    MXL_CHECK_ARG(repetition_penalty > 0.f && typical_p > 0.f && d > 0 && (d % 8) == 0);
    MXL_CHECK_ARG(((uintptr_t)E % 16) == 0 && ((uintptr_t)emb_out % 16) == 0);
    hipLaunchKernelGGL(sample_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scores, ldl, V, (long long*)ids, ld_ids,
                       t_dev, rng_ctr, seed, do_sample, top_k, top_p, temperature, repetition_penalty, typical_p,
                       (const bf16_t*)E, (bf16_t*)emb_out, d, scale, counter);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_decode_advance(int* t_dev, unsigned long long* rng_ctr, void* stream) {
    MXL_CHECK_ARG(t_dev && rng_ctr);
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, t_dev, rng_ctr);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Contrastive search (HF 4.25.1 GenerationMixin.contrastive_search / _ranking_fast, reached from
// musicnlp/trainer/eval.py:296-302 with the mems patch of musicnlp/models/transformer_xl.py:229-234): candidate k of sequence b
// is scored  (1 - alpha) * p[b][k]  -  alpha * max_s cos(h_cand[b*K + k], h_ctx[b][s]),  s over the S context positions.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void row_inv_norm_kernel(const bf16_t* x, long long ld, int d, float* out, int n) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n) return;
    const bf16_t* r = x + (size_t)j * ld;
    float s = 0.f;
    for (int c = lane * 8; c < d; c += 512) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(r + c);
#pragma unroll
        for (int k = 0; k < 8; k++) { const float f = bf2f((bf16_t)v[k]); s += f * f; }
    }
    s = wave_sum(s);
    if (lane == 0) out[j] = rsqrtf(s);
}

// one workgroup per candidate row: four waves stride over the context positions, 8 bf16 per lane per pass over d
__global__ __launch_bounds__(256) void contrastive_score_kernel(const bf16_t* ctx, long long ctx_bs, const float* ctx_inv, int inv_bs,
                                                                int S, const bf16_t* hid, int d, const float* probs, float alpha,
                                                                int K, float* score) {
    __shared__ float wmax[4];
    const int row = blockIdx.x, b = row / K, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bf16_t* h = hid + (size_t)row * d;
    float hn = 0.f;
    for (int c = lane * 8; c < d; c += 512) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(h + c);
#pragma unroll
        for (int k = 0; k < 8; k++) { const float f = bf2f((bf16_t)v[k]); hn += f * f; }
    }
    hn = rsqrtf(wave_sum(hn));
    float best = -INFINITY;
    for (int s = wid; s < S; s += 4) {
        const bf16_t* c_ = ctx + (size_t)b * ctx_bs + (size_t)s * d;
        float dot = 0.f;
        for (int c = lane * 8; c < d; c += 512) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(h + c);
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(c_ + c);
#pragma unroll
            for (int k = 0; k < 8; k++) dot += bf2f((bf16_t)a[k]) * bf2f((bf16_t)v[k]);
        }
        dot = wave_sum(dot) * hn * ctx_inv[(size_t)b * inv_bs + s];
        best = fmaxf(best, dot);
    }
    if (lane == 0) wmax[wid] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float pen = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        score[row] = (1.f - alpha) * probs[row] - alpha * pen;
    }
}

__global__ void contrastive_pick_kernel(const float* score, int K, long long* sel, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int bi = 0;
    float bv = score[(size_t)b * K];
    for (int k = 1; k < K; k++) {
        const float v = score[(size_t)b * K + k];
        if (v > bv) { bv = v; bi = k; }             // first maximum, as torch.max
    }
    sel[b] = bi;
}
}  // namespace

extern "C" int mxl_row_inv_norm_bf16(const void* x, long long ld, int n, int d, float* out, void* stream) {
    MXL_CHECK_ARG(x && out && n > 0 && d > 0 && (d % 8) == 0 && (ld % 8) == 0);
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld, d, out, n);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_contrastive_select(const void* ctx, long long ctx_bs, const float* ctx_inv_norm, int inv_bs, int S, const void* hid,
                                      const float* probs, float alpha, int B, int K, int d, float* score, void* sel, void* stream) {
    MXL_CHECK_ARG(ctx && ctx_inv_norm && hid && probs && score && sel && B > 0 && K > 0 && S > 0 && d > 0 && (d % 8) == 0 &&
                  (ctx_bs % 8) == 0);
    hipLaunchKernelGGL(contrastive_score_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)ctx, ctx_bs,
                       ctx_inv_norm, inv_bs, S, (const bf16_t*)hid, d, probs, alpha, K, score);
    hipLaunchKernelGGL(contrastive_pick_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, score, K, (long long*)sel, B);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}
