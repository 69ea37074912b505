// K4 forward: Transformer-XL relative-position attention, band-only, flash-style (never materialises (T, K)).
// Replaces upstream RelPartialLearnableMultiHeadAttn.forward between qkv_net and o_net:
//
//   score[i,p] = ((q_i + r_w_bias) . k_p  +  (q_i + r_r_bias) . Rd[i - p]) / sqrt(dh)      p = key position
//   visible    : 0 <= i - p <= M-1           (same_length=True with mlen == mem_len: exactly M keys per query)
//   out_i      = softmax_p(score[i,:]) . v
//
// Key positions: current tokens p = 0..T-1, memory p = -M..-1.  The caller supplies Kc >= T rows of K/V covering
// p in [T-Kc, T); positions below that are the zero mems upstream `init_mems` fabricates (k = v = 0 exactly, qkv_net
// has no bias): they add exp(BD) to the softmax denominator and nothing else -- reproduced here by feeding zeros.
// Rd[d] = r_net(pos_emb(min(d, clamp_len))) for d = 0..M-1 (the rel-shift folded into the index).
//
// Layout per workgroup (256 threads = 4 waves): 128 queries, wave w owns 32.  "Swapped" products: S^T = K.Qw^T and
// G^T = Rd.Qr^T on v_mfma_f32_32x32x16_bf16, so a lane owns ONE query (column) and the softmax row reductions are
// lane-local.  The rel-shift (BD[i,p] = G[i, i-p]) is a lane-private LDS round trip: lane writes its column of G^T as a
// row [query][distance & 127] and reads it back at distance i - p.  P stays in registers and feeds O^T += V^T.P^T directly
// (accumulator-as-operand, V^T fragments via ds_read_b64_tr_b16).
#include <type_traits>
#include "common.h"
#include "musicxl_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct RelAttnP {
    const bf16_t *q, *k, *v, *rd;
    const float *rwb, *rrb;
    bf16_t* out;
    float* lse;
    int B, T, H, M, Kc;
    long long q_bs, kv_bs, o_bs;
    int q_rs, kv_rs, rd_rs, o_rs;
    float scale_log2e;
    // training with zero memories (mxl_relattn_fwd_phantom): for the distance blocks whose score gradient the backward does not
    // store (all-phantom 256-distance blocks, see mxl_relattn_bwd_sparse_dg) the forward also accumulates
    //   oph[b,i,h,:] = sum_d 2^(G'[i,d] - mph[b,h,i]) * Rd[d,h,:]      (G' = the positional score in log2 units)
    // -- the "output" of those keys if their values were the Rd rows.  Their dQr is then  -scale * delta_i * 2^(mph - lse2) * oph_i,
    // an elementwise product in the query-owner backward, instead of a second pass over G, exp and an MFMA product there.
    bf16_t* oph;           // (B, T, H*dh) like out, or null
    float* mph;            // (B, H, T) like lse
    // 1: oph sums over EVERY phantom cell (key position below the first stored key tile) -- what mxl_relattn_bwd_fused consumes;
    // 0: over the all-phantom 256-distance blocks only (mxl_relattn_bwd_sparse_dg_oph walks the others itself)
    int oph_all;
    // with oph_all: also the per-tile records of mxl_relattn_drd_phantom (relattn_drd_phantom.hip) -- this wave's 32 scaled
    // (q + r_r_bias) rows in that kernel's LDS image order and -lse2 of its queries -- so that the backward needs no pass over q
    char* ph_rec;          // or null
};

constexpr int QB = 128;      // queries per workgroup
constexpr int KT = 64;       // keys per tile
constexpr int GS = 100;      // skew buffer: 96 distance columns (+4 pad) per query row, fp16 -> 200-byte rows
constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

template <int DH> struct Geo {
    static constexpr int KS = DH / 16;
    static constexpr int EB = (DH + 31) / 32;
    static constexpr int VW = EB * 32;
    static constexpr int ROWB = DH * 2;
    static constexpr int VROWB = VW * 2;
    static constexpr int CH = DH / 8;                   // 16-byte chunks per K / Rd row
    static constexpr int VCH = VW / 8;
    static constexpr int K_BYTES = KT * ROWB;
    static constexpr int V_BYTES = KT * VROWB;
    static constexpr int R_BYTES = 256 * ROWB;
    static constexpr int G_BYTES = 4 * 32 * GS * 2;
    static constexpr int SMEM = K_BYTES + V_BYTES + R_BYTES + G_BYTES;   // 73 KiB at DH = 64: two workgroups per CU
    static constexpr int NLD_K = (KT * CH + 255) / 256;  // 16-byte chunks per thread per tile
    static constexpr int NLD_V = (KT * CH + 255) / 256;
    __device__ static __forceinline__ int koff(int row, int ch) {  // K / Rd image: [row][DH], XOR swizzle for 128-B rows
        if (DH == 64) return row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4);
        return row * ROWB + (ch << 4);
    }
    __device__ static __forceinline__ int eoff(int row, int e) { return koff(row, e >> 3) + ((e & 7) << 1); }
    __device__ static __forceinline__ int voff(int row, int byte) {  // V image: [key][VW]
        if (DH == 64) return row * VROWB + (byte ^ (((row >> 1) & 1) << 6));
        return row * VROWB + byte;
    }
};

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));


// 16 skewed fp16 reads of one 32-key block as genuine ds_read_u16 (the compiler would fuse the constant-offset reads into
// ds_read_b64 at lane-dependent, mis-aligned addresses: 64-cycle replays each, SQ_LDS_UNALIGNED_STALL).  `base` is the LDS
// byte address of column (r - 4*hh + 64 - 32*kb - 27) of this lane's row; register j reads column offset 27 - pat(j),
// pat(j) = (j & 3) + 8 * (j >> 2).  hipcc does not count asm loads, so the waits are written here as well: both key blocks of a
// tile are issued back to back (skew_issue16 for block 0, skew_issue16_w15 for block 1, which then waits for block 0's sixteen --
// LDS returns in order), the second wait is lgkm_wait0().  Everything that consumes the registers is `asm volatile` (add_f16v), so
// it stays behind the waits; compiler-issued LDS or scalar loads in between only make the counts more conservative.
#define MXL_SKEW_READS                                                                                                        \
        "ds_read_u16 %0, %16 offset:54\n\t"  "ds_read_u16 %1, %16 offset:52\n\t"  "ds_read_u16 %2, %16 offset:50\n\t"           \
        "ds_read_u16 %3, %16 offset:48\n\t"  "ds_read_u16 %4, %16 offset:38\n\t"  "ds_read_u16 %5, %16 offset:36\n\t"           \
        "ds_read_u16 %6, %16 offset:34\n\t"  "ds_read_u16 %7, %16 offset:32\n\t"  "ds_read_u16 %8, %16 offset:22\n\t"           \
        "ds_read_u16 %9, %16 offset:20\n\t"  "ds_read_u16 %10, %16 offset:18\n\t" "ds_read_u16 %11, %16 offset:16\n\t"          \
        "ds_read_u16 %12, %16 offset:6\n\t"  "ds_read_u16 %13, %16 offset:4\n\t"  "ds_read_u16 %14, %16 offset:2\n\t"           \
        "ds_read_u16 %15, %16\n\t"
#define MXL_SKEW_OPERANDS                                                                                                     \
        : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7]),               \
          "=&v"(u[8]), "=&v"(u[9]), "=&v"(u[10]), "=&v"(u[11]), "=&v"(u[12]), "=&v"(u[13]), "=&v"(u[14]), "=&v"(u[15])          \
        : "v"(base)                                                                                                           \
        : "memory"
__device__ __forceinline__ void skew_issue16(uint32_t base, uint32_t (&u)[16]) {
    asm volatile(MXL_SKEW_READS "s_nop 0" MXL_SKEW_OPERANDS);
}
__device__ __forceinline__ void skew_issue16_w15(uint32_t base, uint32_t (&u)[16]) {
    asm volatile(MXL_SKEW_READS "s_waitcnt lgkmcnt(15)" MXL_SKEW_OPERANDS);
}
__device__ __forceinline__ void lgkm_wait0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// s + (float)h in ONE VALU issue (v_fma_mix_f32: f32 * 1.0 + f16 taken from the low half of `h16`)
__device__ __forceinline__ float add_f16v(float s, uint32_t h16) {
    float r;
    asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(s), "v"(h16));
    return r;
}

// max(a, b, c) in one issue, without the canonicalising v_max_f32 x, x that fmaxf() gets in front of values produced by inline
// asm (the scores come out of v_fma_mix): 16 instead of 56 VALU issues for the 32 scores of a tile
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// In-kernel stamps (diagnostic builds only: scripts/ab_build.sh relattn_fwd stamp -DMXL_STAMP; the shipped library has none).
// Per wave, shader cycles between consecutive stamps are summed per segment and added to g_fwd_stamps at the end (scripts/stamp_fwd.py).
#ifdef MXL_STAMP
__device__ unsigned long long g_fwd_stamps[16];
#define STAMP_DECL unsigned long long st_last, st_acc[16] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}; \
    { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define STAMP(i) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); st_acc[i] += t_ - st_last; st_last = t_; }
#define STAMP_FLUSH if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 16; i_++) atomicAdd(&g_fwd_stamps[i_], st_acc[i_]); }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

#ifndef FWD_PRIO
#define FWD_PRIO 6                              // s_setprio(1) around MFMA clusters: bit 0 G / S chains of the tile loop, bit 1 P V, bit 2 the phantom loop's chains and
                                                // value-sum products.  Bit 2 pays by itself: 1.630 -> 1.600 ms per layer (-1.8 %, four alternating same-box rounds,
                                                // profiles/r06_fwd_setprio_ab.log); bits 0 and 1 alone move nothing (1.636 / 1.631).  Under the max-ilp scheduling
                                                // strategy this unit is built with (build.py) bits 1 + 2 read another -1.0 % (1.625 -> 1.609 over ten alternating runs),
                                                // bit 0 costs 2 % there
#endif
#define FPRIO_UP(bit_) do { if (FWD_PRIO & (bit_)) __builtin_amdgcn_s_setprio(1); } while (0)
#define FPRIO_DOWN(bit_) do { if (FWD_PRIO & (bit_)) __builtin_amdgcn_s_setprio(0); } while (0)
constexpr float RESCALE_THRESH = 8.0f;   // log2 units: accumulators are re-based when a score exceeds the reference by 2^8
// The common path does not look for the maximum at all.  It exponentiates against the reference as it stands and keeps the block
// when every row's sum of the new terms is at most 2^13 (no term is then more than 2^13 above the reference; fp32 sums and bf16
// probabilities are scale-free, the range is what matters).  Otherwise -- an overflow to inf included -- and for as long as some
// lane of the wave has no reference yet, the scores are computed again and take the path with the maximum, which moves the
// reference exactly as before.  Per 64-key tile that removes 16 v_max3, the cross-half exchange (an LDS round trip) and the
// per-lane decision: ~10 % of the tile's vector issues (round 5).
constexpr float FAST_SUM_MAX = 8192.0f;

template <int DH>
__global__ __launch_bounds__(256, 2) void relattn_fwd_kernel(RelAttnP p) {
    using G = Geo<DH>;
    constexpr int KS = G::KS, EB = G::EB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                          // [64][DH]
    char* sV = sK + G::K_BYTES;               // [64][VW]
    char* sR = sV + G::V_BYTES;               // ring [256][DH]
    _Float16* sG = reinterpret_cast<_Float16*>(sR + G::R_BYTES);  // [4][32][GS] fp16

    STAMP_DECL
    const int tid = threadIdx.x;
    // the wave index in a scalar register: everything derived from it (the wave's query range, which tiles it takes part in, which
    // of its blocks need the mask) is then scalar arithmetic and scalar branches, not per-lane compares and exec juggling
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, r = l & 31, hh = l >> 5;
    int bx_, h, b;
    xcd_block(bx_, h, b);
    // longest-first: late query blocks see the most real keys (early ones mostly phantom distances), and they are dispatched
    // first so the tail of the launch is made of short workgroups
    const int i0 = (gridDim.x - 1 - bx_) * QB;
    const int iw0 = i0 + 32 * wid;
    const int T = p.T, M = p.M;
    const int p0 = T - p.Kc;  // lowest stored key position
    // skew buffer of this wave: row = query, column c = distance - dlo in [0, 96).  All addresses are per-lane constants
    // plus immediates: writes land at column 32*gb + 8*grp + 4*hh, the read of score (kb, j) at column r - jj + 64.
    _Float16* gW = sG + wid * 32 * GS + r * GS + 4 * hh;
    const _Float16* gR = sG + wid * 32 * GS + r * GS + r + 64 - 4 * hh;
    // LDS byte address of column (r - 4hh + 64 - 27) for the skew reads (key block kb subtracts 32 columns = 64 bytes)
    const uint32_t gRb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(const char*)(gR - 27);
    f16x4 carry[4];   // block 0 of the previous tile (= block 2 of this one), kept in registers
    // Rd ring fragments: a 32-distance block starts at a multiple of 32, so its rows are slots (block & 255) + r without a wrap:
    // lane constant (row r, swizzled chunk of k-step ks) + a scalar block offset
    int rfr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
        rfr[ks] = r * G::ROWB + ((DH == 64) ? (((2 * ks + hh) ^ ((r >> 1) & 7)) << 4) : ((2 * ks + hh) << 4));

    const bf16_t* kbase = p.k + (size_t)b * p.kv_bs + (size_t)h * DH;
    const bf16_t* vbase = p.v + (size_t)b * p.kv_bs + (size_t)h * DH;
    const bf16_t* rbase = p.rd + (size_t)h * DH;

    // ---- Q fragments (B operand: lane = query r, k = 16ks + 8hh + j), biases added in fp32
    bf16x8 qw[KS], qr[KS];
    {
        const int i = iw0 + r;
        const bool ok = i < T;
        const bf16_t* qp = p.q + (size_t)b * p.q_bs + (size_t)(ok ? i : 0) * p.q_rs + (size_t)h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int e0 = 16 * ks + 8 * hh;
            bf16x8 qv = *reinterpret_cast<const bf16x8*>(qp + e0);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float qf = ok ? bf2f((bf16_t)qv[j]) : 0.f;
                // scale * log2(e) folded into the operands: the MFMA results are already in exp2 units
                qw[ks][j] = (short)f2bf((qf + p.rwb[h * DH + e0 + j]) * p.scale_log2e);
                qr[ks][j] = (short)f2bf((qf + p.rrb[h * DH + e0 + j]) * p.scale_log2e);
            }
        }
    }

    if (p.ph_rec && iw0 < T) {
        // record of tile iw0 / 32: row r at 128 r, 16-byte chunk c at (c ^ s(r)) << 4, s = (bit 1, bit 2, bit 3) of r -> chunk bits (2, 1, 0)
        char* rec = p.ph_rec + (((size_t)b * p.H + h) * (size_t)(T >> 5) + (iw0 >> 5)) * 4352;
        const int sw = (((r >> 1) & 1) << 2) | (((r >> 2) & 1) << 1) | ((r >> 3) & 1);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) *reinterpret_cast<bf16x8*>(rec + r * 128 + (((2 * ks + hh) ^ sw) << 4)) = qr[ks];
    }

    const int p_lo = i0 - M + 1;
    const int p_hi = min(i0 + QB - 1, T - 1);
    const int kt_lo = floordiv(p_lo, KT), kt_hi = floordiv(p_hi, KT);

    // ---- staging helpers -----------------------------------------------------------------------------------
    // The tile loop keeps one register set per image: the rows of tile t + 1 are requested at the top of tile t and stored behind
    // its first barrier.  (A second set -- requests two tiles ahead -- moved the wait from the stores to the requests and cost 2 %:
    // profiles/r05_fwd_notes.txt.)  The phantom loop, which needs Rd rows only, alternates rr[0] / rr[1].
    typedef u32x4 stage_t[G::NLD_K];
    stage_t rk, rv, rr[2];
    // K / V / Rd rows through buffer descriptors (wave-uniform base in scalar registers, 32-bit lane offset): rows below the first
    // stored key (negative offset = huge unsigned), past the last one, and Rd rows outside [0, M) fall outside num_records and read
    // as zero -- upstream's zero memories for K / V, distances the mask removes for Rd -- without a branch, a select or a clamp.
    // A thread's chunks never change: its global offsets are lane constants plus a scalar (tile start x row stride), its LDS
    // addresses lane constants (the ring adds a scalar 64-row offset: a 64-distance chunk starts at a multiple of 64 and does not
    // wrap, and the swizzle term depends on the row's low bits only) -- one vector add per load, none per store (round 5; the
    // first form rebuilt row, chunk, clamp and swizzle for each of the six loads and six stores of every tile)
    const unsigned kv_bytes = (unsigned)(((long long)(p.Kc - 1) * p.kv_rs + DH) * 2);
    const unsigned rd_bytes = (unsigned)(((long long)(M - 1) * p.rd_rs + DH) * 2);
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)kv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)kv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)rbase, 0, (int)rd_bytes, 0x00020000);
    constexpr bool STG_ALL = (KT * G::CH) % 256 == 0;     // every thread has a chunk in every round
    int kvo[G::NLD_K], rdo[G::NLD_K], kst[G::NLD_K], vst[G::NLD_K];
#pragma unroll
    for (int n = 0; n < G::NLD_K; n++) {
        const int c = tid + n * 256;
        const int row = c / G::CH, ch = c % G::CH;
        kvo[n] = ((row - p0) * p.kv_rs + ch * 8) * 2;
        rdo[n] = (row * p.rd_rs + ch * 8) * 2;
        kst[n] = G::koff(row & (KT - 1), ch);
        vst[n] = G::voff(row & (KT - 1), ch * 16);
    }
    auto load_kv = [&](stage_t& dk, stage_t& dv, int P) __attribute__((always_inline)) {      // the 64 key rows from position P
        const int so = P * p.kv_rs * 2;
#pragma unroll
        for (int n = 0; n < G::NLD_K; n++) {
            if (STG_ALL || tid + n * 256 < KT * G::CH) {
                dk[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_k, kvo[n] + so, 0, 0));
                dv[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, kvo[n] + so, 0, 0));
            } else {
                dk[n] = u32x4{0u, 0u, 0u, 0u}; dv[n] = u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto store_kv = [&](const stage_t& sk, const stage_t& sv) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < G::NLD_K; n++) {
            if (STG_ALL || tid + n * 256 < KT * G::CH) {
                *reinterpret_cast<u32x4*>(sK + kst[n]) = sk[n];
                *reinterpret_cast<u32x4*>(sV + vst[n]) = sv[n];
            }
        }
    };
    // 64 Rd rows d in [dbase, dbase+64), dbase a multiple of 64 -> registers / ring slots (d & 255)
    auto load_r = [&](stage_t& dst, int dbase) __attribute__((always_inline)) {
        const int so = dbase * p.rd_rs * 2;
#pragma unroll
        for (int n = 0; n < G::NLD_K; n++) {
            if (STG_ALL || tid + n * 256 < KT * G::CH)
                dst[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, rdo[n] + so, 0, 0));
            else dst[n] = u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto store_r = [&](const stage_t& src, int dbase) __attribute__((always_inline)) {
        char* ring = sR + (dbase & 255) * G::ROWB;
#pragma unroll
        for (int n = 0; n < G::NLD_K; n++)
            if (STG_ALL || tid + n * 256 < KT * G::CH) *reinterpret_cast<u32x4*>(ring + kst[n]) = src[n];
    };

    // zero the V pad columns once (DH < 32): tr-reads of O^T rows >= DH must see zeros
    if (DH < G::VW) {
        for (int i = tid; i < G::V_BYTES / 4; i += 256) reinterpret_cast<uint32_t*>(sV)[i] = 0u;
        __syncthreads();
    }

    // Online softmax with a LAZY reference: every accumulated quantity is relative to m_run (log2 units), which is only
    // moved when a new score exceeds it by RESCALE_THRESH (or when it is still unset = NEG_BIG).  -m_run is kept in 16
    // registers (`cinit`) and enters the score for free as the C operand of the first MFMA of each chain, so a score costs
    // one v_fma_mix (S + BD), one v_exp and one add (FAST_SUM_MAX above; half a v_max3 more on the path that moves the
    // reference).  Both lanes of a query (hh = 0/1) see the same merged maximum there and therefore take identical decisions.
    float m_run = NEG_BIG, l_run = 0.f;
    bool all_set = false;            // wave-uniform: every lane has a reference
    f32x16 cinit;
#pragma unroll
    for (int j = 0; j < 16; j++) cinit[j] = 0.f;
    f32x16 o[EB];          // O^T accumulators; during the phantom loop they hold oph (v = 0 there: O itself stays 0)
#pragma unroll
    for (int e = 0; e < EB; e++)
#pragma unroll
        for (int j = 0; j < 16; j++) o[e][j] = 0.f;

    // the path with the maximum, shared by both loops: x[0], x[1] hold the relative scores of two 32-cell blocks (masked cells
    // NEG_BIG); moves the reference where a lane needs it, leaves x re-based, and `rescale_acc(alpha)` applied to the accumulators
    // A lane's decision must depend on its own row only (changing a later token must leave earlier positions bit-identical, and
    // a query shares its wave with later ones): after a failed first attempt the lanes that caused it (`bad`) move, the others
    // pass through unchanged; without a first attempt (a lane of the wave still unset -- a matter of positions, not of data) the
    // RESCALE_THRESH rule applies.
    auto rebase = [&](f32x16 (&x)[2], auto&& rescale_acc, bool after_attempt, float rs_attempt) __attribute__((always_inline)) {
        float mx = NEG_BIG;
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int j = 0; j < 16; j += 2) mx = max3(mx, x[kb][j], x[kb][j + 1]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const bool unset = m_run == NEG_BIG;
        const float rs_other = __shfl_xor(rs_attempt, 32, 64);               // the two lanes of a query decide together
        const bool bad_q = !(rs_attempt <= FAST_SUM_MAX) || !(rs_other <= FAST_SUM_MAX);
        const bool need = unset ? (mx > 0.5f * NEG_BIG) : (after_attempt ? bad_q : (mx > RESCALE_THRESH));
        if (__any(need)) {               // rare after the first block: move the reference, re-base the accumulators, l and the scores
            const float delta = need ? mx : 0.f;
            const float alpha = (need && !unset) ? __builtin_amdgcn_exp2f(-delta) : 1.f;
            if (need) m_run = (unset ? 0.f : m_run) + delta;
            const float neg = (m_run == NEG_BIG) ? 0.f : -m_run;
            l_run *= alpha;
#pragma unroll
            for (int j = 0; j < 16; j++) { x[0][j] -= delta; x[1][j] -= delta; cinit[j] = neg; }
            rescale_acc(alpha);
        }
        all_set = !__any(m_run == NEG_BIG);
    };
    auto exp_sum = [&](f32x16 (&x)[2]) __attribute__((always_inline)) {
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int j = 0; j < 16; j++) { x[kb][j] = __builtin_amdgcn_exp2f(x[kb][j]); rs += x[kb][j]; }   // masked cells: exp2(NEG_BIG) = 0
        return rs;
    };

    // ---- phantom keys.  Key positions below the first stored tile (pz) are upstream's zero mems: k = v = 0, so such a key
    // contributes exp(BD) to the softmax denominator and nothing else, and BD depends only on the distance.  Instead of
    // walking those key tiles (S, skew, PV all wasted) walk the DISTANCES d in [i - pz + 1, M - 1] block-wise: G^T blocks
    // straight from the accumulators, no skew, no K/V traffic.  The tile loop then starts at tile pz.
    const int pz = floordiv(p0, KT) * KT;
    int kt_start = kt_lo;
    const int tk0 = G::eoff(4 * hh + ((l & 15) >> 2), 16 * ((l >> 4) & 1) + 4 * (l & 3));
    if (kt_lo * KT < pz) {
        kt_start = pz / KT;
        const int qi = iw0 + r;
        const int db0 = i0 - pz;
        // Rd rows in 64-distance chunks through the 256-row ring: chunk c + 1 is stored (from registers loaded an iteration earlier)
        // while chunk c is read -- different ring slots -- so ONE barrier per chunk covers both "c + 1 is complete" and "everybody is
        // done with c" (the first form stored and read the same chunk between two barriers; an iteration is only 16 MFMAs per wave)
        // Requests run three chunks ahead of the reads (two alternating register sets: chunk c + 3 is requested in iteration c into
        // the set chunk c + 1 was just stored from).  Every iteration requests and stores -- chunks past M - 1 read
        // as zeros through the descriptor and land in ring slots nobody reads any more -- so hipcc's counted waits stay exact.
        load_r(rk, db0);
        load_r(rr[1], db0 + 64);
        load_r(rr[0], db0 + 128);
        store_r(rk, db0);
        __syncthreads();
        auto chunk = [&](auto par_, int db) __attribute__((always_inline)) {
            constexpr int NXT = 1 - decltype(par_)::value;      // chunk c + 1's set when c has parity par
            STAMP(15)
            store_r(rr[NXT], db + 64);
            load_r(rr[NXT], db + 192);
            STAMP(10)
            // the chunk's two 32-distance blocks side by side: both G chains issued before either block's softmax work
            bool on[2], fl[2];
#pragma unroll
            for (int gb = 0; gb < 2; gb++) {
                const int dblk = db + 32 * gb;
                on[gb] = (iw0 < T) && !(dblk + 31 <= iw0 - pz || dblk > M - 1);        // a phantom cell in this block?
                fl[gb] = on[gb] && (dblk >= iw0 + 31 - pz + 1) && (dblk + 31 <= M - 1) && (iw0 + 31 < T);   // every cell phantom and in range
            }
            if (on[0] || on[1]) {
                f32x16 g[2];
                const char* rb = sR + (db & 255) * G::ROWB;      // a multiple of 64 rows: + 0..63 does not wrap
                auto ph_scores = [&]() __attribute__((always_inline)) {
                    bf16x8 ra[2][KS];
#pragma unroll
                    for (int gb = 0; gb < 2; gb++)
#pragma unroll
                        for (int ks = 0; ks < KS; ks++) ra[gb][ks] = *reinterpret_cast<const bf16x8*>(rb + gb * 32 * G::ROWB + rfr[ks]);
                    __builtin_amdgcn_sched_barrier(0);
                    g[0] = cinit; g[1] = cinit;
                    FPRIO_UP(4);
#pragma unroll
                    for (int ks = 0; ks < KS; ks++)
#pragma unroll
                        for (int gb = 0; gb < 2; gb++)
                            g[gb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, ra[gb][ks]),
                                                                            __builtin_bit_cast(mfma_bf16x8, qr[ks]), g[gb], 0, 0, 0);
                    FPRIO_DOWN(4);
#pragma unroll
                    for (int gb = 0; gb < 2; gb++) {
                        if (!fl[gb]) {          // (a block that is off altogether is masked out cell by cell: d <= qi - pz or d > M - 1)
                            const int dblk = db + 32 * gb;
#pragma unroll
                            for (int j = 0; j < 16; j++) {
                                const int d = dblk + (j & 3) + 8 * (j >> 2) + 4 * hh;
                                const bool valid = (d >= qi - pz + 1) && (d <= M - 1) && (qi < T);
                                g[gb][j] = valid ? g[gb][j] : NEG_BIG;
                            }
                        }
                    }
                };
                auto rescale_oph = [&](float alpha) __attribute__((always_inline)) {
                    if (p.oph) {
#pragma unroll
                        for (int e = 0; e < EB; e++)
#pragma unroll
                            for (int j = 0; j < 16; j++) o[e][j] *= alpha;
                    }
                };
                const bool attempt = all_set;
                bool precise = !attempt, bad = false;
                float rs = 0.f;
                if (attempt) {
                    ph_scores();
                    STAMP(11)
                    rs = exp_sum(g);
                    bad = !(rs <= FAST_SUM_MAX);
                    precise = __any(bad);
                }
                if (precise) {
                    ph_scores();
                    rebase(g, rescale_oph, attempt, rs);
                    rs = exp_sum(g);
                }
                l_run += rs;
                STAMP(12)
                // oph: only over the blocks the backward skips (every cell of them is phantom and in range)
                FPRIO_UP(4);
#pragma unroll
                for (int gb = 0; gb < 2; gb++) {
                    const int dblk = db + 32 * gb;
                    if (p.oph && on[gb] && (p.oph_all || (dblk & ~255) > iw0 + 31 - pz)) {
                        const int gq = l >> 4, li = l & 15, q4 = li >> 2, pp = li & 3;
                        const char* rbb = rb + gb * 32 * G::ROWB;
#pragma unroll
                        for (int st = 0; st < 2; st++) {
                            const u32x4 pw = {pack2bf(g[gb][8 * st], g[gb][8 * st + 1]), pack2bf(g[gb][8 * st + 2], g[gb][8 * st + 3]),
                                              pack2bf(g[gb][8 * st + 4], g[gb][8 * st + 5]), pack2bf(g[gb][8 * st + 6], g[gb][8 * st + 7])};
                            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
                            for (int e = 0; e < EB; e++) {
                                const int dist = dblk + 16 * st + 4 * hh + q4;      // accumulator-permuted k order
                                const int ecol = 32 * e + 16 * (gq & 1) + 4 * pp;
                                bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
                                if (DH == 64) {
                                    // one per-lane constant + immediates (relattn_bwd.hip, tk0): the +8 row and the second
                                    // 32-column half flip swizzle bits that do not depend on the lane for this pattern
                                    const char* a0 = rbb + tk0 + 16 * st * G::ROWB;
                                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(a0 + 64 * e));
                                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(a0 + 8 * G::ROWB + 64 * (1 - e)));
                                    a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                                } else if (ecol < DH) {
                                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sR + G::eoff(dist & 255, ecol)));
                                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sR + G::eoff((dist + 8) & 255, ecol)));
                                    a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                                }
                                o[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                               __builtin_bit_cast(mfma_bf16x8, pf), o[e], 0, 0, 0);
                            }
                        }
                    }
                }
                FPRIO_DOWN(4);
            }
            STAMP(13)
            __syncthreads();
            STAMP(14)
        };
#pragma unroll 1
        for (int db = db0; db <= M - 1; db += 128) {
            chunk(std::integral_constant<int, 0>{}, db);
            if (db + 64 > M - 1) break;
            chunk(std::integral_constant<int, 1>{}, db + 64);
        }
    }

    if (p.oph) {           // written for every query (zeros where no block qualified); relative to m_run as it stands now
        const int qi = iw0 + r;
        if (qi < T) {
            bf16_t* op = p.oph + (size_t)b * p.o_bs + (size_t)qi * p.o_rs + (size_t)h * DH;
#pragma unroll
            for (int e = 0; e < EB; e++) {
#pragma unroll
                for (int grp = 0; grp < 4; grp++) {
                    const int e0 = 32 * e + 8 * grp + 4 * hh;
                    if (e0 < DH) {
                        u32x2 w = {pack2bf(o[e][4 * grp], o[e][4 * grp + 1]), pack2bf(o[e][4 * grp + 2], o[e][4 * grp + 3])};
                        *reinterpret_cast<u32x2*>(op + e0) = w;
                    }
                }
            }
            if (hh == 0) p.mph[((size_t)b * p.H + h) * T + qi] = (m_run == NEG_BIG) ? 0.f : m_run;
        }
#pragma unroll
        for (int e = 0; e < EB; e++)
#pragma unroll
            for (int j = 0; j < 16; j++) o[e][j] = 0.f;
    }

    // ---- prologue: first tile + its distance window [i0-P0-64, i0-P0+127] (wave w: dlo_w = i0+32w-P0-64, 96 rows)
    // (every request issued before the first store: one round trip, not four)
    {
        const int P0 = kt_start * KT;
        const int dbase = i0 - P0 - 64;
        stage_t w2;
        load_kv(rk, rv, P0);
        load_r(rr[0], dbase);
        load_r(rr[1], dbase + 64);
        load_r(w2, dbase + 128);
        store_kv(rk, rv);
        store_r(rr[0], dbase);
        store_r(rr[1], dbase + 64);
        store_r(w2, dbase + 128);
    }
    __syncthreads();

    bool have_ring = false;
    auto rescale_o = [&](float alpha) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < EB; e++)
#pragma unroll
            for (int j = 0; j < 16; j++) o[e][j] *= alpha;
    };

    // Requests and stores are unconditional: past the last tile the descriptors return zeros and the images are not read again.
#pragma unroll 1
    for (int kt = kt_start; kt <= kt_hi; kt++) {
        const int P = kt * KT;
        STAMP(15)
        load_kv(rk, rv, P + KT);
        load_r(rr[0], i0 - (P + KT) - 64);   // the next tile's 64 new (lowest) distances
        STAMP(0)
        const int dmin_w = iw0 - P - (KT - 1), dmax_w = iw0 + 31 - P;
        const bool active = (dmax_w >= 0) && (dmin_w <= M - 1) && (iw0 < T);
        if (active) {
            const char* cK = sK;
            const char* cV = sV;
            const int dlo = iw0 - P - 64;       // a multiple of 32
            // ---- G^T = Rd . Qr^T for the new distance blocks -> lane-private skew buffer (fp16)
            auto gblock = [&](int gb, f16x4 (&dst)[4]) __attribute__((always_inline)) {
                f32x16 g;
#pragma unroll
                for (int j = 0; j < 16; j++) g[j] = 0.f;
                const char* rb = sR + ((dlo + 32 * gb) & 255) * G::ROWB;
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(rb + rfr[ks]);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                __builtin_bit_cast(mfma_bf16x8, qr[ks]), g, 0, 0, 0);
                }
#pragma unroll
                for (int grp = 0; grp < 4; grp++) {
                    const f32x4v v4 = {g[4 * grp], g[4 * grp + 1], g[4 * grp + 2], g[4 * grp + 3]};
                    dst[grp] = __builtin_convertvector(v4, f16x4);
                }
            };
            f16x4 b0[4], b1[4];
            if (!have_ring) gblock(2, carry);
            {   // both new distance blocks: eight Rd fragments, then two interleaved chains
                bf16x8 ra[2][KS];
#pragma unroll
                for (int gb = 0; gb < 2; gb++) {
                    const char* rb = sR + ((dlo + 32 * gb) & 255) * G::ROWB;
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) ra[gb][ks] = *reinterpret_cast<const bf16x8*>(rb + rfr[ks]);
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x16 g0, g1;
#pragma unroll
                for (int j = 0; j < 16; j++) { g0[j] = 0.f; g1[j] = 0.f; }
                FPRIO_UP(1);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, ra[0][ks]),
                                                                 __builtin_bit_cast(mfma_bf16x8, qr[ks]), g0, 0, 0, 0);
                    g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, ra[1][ks]),
                                                                 __builtin_bit_cast(mfma_bf16x8, qr[ks]), g1, 0, 0, 0);
                }
                FPRIO_DOWN(1);
                STAMP(1)
#pragma unroll
                for (int grp = 0; grp < 4; grp++) {
                    const f32x4v v0 = {g0[4 * grp], g0[4 * grp + 1], g0[4 * grp + 2], g0[4 * grp + 3]};
                    const f32x4v v1 = {g1[4 * grp], g1[4 * grp + 1], g1[4 * grp + 2], g1[4 * grp + 3]};
                    b0[grp] = __builtin_convertvector(v0, f16x4);
                    b1[grp] = __builtin_convertvector(v1, f16x4);
                }
            }
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                *reinterpret_cast<f16x4*>(gW + 8 * grp) = b0[grp];
                *reinterpret_cast<f16x4*>(gW + 32 + 8 * grp) = b1[grp];
                *reinterpret_cast<f16x4*>(gW + 64 + 8 * grp) = carry[grp];
                carry[grp] = b0[grp];
            }
            have_ring = true;
            STAMP(2)
            // ---- scores: S^T = K . Qw^T (two 32-key blocks, every K fragment of the tile first, a scheduling fence, then the two
            // chains -- left alone hipcc emits [one or two reads, wait, one MFMA] sixteen times) + BD from the skew buffer: the
            // chains run behind the skew writes above, the 32 skew reads are issued behind the chains, so the write -> read round
            // trip of the lane-private buffer passes under the MFMAs.  Band mask on a scalar flag (two code paths: hipcc otherwise
            // if-converts the mask into per-element compares on every tile).
            const bool full = (dmin_w >= 0) && (dmax_w <= M - 1) && (iw0 + 31 < T);
            const int qi = iw0 + r;
            f32x16 s[2];
            auto tile_scores = [&](auto masked) __attribute__((always_inline)) {
                constexpr bool MASKED = decltype(masked)::value;
                {
                    bf16x8 ka[2][KS];
#pragma unroll
                    for (int kb = 0; kb < 2; kb++)
#pragma unroll
                        for (int ks = 0; ks < KS; ks++)
                            ka[kb][ks] = *reinterpret_cast<const bf16x8*>(cK + G::koff(32 * kb + r, 2 * ks + hh));
                    __builtin_amdgcn_sched_barrier(0);
                    s[0] = cinit; s[1] = cinit;      // = -m_run: the scores come out relative to the softmax reference
                    FPRIO_UP(1);
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                        for (int kb = 0; kb < 2; kb++)
                            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, ka[kb][ks]),
                                                                            __builtin_bit_cast(mfma_bf16x8, qw[ks]), s[kb], 0, 0, 0);
                    }
                    FPRIO_DOWN(1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                STAMP(3)
                uint32_t u0[16], u1[16];
                skew_issue16(gRb, u0);
                skew_issue16_w15(gRb - 64, u1);
#pragma unroll
                for (int kb = 0; kb < 2; kb++) {
                    if (kb == 1) lgkm_wait0();
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        float val = add_f16v(s[kb][j], kb == 0 ? u0[j] : u1[j]);
                        if (MASKED) {
                            const int d = qi - P - (32 * kb + (j & 3) + 8 * (j >> 2) + 4 * hh);
                            const bool valid = (d >= 0) && (d <= M - 1) && (qi < T);
                            val = valid ? val : NEG_BIG;
                        }
                        s[kb][j] = val;
                    }
                }
            };
            auto scores = [&]() __attribute__((always_inline)) { if (full) tile_scores(std::false_type{}); else tile_scores(std::true_type{}); };
            const bool attempt = all_set;
            bool precise = !attempt, bad = false;
            float rs = 0.f;
            if (attempt) {
                scores();
                STAMP(4)
                rs = exp_sum(s);
                bad = !(rs <= FAST_SUM_MAX);
                precise = __any(bad);
            }
            if (precise) {
                scores();
                rebase(s, rescale_o, attempt, rs);
                rs = exp_sum(s);
            }
            l_run += rs;
            STAMP(5)
            // ---- O^T += V^T . P^T
            FPRIO_UP(2);
#pragma unroll
            for (int kb = 0; kb < 2; kb++) {
#pragma unroll
                for (int st = 0; st < 2; st++) {
                    const u32x4 pw = {pack2bf(s[kb][8 * st], s[kb][8 * st + 1]), pack2bf(s[kb][8 * st + 2], s[kb][8 * st + 3]),
                                      pack2bf(s[kb][8 * st + 4], s[kb][8 * st + 5]), pack2bf(s[kb][8 * st + 6], s[kb][8 * st + 7])};
                    const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
                    const int gq = l >> 4, li = l & 15, q4 = li >> 2, pp = li & 3;
#pragma unroll
                    for (int e = 0; e < EB; e++) {
                        const int key = 32 * kb + 16 * st + 4 * hh + q4;
                        const int byte = (32 * e + 16 * (gq & 1) + 4 * pp) * 2;
                        const lds_bf16x4* a0 = (const lds_bf16x4*)(cV + G::voff(key, byte));
                        const lds_bf16x4* a1 = (const lds_bf16x4*)(cV + G::voff(key + 8, byte));
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)a0);
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)a1);
                        const bf16x8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        o[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                       __builtin_bit_cast(mfma_bf16x8, pf), o[e], 0, 0, 0);
                    }
                }
            }
            FPRIO_DOWN(2);
        }
        STAMP(6)
        __syncthreads();                     // every wave is done reading this tile
        STAMP(7)
        store_kv(rk, rv);
        store_r(rr[0], i0 - (P + KT) - 64);
        STAMP(8)
        __syncthreads();
        STAMP(9)
    }
    STAMP(15)

    // ---- epilogue: normalise, store O (lane = query, 4 consecutive e per register group) and LSE
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    const int qi = iw0 + r;
    if (qi < T) {
        bf16_t* op = p.out + (size_t)b * p.o_bs + (size_t)qi * p.o_rs + (size_t)h * DH;
#pragma unroll
        for (int e = 0; e < EB; e++) {
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                const int e0 = 32 * e + 8 * grp + 4 * hh;
                if (e0 < DH) {
                    u32x2 w = {pack2bf(o[e][4 * grp] * inv, o[e][4 * grp + 1] * inv),
                               pack2bf(o[e][4 * grp + 2] * inv, o[e][4 * grp + 3] * inv)};
                    *reinterpret_cast<u32x2*>(op + e0) = w;
                }
            }
        }
        const float lse2 = m_run + __builtin_amdgcn_logf(l_tot);          // log2 units
        if (hh == 0 && p.lse) p.lse[((size_t)b * p.H + h) * T + qi] = lse2 * 0.6931471805599453f;
        if (hh == 0 && p.ph_rec)
            reinterpret_cast<float*>(p.ph_rec + (((size_t)b * p.H + h) * (size_t)(T >> 5) + (iw0 >> 5)) * 4352 + 4096)[r] = -lse2;
    }
    STAMP(15)
    STAMP_FLUSH
}

template <int DH>
int launch_fwd(const RelAttnP& p, hipStream_t s) {
    using G = Geo<DH>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&relattn_fwd_kernel<DH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((p.T + QB - 1) / QB, p.H, p.B);
    {
        mxl_kt::Scope kt(MXL_KT_RELATTN_FWD, s);
        hipLaunchKernelGGL((relattn_fwd_kernel<DH>), grid, dim3(256), G::SMEM, s, p);
    }
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

}  // namespace

#ifdef MXL_STAMP
extern "C" int mxl_debug_fwd_stamps(unsigned long long* host_out16) {
    hipError_t e = hipMemcpyFromSymbol(host_out16, HIP_SYMBOL(g_fwd_stamps), sizeof(unsigned long long) * 16);
    if (e != hipSuccess) return (int)e;
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_stamps), z, sizeof(z));
    return (int)e;
}
#endif

static int relattn_fwd_launch(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                              const float* r_r_bias, void* out, float* lse, int B, int T, int H, int dh, int M, int Kc,
                              long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs, long long o_bs, int o_rs,
                              float scale, void* oph, float* mph, void* stream, int oph_all = 0, void* ph_ws = nullptr) {
    MXL_CHECK_ARG(q && k && v && rd && r_w_bias && r_r_bias && out);
    if (oph_all) MXL_CHECK_ARG(oph && ((T - Kc) % 64) == 0);
    if (ph_ws) MXL_CHECK_ARG(oph_all && dh == 64 && ((uintptr_t)ph_ws % 16) == 0);
    // (oph over the all-phantom 256-distance blocks needs whole blocks; over every phantom cell, oph_all, any multiple of 32 distances)
    if (oph) MXL_CHECK_ARG(mph && (M % (oph_all ? 32 : 256)) == 0 && (T % 32) == 0 && ((uintptr_t)oph % 8) == 0);
    MXL_CHECK_ARG(B > 0 && T > 0 && H > 0 && M > 0 && Kc >= T && Kc <= M + T);
    MXL_CHECK_ARG((q_rs % 8) == 0 && (kv_rs % 8) == 0 && (rd_rs % 8) == 0 && (o_rs % 4) == 0);
    MXL_CHECK_ARG((q_bs % 8) == 0 && (kv_bs % 8) == 0 && (o_bs % 4) == 0);
    // 32-bit byte offsets inside one sequence's K / V rows and inside rd (buffer addressing)
    MXL_CHECK_ARG((long long)Kc * kv_rs * 2 < (1ll << 31) && (long long)M * rd_rs * 2 < (1ll << 31));
    MXL_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                  ((uintptr_t)rd % 16) == 0 && ((uintptr_t)out % 8) == 0);
    RelAttnP p;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.rd = (const bf16_t*)rd;
    p.rwb = r_w_bias; p.rrb = r_r_bias; p.out = (bf16_t*)out; p.lse = lse;
    p.B = B; p.T = T; p.H = H; p.M = M; p.Kc = Kc;
    p.q_bs = q_bs; p.kv_bs = kv_bs; p.o_bs = o_bs; p.q_rs = q_rs; p.kv_rs = kv_rs; p.rd_rs = rd_rs; p.o_rs = o_rs;
    p.scale_log2e = scale * 1.4426950408889634f;
    p.oph = (bf16_t*)oph; p.mph = mph; p.oph_all = oph_all ? 1 : 0; p.ph_rec = (char*)ph_ws;
    hipStream_t s = (hipStream_t)stream;
    switch (dh) {
        case 16: return launch_fwd<16>(p, s);
        case 32: return launch_fwd<32>(p, s);
        case 64: return launch_fwd<64>(p, s);
        default: return MXL_EUNSUPPORTED;
    }
}

extern "C" int mxl_relattn_fwd(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                               const float* r_r_bias, void* out, float* lse, int B, int T, int H, int dh, int M, int Kc,
                               long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs, long long o_bs, int o_rs,
                               float scale, void* stream) {
    return relattn_fwd_launch(q, k, v, rd, r_w_bias, r_r_bias, out, lse, B, T, H, dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs,
                              o_rs, scale, nullptr, nullptr, stream);
}

extern "C" int mxl_relattn_fwd_phantom(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                                       const float* r_r_bias, void* out, float* lse, void* oph, float* mph, int B, int T, int H,
                                       int dh, int M, int Kc, long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs,
                                       long long o_bs, int o_rs, float scale, void* stream) {
    MXL_CHECK_ARG(oph && mph);
    return relattn_fwd_launch(q, k, v, rd, r_w_bias, r_r_bias, out, lse, B, T, H, dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs,
                              o_rs, scale, oph, mph, stream);
}

extern "C" int mxl_relattn_fwd_phantom2(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                                        const float* r_r_bias, void* out, float* lse, void* oph, float* mph, int oph_all, void* ph_ws,
                                        int B, int T, int H, int dh, int M, int Kc, long long q_bs, int q_rs, long long kv_bs, int kv_rs,
                                        int rd_rs, long long o_bs, int o_rs, float scale, void* stream) {
    MXL_CHECK_ARG(oph && mph);
    return relattn_fwd_launch(q, k, v, rd, r_w_bias, r_r_bias, out, lse, B, T, H, dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs,
                              o_rs, scale, oph, mph, stream, oph_all, ph_ws);
}
