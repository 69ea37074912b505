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
// pat(j) = (j & 3) + 8 * (j >> 2).  Loads and their wait sit in ONE asm statement (hipcc does not count asm loads).
__device__ __forceinline__ void skew_read16(uint32_t base, uint32_t (&u)[16]) {
    asm volatile(
        "ds_read_u16 %0, %16 offset:54\n\t"  "ds_read_u16 %1, %16 offset:52\n\t"  "ds_read_u16 %2, %16 offset:50\n\t"
        "ds_read_u16 %3, %16 offset:48\n\t"  "ds_read_u16 %4, %16 offset:38\n\t"  "ds_read_u16 %5, %16 offset:36\n\t"
        "ds_read_u16 %6, %16 offset:34\n\t"  "ds_read_u16 %7, %16 offset:32\n\t"  "ds_read_u16 %8, %16 offset:22\n\t"
        "ds_read_u16 %9, %16 offset:20\n\t"  "ds_read_u16 %10, %16 offset:18\n\t" "ds_read_u16 %11, %16 offset:16\n\t"
        "ds_read_u16 %12, %16 offset:6\n\t"  "ds_read_u16 %13, %16 offset:4\n\t"  "ds_read_u16 %14, %16 offset:2\n\t"
        "ds_read_u16 %15, %16\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7]),
          "=&v"(u[8]), "=&v"(u[9]), "=&v"(u[10]), "=&v"(u[11]), "=&v"(u[12]), "=&v"(u[13]), "=&v"(u[14]), "=&v"(u[15])
        : "v"(base)
        : "memory");
}

// s + (float)h in ONE VALU issue (v_fma_mix_f32: f32 * 1.0 + f16 taken from the low half of `h16`)
__device__ __forceinline__ float add_f16(float s, uint32_t h16) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(s), "v"(h16));
    return r;
}

// max(a, b, c) in one issue, without the canonicalising v_max_f32 x, x that fmaxf() gets in front of values produced by inline
// asm (the scores come out of v_fma_mix): 16 instead of 56 VALU issues for the 32 scores of a tile
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

constexpr float RESCALE_THRESH = 8.0f;   // log2 units: accumulators are re-based when a score exceeds the reference by 2^8

template <int DH>
__global__ __launch_bounds__(256, 2) void relattn_fwd_kernel(RelAttnP p) {
    using G = Geo<DH>;
    constexpr int KS = G::KS, EB = G::EB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                          // [64][DH]
    char* sV = sK + G::K_BYTES;               // [64][VW]
    char* sR = sV + G::V_BYTES;               // ring [256][DH]
    _Float16* sG = reinterpret_cast<_Float16*>(sR + G::R_BYTES);  // [4][32][GS] fp16

    const int tid = threadIdx.x;
    const int wid = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
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
    // LDS byte address of column (r - 4hh + 64 - 27) for skew_read16 (key block kb subtracts 32 columns = 64 bytes)
    const uint32_t gRb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(const char*)(gR - 27);
    f16x4 carry[4];   // block 0 of the previous tile (= block 2 of this one), kept in registers
    // Rd ring fragments: slot = (16-aligned window base + r) & 255, so the XOR-swizzle term of its row depends on the lane only
    int rswz[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) rswz[ks] = (DH == 64) ? (((2 * ks + hh) ^ ((r >> 1) & 7)) << 4) : ((2 * ks + hh) << 4);

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
    u32x4 rk[G::NLD_K], rv[G::NLD_V], rr[G::NLD_K];
    // K / V / Rd rows through buffer descriptors (wave-uniform base in scalar registers, 32-bit lane offset): rows below the first
    // stored key (negative offset = huge unsigned) and past the last one fall outside num_records and read as zero -- upstream's zero
    // memories -- without a branch, a select or 64-bit lane address arithmetic per load (the flat-pointer form spent ~15
    // instructions and an exec-masked branch on each of the six loads at the top of every tile)
    const unsigned kv_bytes = (unsigned)(((long long)(p.Kc - 1) * p.kv_rs + DH) * 2);
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)kv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)kv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)rbase, 0, -1, 0x00020000);
    auto load_kv = [&](int kt) {
        const int P = kt * KT;
#pragma unroll
        for (int n = 0; n < G::NLD_K; n++) {
            const int c = tid + n * 256;
            const int row = c / G::CH, ch = c % G::CH;
            const int srow = P + row - p0;
            if (KT * G::CH % 256 == 0 || c < KT * G::CH) {
                const int off = (srow * p.kv_rs + ch * 8) * 2;
                rk[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_k, off, 0, 0));
                rv[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, off, 0, 0));
            } else {
                rk[n] = u32x4{0u, 0u, 0u, 0u}; rv[n] = u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto store_kv = [&]() {
#pragma unroll
        for (int n = 0; n < G::NLD_K; n++) {
            const int c = tid + n * 256;
            if (c < KT * G::CH) {
                const int row = c / G::CH, ch = c % G::CH;
                *reinterpret_cast<u32x4*>(sK + G::koff(row, ch)) = rk[n];
                *reinterpret_cast<u32x4*>(sV + G::voff(row, ch * 16)) = rv[n];
            }
        }
    };
    // 64 Rd rows d in [dbase, dbase+64) -> registers / ring slots (d & 255); row index clamped (masked anyway)
    auto load_r = [&](int dbase) {
#pragma unroll
        for (int n = 0; n < G::NLD_K; n++) {
            const int c = tid + n * 256;
            const int row = c / G::CH, ch = c % G::CH;
            int d = dbase + row;
            d = d < 0 ? 0 : (d > M - 1 ? M - 1 : d);
            if (KT * G::CH % 256 == 0 || c < KT * G::CH)
                rr[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, (d * p.rd_rs + ch * 8) * 2, 0, 0));
            else rr[n] = u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto store_r = [&](int dbase) {
#pragma unroll
        for (int n = 0; n < G::NLD_K; n++) {
            const int c = tid + n * 256;
            if (c < KT * G::CH) {
                const int row = c / G::CH, ch = c % G::CH;
                const int slot = (dbase + row) & 255;
                *reinterpret_cast<u32x4*>(sR + G::koff(slot, ch)) = rr[n];
            }
        }
    };

    // zero the V pad columns once (DH < 32): tr-reads of O^T rows >= DH must see zeros
    if (DH < G::VW) {
        for (int i = tid; i < G::V_BYTES / 4; i += 256) reinterpret_cast<uint32_t*>(sV)[i] = 0u;
        __syncthreads();
    }

    // Online softmax with a LAZY reference: every accumulated quantity is relative to m_run (log2 units), which is only
    // moved when a new score exceeds it by RESCALE_THRESH (or when it is still unset = NEG_BIG).  -m_run is kept in 16
    // registers (`cinit`) and enters the score for free as the C operand of the first MFMA of each chain, so a score costs
    // one v_fma_mix (S + BD), half a v_max3, one v_exp and one add.  Both lanes of a query (hh = 0/1) see the same merged
    // maximum and therefore take identical decisions.
    float m_run = NEG_BIG, l_run = 0.f;
    f32x16 cinit;
#pragma unroll
    for (int j = 0; j < 16; j++) cinit[j] = 0.f;

    // ---- phantom keys.  Key positions below the first stored tile (pz) are upstream's zero mems: k = v = 0, so such a key
    // contributes exp(BD) to the softmax denominator and nothing else, and BD depends only on the distance.  Instead of
    // walking those key tiles (S, skew, PV all wasted) walk the DISTANCES d in [i - pz + 1, M - 1] block-wise: G^T blocks
    // straight from the accumulators, no skew, no K/V traffic.  The tile loop then starts at tile pz.
    const int pz = floordiv(p0, KT) * KT;
    int kt_start = kt_lo;
    const int tk0 = G::eoff(4 * hh + ((l & 15) >> 2), 16 * ((l >> 4) & 1) + 4 * (l & 3));
    f32x16 o[EB];          // O^T accumulators; during the phantom loop they hold oph (v = 0 there: O itself stays 0)
#pragma unroll
    for (int e = 0; e < EB; e++)
#pragma unroll
        for (int j = 0; j < 16; j++) o[e][j] = 0.f;
    if (kt_lo * KT < pz) {
        kt_start = pz / KT;
        const int qi = iw0 + r;
        const int db0 = i0 - pz;
        // Rd rows in 64-distance chunks through the 256-row ring: chunk c + 1 is stored (from registers loaded an iteration earlier)
        // while chunk c is read -- different ring slots -- so ONE barrier per chunk covers both "c + 1 is complete" and "everybody is
        // done with c" (the first form stored and read the same chunk between two barriers; an iteration is only 16 MFMAs per wave)
        load_r(db0);
        store_r(db0);
        if (db0 + 64 <= M - 1) load_r(db0 + 64);
        __syncthreads();
#pragma unroll 1
        for (int db = db0; db <= M - 1; db += 64) {
            if (db + 64 <= M - 1) store_r(db + 64);
            if (db + 128 <= M - 1) load_r(db + 128);
            // the chunk's two 32-distance blocks side by side: both G chains issued before either block's softmax work
            bool on[2], fullb[2];
#pragma unroll
            for (int gb = 0; gb < 2; gb++) {
                const int dblk = db + 32 * gb;
                on[gb] = (iw0 < T) && !(dblk + 31 <= iw0 - pz || dblk > M - 1);        // wave-uniform: a phantom cell in this block?
                fullb[gb] = (dblk >= iw0 + 31 - pz + 1) && (dblk + 31 <= M - 1) && (iw0 + 31 < T);   // every cell phantom and in range
            }
            if (on[0] || on[1]) {
                f32x16 g[2];
                bf16x8 ra[2][KS];
#pragma unroll
                for (int gb = 0; gb < 2; gb++) {
                    const int slot = (db + 32 * gb + r) & 255;
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) ra[gb][ks] = *reinterpret_cast<const bf16x8*>(sR + slot * G::ROWB + rswz[ks]);
                }
                __builtin_amdgcn_sched_barrier(0);
                g[0] = cinit; g[1] = cinit;
#pragma unroll
                for (int ks = 0; ks < KS; ks++)
#pragma unroll
                    for (int gb = 0; gb < 2; gb++)
                        g[gb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, ra[gb][ks]),
                                                                        __builtin_bit_cast(mfma_bf16x8, qr[ks]), g[gb], 0, 0, 0);
                float mx = NEG_BIG;
#pragma unroll
                for (int gb = 0; gb < 2; gb++) {
                    const int dblk = db + 32 * gb;
                    const bool fl = __builtin_amdgcn_readfirstlane((int)(on[gb] && fullb[gb])) != 0;
                    if (!fl) {          // (a block that is off altogether is masked out cell by cell: d <= qi - pz or d > M - 1)
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const int d = dblk + (j & 3) + 8 * (j >> 2) + 4 * hh;
                            const bool valid = (d >= qi - pz + 1) && (d <= M - 1) && (qi < T);
                            g[gb][j] = valid ? g[gb][j] : NEG_BIG;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 16; j += 2) mx = max3(mx, g[gb][j], g[gb][j + 1]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const bool unset = m_run == NEG_BIG;
                const bool need = unset ? (mx > 0.5f * NEG_BIG) : (mx > RESCALE_THRESH);
                if (__any(need)) {
                    const float delta = need ? mx : 0.f;
                    const float alpha = (need && !unset) ? __builtin_amdgcn_exp2f(-delta) : 1.f;
                    l_run *= alpha;
                    if (need) m_run = (unset ? 0.f : m_run) + delta;
                    const float neg = (m_run == NEG_BIG) ? 0.f : -m_run;
#pragma unroll
                    for (int j = 0; j < 16; j++) { g[0][j] -= delta; g[1][j] -= delta; cinit[j] = neg; }
                    if (p.oph) {
#pragma unroll
                        for (int e = 0; e < EB; e++)
#pragma unroll
                            for (int j = 0; j < 16; j++) o[e][j] *= alpha;
                    }
                }
                float rs = 0.f;
#pragma unroll
                for (int gb = 0; gb < 2; gb++)
#pragma unroll
                    for (int j = 0; j < 16; j++) { g[gb][j] = __builtin_amdgcn_exp2f(g[gb][j]); rs += g[gb][j]; }   // exp2(NEG_BIG) = 0
                l_run += rs;
                // oph: only over the blocks the backward skips (every cell of them is phantom and in range)
#pragma unroll
                for (int gb = 0; gb < 2; gb++) {
                    const int dblk = db + 32 * gb;
                    if (p.oph && on[gb] && (p.oph_all || (dblk & ~255) > iw0 + 31 - pz)) {
                        const int gq = l >> 4, li = l & 15, q4 = li >> 2, pp = li & 3;
                        const char* rb = sR + (dblk & 255) * G::ROWB;      // a multiple of 32 rows: + 0..31 does not wrap
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
                                    const char* a0 = rb + tk0 + 16 * st * G::ROWB;
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
            }
            __syncthreads();
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
    {
        const int P0 = kt_start * KT;
        load_kv(kt_start);
        store_kv();
#pragma unroll 1
        for (int q4 = 0; q4 < 3; q4++) {
            const int dbase = i0 - P0 - 64 + 64 * q4;
            load_r(dbase);
            store_r(dbase);
        }
    }
    __syncthreads();

    bool have_ring = false;

#pragma unroll 1
    for (int kt = kt_start; kt <= kt_hi; kt++) {
        const int P = kt * KT;
        const bool more = kt < kt_hi;
        if (more) {
            load_kv(kt + 1);
            load_r(i0 - (P + KT) - 64);   // next step's 64 new (lowest) distances
        }
        const int dmin_w = iw0 - P - (KT - 1), dmax_w = iw0 + 31 - P;
        const bool active = (dmax_w >= 0) && (dmin_w <= M - 1) && (iw0 < T);
        if (active) {
            const char* cK = sK;
            const char* cV = sV;
            const int dlo = iw0 - P - 64;
            // ---- S^T = K . Qw^T : two 32-key blocks
            // (round 4) every K fragment of the tile first, a scheduling fence, then the two chains: left alone hipcc emits
            // [one or two reads, wait, one MFMA] sixteen times -- an LDS round trip per MFMA with two waves per SIMD to hide it
            f32x16 s[2];
            {
                bf16x8 ka[2][KS];
#pragma unroll
                for (int kb = 0; kb < 2; kb++)
#pragma unroll
                    for (int ks = 0; ks < KS; ks++)
                        ka[kb][ks] = *reinterpret_cast<const bf16x8*>(cK + G::koff(32 * kb + r, 2 * ks + hh));
                __builtin_amdgcn_sched_barrier(0);
                s[0] = cinit; s[1] = cinit;      // = -m_run: the scores come out relative to the softmax reference
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                    for (int kb = 0; kb < 2; kb++)
                        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, ka[kb][ks]),
                                                                        __builtin_bit_cast(mfma_bf16x8, qw[ks]), s[kb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- G^T = Rd . Qr^T for the new distance blocks -> lane-private skew buffer (fp16)
            auto gblock = [&](int gb, f16x4 (&dst)[4]) {
                f32x16 g;
#pragma unroll
                for (int j = 0; j < 16; j++) g[j] = 0.f;
                const int slot = (dlo + 32 * gb + r) & 255;
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(sR + slot * G::ROWB + rswz[ks]);
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
                    const int slot = (dlo + 32 * gb + r) & 255;
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) ra[gb][ks] = *reinterpret_cast<const bf16x8*>(sR + slot * G::ROWB + rswz[ks]);
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x16 g0, g1;
#pragma unroll
                for (int j = 0; j < 16; j++) { g0[j] = 0.f; g1[j] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, ra[0][ks]),
                                                                 __builtin_bit_cast(mfma_bf16x8, qr[ks]), g0, 0, 0, 0);
                    g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, ra[1][ks]),
                                                                 __builtin_bit_cast(mfma_bf16x8, qr[ks]), g1, 0, 0, 0);
                }
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
            // ---- scores: (AC + BD) * scale, band mask, online softmax (lane = query).  Two explicit code paths on a
            // scalar (readfirstlane) flag: hipcc otherwise if-converts the mask into per-element compares on every tile.
            const bool full = __builtin_amdgcn_readfirstlane((int)((dmin_w >= 0) && (dmax_w <= M - 1) && (iw0 + 31 < T))) != 0;
            const int qi = iw0 + r;
            float mx = NEG_BIG;
            auto scores = [&](auto masked) {
                constexpr bool MASKED = decltype(masked)::value;
#pragma unroll
                for (int kb = 0; kb < 2; kb++) {
                    uint32_t bdu[16];
                    skew_read16(gRb - 64 * kb, bdu);
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        float val = add_f16(s[kb][j], bdu[j]);
                        if (MASKED) {
                            const int d = qi - P - (32 * kb + (j & 3) + 8 * (j >> 2) + 4 * hh);
                            const bool valid = (d >= 0) && (d <= M - 1) && (qi < T);
                            val = valid ? val : NEG_BIG;
                        }
                        s[kb][j] = val;
                    }
#pragma unroll
                    for (int j = 0; j < 16; j += 2) mx = max3(mx, s[kb][j], s[kb][j + 1]);
                }
            };
            if (full) scores(std::false_type{}); else scores(std::true_type{});
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            {
                const bool unset = m_run == NEG_BIG;
                const bool need = unset ? (mx > 0.5f * NEG_BIG) : (mx > RESCALE_THRESH);
                if (__any(need)) {               // rare after the first tile: move the reference, re-base O, l and the scores
                    const float delta = need ? mx : 0.f;
                    const float alpha = (need && !unset) ? __builtin_amdgcn_exp2f(-delta) : 1.f;
                    if (need) m_run = (unset ? 0.f : m_run) + delta;
                    const float neg = (m_run == NEG_BIG) ? 0.f : -m_run;
                    l_run *= alpha;
#pragma unroll
                    for (int j = 0; j < 16; j++) { s[0][j] -= delta; s[1][j] -= delta; cinit[j] = neg; }
#pragma unroll
                    for (int e = 0; e < EB; e++)
#pragma unroll
                        for (int j = 0; j < 16; j++) o[e][j] *= alpha;
                }
            }
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; kb++) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const float pv = __builtin_amdgcn_exp2f(s[kb][j]);     // masked cells: exp2(NEG_BIG) = 0
                    s[kb][j] = pv;
                    rs += pv;
                }
            }
            l_run += rs;
            // ---- O^T += V^T . P^T
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
        }
        __syncthreads();                     // every wave is done reading this tile
        if (more) {
            store_kv();
            store_r(i0 - (P + KT) - 64);
        }
        __syncthreads();
    }

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

static int relattn_fwd_launch(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                              const float* r_r_bias, void* out, float* lse, int B, int T, int H, int dh, int M, int Kc,
                              long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs, long long o_bs, int o_rs,
                              float scale, void* oph, float* mph, void* stream, int oph_all = 0, void* ph_ws = nullptr) {
    MXL_CHECK_ARG(q && k && v && rd && r_w_bias && r_r_bias && out);
    if (oph_all) MXL_CHECK_ARG(oph && ((T - Kc) % 64) == 0);
    if (ph_ws) MXL_CHECK_ARG(oph_all && dh == 64 && ((uintptr_t)ph_ws % 16) == 0);
    if (oph) MXL_CHECK_ARG(mph && (M % 256) == 0 && (T % 32) == 0 && ((uintptr_t)oph % 8) == 0);
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
