// K4 backward: gradients of the banded relative-position attention (see relattn_fwd.hip for the forward statement).
//
// With raw = AC + BD, s = raw * scale, P = exp(s - lse), dP[i,p] = dO_i . v_p, delta_i = dO_i . O_i:
//     dSr = scale * P * (dP - delta)                        (gradient w.r.t. raw scores)
//     dQw_i = sum_p dSr[i,p] k_p          dk_p = sum_i dSr[i,p] (q_i + r_w_bias)        dv_p = sum_i P[i,p] dO_i
//     dQr_i = sum_d dG[i,d] Rd[d]         dRd[d] = sum_i dG[i,d] (q_i + r_r_bias)       dG[i,d] = dSr[i, i-d]
//     dq = dQw + dQr,  d r_w_bias = sum_i dQw_i,  d r_r_bias = sum_i dQr_i
//
// Three launches, owner-computes, no atomics on the big tensors (deterministic):
//   relattn_bwd_delta : delta
//   relattn_bwd_dq    : query-owner (lane = query, same structure as the forward).  Produces dq, the bias gradients and
//                       dG (B,H,T,M) in bf16 -- the un-skewed score gradient -- which the host contracts with
//                       (q + r_r_bias) by the batched TT GEMM to obtain dRd (a correlation along diagonals that neither a
//                       query- nor a key-owner can accumulate on chip).
//   relattn_bwd_dkv   : key-owner (lane = key): dk, dv.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "musicxl_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;

// In-kernel stamps of the query-owner kernel's tile loop (diagnostic builds only: scripts/ab_build.sh relattn_bwd stamp -DMXL_STAMP;
// the shipped library contains none of this).  Per wave, shader cycles between consecutive stamps are summed per segment in
// scalar registers and added to g_dq8_stamps once at the end; read them back with mxl_debug_dq8_stamps.  A stamp waits lgkmcnt(0),
// so read the SHARES, not the run time (cdna_hip_programming.md, In-kernel stamps).
#ifdef MXL_STAMP
__device__ unsigned long long g_dq8_stamps[16];
#define STAMP_DECL unsigned long long st_last, st_acc[12] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}; \
    { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define STAMP(i) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); st_acc[i] += t_ - st_last; st_last = t_; }
#define STAMP_FLUSH if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 12; i_++) atomicAdd(&g_dq8_stamps[i_], st_acc[i_]); }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

__device__ __forceinline__ int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

struct BwdP {
    const bf16_t *q, *k, *v, *rd, *o, *dout;
    const float *rwb, *rrb, *lse, *delta;
    bf16_t *dq, *dk, *dv, *dg;
    float *d_rwb, *d_rrb;
    float* delta_out;
    int B, T, H, M, Kc;
    long long q_bs, kv_bs, o_bs, dq_bs, dkv_bs;
    int q_rs, kv_rs, rd_rs, o_rs, dq_rs, dkv_rs;
    float scale, scale_log2e;
    // 8-wave query-owner kernel only: do not store dG for 32-distance blocks whose 256-distance block lies entirely on phantom
    // distances of the wave's 32 queries -- mxl_relattn_drd_recompute rebuilds exactly those cells itself
    int dg_skip_phantom;
    // with dg_skip_phantom: the forward's phantom value-sum over exactly those blocks (mxl_relattn_fwd_phantom); the query-owner
    // kernel then does not visit them at all -- their dQr is -scale * delta_i * 2^(mph_i - lse2_i) * oph_i, added in its epilogue
    const bf16_t* oph;
    const float* mph;
};

// ---------------------------------------------------------------------------------------------------------------
// delta[b,h,i] = sum_e dO[b,i,h,e] * O[b,i,h,e]        one wave per (b, i): lanes sweep the d = H*dh row
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relattn_bwd_delta_kernel(BwdP p, int dh) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.B * p.T) return;
    const int lane = threadIdx.x & 63;
    const int b = row / p.T, i = row % p.T;
    const bf16_t* op = p.o + (size_t)b * p.o_bs + (size_t)i * p.o_rs;
    const bf16_t* dp = p.dout + (size_t)b * p.o_bs + (size_t)i * p.o_rs;
    // each lane handles 8-element chunks; chunk c belongs to head (c*8)/dh
    const int chunks = p.H * dh / 8;
    const int cph = dh / 8;  // chunks per head (2, 4, 8)
    for (int c0 = 0; c0 < chunks; c0 += 64) {
        const int c = c0 + lane;
        float s = 0.f;
        if (c < chunks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(op + c * 8);
            const bf16x8 d = *reinterpret_cast<const bf16x8*>(dp + c * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) s += bf2f((bf16_t)a[j]) * bf2f((bf16_t)d[j]);
        }
        // reduce groups of cph consecutive lanes
        for (int off = 1; off < cph; off <<= 1) s += __shfl_xor(s, off, 64);
        if (c < chunks && (lane % cph) == 0) {
            const int h = c / cph;
            p.delta_out[((size_t)b * p.H + h) * p.T + i] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// query-owner kernel
// ---------------------------------------------------------------------------------------------------------------
constexpr int QB = 128, KT = 64;
constexpr int GS = 100;    // fp16 skew buffer: 96 distance columns (+pad) per query row (see relattn_fwd.hip)
constexpr int DGS = 104;   // bf16 un-skew buffer: 96 columns (+pad): 208-byte rows keep ds_read_b128 aligned

template <int DH> struct GeoQ {
    static constexpr int KS = DH / 16;
    static constexpr int EB = (DH + 31) / 32;
    static constexpr int ROWB = DH * 2;
    static constexpr int CH = DH / 8;
    static constexpr int K_BYTES = KT * ROWB;
    static constexpr int R_BYTES = 256 * ROWB;
    static constexpr int G_BYTES = 4 * 32 * GS * 2;
    static constexpr int DG_BYTES = 4 * 32 * DGS * 2;
    static constexpr int SMEM = 4 * K_BYTES + R_BYTES + G_BYTES + DG_BYTES;
    static constexpr int NLD = (KT * CH + 255) / 256;
    __device__ static __forceinline__ int koff(int row, int ch) {
        if (DH == 64) return row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4);
        return row * ROWB + (ch << 4);
    }
    // byte offset of element e (multiple of 4) of a row, for 8-byte transposed reads out of the same image
    __device__ static __forceinline__ int eoff(int row, int e) { return koff(row, e >> 3) + ((e & 7) << 1); }
};

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// see relattn_fwd.hip: 16 genuine ds_read_u16 at column offsets 27 - pat(j) from `base` (bytes), pat(j) = (j&3) + 8*(j>>2)
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
// the mirror image of skew_read16 for dSr: 16 genuine 16-bit writes at the same column pattern.
// packed variant: w[m] holds (value 2m, value 2m+1) as a bf16 pair; the odd one leaves through ds_write_b16_d16_hi
__device__ __forceinline__ void skew_write16p(uint32_t base, const uint32_t (&w)[8]) {
    asm volatile(
        "ds_write_b16 %8, %0 offset:54\n\t"         "ds_write_b16_d16_hi %8, %0 offset:52\n\t"
        "ds_write_b16 %8, %1 offset:50\n\t"         "ds_write_b16_d16_hi %8, %1 offset:48\n\t"
        "ds_write_b16 %8, %2 offset:38\n\t"         "ds_write_b16_d16_hi %8, %2 offset:36\n\t"
        "ds_write_b16 %8, %3 offset:34\n\t"         "ds_write_b16_d16_hi %8, %3 offset:32\n\t"
        "ds_write_b16 %8, %4 offset:22\n\t"         "ds_write_b16_d16_hi %8, %4 offset:20\n\t"
        "ds_write_b16 %8, %5 offset:18\n\t"         "ds_write_b16_d16_hi %8, %5 offset:16\n\t"
        "ds_write_b16 %8, %6 offset:6\n\t"          "ds_write_b16_d16_hi %8, %6 offset:4\n\t"
        "ds_write_b16 %8, %7 offset:2\n\t"          "ds_write_b16_d16_hi %8, %7"
        :
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(base)
        : "memory");
}
// s + (float)h in ONE VALU issue (v_fma_mix_f32: f32 * 1.0 + f16 from the low half of `h16`)
__device__ __forceinline__ float add_f16(float s, uint32_t h16) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(s), "v"(h16));
    return r;
}

// Operand scaling shared by the three attention kernels (relattn_fwd.hip builds its fragments the same way, so the
// recomputed scores match the forward's LSE bit for bit):  Qw, Qr carry scale*log2(e) (scores come out in exp2 units),
// dO carries `scale` (dP comes out as scale*dP).  -lse and -scale*delta enter as the C operand of the first MFMA of each
// chain, so the per-score VALU work is one v_fma_mix (S + BD), one v_exp and one multiply.
template <int DH>
__global__ __launch_bounds__(256, 1) void relattn_bwd_dq_kernel(BwdP p) {
    using G = GeoQ<DH>;
    constexpr int KS = G::KS, EB = G::EB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                       // [2][64][DH]   double-buffered: one barrier per tile, loads hidden behind compute
    char* sV = sK + 2 * G::K_BYTES;        // [2][64][DH]
    char* sR = sV + 2 * G::K_BYTES;        // ring [256][DH]
    _Float16* sG = reinterpret_cast<_Float16*>(sR + G::R_BYTES);                  // [4][32][GS] fp16
    bf16_t* sDG = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(sG) + G::G_BYTES);  // [4][32][DGS] bf16

    const int tid = threadIdx.x;
    const int wid = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
    int bx_, h, b;
    xcd_block(bx_, h, b);
    // longest-first: late query blocks see the most real keys (early ones mostly phantom distances), and they are dispatched
    // first so the tail of the launch is made of short workgroups
    const int i0 = (gridDim.x - 1 - bx_) * QB;
    const int iw0 = i0 + 32 * wid;
    const int T = p.T, M = p.M;
    const int p0 = T - p.Kc;
    // both per-wave buffers use FIXED columns c = distance - dlo in [0, 96): every address is a per-lane constant plus an
    // immediate.  Carry between tiles (block 0 -> block 2): G in registers, dG by an LDS move of the lane's own row.
    _Float16* gW = sG + wid * 32 * GS + r * GS + 4 * hh;
    const _Float16* gR = sG + wid * 32 * GS + r * GS + r + 64 - 4 * hh;
    const uint32_t gRb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(const char*)(gR - 27);
    bf16_t* myDG = sDG + wid * 32 * DGS + r * DGS;                 // row of this lane's query
    const uint32_t dgWb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(const char*)(myDG + r + 64 - 4 * hh - 27);
    f16x4 carry[4];
    // Rd ring fragments: slot = (16-aligned window base + r) & 255, so the XOR-swizzle term of its row depends on the lane only
    int rswz[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) rswz[ks] = (DH == 64) ? (((2 * ks + hh) ^ ((r >> 1) & 7)) << 4) : ((2 * ks + hh) << 4);

    const bf16_t* kbase = p.k + (size_t)b * p.kv_bs + (size_t)h * DH;
    const bf16_t* vbase = p.v + (size_t)b * p.kv_bs + (size_t)h * DH;
    const bf16_t* rbase = p.rd + (size_t)h * DH;
    const int qi = iw0 + r;
    const bool qok = qi < T;

    bf16x8 qw[KS], qr[KS], dof[KS];
    {
        const size_t qrow = (size_t)(qok ? qi : 0);
        const bf16_t* qp = p.q + (size_t)b * p.q_bs + qrow * p.q_rs + (size_t)h * DH;
        const bf16_t* dop = p.dout + (size_t)b * p.o_bs + qrow * p.o_rs + (size_t)h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int e0 = 16 * ks + 8 * hh;
            const bf16x8 qv = *reinterpret_cast<const bf16x8*>(qp + e0);
            const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dop + e0);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float qf = qok ? bf2f((bf16_t)qv[j]) : 0.f;
                qw[ks][j] = (short)f2bf((qf + p.rwb[h * DH + e0 + j]) * p.scale_log2e);
                qr[ks][j] = (short)f2bf((qf + p.rrb[h * DH + e0 + j]) * p.scale_log2e);
                dof[ks][j] = qok ? (short)f2bf(bf2f((bf16_t)dv[j]) * p.scale) : (short)0;
            }
        }
    }
    const size_t sidx = ((size_t)b * p.H + h) * T + (qok ? qi : 0);
    const float lse2 = qok ? p.lse[sidx] * LOG2E : 0.f;
    const float ndlt = qok ? -p.scale * p.delta[sidx] : 0.f;
    f32x16 c_lse, c_dlt;          // MFMA C operands: -lse (log2 units) and -scale*delta of this lane's query
#pragma unroll
    for (int j = 0; j < 16; j++) { c_lse[j] = -lse2; c_dlt[j] = ndlt; }

    // zero this wave's un-skew ring: never-written cells must read as 0
    {
        uint32_t* z = reinterpret_cast<uint32_t*>(sDG + wid * 32 * DGS);
        for (int i = l; i < 32 * DGS / 2; i += 64) z[i] = 0u;
    }

    const int p_lo = i0 - M + 1;
    const int p_hi = min(i0 + QB - 1, T - 1);
    const int kt_lo = floordiv(p_lo, KT), kt_hi = floordiv(p_hi, KT);

    u32x4 rk[G::NLD], rv[G::NLD], rr[G::NLD];
    auto load_kv = [&](int kt) {
        const int P = kt * KT;
#pragma unroll
        for (int n = 0; n < G::NLD; n++) {
            const int c = tid + n * 256;
            const int row = c / G::CH, ch = c % G::CH;
            const int srow = P + row - p0;
            u32x4 z = {0u, 0u, 0u, 0u};
            const bool ok = (c < KT * G::CH) && (srow >= 0) && (srow < p.Kc);
            rk[n] = ok ? *reinterpret_cast<const u32x4*>(kbase + (size_t)srow * p.kv_rs + ch * 8) : z;
            rv[n] = ok ? *reinterpret_cast<const u32x4*>(vbase + (size_t)srow * p.kv_rs + ch * 8) : z;
        }
    };
    auto store_kv = [&](int buf) {
#pragma unroll
        for (int n = 0; n < G::NLD; n++) {
            const int c = tid + n * 256;
            if (c < KT * G::CH) {
                const int row = c / G::CH, ch = c % G::CH;
                *reinterpret_cast<u32x4*>(sK + buf * G::K_BYTES + G::koff(row, ch)) = rk[n];
                *reinterpret_cast<u32x4*>(sV + buf * G::K_BYTES + G::koff(row, ch)) = rv[n];
            }
        }
    };
    auto load_r = [&](int dbase) {
#pragma unroll
        for (int n = 0; n < G::NLD; n++) {
            const int c = tid + n * 256;
            const int row = c / G::CH, ch = c % G::CH;
            int d = dbase + row;
            d = d < 0 ? 0 : (d > M - 1 ? M - 1 : d);
            u32x4 z = {0u, 0u, 0u, 0u};
            rr[n] = (c < KT * G::CH) ? *reinterpret_cast<const u32x4*>(rbase + (size_t)d * p.rd_rs + ch * 8) : z;
        }
    };
    auto store_r = [&](int dbase) {
#pragma unroll
        for (int n = 0; n < G::NLD; n++) {
            const int c = tid + n * 256;
            if (c < KT * G::CH) {
                const int row = c / G::CH, ch = c % G::CH;
                *reinterpret_cast<u32x4*>(sR + G::koff((dbase + row) & 255, ch)) = rr[n];
            }
        }
    };

    f32x16 aw[EB], ar[EB];  // dQw^T, dQr^T : [e][query]
#pragma unroll
    for (int e = 0; e < EB; e++)
#pragma unroll
        for (int j = 0; j < 16; j++) { aw[e][j] = 0.f; ar[e][j] = 0.f; }
    bool have_ring = false;
    int cur = 0;
    bf16_t* dgrow = p.dg ? p.dg + (((size_t)b * p.H + h) * T + (qok ? qi : 0)) * (size_t)M : nullptr;
    __syncthreads();   // dG buffer zeroed

    // ---- phantom keys (see relattn_fwd.hip): for key positions below the first stored tile pz, k = v = 0, hence S = 0 and
    // dP = 0: dSr = -scale * P * delta depends only on the distance.  Walk those DISTANCES d in [i - pz + 1, M - 1] block-wise:
    // G^T (no skew) -> P -> dG^T straight from the accumulators -> dQr MFMA + dG store.  No K/V, no S/dP/dQw, no LDS rings.
    // The one block that straddles real and phantom distances (d in [iw0-pz, iw0-pz+31]) is deposited into the un-skew buffer
    // (block-0 columns) so that the first real tile completes and emits it.
    const int pz = floordiv(p0, KT) * KT;
    int kt_start = kt_lo;
    if (kt_lo * KT < pz) {
        kt_start = pz / KT;
        const int gq = l >> 4, li = l & 15, q4 = li >> 2, pp = li & 3;
        load_r(i0 - pz);
#pragma unroll 1
        for (int db = i0 - pz; db <= M - 1; db += 64) {
            store_r(db);
            __syncthreads();
            if (db + 64 <= M - 1) load_r(db + 64);
            if (iw0 < T) {
#pragma unroll 1
                for (int gb = 0; gb < 2; gb++) {
                    const int dblk = db + 32 * gb;
                    if (dblk + 31 <= iw0 - pz || dblk > M - 1) continue;   // wave-uniform
                    f32x16 g = c_lse;
                    const int slot = (dblk + r) & 255;
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(sR + slot * G::ROWB + rswz[ks]);
                        g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                    __builtin_bit_cast(mfma_bf16x8, qr[ks]), g, 0, 0, 0);
                    }
                    // dSr = -scale * P * delta.  Only boundary blocks need the per-cell validity test (scalar branch).
                    const bool fullblk = __builtin_amdgcn_readfirstlane(
                        (int)((dblk >= iw0 + 31 - pz + 1) && (dblk + 31 <= M - 1) && (iw0 + 31 < T))) != 0;
                    if (fullblk) {
#pragma unroll
                        for (int j = 0; j < 16; j++) g[j] = __builtin_amdgcn_exp2f(g[j]) * ndlt;
                    } else {
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const int d = dblk + (j & 3) + 8 * (j >> 2) + 4 * hh;
                            const bool valid = (d >= qi - pz + 1) && (d <= M - 1) && qok;
                            g[j] = valid ? __builtin_amdgcn_exp2f(g[j]) * ndlt : 0.f;
                        }
                    }
                    if (dblk == iw0 - pz) {
                        // straddling block: into block-0 columns of the un-skew buffer (4 consecutive distances per store)
#pragma unroll
                        for (int grp = 0; grp < 4; grp++) {
                            const u32x2 w = {pack2bf(g[4 * grp], g[4 * grp + 1]), pack2bf(g[4 * grp + 2], g[4 * grp + 3])};
                            *reinterpret_cast<u32x2*>(myDG + 8 * grp + 4 * hh) = w;
                        }
                    } else {
                        if (dgrow && (M & 7) == 0) {
                            // lanes l and l + 32 (same query, distance groups 4 apart) trade one packed quad each, so that every
                            // lane stores 8 consecutive distances with one 16-byte store instead of two 8-byte ones (the store
                            // path's cost is per instruction and per row segment: scripts/ubench/stores.hip)
#pragma unroll
                            for (int gp = 0; gp < 2; gp++) {
                                const unsigned a0 = pack2bf(g[8 * gp], g[8 * gp + 1]), a1 = pack2bf(g[8 * gp + 2], g[8 * gp + 3]);
                                const unsigned b0 = pack2bf(g[8 * gp + 4], g[8 * gp + 5]), b1 = pack2bf(g[8 * gp + 6], g[8 * gp + 7]);
                                const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                                const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                                const int d8 = dblk + 16 * gp + 8 * hh;
                                if (qok && d8 + 7 <= M - 1) *reinterpret_cast<u32x4*>(dgrow + d8) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                            }
                        } else if (dgrow && qok) {
#pragma unroll
                            for (int grp = 0; grp < 4; grp++) {
                                const int d4 = dblk + 8 * grp + 4 * hh;
                                if (d4 + 3 <= M - 1) {
                                    const u32x2 w = {pack2bf(g[4 * grp], g[4 * grp + 1]), pack2bf(g[4 * grp + 2], g[4 * grp + 3])};
                                    *reinterpret_cast<u32x2*>(dgrow + d4) = w;
                                }
                            }
                        }
#pragma unroll
                        for (int st = 0; st < 2; st++) {
                            const u32x4 pw = {pack2bf(g[8 * st], g[8 * st + 1]), pack2bf(g[8 * st + 2], g[8 * st + 3]),
                                              pack2bf(g[8 * st + 4], g[8 * st + 5]), pack2bf(g[8 * st + 6], g[8 * st + 7])};
                            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
                            for (int e = 0; e < EB; e++) {
                                const int dist = dblk + 16 * st + 4 * hh + q4;      // accumulator-permuted k order
                                const int ecol = 32 * e + 16 * (gq & 1) + 4 * pp;
                                bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
                                if (ecol < DH) {
                                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sR + G::eoff(dist & 255, ecol)));
                                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sR + G::eoff((dist + 8) & 255, ecol)));
                                    a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                                }
                                ar[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                                __builtin_bit_cast(mfma_bf16x8, pf), ar[e], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }

    {
        const int P0 = kt_start * KT;
        load_kv(kt_start);
        store_kv(0);
#pragma unroll 1
        for (int q4 = 0; q4 < 3; q4++) {
            const int dbase = i0 - P0 - 64 + 64 * q4;
            load_r(dbase);
            store_r(dbase);
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int kt = kt_start; kt <= kt_hi; kt++) {
        const int P = kt * KT;
        const bool more = kt < kt_hi;
        if (more) {
            load_kv(kt + 1);
            load_r(i0 - (P + KT) - 64);
        }
        const int dmin_w = iw0 - P - (KT - 1), dmax_w = iw0 + 31 - P;
        const bool active = (dmax_w >= 0) && (dmin_w <= M - 1) && (iw0 < T);
        if (active) {
            const int dlo = iw0 - P - 64;
            const char* cK = sK + cur * G::K_BYTES;
            const char* cV = sV + cur * G::K_BYTES;
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
            gblock(0, b0);
            gblock(1, b1);
            // move the carried dG block: columns [0,32) of the previous tile are columns [64,96) of this one
            {
                const u32x4 m0 = *reinterpret_cast<const u32x4*>(myDG + 16 * hh);
                const u32x4 m1 = *reinterpret_cast<const u32x4*>(myDG + 16 * hh + 8);
                *reinterpret_cast<u32x4*>(myDG + 64 + 16 * hh) = m0;
                *reinterpret_cast<u32x4*>(myDG + 64 + 16 * hh + 8) = m1;
            }
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                *reinterpret_cast<f16x4*>(gW + 8 * grp) = b0[grp];
                *reinterpret_cast<f16x4*>(gW + 32 + 8 * grp) = b1[grp];
                *reinterpret_cast<f16x4*>(gW + 64 + 8 * grp) = carry[grp];
                carry[grp] = b0[grp];
            }
            have_ring = true;
            // The two 32-key blocks run one after the other (S/dP chains, skew read, P and dSr, skew write, dQw products) with a
            // scheduling fence between them: at most one block's scores are live, which keeps the kernel's working set in the
            // architectural VGPRs (the accumulators sit in AGPRs) instead of shuttling values through v_accvgpr moves.
            const bool full = __builtin_amdgcn_readfirstlane((int)((dmin_w >= 0) && (dmax_w <= M - 1) && (iw0 + 31 < T))) != 0;
            const int gq = l >> 4, li = l & 15, q4 = li >> 2, pp = li & 3;
#pragma unroll
            for (int kb = 0; kb < 2; kb++) {
                f32x16 s = c_lse, dp = c_dlt;
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(cK + G::koff(32 * kb + r, 2 * ks + hh));
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                __builtin_bit_cast(mfma_bf16x8, qw[ks]), s, 0, 0, 0);
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(cV + G::koff(32 * kb + r, 2 * ks + hh));
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, av),
                                                                 __builtin_bit_cast(mfma_bf16x8, dof[ks]), dp, 0, 0, 0);
                }
                uint32_t bdu[16];
                skew_read16(gRb - 64 * kb, bdu);
                if (full) {
#pragma unroll
                    for (int j = 0; j < 16; j++) s[j] = __builtin_amdgcn_exp2f(add_f16(s[j], bdu[j])) * dp[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const int d = qi - P - (32 * kb + (j & 3) + 8 * (j >> 2) + 4 * hh);
                        const bool valid = (d >= 0) && (d <= M - 1) && qok;
                        const float pv = __builtin_amdgcn_exp2f(add_f16(s[j], bdu[j]));
                        s[j] = valid ? pv * dp[j] : 0.f;
                    }
                }
                uint32_t dsw[8];          // dSr as bf16 pairs (2m, 2m+1): MFMA operand and skew-write source
#pragma unroll
                for (int m = 0; m < 8; m++) dsw[m] = pack2bf(s[2 * m], s[2 * m + 1]);
                skew_write16p(dgWb - 64 * kb, dsw);
                // dQw^T += K^T . dSr^T   (A = K^T through transposed reads of the K image, accumulator-permuted k order)
#pragma unroll
                for (int st = 0; st < 2; st++) {
                    const u32x4 pw = {dsw[4 * st], dsw[4 * st + 1], dsw[4 * st + 2], dsw[4 * st + 3]};
                    const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
                    for (int e = 0; e < EB; e++) {
                        const int key = 32 * kb + 16 * st + 4 * hh + q4;
                        const int ecol = 32 * e + 16 * (gq & 1) + 4 * pp;
                        bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (ecol < DH) {
                            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(cK + G::eoff(key, ecol)));
                            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(cK + G::eoff(key + 8, ecol)));
                            a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        }
                        aw[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                        __builtin_bit_cast(mfma_bf16x8, pf), aw[e], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // completed distance blocks 1 and 2 of the window: d in [dlo+32, dlo+95]
#pragma unroll
            for (int blk = 1; blk < 3; blk++) {
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ks2++) {
                    const int d8 = dlo + 32 * blk + 16 * ks2 + 8 * hh;  // 8 consecutive distances (natural k order)
                    const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(myDG + 32 * blk + 16 * ks2 + 8 * hh);
                    if (dgrow && qok && d8 >= 0 && d8 + 7 <= M - 1)
                        *reinterpret_cast<bf16x8*>(dgrow + d8) = bfrag;
#pragma unroll
                    for (int e = 0; e < EB; e++) {
                        const int dist = dlo + 32 * blk + 16 * ks2 + 8 * (gq >> 1) + q4;
                        const int ecol = 32 * e + 16 * (gq & 1) + 4 * pp;
                        bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (ecol < DH) {
                            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sR + G::eoff(dist & 255, ecol)));
                            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sR + G::eoff((dist + 4) & 255, ecol)));
                            a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        }
                        ar[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                        __builtin_bit_cast(mfma_bf16x8, bfrag), ar[e], 0, 0, 0);
                    }
                }
            }
        }
        if (more) {
            store_kv(cur ^ 1);
            store_r(i0 - (P + KT) - 64);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: dq = dQw + dQr (lane = query); bias gradients = sums over the wave's queries
    if (qok) {
        bf16_t* dqp = p.dq + (size_t)b * p.dq_bs + (size_t)qi * p.dq_rs + (size_t)h * DH;
#pragma unroll
        for (int e = 0; e < EB; e++) {
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                const int e0 = 32 * e + 8 * grp + 4 * hh;
                if (e0 < DH) {
                    u32x2 w = {pack2bf(aw[e][4 * grp] + ar[e][4 * grp], aw[e][4 * grp + 1] + ar[e][4 * grp + 1]),
                               pack2bf(aw[e][4 * grp + 2] + ar[e][4 * grp + 2], aw[e][4 * grp + 3] + ar[e][4 * grp + 3])};
                    *reinterpret_cast<u32x2*>(dqp + e0) = w;
                }
            }
        }
    }
    // per-wave column sums -> LDS, summed over the waves, ONE atomic instruction per workgroup and bias (lane = e).  Issued per
    // wave they are 64 two-lane atomic instructions each, 786 k instructions onto 1536 addresses per layer: measured 0.3 ms of
    // this kernel.
    __syncthreads();                                       // the K / V images are dead: reuse them
    float* sred = reinterpret_cast<float*>(smem);          // [4 waves][2][64]
#pragma unroll
    for (int e = 0; e < EB; e++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            float a = qok ? aw[e][j] : 0.f, c = qok ? ar[e][j] : 0.f;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); c += __shfl_xor(c, off, 64); }
            const int ee = 32 * e + (j & 3) + 8 * (j >> 2) + 4 * hh;
            if (r == 0 && ee < DH) { sred[(wid * 2 + 0) * 64 + ee] = a; sred[(wid * 2 + 1) * 64 + ee] = c; }
        }
    }
    __syncthreads();
    if (tid < 2 * DH) {
        const int which = tid / DH, ee = tid % DH;
        const float t = (sred[(0 * 2 + which) * 64 + ee] + sred[(1 * 2 + which) * 64 + ee]) +
                        (sred[(2 * 2 + which) * 64 + ee] + sred[(3 * 2 + which) * 64 + ee]);
        atomicAdd((which ? p.d_rrb : p.d_rwb) + h * DH + ee, t);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// query-owner kernel, 8-wave form (used when the caller takes d(r_r_bias) from the dRd kernel: d_r_r_bias == NULL).
// Two waves per 32-query group, split by key half, so every SIMD holds two waves instead of one -- the 4-wave kernel above is
// latency-bound at one.  At 256 registers per wave there is room for ONE dq accumulator: dQw and dQr are summed in it, its
// column sums (= d r_w_bias + d r_r_bias) go to d_rwb, and mxl_relattn_drd, which streams dG anyway, computes
// d r_r_bias = colsum(dG) . Rd and moves it from d_rwb to d_rrb.  Two barriers per tile (dS visible / next tile's G blocks and
// buffers ready).  Measured at C3: backward 1.74 -> 1.51 ms per layer against the 4-wave form.
// Every LDS address is (loop-invariant per-lane constant) + (wave-uniform offset) + (immediate): the XOR swizzle of the K / V /
// Rd images makes the addresses non-affine, and left to the compiler they were recomputed from the lane id for every read of
// every tile -- half of the kernel's vector instructions (14 per MFMA; profiles/r02_attn_issue_wait_anatomy.txt), on the issue
// port the two waves of a SIMD share with the MFMAs.  Worked out once instead (DH = 64):
//   rrow[ks]  = r * ROWB + swizzle(2ks + hh, r)        row fragments (ds_read_b128) of the K, V images and the Rd ring
//   tk0       = transposed-read (ds_read_b64_tr_b16) base of the pattern {row = 4hh + q4 (+8), cols 16(gq&1) + 4pp}: the +8 row
//               and the second 32-column half are the immediates 8 ROWB + 64 (1 - e) and 64 e -- the swizzle bits they flip are
//               lane-independent for this pattern (K^T fragments of the dQw product; Rd^T fragments of the phantom blocks)
//   tr3[v][e] = the same for the pattern {row = 8 (gq >> 1) + q4 + 4v} of the completed-block phase (four constants: there the
//               flipped swizzle bits depend on the lane)
// and the phantom-distance loop has a branch-free body for blocks whose 32 x 32 cells are all valid (no per-cell tests, the bf16
// pairs packed once for both the dG store and the MFMA operand).  Worth 2.5 % of the backward once the phantom blocks' dG stores
// were gone (dg_skip_phantom); inside the noise before that, when the store rate of dG set the pace of the phantom loop.
// ---------------------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(512, 1) void relattn_bwd_dq8_kernel(BwdP p) {
    using G = GeoQ<DH>;
    constexpr int KS = G::KS, EB = G::EB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                       // [2][64][DH]   double-buffered: one barrier per tile, loads hidden behind compute
    char* sV = sK + 2 * G::K_BYTES;        // [2][64][DH]
    char* sR = sV + 2 * G::K_BYTES;        // ring [256][DH]
    _Float16* sG = reinterpret_cast<_Float16*>(sR + G::R_BYTES);                  // [4][32][GS] fp16
    bf16_t* sDG = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(sG) + G::G_BYTES);  // [4][32][DGS] bf16

    STAMP_DECL
    const int tid = threadIdx.x;
    // 8 waves: waves w and w + 4 share query group w (32 queries, one LDS skew row per query between them); kbw = 0 takes the
    // first 32 keys of every 64-key tile, kbw = 1 the last 32 -- two waves per SIMD instead of one.
    const int wid = (tid >> 6) & 3, kbw = tid >> 8, l = tid & 63, r = l & 31, hh = l >> 5;
    int bx_, h, b;
    xcd_block(bx_, h, b);
    // longest-first: late query blocks see the most real keys (early ones mostly phantom distances), and they are dispatched
    // first so the tail of the launch is made of short workgroups
    const int i0 = (gridDim.x - 1 - bx_) * QB;
    const int iw0 = i0 + 32 * wid;
    const int T = p.T, M = p.M;
    const int p0 = T - p.Kc;
    // both per-wave buffers use FIXED columns c = distance - dlo in [0, 96): every address is a per-lane constant plus an
    // immediate.  Carry between tiles (block 0 -> block 2): G in registers, dG by an LDS move of the lane's own row.
    _Float16* gW = sG + wid * 32 * GS + r * GS + 4 * hh;
    const _Float16* gR = sG + wid * 32 * GS + r * GS + r + 64 - 4 * hh;
    const uint32_t gRb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(const char*)(gR - 27);
    bf16_t* myDG = sDG + wid * 32 * DGS + r * DGS;                 // row of this lane's query
    const uint32_t dgWb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(const char*)(myDG + r + 64 - 4 * hh - 27);
    f16x4 carry[4];
    // Rd ring fragments: slot = (16-aligned window base + r) & 255, so the XOR-swizzle term of its row depends on the lane only
    int rswz[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) rswz[ks] = (DH == 64) ? (((2 * ks + hh) ^ ((r >> 1) & 7)) << 4) : ((2 * ks + hh) << 4);
    int rrow[KS];           // row-fragment offset of lane row r inside a 32-row group: r * ROWB + swizzled chunk
#pragma unroll
    for (int ks = 0; ks < KS; ks++) rrow[ks] = r * G::ROWB + rswz[ks];
    const int gq_ = l >> 4, q4_ = (l & 15) >> 2, pp_ = l & 3;
    const int tk0 = G::eoff(4 * hh + q4_, 16 * (gq_ & 1) + 4 * pp_);
    int tr3[2][2];
#pragma unroll
    for (int v = 0; v < 2; v++)
#pragma unroll
        for (int e = 0; e < 2; e++) tr3[v][e] = G::eoff(8 * (gq_ >> 1) + q4_ + 4 * v, 32 * e + 16 * (gq_ & 1) + 4 * pp_);

    const bf16_t* kbase = p.k + (size_t)b * p.kv_bs + (size_t)h * DH;
    const bf16_t* vbase = p.v + (size_t)b * p.kv_bs + (size_t)h * DH;
    const bf16_t* rbase = p.rd + (size_t)h * DH;
    const int qi = iw0 + r;
    const bool qok = qi < T;

    bf16x8 qw[KS], qr[KS], dof[KS];
    {
        const size_t qrow = (size_t)(qok ? qi : 0);
        const bf16_t* qp = p.q + (size_t)b * p.q_bs + qrow * p.q_rs + (size_t)h * DH;
        const bf16_t* dop = p.dout + (size_t)b * p.o_bs + qrow * p.o_rs + (size_t)h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int e0 = 16 * ks + 8 * hh;
            const bf16x8 qv = *reinterpret_cast<const bf16x8*>(qp + e0);
            const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dop + e0);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float qf = qok ? bf2f((bf16_t)qv[j]) : 0.f;
                qw[ks][j] = (short)f2bf((qf + p.rwb[h * DH + e0 + j]) * p.scale_log2e);
                qr[ks][j] = (short)f2bf((qf + p.rrb[h * DH + e0 + j]) * p.scale_log2e);
                dof[ks][j] = qok ? (short)f2bf(bf2f((bf16_t)dv[j]) * p.scale) : (short)0;
            }
        }
    }
    const size_t sidx = ((size_t)b * p.H + h) * T + (qok ? qi : 0);
    const float lse2 = qok ? p.lse[sidx] * LOG2E : 0.f;
    const float ndlt = qok ? -p.scale * p.delta[sidx] : 0.f;
    f32x16 c_lse, c_dlt;          // MFMA C operands: -lse (log2 units) and -scale*delta of this lane's query
#pragma unroll
    for (int j = 0; j < 16; j++) { c_lse[j] = -lse2; c_dlt[j] = ndlt; }

    // zero this wave's un-skew ring: never-written cells must read as 0
    {
        uint32_t* z = reinterpret_cast<uint32_t*>(sDG + wid * 32 * DGS);
        if (kbw == 0)
            for (int i = l; i < 32 * DGS / 2; i += 64) z[i] = 0u;
    }

    const int p_lo = i0 - M + 1;
    const int p_hi = min(i0 + QB - 1, T - 1);
    const int kt_lo = floordiv(p_lo, KT), kt_hi = floordiv(p_hi, KT);

    constexpr int NLD8 = (KT * G::CH + 511) / 512;
    u32x4 rk[NLD8], rv[NLD8], rr[NLD8];
    // DH = 64: every thread owns exactly one 16-byte chunk of a tile.  The loads are then UNCONDITIONAL (row clamped) unless the tile
    // reaches below the first stored key: with the `ok ? load : zero` form hipcc zero-fills the destination registers first and, to
    // do that, waits s_waitcnt vmcnt(0) at the top of every tile -- which also waits for the previous tile's dG stores (vmcnt counts
    // stores): each wave then idles until its stores are acknowledged before it even requests the next tile.  Rows past the last
    // key (P + row >= T) may hold any finite value: their scores are masked (distance < 0).
    constexpr bool ONE_CHUNK = (KT * G::CH == 512) && (NLD8 == 1);
    auto load_kv = [&](int kt) {
        const int P = kt * KT;
        if (ONE_CHUNK) {      // always loaded (row clamped); rows below the first stored key are zeroed when the tile is STORED
            const int row = tid / G::CH, ch = tid % G::CH;
            const int srow = max(min(P + row - p0, p.Kc - 1), 0);
            rk[0] = *reinterpret_cast<const u32x4*>(kbase + (size_t)srow * p.kv_rs + ch * 8);
            rv[0] = *reinterpret_cast<const u32x4*>(vbase + (size_t)srow * p.kv_rs + ch * 8);
            return;
        }
#pragma unroll
        for (int n = 0; n < NLD8; n++) {
            const int c = tid + n * 512;
            const int row = c / G::CH, ch = c % G::CH;
            const int srow = P + row - p0;
            u32x4 z = {0u, 0u, 0u, 0u};
            const bool ok = (c < KT * G::CH) && (srow >= 0) && (srow < p.Kc);
            rk[n] = ok ? *reinterpret_cast<const u32x4*>(kbase + (size_t)srow * p.kv_rs + ch * 8) : z;
            rv[n] = ok ? *reinterpret_cast<const u32x4*>(vbase + (size_t)srow * p.kv_rs + ch * 8) : z;
        }
    };
    // `P`: first key position of the tile being stored (rows below p0 are the zero memories: k = v = 0)
    auto store_kv = [&](int buf, int P) {
#pragma unroll
        for (int n = 0; n < NLD8; n++) {
            const int c = tid + n * 512;
            if (c < KT * G::CH) {
                const int row = c / G::CH, ch = c % G::CH;
                u32x4 wk = rk[n], wv = rv[n];
                if (ONE_CHUNK && P + row < p0) { wk = u32x4{0u, 0u, 0u, 0u}; wv = wk; }
                *reinterpret_cast<u32x4*>(sK + buf * G::K_BYTES + G::koff(row, ch)) = wk;
                *reinterpret_cast<u32x4*>(sV + buf * G::K_BYTES + G::koff(row, ch)) = wv;
            }
        }
    };
    auto load_r = [&](int dbase) {
        if (ONE_CHUNK) {
            const int row = tid / G::CH, ch = tid % G::CH;
            int d = dbase + row;
            d = d < 0 ? 0 : (d > M - 1 ? M - 1 : d);
            rr[0] = *reinterpret_cast<const u32x4*>(rbase + (size_t)d * p.rd_rs + ch * 8);
            return;
        }
#pragma unroll
        for (int n = 0; n < NLD8; n++) {
            const int c = tid + n * 512;
            const int row = c / G::CH, ch = c % G::CH;
            int d = dbase + row;
            d = d < 0 ? 0 : (d > M - 1 ? M - 1 : d);
            u32x4 z = {0u, 0u, 0u, 0u};
            rr[n] = (c < KT * G::CH) ? *reinterpret_cast<const u32x4*>(rbase + (size_t)d * p.rd_rs + ch * 8) : z;
        }
    };
    auto store_r = [&](int dbase) {
#pragma unroll
        for (int n = 0; n < NLD8; n++) {
            const int c = tid + n * 512;
            if (c < KT * G::CH) {
                const int row = c / G::CH, ch = c % G::CH;
                *reinterpret_cast<u32x4*>(sR + G::koff((dbase + row) & 255, ch)) = rr[n];
            }
        }
    };

    f32x16 aw[EB];          // (dQw + dQr)^T : [e][query] -- ONE accumulator: see the note above the kernel
#pragma unroll
    for (int e = 0; e < EB; e++)
#pragma unroll
        for (int j = 0; j < 16; j++) aw[e][j] = 0.f;
    bool have_ring = false;
    int cur = 0;
    bf16_t* dgrow = p.dg ? p.dg + (((size_t)b * p.H + h) * T + (qok ? qi : 0)) * (size_t)M : nullptr;
    __syncthreads();   // dG buffer zeroed

    // ---- phantom keys (see relattn_fwd.hip): for key positions below the first stored tile pz, k = v = 0, hence S = 0 and
    // dP = 0: dSr = -scale * P * delta depends only on the distance.  Walk those DISTANCES d in [i - pz + 1, M - 1] block-wise:
    // G^T (no skew) -> P -> dG^T straight from the accumulators -> dQr MFMA + dG store.  No K/V, no S/dP/dQw, no LDS rings.
    // The one block that straddles real and phantom distances (d in [iw0-pz, iw0-pz+31]) is deposited into the un-skew buffer
    // (block-0 columns) so that the first real tile completes and emits it.
    const int pz = floordiv(p0, KT) * KT;
    int kt_start = kt_lo;
    if (kt_lo * KT < pz) {
        kt_start = pz / KT;
        const int gq = l >> 4, li = l & 15, q4 = li >> 2, pp = li & 3;
        load_r(i0 - pz);
        // with the forward's phantom value-sum the walk ends with the 256-distance block that holds the last query's first
        // phantom distance: everything above it is an all-phantom block for every wave of the workgroup
        const int db_end = p.oph ? min(M - 1, (i0 + QB - 1 - pz) | 255) : M - 1;
#pragma unroll 1
        for (int db = i0 - pz; db <= db_end; db += 64) {
            store_r(db);
            __syncthreads();
            if (db + 64 <= db_end) load_r(db + 64);
            if (iw0 < T) {
                for (int gb = kbw; gb == kbw; gb += 2) {                  // each wave of the pair takes one 32-distance block
                    const int dblk = db + 32 * gb;
                    if (dblk + 31 <= iw0 - pz || dblk > M - 1) continue;   // wave-uniform
                    if (p.oph && (dblk & ~255) > iw0 + 31 - pz) continue;  // all-phantom 256-block: the epilogue's oph term
                    f32x16 g = c_lse;
                    const char* rb = sR + (dblk & 255) * G::ROWB;      // ring rows of this block: (dblk & 255) + 0..31, no wrap inside
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(rb + rrow[ks]);
                        g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                    __builtin_bit_cast(mfma_bf16x8, qr[ks]), g, 0, 0, 0);
                    }
                    // dSr = -scale * P * delta.  Only boundary blocks need the per-cell validity test (scalar branch).
                    const bool fullblk = __builtin_amdgcn_readfirstlane(
                        (int)((dblk >= iw0 + 31 - pz + 1) && (dblk + 31 <= M - 1) && (iw0 + 31 < T))) != 0;
                    if (DH == 64 && fullblk && dgrow && (M & 7) == 0) {
                        // every cell of the block is a phantom distance of a valid query (the block is above the straddling one by
                        // construction): no tests, one packing for the dG store and the MFMA operand, transposed Rd reads at
                        // tk0 + immediates
#pragma unroll
                        for (int j = 0; j < 16; j++) g[j] = __builtin_amdgcn_exp2f(g[j]) * ndlt;
                        uint32_t w[8];
#pragma unroll
                        for (int m = 0; m < 8; m++) w[m] = pack2bf(g[2 * m], g[2 * m + 1]);
                        if (!(p.dg_skip_phantom && ((dblk & ~255) > iw0 + 31 - pz))) {
#pragma unroll
                            for (int gp = 0; gp < 2; gp++) {
                                const auto r0 = __builtin_amdgcn_permlane32_swap(w[4 * gp], w[4 * gp + 2], false, false);
                                const auto r1 = __builtin_amdgcn_permlane32_swap(w[4 * gp + 1], w[4 * gp + 3], false, false);
                                *reinterpret_cast<u32x4*>(dgrow + dblk + 16 * gp + 8 * hh) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                            }
                        }
#pragma unroll
                        for (int st = 0; st < 2; st++) {
                            const u32x4 pw = {w[4 * st], w[4 * st + 1], w[4 * st + 2], w[4 * st + 3]};
                            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
                            for (int e = 0; e < EB; e++) {
                                const char* a0 = rb + tk0 + 16 * st * G::ROWB;
                                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(a0 + 64 * e));
                                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(a0 + 8 * G::ROWB + 64 * (1 - e)));
                                const bf16x8 a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                                aw[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                                __builtin_bit_cast(mfma_bf16x8, pf), aw[e], 0, 0, 0);
                            }
                        }
                        continue;
                    }
                    if (fullblk) {
#pragma unroll
                        for (int j = 0; j < 16; j++) g[j] = __builtin_amdgcn_exp2f(g[j]) * ndlt;
                    } else {
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const int d = dblk + (j & 3) + 8 * (j >> 2) + 4 * hh;
                            const bool valid = (d >= qi - pz + 1) && (d <= M - 1) && qok;
                            g[j] = valid ? __builtin_amdgcn_exp2f(g[j]) * ndlt : 0.f;
                        }
                    }
                    if (dblk == iw0 - pz) {
                        // straddling block: into block-0 columns of the un-skew buffer (4 consecutive distances per store)
#pragma unroll
                        for (int grp = 0; grp < 4; grp++) {
                            const u32x2 w = {pack2bf(g[4 * grp], g[4 * grp + 1]), pack2bf(g[4 * grp + 2], g[4 * grp + 3])};
                            *reinterpret_cast<u32x2*>(myDG + 8 * grp + 4 * hh) = w;
                        }
                    } else {
                        const bool skipdg = p.dg_skip_phantom && ((dblk & ~255) > iw0 + 31 - pz);
                        if (skipdg) {
                        } else if (dgrow && (M & 7) == 0) {
                            // lanes l and l + 32 (same query, distance groups 4 apart) trade one packed quad each, so that every
                            // lane stores 8 consecutive distances with one 16-byte store instead of two 8-byte ones (the store
                            // path's cost is per instruction and per row segment: scripts/ubench/stores.hip)
#pragma unroll
                            for (int gp = 0; gp < 2; gp++) {
                                const unsigned a0 = pack2bf(g[8 * gp], g[8 * gp + 1]), a1 = pack2bf(g[8 * gp + 2], g[8 * gp + 3]);
                                const unsigned b0 = pack2bf(g[8 * gp + 4], g[8 * gp + 5]), b1 = pack2bf(g[8 * gp + 6], g[8 * gp + 7]);
                                const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                                const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                                const int d8 = dblk + 16 * gp + 8 * hh;
                                if (qok && d8 + 7 <= M - 1) *reinterpret_cast<u32x4*>(dgrow + d8) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                            }
                        } else if (dgrow && qok) {
#pragma unroll
                            for (int grp = 0; grp < 4; grp++) {
                                const int d4 = dblk + 8 * grp + 4 * hh;
                                if (d4 + 3 <= M - 1) {
                                    const u32x2 w = {pack2bf(g[4 * grp], g[4 * grp + 1]), pack2bf(g[4 * grp + 2], g[4 * grp + 3])};
                                    *reinterpret_cast<u32x2*>(dgrow + d4) = w;
                                }
                            }
                        }
#pragma unroll
                        for (int st = 0; st < 2; st++) {
                            const u32x4 pw = {pack2bf(g[8 * st], g[8 * st + 1]), pack2bf(g[8 * st + 2], g[8 * st + 3]),
                                              pack2bf(g[8 * st + 4], g[8 * st + 5]), pack2bf(g[8 * st + 6], g[8 * st + 7])};
                            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
                            for (int e = 0; e < EB; e++) {
                                const int dist = dblk + 16 * st + 4 * hh + q4;      // accumulator-permuted k order
                                const int ecol = 32 * e + 16 * (gq & 1) + 4 * pp;
                                bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
                                if (ecol < DH) {
                                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sR + G::eoff(dist & 255, ecol)));
                                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sR + G::eoff((dist + 8) & 255, ecol)));
                                    a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                                }
                                aw[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                                __builtin_bit_cast(mfma_bf16x8, pf), aw[e], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }

    STAMP(10)
    {
        const int P0 = kt_start * KT;
        load_kv(kt_start);
        store_kv(0, P0);
#pragma unroll 1
        for (int q4 = 0; q4 < 3; q4++) {
            const int dbase = i0 - P0 - 64 + 64 * q4;
            load_r(dbase);
            store_r(dbase);
        }
    }
    __syncthreads();

    // activity of this query group for the tile at key position P (the same for both waves of a pair)
    auto is_active = [&](int P) { return (iw0 + 31 - P >= 0) && (iw0 - P - (KT - 1) <= M - 1) && (iw0 < T); };
    // "pre-phase" of the tile at P: the G blocks of its 96-column window.  Wave 1 (keys 32..63: columns 0..63) computes block 0 and
    // carries it as the next tile's block 2; wave 0 (keys 0..31: columns 32..95) computes block 1 and moves the un-skew buffer's
    // carried block (columns [0,32) of the previous tile are columns [64,96) of this one).  It runs in the same barrier interval
    // as the previous tile's completed-block phase (two barriers per tile instead of three).
    auto pre_phase = [&](int P) {
        const int dlo = iw0 - P - 64;
        const bool active = is_active(P);
        if (active) {
        auto gblock = [&](int gb, f16x4 (&dst)[4]) {
            f32x16 g;
#pragma unroll
            for (int j = 0; j < 16; j++) g[j] = 0.f;
            const char* rb = sR + ((dlo + 32 * gb) & 255) * G::ROWB;      // a multiple of 32 rows: + r does not wrap
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(rb + rrow[ks]);
                g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                            __builtin_bit_cast(mfma_bf16x8, qr[ks]), g, 0, 0, 0);
            }
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                const f32x4v v4 = {g[4 * grp], g[4 * grp + 1], g[4 * grp + 2], g[4 * grp + 3]};
                dst[grp] = __builtin_convertvector(v4, f16x4);
            }
        };
        if (kbw == 1) {
            f16x4 b0[4];
            if (!have_ring) gblock(2, carry);
            gblock(0, b0);
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                *reinterpret_cast<f16x4*>(gW + 8 * grp) = b0[grp];
                *reinterpret_cast<f16x4*>(gW + 64 + 8 * grp) = carry[grp];
                carry[grp] = b0[grp];
            }
        } else {
            f16x4 b1[4];
            gblock(1, b1);
            const u32x4 m0 = *reinterpret_cast<const u32x4*>(myDG + 16 * hh);
            const u32x4 m1 = *reinterpret_cast<const u32x4*>(myDG + 16 * hh + 8);
            *reinterpret_cast<u32x4*>(myDG + 64 + 16 * hh) = m0;
            *reinterpret_cast<u32x4*>(myDG + 64 + 16 * hh + 8) = m1;
#pragma unroll
            for (int grp = 0; grp < 4; grp++) *reinterpret_cast<f16x4*>(gW + 32 + 8 * grp) = b1[grp];
        }
        have_ring = true;
    }
    };
    pre_phase(kt_start * KT);
    __syncthreads();
    STAMP(11)

#pragma unroll 1
    for (int kt = kt_start; kt <= kt_hi; kt++) {
        const int P = kt * KT;
        const bool more = kt < kt_hi;
        if (more) {
            load_kv(kt + 1);
            load_r(i0 - (P + KT) - 64);
        }
        const int dmin_w = iw0 - P - (KT - 1), dmax_w = iw0 + 31 - P;
        const bool active = (dmax_w >= 0) && (dmin_w <= M - 1) && (iw0 < T);       // same for both waves of a pair
        const int dlo = iw0 - P - 64;
        const char* cK = sK + cur * G::K_BYTES;
        const char* cV = sV + cur * G::K_BYTES;
        const int gq = l >> 4, li = l & 15, q4 = li >> 2, pp = li & 3;
        STAMP(0)
        // ---- phase 2: this wave's 32 keys: S / dP chains, skew read, P and dSr, skew write, dQw products
        if (active) {
            const bool full = __builtin_amdgcn_readfirstlane((int)((dmin_w >= 0) && (dmax_w <= M - 1) && (iw0 + 31 < T))) != 0;
            const int kb = kbw;
            f32x16 s = c_lse, dp = c_dlt;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(cK + 32 * kb * G::ROWB + rrow[ks]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                            __builtin_bit_cast(mfma_bf16x8, qw[ks]), s, 0, 0, 0);
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(cV + 32 * kb * G::ROWB + rrow[ks]);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, av),
                                                             __builtin_bit_cast(mfma_bf16x8, dof[ks]), dp, 0, 0, 0);
            }
            STAMP(1)
            uint32_t bdu[16];
            skew_read16(gRb - 64 * kb, bdu);
            STAMP(2)
            if (full) {
#pragma unroll
                for (int j = 0; j < 16; j++) s[j] = __builtin_amdgcn_exp2f(add_f16(s[j], bdu[j])) * dp[j];
            } else {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int d = qi - P - (32 * kb + (j & 3) + 8 * (j >> 2) + 4 * hh);
                    const bool valid = (d >= 0) && (d <= M - 1) && qok;
                    const float pv = __builtin_amdgcn_exp2f(add_f16(s[j], bdu[j]));
                    s[j] = valid ? pv * dp[j] : 0.f;
                }
            }
            uint32_t dsw[8];          // dSr as bf16 pairs (2m, 2m+1): MFMA operand and skew-write source
#pragma unroll
            for (int m = 0; m < 8; m++) dsw[m] = pack2bf(s[2 * m], s[2 * m + 1]);
            skew_write16p(dgWb - 64 * kb, dsw);
            STAMP(3)
#pragma unroll
            for (int st = 0; st < 2; st++) {
                const u32x4 pw = {dsw[4 * st], dsw[4 * st + 1], dsw[4 * st + 2], dsw[4 * st + 3]};
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
                for (int e = 0; e < EB; e++) {
                    const int key = 32 * kb + 16 * st + 4 * hh + q4;
                    const int ecol = 32 * e + 16 * (gq & 1) + 4 * pp;
                    bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (DH == 64) {
                        const char* a0 = cK + (32 * kb + 16 * st) * G::ROWB + tk0;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(a0 + 64 * e));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(a0 + 8 * G::ROWB + 64 * (1 - e)));
                        a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    } else if (ecol < DH) {
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(cK + G::eoff(key, ecol)));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(cK + G::eoff(key + 8, ecol)));
                        a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                    aw[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                    __builtin_bit_cast(mfma_bf16x8, pf), aw[e], 0, 0, 0);
                }
            }
        }
        STAMP(4)
        // the next tile's K/V image and ring rows go to LDS before the barrier: the ring slots they overwrite lie above this tile's
        // window, and the next tile's G blocks (below) need the new rows
        if (more) {
            store_kv(cur ^ 1, P + KT);
            store_r(i0 - (P + KT) - 64);
        }
        STAMP(5)
        __syncthreads();
        STAMP(6)
        // ---- phase 3: completed distance blocks of the window: wave 0 emits block 2 (d in [dlo+64, dlo+95]), wave 1 block 1
        if (active) {
            const int blk = 2 - kbw;
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ks2++) {
                const int d8 = dlo + 32 * blk + 16 * ks2 + 8 * hh;  // 8 consecutive distances (natural k order)
                const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(myDG + 32 * blk + 16 * ks2 + 8 * hh);
                if (dgrow && qok && d8 >= 0 && d8 + 7 <= M - 1)
                    *reinterpret_cast<bf16x8*>(dgrow + d8) = bfrag;
#pragma unroll
                for (int e = 0; e < EB; e++) {
                    const int dist = dlo + 32 * blk + 16 * ks2 + 8 * (gq >> 1) + q4;
                    const int ecol = 32 * e + 16 * (gq & 1) + 4 * pp;
                    bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (DH == 64) {
                        const char* rb3 = sR + ((dlo + 32 * blk + 16 * ks2) & 255) * G::ROWB;     // a multiple of 16 rows: + 0..15 does not wrap
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(rb3 + tr3[0][e]));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(rb3 + tr3[1][e]));
                        a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    } else if (ecol < DH) {
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sR + G::eoff(dist & 255, ecol)));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sR + G::eoff((dist + 4) & 255, ecol)));
                        a = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                    aw[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                                    __builtin_bit_cast(mfma_bf16x8, bfrag), aw[e], 0, 0, 0);
                }
            }
        }
        STAMP(7)
        if (more) pre_phase(P + KT);
        STAMP(8)
        __syncthreads();
        STAMP(9)
        cur ^= 1;
    }

    STAMP_FLUSH
    // ---- the skipped all-phantom blocks: dQr_i += -scale * delta_i * 2^(mph_i - lse2_i) * oph_i (one wave of the pair adds it)
    if (p.oph && kbw == 0 && qok) {
        const float f = ndlt * __builtin_amdgcn_exp2f(p.mph[sidx] - lse2);
        const bf16_t* op = p.oph + (size_t)b * p.o_bs + (size_t)qi * p.o_rs + (size_t)h * DH;
#pragma unroll
        for (int e = 0; e < EB; e++) {
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                const int e0 = 32 * e + 8 * grp + 4 * hh;
                if (e0 < DH) {
                    const u32x2 w = *reinterpret_cast<const u32x2*>(op + e0);
                    aw[e][4 * grp] += f * bf2f((bf16_t)(w[0] & 0xffffu));
                    aw[e][4 * grp + 1] += f * bf2f((bf16_t)(w[0] >> 16));
                    aw[e][4 * grp + 2] += f * bf2f((bf16_t)(w[1] & 0xffffu));
                    aw[e][4 * grp + 3] += f * bf2f((bf16_t)(w[1] >> 16));
                }
            }
        }
    }
    // ---- epilogue.  The column sums of the wave's accumulator over its queries are its part of d(r_w_bias) + d(r_r_bias): they
    // go to d_rwb; the dRd kernel moves the r_r_bias part over (colsum(dG) . Rd).  dq = the two waves' accumulators, summed
    // through LDS (the K / V images are dead now).
    __syncthreads();                                       // the K / V images are dead: reuse them
    float* sred = reinterpret_cast<float*>(smem) + 4 * EB * 16 * 64;     // [8 waves][64], behind the dq exchange area below
#pragma unroll
    for (int e = 0; e < EB; e++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            float a = qok ? aw[e][j] : 0.f;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
            const int ee = 32 * e + (j & 3) + 8 * (j >> 2) + 4 * hh;
            if (r == 0 && ee < DH) sred[(kbw * 4 + wid) * 64 + ee] = a;
        }
    }
    float* red = reinterpret_cast<float*>(smem);          // [4 groups][EB * 16][64 lanes] f32 <= 32 KB
    if (kbw == 1) {
#pragma unroll
        for (int e = 0; e < EB; e++)
#pragma unroll
            for (int j = 0; j < 16; j++) red[(wid * EB * 16 + e * 16 + j) * 64 + l] = aw[e][j];
    }
    __syncthreads();
    if (tid < DH) {          // one atomic instruction per workgroup (lane = e): see the 4-wave kernel's epilogue
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; w++) t += sred[w * 64 + tid];
        atomicAdd(p.d_rwb + h * DH + tid, t);
    }
    if (kbw == 0 && qok) {
        bf16_t* dqp = p.dq + (size_t)b * p.dq_bs + (size_t)qi * p.dq_rs + (size_t)h * DH;
#pragma unroll
        for (int e = 0; e < EB; e++) {
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                const int e0 = 32 * e + 8 * grp + 4 * hh;
                if (e0 < DH) {
                    float o[4];
#pragma unroll
                    for (int t = 0; t < 4; t++) o[t] = aw[e][4 * grp + t] + red[(wid * EB * 16 + e * 16 + 4 * grp + t) * 64 + l];
                    u32x2 w = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                    *reinterpret_cast<u32x2*>(dqp + e0) = w;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// key-owner kernel: wave owns 32 keys (lane = key), workgroup 128 keys; streams 32-query tiles
// ---------------------------------------------------------------------------------------------------------------
constexpr int KB = 128, QT = 32;
constexpr int SKS = 68;  // fp16 skew row stride (64 distance columns + pad): 136-byte rows, 8-byte aligned groups

template <int DH> struct GeoK {
    static constexpr int KS = DH / 16;
    static constexpr int EB = (DH + 31) / 32;
    static constexpr int ROWB = DH * 2;
    static constexpr int CH = DH / 8;
    static constexpr int Q_BYTES = QT * ROWB;          // one 32-row tile image
    static constexpr int R_BYTES = 256 * ROWB;
    static constexpr int S_BYTES = 4 * 32 * SKS * 2;
    static constexpr int QSET = 3 * Q_BYTES + 2 * QT * 4;       // one query tile: Qw, Qr, dO images + lse, delta
    static constexpr int SMEM = 2 * QSET + R_BYTES + S_BYTES;   // two query tiles (double-buffered)
    __device__ static __forceinline__ int koff(int row, int ch) {
        if (DH == 64) return row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4);
        return row * ROWB + (ch << 4);
    }
    __device__ static __forceinline__ int eoff(int row, int e) { return koff(row, e >> 3) + ((e & 7) << 1); }
};

// key-owner skew read: register j belongs to query ii = pat(j) + 4*hh and needs column ii - r + 32 of row ii, i.e. the
// per-lane base plus pat(j) * (SKS + 1) elements: 16 genuine ds_read_u16 with immediate offsets (see relattn_fwd.hip)
__device__ __forceinline__ void skew_read16k(uint32_t base, uint32_t (&u)[16]) {
    static_assert(SKS == 68, "offsets below are pat(j) * (SKS + 1) * 2 bytes");
    asm volatile(
        "ds_read_u16 %0, %16\n\t"              "ds_read_u16 %1, %16 offset:138\n\t"   "ds_read_u16 %2, %16 offset:276\n\t"
        "ds_read_u16 %3, %16 offset:414\n\t"   "ds_read_u16 %4, %16 offset:1104\n\t"  "ds_read_u16 %5, %16 offset:1242\n\t"
        "ds_read_u16 %6, %16 offset:1380\n\t"  "ds_read_u16 %7, %16 offset:1518\n\t"  "ds_read_u16 %8, %16 offset:2208\n\t"
        "ds_read_u16 %9, %16 offset:2346\n\t"  "ds_read_u16 %10, %16 offset:2484\n\t" "ds_read_u16 %11, %16 offset:2622\n\t"
        "ds_read_u16 %12, %16 offset:3312\n\t" "ds_read_u16 %13, %16 offset:3450\n\t" "ds_read_u16 %14, %16 offset:3588\n\t"
        "ds_read_u16 %15, %16 offset:3726\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7]),
          "=&v"(u[8]), "=&v"(u[9]), "=&v"(u[10]), "=&v"(u[11]), "=&v"(u[12]), "=&v"(u[13]), "=&v"(u[14]), "=&v"(u[15])
        : "v"(base)
        : "memory");
}

template <int DH>
__global__ __launch_bounds__(256, 2) void relattn_bwd_dkv_kernel(BwdP p) {
    using G = GeoK<DH>;
    constexpr int KS = G::KS, EB = G::EB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // query tile set c (0 / 1, double-buffered: the next tile is stored while this one is read, one barrier per tile):
    //   Qw, Qr, dO images, then -lse (log2 units) [32] and -scale * delta [32]
    char* sQ0 = smem;
    char* sR = smem + 2 * G::QSET;                              // ring [256][DH]
    _Float16* sS = reinterpret_cast<_Float16*>(sR + G::R_BYTES);   // [4][32 queries][SKS] fp16: G rows, lane-private per wave

    const int tid = threadIdx.x;
    const int wid = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
    int bx_, h, b;
    xcd_block(bx_, h, b);
    const int T = p.T, M = p.M;
    const int p0 = T - p.Kc;
    const int P0 = p0 + bx_ * KB;   // first key position of the workgroup
    const int Pw = P0 + 32 * wid;
    const int pk = Pw + r;                 // this lane's key position
    const bool kok = pk < T;               // pk >= p0 always
    _Float16* myS = sS + wid * 32 * SKS;
    _Float16* sW = myS + r * SKS + 4 * hh;                                  // G^T column of query r: row r, 4 distances per store
    const uint32_t sRb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(const char*)
                         (myS + 4 * hh * (SKS + 1) + 32 - r);

    const bf16_t* qbase = p.q + (size_t)b * p.q_bs + (size_t)h * DH;
    const bf16_t* dobase = p.dout + (size_t)b * p.o_bs + (size_t)h * DH;
    const bf16_t* rbase = p.rd + (size_t)h * DH;

    // K, V fragments (B operands: lane = key, k = 16ks + 8hh + j)
    bf16x8 kf[KS], vf[KS];
    {
        const size_t srow = (size_t)(kok ? pk - p0 : 0);
        const bf16_t* kp = p.k + (size_t)b * p.kv_bs + srow * p.kv_rs + (size_t)h * DH;
        const bf16_t* vp = p.v + (size_t)b * p.kv_bs + srow * p.kv_rs + (size_t)h * DH;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int e0 = 16 * ks + 8 * hh;
            const bf16x8 kv = *reinterpret_cast<const bf16x8*>(kp + e0);
            const bf16x8 vv = *reinterpret_cast<const bf16x8*>(vp + e0);
#pragma unroll
            for (int j = 0; j < 8; j++) { kf[ks][j] = kok ? kv[j] : (short)0; vf[ks][j] = kok ? vv[j] : (short)0; }
        }
    }

    // queries that can see any key of this workgroup: i in [P0, P0 + KB - 1 + M - 1], clipped to [0, T)
    const int i_lo = max(P0, 0), i_hi = min(P0 + KB - 1 + M - 1, T - 1);
    const int it_lo = i_lo / QT, it_hi = i_hi / QT;

    // staging: thread t < 32*CH handles one 16-byte chunk of each of the three tiles
    u32x4 tq, tdo, rr;
    float tl = 0.f, tdl = 0.f;
    bool tok = false;
    auto load_q = [&](int it) {
        const int I = it * QT;
        const int row = tid / G::CH, ch = tid % G::CH;
        u32x4 z = {0u, 0u, 0u, 0u};
        const bool ok = (tid < QT * G::CH) && (I + row < T);
        tq = ok ? *reinterpret_cast<const u32x4*>(qbase + (size_t)(I + row) * p.q_rs + ch * 8) : z;
        tdo = ok ? *reinterpret_cast<const u32x4*>(dobase + (size_t)(I + row) * p.o_rs + ch * 8) : z;
        if (tid < QT) {
            // RAW values, from a clamped index; scaled and masked when they are stored (store_q).  Scaling them here puts an
            // s_waitcnt vmcnt(0) right behind the loads -- at the top of every tile wave 0 then waited for these AND for the
            // tile's Q / dO rows it had just requested (the wait is a drain), and the other waves waited for wave 0 at the barrier:
            // the loads' latency was exposed once per tile instead of hidden behind the tile's products
            const size_t sidx = ((size_t)b * p.H + h) * T + min(I + tid, T - 1);
            tl = p.lse[sidx];
            tdl = p.delta[sidx];
            tok = I + tid < T;
        }
    };
    auto store_q = [&](int buf) {
        char* sQw = sQ0 + buf * G::QSET;
        char* sQr = sQw + G::Q_BYTES;
        char* sDO = sQr + G::Q_BYTES;
        float* sLse = reinterpret_cast<float*>(sDO + G::Q_BYTES);
        float* sDl = sLse + QT;
        if (tid < QT * G::CH) {
            const int row = tid / G::CH, ch = tid % G::CH;
            u32x4 w, rq, wd;
            const bf16_t* src = reinterpret_cast<const bf16_t*>(&tq);
            const bf16_t* sdo = reinterpret_cast<const bf16_t*>(&tdo);
            bf16_t* dw = reinterpret_cast<bf16_t*>(&w);
            bf16_t* dr = reinterpret_cast<bf16_t*>(&rq);
            bf16_t* dd = reinterpret_cast<bf16_t*>(&wd);
#pragma unroll
            for (int j = 0; j < 8; j++) {     // operand scaling: see the note above relattn_bwd_dq_kernel
                const float qf = bf2f(src[j]);
                dw[j] = f2bf((qf + p.rwb[h * DH + ch * 8 + j]) * p.scale_log2e);
                dr[j] = f2bf((qf + p.rrb[h * DH + ch * 8 + j]) * p.scale_log2e);
                dd[j] = f2bf(bf2f(sdo[j]) * p.scale);
            }
            *reinterpret_cast<u32x4*>(sQw + G::koff(row, ch)) = w;
            *reinterpret_cast<u32x4*>(sQr + G::koff(row, ch)) = rq;
            *reinterpret_cast<u32x4*>(sDO + G::koff(row, ch)) = wd;
        }
        if (tid < QT) { sLse[tid] = tok ? -tl * LOG2E : 0.f; sDl[tid] = tok ? -p.scale * tdl : 0.f; }
    };
    // 32 Rd rows [dbase, dbase+32)
    auto load_r = [&](int dbase) {
        const int row = tid / G::CH, ch = tid % G::CH;
        int d = dbase + row;
        d = d < 0 ? 0 : (d > M - 1 ? M - 1 : d);
        u32x4 z = {0u, 0u, 0u, 0u};
        rr = (tid < 32 * G::CH) ? *reinterpret_cast<const u32x4*>(rbase + (size_t)d * p.rd_rs + ch * 8) : z;
    };
    auto store_r = [&](int dbase) {
        if (tid < 32 * G::CH) {
            const int row = tid / G::CH, ch = tid % G::CH;
            *reinterpret_cast<u32x4*>(sR + G::koff((dbase + row) & 255, ch)) = rr;
        }
    };

    // prologue: first query tile and its distance window [I - P0 - 128, I - P0 + 31]
    {
        const int I = it_lo * QT;
        load_q(it_lo);
        store_q(0);
#pragma unroll 1
        for (int c5 = 0; c5 < 5; c5++) {
            const int dbase = I - P0 - 128 + 32 * c5;
            load_r(dbase);
            store_r(dbase);
        }
    }
    __syncthreads();

    f32x16 ak[EB], av[EB];  // dK^T, dV^T : [e][key]
#pragma unroll
    for (int e = 0; e < EB; e++)
#pragma unroll
        for (int j = 0; j < 16; j++) { ak[e][j] = 0.f; av[e][j] = 0.f; }

    int cur = 0;
#pragma unroll 1
    for (int it = it_lo; it <= it_hi; it++) {
        const int I = it * QT;
        const bool more = it < it_hi;
        const char* sQw = sQ0 + cur * G::QSET;
        const char* sQr = sQw + G::Q_BYTES;
        const char* sDO = sQr + G::Q_BYTES;
        const float* sLse = reinterpret_cast<const float*>(sDO + G::Q_BYTES);
        const float* sDl = sLse + QT;
        if (more) {
            load_q(it + 1);
            load_r(I + QT - P0);  // next window's 32 new (highest) distances: [I+32-P0, I+63-P0]
        }
        const int dmin_w = I - Pw - 31, dmax_w = I + 31 - Pw;
        const bool active = (dmax_w >= 0) && (dmin_w <= M - 1) && (Pw < T);
        if (active) {
            const int dlo = I - Pw - 32;
            // S = Qw . K^T - lse and dP = scale * (dO . V^T - delta)  (rows = queries, lane = key); the per-query constants
            // enter as the C operand (register j <-> query (j&3) + 8*(j>>2) + 4*hh: four 16-byte LDS reads each)
            f32x16 s, dp;
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                const f32x4 cl = *reinterpret_cast<const f32x4*>(sLse + 8 * grp + 4 * hh);
                const f32x4 cd = *reinterpret_cast<const f32x4*>(sDl + 8 * grp + 4 * hh);
#pragma unroll
                for (int t = 0; t < 4; t++) { s[4 * grp + t] = cl[t]; dp[4 * grp + t] = cd[t]; }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(sQw + G::koff(r, 2 * ks + hh));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                            __builtin_bit_cast(mfma_bf16x8, kf[ks]), s, 0, 0, 0);
                const bf16x8 ad = *reinterpret_cast<const bf16x8*>(sDO + G::koff(r, 2 * ks + hh));
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, ad),
                                                             __builtin_bit_cast(mfma_bf16x8, vf[ks]), dp, 0, 0, 0);
            }
            // G^T = Rd . Qr^T over the 64-distance window (lane = query, registers = distances): the lane writes its query's
            // row of the fp16 skew buffer (4 consecutive distances per store); key lane r then reads column ii - r + 32 of
            // row ii
#pragma unroll
            for (int gb = 0; gb < 2; gb++) {
                f32x16 g;
#pragma unroll
                for (int j = 0; j < 16; j++) g[j] = 0.f;
                const int slot = (dlo + 32 * gb + r) & 255;
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const bf16x8 bb = *reinterpret_cast<const bf16x8*>(sR + G::koff(slot, 2 * ks + hh));
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(sQr + G::koff(r, 2 * ks + hh));
                    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, bb),
                                                                __builtin_bit_cast(mfma_bf16x8, a), g, 0, 0, 0);
                }
#pragma unroll
                for (int grp = 0; grp < 4; grp++) {
                    const f32x4v v4 = {g[4 * grp], g[4 * grp + 1], g[4 * grp + 2], g[4 * grp + 3]};
                    *reinterpret_cast<f16x4*>(sW + 32 * gb + 8 * grp) = __builtin_convertvector(v4, f16x4);
                }
            }
            const bool full = __builtin_amdgcn_readfirstlane((int)((dmin_w >= 0) && (dmax_w <= M - 1) && (I + 31 < T))) != 0;
            f32x16 pr;
            uint32_t bdu[16];
            skew_read16k(sRb, bdu);
            auto grads = [&](auto masked) {
                constexpr bool MASKED = decltype(masked)::value;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    float pv = __builtin_amdgcn_exp2f(add_f16(s[j], bdu[j]));
                    if (MASKED) {
                        const int ii = (j & 3) + 8 * (j >> 2) + 4 * hh;
                        const int d = I + ii - pk;             // = dlo + (ii - r + 32)
                        const bool valid = (d >= 0) && (d <= M - 1) && (I + ii < T);
                        pv = valid ? pv : 0.f;
                    }
                    pr[j] = pv;
                    s[j] = pv * dp[j];
                }
            };
            if (full) grads(std::false_type{}); else grads(std::true_type{});
            // dV^T += dO^T . P ; dK^T += Qw^T . dSr   (A through transposed reads, accumulator-permuted k order)
            const int gq = l >> 4, li = l & 15, q4 = li >> 2, pp = li & 3;
#pragma unroll
            for (int st = 0; st < 2; st++) {
                const u32x4 pw = {pack2bf(pr[8 * st], pr[8 * st + 1]), pack2bf(pr[8 * st + 2], pr[8 * st + 3]),
                                  pack2bf(pr[8 * st + 4], pr[8 * st + 5]), pack2bf(pr[8 * st + 6], pr[8 * st + 7])};
                const u32x4 dw = {pack2bf(s[8 * st], s[8 * st + 1]), pack2bf(s[8 * st + 2], s[8 * st + 3]),
                                  pack2bf(s[8 * st + 4], s[8 * st + 5]), pack2bf(s[8 * st + 6], s[8 * st + 7])};
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pw), df = __builtin_bit_cast(bf16x8, dw);
#pragma unroll
                for (int e = 0; e < EB; e++) {
                    const int qrow = 16 * st + 4 * hh + q4;
                    const int ecol = 32 * e + 16 * (gq & 1) + 4 * pp;
                    bf16x8 a1 = {0, 0, 0, 0, 0, 0, 0, 0}, a2 = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (ecol < DH) {
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sDO + G::eoff(qrow, ecol)));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sDO + G::eoff(qrow + 8, ecol)));
                        a1 = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        const bf16x4 lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sQw + G::eoff(qrow, ecol)));
                        const bf16x4 hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sQw + G::eoff(qrow + 8, ecol)));
                        a2 = bf16x8{lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
                    }
                    av[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a1),
                                                                    __builtin_bit_cast(mfma_bf16x8, pf), av[e], 0, 0, 0);
                    ak[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a2),
                                                                    __builtin_bit_cast(mfma_bf16x8, df), ak[e], 0, 0, 0);
                }
            }
        }
        // the next tile goes into the other tile set, its 32 new ring rows into slots below this tile's window: one barrier
        if (more) {
            store_q(cur ^ 1);
            store_r(I + QT - P0);
        }
        __syncthreads();
        cur ^= 1;
    }

    // undo the operand scaling: dK was accumulated against scale*log2(e)*Qw, dV against scale*dO
    {
        const float fk = 1.f / p.scale_log2e, fv = 1.f / p.scale;
#pragma unroll
        for (int e = 0; e < EB; e++)
#pragma unroll
            for (int j = 0; j < 16; j++) { ak[e][j] *= fk; av[e][j] *= fv; }
    }
    if (kok) {
        const size_t srow = (size_t)(pk - p0);
        bf16_t* dkp = p.dk + (size_t)b * p.dkv_bs + srow * p.dkv_rs + (size_t)h * DH;
        bf16_t* dvp = p.dv + (size_t)b * p.dkv_bs + srow * p.dkv_rs + (size_t)h * DH;
#pragma unroll
        for (int e = 0; e < EB; e++) {
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                const int e0 = 32 * e + 8 * grp + 4 * hh;
                if (e0 < DH) {
                    u32x2 wk = {pack2bf(ak[e][4 * grp], ak[e][4 * grp + 1]), pack2bf(ak[e][4 * grp + 2], ak[e][4 * grp + 3])};
                    u32x2 wv = {pack2bf(av[e][4 * grp], av[e][4 * grp + 1]), pack2bf(av[e][4 * grp + 2], av[e][4 * grp + 3])};
                    *reinterpret_cast<u32x2*>(dkp + e0) = wk;
                    *reinterpret_cast<u32x2*>(dvp + e0) = wv;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// d Rd[delta, h, :] += sum_{b,i} dG[b,h,i,delta] * Qr[b,i,h,:]       (Qr = q + r_r_bias, bf16)
// An HBM-streaming contraction: 1.6 GB of dG per layer at C3 against 0.1 GFLOP per MB, so the kernel is a DMA ring with a
// few MFMAs attached.  Workgroup = (256 distances, head, batch group): the wide distance tile keeps the re-reads of Qr (once
// per distance tile) at a quarter of the dG bytes.  Per step one [32 i][256 delta] tile of dG and one [32 i][64 e] tile of
// Qr (both contracted over their ROW index) go HBM -> LDS by global_load_lds into a four-stage ring (three steps in flight
// across one raw barrier per step, counted vmcnt); fragments by transposed LDS reads (32-byte blocks XOR-swizzled on the
// source side of the DMA).  80 KB LDS: two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------
struct DrdP {
    const bf16_t* dg; const bf16_t* qr; float* drd;
    int B, T, H, M, bgroup;            // bgroup = batches per workgroup
    long long qr_bs; int qr_rs, drd_ld;
    // optional: d r_r_bias[h, :] += colsum_i(dG)[delta] . Rd[delta, h, :], and the same amount is taken OUT of d_rwb (the 8-wave
    // query-owner kernel leaves d r_w_bias + d r_r_bias there)
    const bf16_t* rd; int rd_rs; float* d_rrb; float* d_rwb;
    // recompute form (mxl_relattn_drd_recompute): the cells the query-owner kernel did not store are phantom distances, whose
    // score gradient needs no K / V:  dG[i, d] = -scale * delta_i * exp(scale * (q_i + r_r_bias) . Rd[d] - lse_i)
    const float* lse; const float* delta; float scale; int pz; int recompute;
};
constexpr int DRD_A = 32 * 512;        // dG tile  [32 i][256 delta] bf16
constexpr int DRD_B = 32 * 128;        // Qr tile  [32 i][64 e] bf16
constexpr int DRD_STAGE = DRD_A + DRD_B;
constexpr int DRD_SMEM = 4 * DRD_STAGE;                 // 80 KB: two workgroups per CU
__device__ __forceinline__ int drd_swzA(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)); }

__global__ __launch_bounds__(256, 2) void relattn_drd_kernel(DrdP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    int bx_, h, bz_;
    xcd_block(bx_, h, bz_);      // the distance blocks of one (head, batch group) read the same Qr tiles: keep them on one XCD
    const int b0 = bz_ * p.bgroup;
    const int nb = min(p.bgroup, p.B - b0);
    const int spb = p.T >> 5;                       // 32-row steps per batch item

    // Qr tile: one DMA instruction per wave, rows 8w .. 8w+7 (128-byte rows): lane -> row l >> 3, slot l & 7, block
    // ((s >> 1) ^ (k & 3)).
    const int brow = 8 * wid + (l >> 3);
    const int bcol = ((((l & 7) >> 1) ^ (brow & 3)) << 4) + ((l & 1) << 3);
    // transposed fragments for v_mfma_f32_16x16x32_bf16 (one K-step = 32 rows): lane (g = l >> 4, q = (l & 15) >> 2, pp = l & 3)
    // addresses k-row 8g + q, columns rb + 4pp..+3; rows +0 / +4 give the lane its 8 k-values of output row rb + (l & 15)
    const int fk = 8 * (l >> 4) + ((l & 15) >> 2);
    int ao[4], bo[4];                                // wave tile: distances 64 wid .. +63 (4 fragments) x all 64 e (4 fragments)
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int ca = 64 * wid + 16 * t + 4 * (l & 3), cb = 16 * t + 4 * (l & 3);
        ao[t] = fk * 512 + (((ca >> 4) ^ drd_swzA(fk)) << 5) + ((ca & 15) << 1);
        bo[t] = DRD_A + fk * 128 + (((cb >> 4) ^ (fk & 3)) << 5) + ((cb & 15) << 1);
    }
    auto trfrag = [&](const char* a, int pitch4) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(a));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(a + pitch4));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    const bool want_rrb = p.d_rrb != nullptr;
    const u32x4 ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};   // eight bf16 1.0
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);

    const int d0 = bx_ * 256;
    // tiles [0, nph) of every batch item lie entirely on phantom distances for this 256-distance block (i0 + 31 - pz < d0, the
    // rule by which the query-owner kernel skipped their dG stores): recomputed below; the others are streamed
    const int tph = (d0 + p.pz) >> 5;                // first tile with a stored-key cell in this distance block (may be < 0)
    const int nph = p.recompute ? max(0, min(spb, tph)) : 0;
    const int nst = spb - nph;
    const int S = nb * nst;

    // dG tile: DMA instruction j (0..3) of wave w fills rows 2(4j + w), +1 (512-byte rows): lane -> row l >> 5, 16-byte slot
    // l & 31; slot s of row k holds source block ((s >> 1) ^ drd_swzA(k)), half s & 1.
    int arow[4], acol[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        arow[j] = 2 * (4 * j + wid) + (l >> 5);
        acol[j] = min(d0 + ((((l & 31) >> 1) ^ drd_swzA(arow[j])) << 4) + ((l & 1) << 3), p.M - 8);
    }
    auto issue = [&](int g) {
        char* st = smem + (g & 3) * DRD_STAGE;
        const int b = b0 + g / nst, i0 = (nph + g % nst) << 5;
        const bf16_t* a = p.dg + (((size_t)b * p.H + h) * p.T + i0) * (size_t)p.M;
        const bf16_t* q = p.qr + (size_t)b * p.qr_bs + (size_t)i0 * p.qr_rs + (size_t)h * 64;
#pragma unroll
        for (int j = 0; j < 4; j++)
            __builtin_amdgcn_global_load_lds((gptr_t)(a + (size_t)arow[j] * p.M + acol[j]), (lptr_t)(st + (4 * j + wid) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(q + (size_t)brow * p.qr_rs + bcol), (lptr_t)(st + DRD_A + wid * 1024), 16, 0, 0);
    };

    f32x4 acc[4][4], accs[4];          // accs: dG^T . 1 = column sums of dG over i (every output column the same)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- recomputed tiles.  A step needs only the [32 i][64 e] Qr tile and lse / delta of its 32 queries (a 4-byte DMA into
    // this wave's own 256 bytes), 5 KB: the 80 KB of LDS hold sixteen such stages, loads run twelve steps ahead.
    // G[i, delta] = Qr . Rd^T on MFMA with the wave's 64 Rd rows held in registers for the whole pass, dG from the
    // accumulators -- whose layout, lane = delta with rows i = 4 (l >> 4) + r of the two 16-row halves, IS the A operand of the
    // contraction once the Qr^T fragments are read in that same k order (k-rows 4g + q and 16 + 4g + q instead of 8g + q and
    // 8g + 4 + q): no LDS round trip, no HBM bytes.
    const int SA = nb * nph;
    if (SA > 0) {
        constexpr int STA = 5120, NSTA = 16, PA = 12;      // stage bytes (4 KB Qr + 4 x 256 B), stages, prefetch distance
        static_assert(STA * NSTA <= DRD_SMEM, "recompute ring exceeds the kernel's LDS");
        const int gq = l >> 4, q4 = (l & 15) >> 2;
        bf16x8 rdf[4][2];
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const int dd = min(d0 + 64 * wid + 16 * f + (l & 15), p.M - 1);
                const bf16x8 raw = *reinterpret_cast<const bf16x8*>(p.rd + (size_t)dd * p.rd_rs + h * 64 + 32 * ks + 8 * gq);
                // scale * log2(e) rides on the Rd operand (the attention kernels put it on the query operand: the same single
                // bf16 rounding of one factor), so the MFMA result is the exponent and -lse enters as the accumulator's start
#pragma unroll
                for (int j = 0; j < 8; j++) rdf[f][ks][j] = (short)f2bf(bf2f((bf16_t)raw[j]) * (p.scale * LOG2E));
            }
        int qo[2][2], bo2[4];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const int row = 16 * t + (l & 15), blk = 2 * ks + (gq >> 1);
                qo[t][ks] = row * 128 + ((blk ^ (row & 3)) << 5) + ((gq & 1) << 4);
            }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int fk2 = 4 * gq + q4, cb = 16 * t + 4 * (l & 3);
            bo2[t] = fk2 * 128 + (((cb >> 4) ^ (fk2 & 3)) << 5) + ((cb & 15) << 1);
        }
        auto issueA = [&](int g) {
            char* st = smem + (g & (NSTA - 1)) * STA;
            const int b = b0 + g / nph, i0 = (g % nph) << 5;
            const bf16_t* q = p.qr + (size_t)b * p.qr_bs + (size_t)i0 * p.qr_rs + (size_t)h * 64;
            __builtin_amdgcn_global_load_lds((gptr_t)(q + (size_t)brow * p.qr_rs + bcol), (lptr_t)(st + wid * 1024), 16, 0, 0);
            const size_t row = ((size_t)b * p.H + h) * p.T + i0 + (l & 31);
            const float* src = (l < 32) ? p.lse + row : p.delta + row;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + 4096 + wid * 256), 4, 0, 0);
        };
        const float nsc = -p.scale;
        for (int g = 0; g < min(PA, SA); g++) issueA(g);
        if (SA > PA) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");      // 2 (PA - 2): steps 0 and 1 have landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // TWO steps per barrier (round 4): a step is 36 short MFMAs between an LDS round trip and a workgroup barrier, and at one
        // barrier per step the four waves spent more time meeting than computing (1.04 ms for 0.24 ms of MFMA at the bench shape)
        auto step = [&](int g) {
            const char* st = smem + (g & (NSTA - 1)) * STA;
            const float* sv = reinterpret_cast<const float*>(st + 4096 + wid * 256);
            const f32x4 ls0 = *reinterpret_cast<const f32x4*>(sv + 4 * gq), ls1 = *reinterpret_cast<const f32x4*>(sv + 16 + 4 * gq);
            const f32x4 dl0 = *reinterpret_cast<const f32x4*>(sv + 32 + 4 * gq), dl1 = *reinterpret_cast<const f32x4*>(sv + 48 + 4 * gq);
            float l0[4], l1[4], n0[4], n1[4];
#pragma unroll
            for (int r = 0; r < 4; r++) { l0[r] = -ls0[r] * LOG2E; l1[r] = -ls1[r] * LOG2E; n0[r] = nsc * dl0[r]; n1[r] = nsc * dl1[r]; }
            bf16x8 qa[2][2], fb[4];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) qa[t][ks] = *reinterpret_cast<const bf16x8*>(st + qo[t][ks]);
#pragma unroll
            for (int t = 0; t < 4; t++) fb[t] = trfrag(st + bo2[t], 16 * 128);
            // the sixteen G MFMAs of the step first (all four 16-distance fragments), then per fragment exp -> pack -> contraction:
            // the contraction MFMAs of fragment f then run under the exponentials of f + 1 instead of every fragment waiting for
            // the G products it has just issued (round 4)
            f32x4 cg0[4], cg1[4];
#pragma unroll
            for (int f = 0; f < 4; f++) {
                cg0[f] = f32x4{l0[0], l0[1], l0[2], l0[3]};
                cg1[f] = f32x4{l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    cg0[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, qa[0][ks]),
                                                                     __builtin_bit_cast(mfma_bf16x8, rdf[f][ks]), cg0[f], 0, 0, 0);
                    cg1[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, qa[1][ks]),
                                                                     __builtin_bit_cast(mfma_bf16x8, rdf[f][ks]), cg1[f], 0, 0, 0);
                }
            }
#pragma unroll
            for (int f = 0; f < 4; f++) {
                const f32x4 c0 = cg0[f], c1 = cg1[f];
                float g0[4], g1[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    g0[r] = __builtin_amdgcn_exp2f(c0[r]) * n0[r];
                    g1[r] = __builtin_amdgcn_exp2f(c1[r]) * n1[r];
                }
                const u32x4 fw = {pack2bf(g0[0], g0[1]), pack2bf(g0[2], g0[3]), pack2bf(g1[0], g1[1]), pack2bf(g1[2], g1[3])};
                const bf16x8 fa = __builtin_bit_cast(bf16x8, fw);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, fa),
                                                                        __builtin_bit_cast(mfma_bf16x8, fb[j]), acc[f][j], 0, 0, 0);
                if (want_rrb)
                    accs[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, fa),
                                                                      __builtin_bit_cast(mfma_bf16x8, ones), accs[f], 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int g = 0; g < SA; g += 2) {
            // stages (g + PA) % 16 and (g + PA + 1) % 16 were last read in steps g - 4 and g - 3: that pair's barrier is behind every wave
            const bool two = g + PA + 1 < SA;
            if (g + PA < SA) issueA(g + PA);
            if (two) issueA(g + PA + 1);
            step(g);
            if (g + 1 < SA) step(g + 1);
            // steps g + 2 and g + 3 must have landed before the next pair reads them; in the tail (nothing left to issue) drain once
            if (two) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }

    if (S > 0) {
        issue(0);
        if (S > 1) issue(1);
        if (S > 2) issue(2);
        if (S > 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (S > 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll 1
    for (int g = 0; g < S; g++) {
        // stage (g + 3) & 3 == (g - 1) & 3 was read in step g - 1, before that step's barrier
        const bool issued = g + 3 < S;
        if (issued) issue(g + 3);
        const char* st = smem + (g & 3) * DRD_STAGE;
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; t++) { fa[t] = trfrag(st + ao[t], 4 * 512); fb[t] = trfrag(st + bo[t], 4 * 128); }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, fa[i]),
                                                                    __builtin_bit_cast(mfma_bf16x8, fb[j]), acc[i][j], 0, 0, 0);
        if (want_rrb) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, fa[i]),
                                                                  __builtin_bit_cast(mfma_bf16x8, ones), accs[i], 0, 0, 0);
        }
        // step g + 1 must have landed before the next iteration reads it: leave the two youngest steps in flight
        if (issued) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (g + 2 < S) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // acc[i][j][r]: distance d0 + 64 wid + 16i + 4*(l >> 4) + r, e = 16j + (l & 15)   (MFMA issued (A, B): a lane owns one
    // column, so an atomic instruction covers 4 rows x 64 contiguous bytes -- the (B, A) layout's 16 rows x 4 separate dwords
    // run at a quarter of the atomic throughput, scripts/ubench/atomic_tiles.hip)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int dd = d0 + 64 * wid + 16 * i + 4 * (l >> 4) + r;
            if (dd >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) atomicAdd(p.drd + (size_t)dd * p.drd_ld + h * 64 + 16 * j + (l & 15), acc[i][j][r]);
        }
    if (want_rrb) {
        // this wave's 64 column sums (lanes with l & 15 == 0 hold them: rows 16i + 4*(l >> 4) + r) -> LDS (the ring is drained),
        // then lane = e: sum over the wave's distances of cs[delta] * Rd[delta, h, e]
        float* scs = reinterpret_cast<float*>(smem) + wid * 64;
        if ((l & 15) == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) scs[16 * i + 4 * (l >> 4) + r] = accs[i][r];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        float sum = 0.f;
        for (int dl = 0; dl < 64; dl++) {
            const int dd = d0 + 64 * wid + dl;
            if (dd < p.M) sum += scs[dl] * bf2f(p.rd[(size_t)dd * p.rd_rs + h * 64 + l]);
        }
        atomicAdd(p.d_rrb + h * 64 + l, sum);
        if (p.d_rwb) atomicAdd(p.d_rwb + h * 64 + l, -sum);
    }
}

template <int DH>
int launch_bwd(const BwdP& p, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&relattn_bwd_dq_kernel<DH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GeoQ<DH>::SMEM);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&relattn_bwd_dkv_kernel<DH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, GeoK<DH>::SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    {
        mxl_kt::Scope kt(MXL_KT_RELATTN_DELTA, s);
        hipLaunchKernelGGL(relattn_bwd_delta_kernel, dim3((p.B * p.T + 3) / 4), dim3(256), 0, s, p, DH);
    }
    if (p.d_rrb == nullptr) {
        static bool attr8 = false;
        if (!attr8) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&relattn_bwd_dq8_kernel<DH>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, GeoQ<DH>::SMEM);
            if (e != hipSuccess) return (int)e;
            attr8 = true;
        }
        mxl_kt::Scope kt(MXL_KT_RELATTN_DQ, s);
        hipLaunchKernelGGL((relattn_bwd_dq8_kernel<DH>), dim3((p.T + QB - 1) / QB, p.H, p.B), dim3(512), GeoQ<DH>::SMEM, s, p);
    } else {
        mxl_kt::Scope kt(MXL_KT_RELATTN_DQ, s);
        hipLaunchKernelGGL((relattn_bwd_dq_kernel<DH>), dim3((p.T + QB - 1) / QB, p.H, p.B), dim3(256), GeoQ<DH>::SMEM, s, p);
    }
    {
        mxl_kt::Scope kt(MXL_KT_RELATTN_DKV, s);
        hipLaunchKernelGGL((relattn_bwd_dkv_kernel<DH>), dim3((p.Kc + KB - 1) / KB, p.H, p.B), dim3(256), GeoK<DH>::SMEM, s, p);
    }
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

}  // namespace

static int relattn_bwd_impl(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                            const float* r_r_bias, const void* out, const void* dout, const float* lse, float* delta,
                            void* dq, void* dk, void* dv, void* dg, float* d_r_w_bias, float* d_r_r_bias, int B, int T,
                            int H, int dh, int M, int Kc, long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs,
                            long long o_bs, int o_rs, long long dq_bs, int dq_rs, long long dkv_bs, int dkv_rs,
                            float scale, int skip_phantom_dg, void* stream, const void* oph = nullptr, const float* mph = nullptr) {
    MXL_CHECK_ARG(q && k && v && rd && r_w_bias && r_r_bias && out && dout && lse && delta && dq && dk && dv);
    if (oph) MXL_CHECK_ARG(skip_phantom_dg && mph && ((uintptr_t)oph % 8) == 0);
    // the skipped cells are rebuilt by mxl_relattn_drd_recompute, which tiles 256 distances x 32 queries and needs d r_r_bias
    // left to it (the 8-wave query-owner kernel)
    if (skip_phantom_dg) MXL_CHECK_ARG(dg && d_r_r_bias == nullptr && dh == 64 && (M % 256) == 0 && (T % 32) == 0);
    MXL_CHECK_ARG(d_r_w_bias);      // d_r_r_bias == NULL: it is left to mxl_relattn_drd (see there)
    MXL_CHECK_ARG(B > 0 && T > 0 && H > 0 && M > 0 && (M % 8) == 0 && Kc >= T && Kc <= M + T);
    MXL_CHECK_ARG((q_rs % 8) == 0 && (kv_rs % 8) == 0 && (rd_rs % 8) == 0 && (o_rs % 8) == 0 && (dq_rs % 4) == 0 && (dkv_rs % 4) == 0);
    MXL_CHECK_ARG((q_bs % 8) == 0 && (kv_bs % 8) == 0 && (o_bs % 8) == 0 && (dq_bs % 4) == 0 && (dkv_bs % 4) == 0);
    BwdP p;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.rd = (const bf16_t*)rd;
    p.o = (const bf16_t*)out; p.dout = (const bf16_t*)dout;
    p.rwb = r_w_bias; p.rrb = r_r_bias; p.lse = lse; p.delta = delta; p.delta_out = delta;
    p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.dg = (bf16_t*)dg;
    p.d_rwb = d_r_w_bias; p.d_rrb = d_r_r_bias;
    p.B = B; p.T = T; p.H = H; p.M = M; p.Kc = Kc;
    p.q_bs = q_bs; p.kv_bs = kv_bs; p.o_bs = o_bs; p.dq_bs = dq_bs; p.dkv_bs = dkv_bs;
    p.q_rs = q_rs; p.kv_rs = kv_rs; p.rd_rs = rd_rs; p.o_rs = o_rs; p.dq_rs = dq_rs; p.dkv_rs = dkv_rs;
    p.scale = scale; p.scale_log2e = scale * LOG2E;
    p.dg_skip_phantom = skip_phantom_dg ? 1 : 0;
    p.oph = (const bf16_t*)oph; p.mph = mph;
    hipStream_t s = (hipStream_t)stream;
    switch (dh) {
        case 16: return launch_bwd<16>(p, s);
        case 32: return launch_bwd<32>(p, s);
        case 64: return launch_bwd<64>(p, s);
        default: return MXL_EUNSUPPORTED;
    }
}

#ifdef MXL_STAMP
extern "C" int mxl_debug_dq8_stamps(unsigned long long* host_out16) {
    hipError_t e = hipMemcpyFromSymbol(host_out16, HIP_SYMBOL(g_dq8_stamps), sizeof(unsigned long long) * 16);
    if (e != hipSuccess) return (int)e;
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_dq8_stamps), z, sizeof(z));
    return (int)e;
}
#endif

extern "C" int mxl_relattn_bwd(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                               const float* r_r_bias, const void* out, const void* dout, const float* lse, float* delta,
                               void* dq, void* dk, void* dv, void* dg, float* d_r_w_bias, float* d_r_r_bias, int B, int T,
                               int H, int dh, int M, int Kc, long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs,
                               long long o_bs, int o_rs, long long dq_bs, int dq_rs, long long dkv_bs, int dkv_rs,
                               float scale, void* stream) {
    return relattn_bwd_impl(q, k, v, rd, r_w_bias, r_r_bias, out, dout, lse, delta, dq, dk, dv, dg, d_r_w_bias, d_r_r_bias, B, T, H,
                            dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs, o_rs, dq_bs, dq_rs, dkv_bs, dkv_rs, scale, 0, stream);
}

extern "C" int mxl_relattn_bwd_sparse_dg(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                                         const float* r_r_bias, const void* out, const void* dout, const float* lse,
                                         float* delta, void* dq, void* dk, void* dv, void* dg, float* d_r_w_bias, int B, int T,
                                         int H, int dh, int M, int Kc, long long q_bs, int q_rs, long long kv_bs, int kv_rs,
                                         int rd_rs, long long o_bs, int o_rs, long long dq_bs, int dq_rs, long long dkv_bs,
                                         int dkv_rs, float scale, void* stream) {
    return relattn_bwd_impl(q, k, v, rd, r_w_bias, r_r_bias, out, dout, lse, delta, dq, dk, dv, dg, d_r_w_bias, nullptr, B, T, H,
                            dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs, o_rs, dq_bs, dq_rs, dkv_bs, dkv_rs, scale, 1, stream);
}

extern "C" int mxl_relattn_bwd_sparse_dg_oph(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                                             const float* r_r_bias, const void* out, const void* dout, const float* lse,
                                             float* delta, void* dq, void* dk, void* dv, void* dg, float* d_r_w_bias,
                                             const void* oph, const float* mph, int B, int T, int H, int dh, int M, int Kc,
                                             long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs, long long o_bs, int o_rs,
                                             long long dq_bs, int dq_rs, long long dkv_bs, int dkv_rs, float scale, void* stream) {
    MXL_CHECK_ARG(oph && mph);
    return relattn_bwd_impl(q, k, v, rd, r_w_bias, r_r_bias, out, dout, lse, delta, dq, dk, dv, dg, d_r_w_bias, nullptr, B, T, H,
                            dh, M, Kc, q_bs, q_rs, kv_bs, kv_rs, rd_rs, o_bs, o_rs, dq_bs, dq_rs, dkv_bs, dkv_rs, scale, 1, stream,
                            oph, mph);
}

static int relattn_drd_impl(const void* dg, const void* qr, float* d_rd, int B, int T, int H, int dh, int M,
                            long long qr_bs, int qr_rs, int drd_ld, const void* rd, int rd_rs, float* d_r_r_bias,
                            float* d_r_w_bias_fix, const float* lse, const float* delta, float scale, int Kc, int recompute,
                            void* stream) {
    MXL_CHECK_ARG(dg && qr && d_rd && B > 0 && T > 0 && H > 0 && M > 0);
    if (recompute) MXL_CHECK_ARG(rd && lse && delta && (M % 256) == 0 && Kc >= T && Kc <= M + T);
    if (dh != 64 || (T % 32) != 0 || (M % 8) != 0 || M < 8) return MXL_EUNSUPPORTED;
    MXL_CHECK_ARG((qr_rs % 8) == 0 && (qr_bs % 8) == 0 && drd_ld >= H * dh && ((uintptr_t)dg % 16) == 0 && ((uintptr_t)qr % 16) == 0);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&relattn_drd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, DRD_SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    DrdP p;
    p.dg = (const bf16_t*)dg; p.qr = (const bf16_t*)qr; p.drd = d_rd;
    p.B = B; p.T = T; p.H = H; p.M = M; p.qr_bs = qr_bs; p.qr_rs = qr_rs; p.drd_ld = drd_ld;
    MXL_CHECK_ARG(!d_r_r_bias || rd);
    p.rd = (const bf16_t*)rd; p.rd_rs = rd_rs; p.d_rrb = d_r_r_bias; p.d_rwb = d_r_w_bias_fix;
    p.lse = lse; p.delta = delta; p.scale = scale; p.recompute = recompute ? 1 : 0;
    {   // first stored key tile of the attention kernels (floor(p0 / 64) * 64, p0 = T - Kc <= 0): key positions below it are phantom
        const int p0 = T - Kc;
        p.pz = recompute ? -(((-p0) + 63) / 64) * 64 : 0;
    }
    // batch groups: fill the 512 resident workgroup slots about once (each workgroup ends with 64 KB of fp32 atomics)
    const int tiles = ((M + 255) / 256) * H;
    int groups = 512 / tiles;
    if (groups < 1) groups = 1;
    if (groups > B) groups = B;
    p.bgroup = (B + groups - 1) / groups;
    groups = (B + p.bgroup - 1) / p.bgroup;
    {
        mxl_kt::Scope kt(MXL_KT_RELATTN_DRD, (hipStream_t)stream);
        hipLaunchKernelGGL(relattn_drd_kernel, dim3((M + 255) / 256, H, groups), dim3(256), DRD_SMEM, (hipStream_t)stream, p);
    }
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_relattn_drd(const void* dg, const void* qr, float* d_rd, int B, int T, int H, int dh, int M,
                               long long qr_bs, int qr_rs, int drd_ld, const void* rd, int rd_rs, float* d_r_r_bias,
                               float* d_r_w_bias_fix, void* stream) {
    return relattn_drd_impl(dg, qr, d_rd, B, T, H, dh, M, qr_bs, qr_rs, drd_ld, rd, rd_rs, d_r_r_bias, d_r_w_bias_fix, nullptr,
                            nullptr, 0.f, T, 0, stream);
}

extern "C" int mxl_relattn_drd_recompute(const void* dg, const void* qr, float* d_rd, int B, int T, int H, int dh, int M,
                                         long long qr_bs, int qr_rs, int drd_ld, const void* rd, int rd_rs, float* d_r_r_bias,
                                         float* d_r_w_bias_fix, const float* lse, const float* delta, float scale, int Kc,
                                         void* stream) {
    return relattn_drd_impl(dg, qr, d_rd, B, T, H, dh, M, qr_bs, qr_rs, drd_ld, rd, rd_rs, d_r_r_bias, d_r_w_bias_fix, lse, delta,
                            scale, Kc, 1, stream);
}
