// K4 backward, fused key-owner form (round 4): ONE pass over the score cells for every gradient of the banded relative-position
// attention (forward statement: relattn_fwd.hip; the three-kernel form this replaces at dh = 64: relattn_bwd.hip).
//
// With raw = AC + BD, P = exp(raw * scale - lse), dP[i,p] = dO_i . v_p, delta_i = dO_i . O_i, dS = scale * P * (dP - delta):
//     dv_p = sum_i P[i,p] dO_i            dk_p = sum_i dS[i,p] (q_i + r_w_bias)
//     dq_i = sum_p dS[i,p] k_p + sum_d dG[i,d] Rd[d]            dG[i,d] = dS[i, i-d]   (the un-skewed score gradient)
//     dRd[d] = sum_{b,i} dG[i,d] (q_i + r_r_bias)               d r_w_bias = sum_i dQw_i,  d r_r_bias = sum_i dQr_i
//
// The three-kernel form computes the scores twice (query-owner for dq and dG, key-owner for dk and dv: 10 score-sized MFMA
// products per cell for 6 algorithmic ones) and sends dG (B,H,T,M) through HBM to a third, streaming kernel for dRd: 3.8 GB
// written + 3.2 GB read per layer at the bench shape.  Here a workgroup owns 256 keys of one (sequence, head) and sweeps the
// 32-query tiles that see them (exactly M keys per query):
//   * S and dP once, key on the lane (wave w = keys 32w .. 32w+31): their accumulators ARE the B operands of dV^T += dO^T P and
//     dK^T += Qw^T dS, which stay in registers for the whole sweep -- dk, dv are written once, no atomics.
//   * dS crosses LDS once, in two images: X[key][query] (straight, 8-byte writes) and Y[query][distance - dlo] (un-skewed, 16-bit
//     writes at lane-constant + immediate addresses: the rel-shift).  After the barrier every wave forms one 16 x 16 piece of
//     dq[32 queries][64] = X^T K + Y Rd over the whole contraction (256 keys + 288 distances, v_mfma_f32_16x16x32_bf16), so the
//     8 waves' pieces need no sum: the tile's partial dq goes out with plain stores into a per-key-block fp32 slab, and
//     relattn_dq_finish_kernel adds the (at most M/256 + 1) slabs of a query, the forward's phantom value-sum term, rounds to
//     bf16 once.  Plain stores run 4-5x the float-atomic rate (MI355X_MICROARCH.md, Global float atomics) and the sum is
//     reproducible.
//   * dG never leaves the chip.  The 32-distance blocks of the tile's window slide down by one block per query tile; wave w
//     keeps the fp32 accumulator of "its" block -- the one whose index is w mod 8 -- for the nine tiles the block stays in the
//     window (dRd_blk += Y_blk^T Qr, 4 MFMAs per tile), then adds it to d_rd with 32 float atomic instructions (two 128-byte
//     segments each) and takes the next block of its residue class: 8 KB of atomics per workgroup and tile, 1.8 GB per layer
//     at the bench shape instead of the 7 GB round trip of dG, and no third kernel.
//   * the positional term G = Qr Rd^T of the NEXT query tile is formed (nine 32 x 32 blocks over the eight waves) while this
//     tile's dq / dRd products run, into an fp16 skew buffer that the score phase reads at column (query - key) with sixteen
//     immediate-offset 16-bit reads.
// Score-sized MFMA products per cell: S, dP, G, dV, dK, dQw, dQr, dRd = 8 for 8 algorithmic ones (the old form: 10 + the
// streaming contraction).
//
// Zero memories (the reference's training, mode R): key positions below the first stored one exist only as distances.  Their
// part of dq comes from the forward's value-sum over ALL such cells (mxl_relattn_fwd_phantom2, oph_all = 1), added by the
// finishing kernel; their part of d_rd is rebuilt on MFMA from q, rd, lse, delta by mxl_relattn_drd_phantom (relattn_bwd.hip).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "musicxl_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// ---- compile-time knobs of the fused kernel (A/B builds: scripts/ab_build.sh relattn_bwd_fused <tag> -D...)
#ifndef FUSED_PF
#define FUSED_PF 3                              // phase B: units of operand reads in flight ahead of the MFMA that consumes them
#endif
#ifndef FUSED_VF_RELOAD
#define FUSED_VF_RELOAD 0
#endif
#ifndef FUSED_PARK
#define FUSED_PARK 0                            // the leaving dRd block through an LDS park buffer (1) or straight from each wave's registers (0)
#endif
#ifndef FUSED_SB
#define FUSED_SB 1                              // scheduling fences between the units of phase B
#endif
#define UNIT_FENCE() do { if (FUSED_SB) __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef FUSED_DVDK_LATE
#define FUSED_DVDK_LATE 1                       // dV / dK MFMAs behind barrier 1 (1) or in front of it, as in round 4 (0)
#endif
#ifndef FUSED_PRIO
#define FUSED_PRIO 3                            // s_setprio(1) around MFMA clusters.  bit 0: S / dP chains and dV / dK; bit 1: the 26 units of phase B; bit 2: the next
                                                // tile's G.  The wave that is issuing MFMAs wins the SIMD's issue port over the other wave's vector work, which fills
                                                // the gaps instead of delaying the chain: 3.877 -> 3.726 ms per layer (-3.9 %, four alternating same-box rounds,
                                                // profiles/r06_fused_bwd_setprio_ab.log; bit 0 alone -1.7 %, bit 1 alone -1.6 %, bit 2 adds nothing)
#endif
#define PRIO_UP() do { if (FUSED_PRIO & 1) __builtin_amdgcn_s_setprio(1); } while (0)
#define PRIO_DOWN() do { if (FUSED_PRIO & 1) __builtin_amdgcn_s_setprio(0); } while (0)
#define PRIO_UP_B(bit_) do { if (FUSED_PRIO & (bit_)) __builtin_amdgcn_s_setprio(1); } while (0)
#define PRIO_DOWN_B(bit_) do { if (FUSED_PRIO & (bit_)) __builtin_amdgcn_s_setprio(0); } while (0)
#ifndef FUSED_TRF_EARLY
#define FUSED_TRF_EARLY 1                       // transposed dO / Qw fragments requested ahead of barrier 1 (1) or behind it (0)
#endif
constexpr float LOG2E = 1.4426950408889634f;
constexpr int KBLK = 256;                       // keys per workgroup
constexpr int QT = 32;                          // queries per tile
constexpr int ROWB = 128;                       // bytes per 64-element bf16 row
constexpr int K_BYTES = KBLK * ROWB;            // K image                      32 KB
constexpr int RING_BLKS = 10;                   // Rd ring: ten 32-row blocks   40 KB (nine of a window + the next tile's new one)
constexpr int RBLK_BYTES = 32 * ROWB;
constexpr int RING_BYTES = RING_BLKS * RBLK_BYTES;
constexpr int QIMG = QT * ROWB;
constexpr int QSET = 3 * QIMG + 2 * QT * 4;     // Qw, Qr, dO images + -lse (log2) + -scale*delta          12.25 KB, two sets
constexpr int GP = 584;                         // fp16 skew buffer pitch (bytes): 288 columns + pad; 146 dwords = 2 mod 16
constexpr int G_BYTES = QT * GP;
constexpr int X_BYTES = KBLK * 64;              // X[key][32 queries] bf16
constexpr int YP = 608;                         // Y[query][288 columns + pad] bf16: 152 dwords = 24 mod 64: the 16-byte row reads of the dq piece
                                                // (lane groups {0-3, 12-15, 20-27} ..: MI355X_MICROARCH.md, LDS) are conflict-free; at 592
                                                // (rounds 4) each was a two-way conflict (scripts/lds_conflicts.py)
constexpr int Y_BYTES = QT * YP;
constexpr int BIAS_BYTES = 2 * 64 * 4;
constexpr int PARK_BYTES = FUSED_PARK ? 32 * 64 * 4 : 0;   // (FUSED_PARK) the dRd block that left the window in the previous tile: [32 distances][64] fp32
constexpr int SMEM = K_BYTES + RING_BYTES + 2 * QSET + G_BYTES + X_BYTES + Y_BYTES + BIAS_BYTES + PARK_BYTES;   // 153 856 B (+ 8 KB with FUSED_PARK): one workgroup per CU
constexpr int RING_OFF = 10240;                 // multiple of RING_BLKS added to (possibly negative) block indices before the modulo

// In-kernel stamps (diagnostic builds only: scripts/ab_build.sh relattn_bwd_fused stamp -DMXL_STAMP; the shipped library has none).
// Per wave, shader cycles between consecutive stamps are summed per segment and added to g_fused_stamps at the end.
#ifdef MXL_STAMP
__device__ unsigned long long g_fused_stamps[16];
#define STAMP_DECL unsigned long long st_last, st_acc[16] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}; \
    { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define STAMP(i) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); st_acc[i] += t_ - st_last; st_last = t_; }
#define STAMP_FLUSH if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 16; i_++) atomicAdd(&g_fused_stamps[i_], st_acc[i_]); }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

// Partial-dq slabs: fp32 (MXL_SLAB_BF16 = 0) or bf16.  bf16 halves the slab round trip (dq finish 0.43 -> 0.25 ms, fused pass
// -0.03 ms at the bench shape) at the price of one bf16 rounding per key block and query in front of the fp32 sum.
#ifndef MXL_SLAB_BF16
#define MXL_SLAB_BF16 1
#endif
#if MXL_SLAB_BF16
typedef bf16_t slab_t;
#else
typedef float slab_t;
#endif

struct FusedP {
    const bf16_t *q, *k, *v, *rd, *dout;
    const float *rwb, *rrb, *lse, *delta;
    bf16_t *dk, *dv;
    slab_t* slab;         // [nslot][B][T][H*64] partial dq, slot = key block - first key block that sees the query tile
    float* drd;           // (M, drd_ld) fp32, +=
    float *d_rwb, *d_rrb; // (H, 64) fp32, +=  (the stored keys' part)
    int B, T, H, M, Kc;
    long long q_bs, kv_bs, o_bs, dkv_bs, slab_stride;
    int q_rs, kv_rs, rd_rs, o_rs, dkv_rs, drd_ld;
    float scale, scale_log2e;
};

// Q-set images: [32][64] bf16, 16-byte chunk c of row at c ^ qswz(row), qswz = (bit 1, bit 2, bit 3) of the row index as chunk bits
// (2, 1, 0) -- the image of the phantom-cell dRd kernel's records (relattn_drd_phantom.hip, ph_swz).  Row reads (32x32x16 A operand:
// ds_read_b128 lane groups of 16 rows) cover the 64 banks once, and so does each 32-lane half of the transposed reads (4 rows x 64
// bytes: rows q, q + 2 land in different 64-byte halves).  Until round 5 the swizzle was (row >> 1) & 7, under which every
// transposed read of these images was a two-way bank conflict (scripts/lds_conflicts.py; 24 of them per wave and tile).
__device__ __forceinline__ int qswz(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }
__device__ __forceinline__ int qoff(int row, int ch) { return row * ROWB + ((ch ^ qswz(row)) << 4); }
__device__ __forceinline__ int qeoff(int row, int e) { return qoff(row, e >> 3) + ((e & 7) << 1); }
// K image and Rd ring: chunk c of a row at c ^ s(row), s = (bit 3, bit 1, bit 2) of the row index.  A bijection of bits 1..3, so
// 16-byte row reads by 16 different rows hit 64 different banks; and for the transposed 8-byte reads of a 16x16x32 B fragment
// (rows 8g + q (+4), 4 columns per lane) the two bits that separate the four row pairs of a half-wave land in chunk bits 1, 2.
__device__ __forceinline__ int sswz(int row) { return (((row >> 3) & 1) << 2) | (((row >> 1) & 1) << 1) | ((row >> 2) & 1); }
__device__ __forceinline__ int soff(int row, int ch) { return row * ROWB + ((ch ^ sswz(row)) << 4); }

// workgroup barrier for LDS hand-offs: waits for this wave's LDS operations only.  __syncthreads() also waits s_waitcnt vmcnt(0), i.e.
// for every global store and atomic the wave has in flight -- the tile loop's slab stores and dRd atomics would be drained twice
// per tile, with the other waves of the workgroup waiting at the barrier for the slowest drain.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(const char*)p;
}

// sixteen genuine 16-bit LDS reads: register j belongs to query ii = pat(j) + 4 hh, pat(j) = (j & 3) + 8 (j >> 2), and needs
// column ii + 256 - kloc of row ii of the skew buffer: lane base + pat(j) * (GP + 2)
// (issued WITHOUT a wait: LDS operations complete in order, so any later wait of the wave's own LDS reads covers these; the
// caller places gskew_wait() in front of the first use -- hipcc does not count loads made by inline asm)
__device__ __forceinline__ void gskew_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void gskew_read16(uint32_t base, uint32_t (&u)[16]) {
    static_assert(GP == 584, "offsets below are pat(j) * (GP + 2)");
    asm volatile(
        "ds_read_u16 %0, %16\n\t"               "ds_read_u16 %1, %16 offset:586\n\t"    "ds_read_u16 %2, %16 offset:1172\n\t"
        "ds_read_u16 %3, %16 offset:1758\n\t"   "ds_read_u16 %4, %16 offset:4688\n\t"   "ds_read_u16 %5, %16 offset:5274\n\t"
        "ds_read_u16 %6, %16 offset:5860\n\t"   "ds_read_u16 %7, %16 offset:6446\n\t"   "ds_read_u16 %8, %16 offset:9376\n\t"
        "ds_read_u16 %9, %16 offset:9962\n\t"   "ds_read_u16 %10, %16 offset:10548\n\t" "ds_read_u16 %11, %16 offset:11134\n\t"
        "ds_read_u16 %12, %16 offset:14064\n\t" "ds_read_u16 %13, %16 offset:14650\n\t" "ds_read_u16 %14, %16 offset:15236\n\t"
        "ds_read_u16 %15, %16 offset:15822"
        : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7]),
          "=&v"(u[8]), "=&v"(u[9]), "=&v"(u[10]), "=&v"(u[11]), "=&v"(u[12]), "=&v"(u[13]), "=&v"(u[14]), "=&v"(u[15])
        : "v"(base)
        : "memory");
}
// the mirror image for dS into Y: w[m] holds (value 2m, value 2m + 1) as a bf16 pair; lane base + pat(j) * (YP + 2)
__device__ __forceinline__ void yskew_write16(uint32_t base, const uint32_t (&w)[8]) {
    static_assert(YP == 608, "offsets below are pat(j) * (YP + 2)");
    asm volatile(
        "ds_write_b16 %8, %0\n\t"               "ds_write_b16_d16_hi %8, %0 offset:610\n\t"
        "ds_write_b16 %8, %1 offset:1220\n\t"   "ds_write_b16_d16_hi %8, %1 offset:1830\n\t"
        "ds_write_b16 %8, %2 offset:4880\n\t"   "ds_write_b16_d16_hi %8, %2 offset:5490\n\t"
        "ds_write_b16 %8, %3 offset:6100\n\t"   "ds_write_b16_d16_hi %8, %3 offset:6710\n\t"
        "ds_write_b16 %8, %4 offset:9760\n\t"   "ds_write_b16_d16_hi %8, %4 offset:10370\n\t"
        "ds_write_b16 %8, %5 offset:10980\n\t"   "ds_write_b16_d16_hi %8, %5 offset:11590\n\t"
        "ds_write_b16 %8, %6 offset:14640\n\t"   "ds_write_b16_d16_hi %8, %6 offset:15250\n\t"
        "ds_write_b16 %8, %7 offset:15860\n\t"   "ds_write_b16_d16_hi %8, %7 offset:16470"
        :
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(base)
        : "memory");
}
__device__ __forceinline__ float add_f16(float s, uint32_t h16) {     // s + (float)h in ONE VALU issue
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(s), "v"(h16));
    return r;
}
__device__ __forceinline__ bf16x8 tr_pair(const char* lo_p, const char* hi_p) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)lo_p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)hi_p);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a), __builtin_bit_cast(mfma_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, a), __builtin_bit_cast(mfma_bf16x8, b), c, 0, 0, 0);
}

// Vector-memory traffic of the tile loop and hipcc's s_waitcnt pass.  The vmcnt counter is in order, and hipcc places the wait for
// the rows staged a tile ahead from what it can PROVE is younger than them on every path.  Until round 5 one wave per tile issued the
// 32 float atomics of the dRd block that left the window, the staging loads sat behind `if` conditions, and the loop was entered
// with the first rows just requested: the wait came out as s_waitcnt vmcnt(0) at the top of every tile -- a drain of the tile's slab
// stores and, on the wave that had just flushed, of 32 atomics (~3 k cycles each with every CU adding), the other seven waves
// waiting at barrier 1: 0.5 ms of the 4.0 ms kernel (ablation: profiles/r05_fused_bwd_notes.txt).  Now every wave issues the same
// operations on every path of the loop -- its staging loads, four atomics of the block parked in LDS, two slab stores -- also on the
// last tiles (clamped tile index, stores into the image nobody reads any more) and in front of the loop (VM_PER_TILE dummy
// atomics of 0.0), so the wait hipcc computes leaves the younger six in flight.
constexpr int VM_PER_TILE = 6;                  // vector-memory operations a wave issues per tile behind its staging loads: 4 atomics + 2 slab stores

// delta[b,h,i] = sum_e dO[b,i,h,e] * O[b,i,h,e]   (dh = 64: 8 lanes per head, one wave per token row per sweep of 8 heads)
__global__ __launch_bounds__(256) void fused_delta_kernel(const bf16_t* o, const bf16_t* dout, float* delta, int B, int T, int H,
                                                          long long o_bs, int o_rs) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * T) return;
    const int lane = threadIdx.x & 63;
    const int b = row / T, i = row % T;
    const bf16_t* op = o + (size_t)b * o_bs + (size_t)i * o_rs;
    const bf16_t* dp = dout + (size_t)b * o_bs + (size_t)i * o_rs;
    const int chunks = H * 8;
    for (int c0 = 0; c0 < chunks; c0 += 64) {
        const int c = c0 + lane;
        float s = 0.f;
        if (c < chunks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(op + c * 8);
            const bf16x8 d = *reinterpret_cast<const bf16x8*>(dp + c * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) s += bf2f((bf16_t)a[j]) * bf2f((bf16_t)d[j]);
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (c < chunks && (lane & 7) == 0) delta[((size_t)b * H + (c >> 3)) * T + i] = s;
    }
}

// Eight waves, two per SIMD, 256 registers each: wave w = keys 32 w .. 32 w + 31.  (Round 4 also carried a four-wave form -- one wave
// per SIMD, 512 registers, 64 keys per wave: 5.07 ms against 4.08 ms with every refinement applied, profiles/r04_experiments_not_kept.txt
// -- one wave per SIMD exposes every LDS round trip.  Removed in round 5.)
//
// A tile (round 5 order):
//   top      the rows of tile it + 1 (requested a tile ago) go into their LDS images, the rows of tile it + 2 are requested
//   phase A  S and dP chains (8 MFMAs), exponentials, dS -> X / Y; the transposed dO / Qw fragments of the dV / dK products are requested
//   barrier 1
//   phase B  one software-pipelined stream of (operand reads -> MFMA) units, three units of reads in flight ahead of the MFMA that
//            consumes them: the first three units' reads, then the eight dV / dK MFMAs (operands in registers since phase A: they fill
//            the LDS latency behind the barrier), the 8 key units and 9 distance units of the dq piece, the slab stores, the dRd
//            block, the next tile's G block(s)
//   barrier 2
// Until round 5 phase B was three rounds "all reads of the round, wait, all MFMAs of the round" (256 registers hold one round), each
// of which exposed a full LDS round trip with all eight waves reading at once, and dV / dK sat in front of barrier 1.
__global__ __launch_bounds__(512, 1) void relattn_bwd_fused_kernel(FusedP p) {
    constexpr int NT = 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sR = sK + K_BYTES;
    char* sQ = sR + RING_BYTES;
    char* sG = sQ + 2 * QSET;
    char* sX = sG + G_BYTES;
    char* sY = sX + X_BYTES;
    float* sBias = reinterpret_cast<float*>(sY + Y_BYTES);       // r_w_bias[64], r_r_bias[64] of this head
    float* sPark = sBias + BIAS_BYTES / 4;                       // the parked dRd block

    STAMP_DECL
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, r = l & 31, hh = l >> 5;
    int bx_, h, b;
    xcd_block(bx_, h, b);
    const int T = p.T, M = p.M;
    const int p0 = T - p.Kc;                    // lowest stored key position (a multiple of 32)
    const int P0 = p0 + KBLK * bx_;             // first key position of the workgroup
    const int Pw = P0 + 32 * w;                 // first key position of the wave
    const int kloc0 = 32 * w + r;               // this lane's key inside the workgroup
    const int MB = M >> 5;

    const bf16_t* qbase = p.q + (size_t)b * p.q_bs + (size_t)h * 64;
    const bf16_t* dobase = p.dout + (size_t)b * p.o_bs + (size_t)h * 64;
    const bf16_t* rbase = p.rd + (size_t)h * 64;
    const bf16_t* kbase = p.k + (size_t)b * p.kv_bs + (size_t)h * 64;
    const bf16_t* vbase = p.v + (size_t)b * p.kv_bs + (size_t)h * 64;

    // ---- lane constants.  gq / q4 / pp: the decomposition the transposed 8-byte reads use (16 lanes read 4 rows x 16 columns)
    const int gq = l >> 4, q4 = (l & 15) >> 2, pp = l & 3;
    int rowq = qoff(r, hh);                                  // Q-set row fragment, k-step ks: rowq ^ (ks << 5)
    int rows = r * ROWB + ((hh ^ sswz(r)) << 4);             // ring row fragment,   k-step ks: rows ^ (ks << 5)
    // Q-set transposed pattern: rows 4 hh + q4 (+ 8: chunk bit 0 flips, + 16: nothing), elements 16 (gq & 1) + 4 pp .. + 3 (+ 32 e: chunk bit 2 flips)
    int tkb = qeoff(4 * hh + q4, 16 * (gq & 1) + 4 * pp);
    // skew buffers
    const uint32_t gRb = lds_addr(sG) + 4 * hh * (GP + 2) + (256 - kloc0) * 2;
    const uint32_t yWb = lds_addr(sY) + 4 * hh * (YP + 2) + (256 - kloc0) * 2;
    // X[key][query]: 8-byte granule g of row p at g ^ ((p >> 1) & 7): this lane's group `grp` (queries 8 grp + 4 hh ..) at xw ^ (grp << 4)
    const int xf = (kloc0 >> 1) & 7;
    int xw = kloc0 * 64 + ((((hh ^ (xf & 1)) | (xf & 6))) << 3);
    // dRd operands (16x16x32, this wave's slice: distances 16 (w & 1) .. + 15 of a block, elements 16 (w >> 1) .. + 15): transposed reads
    // with the contraction index (the query) permuted -- lane group g reads image rows 4 g .. 4 g + 3 and 16 + 4 g .. 19 + 4 g, in Y and
    // in the Qr image alike -- so that the eight rows a 32-lane half touches are consecutive (conflict-free at this pitch)
    int yt0 = (4 * (l >> 4) + ((l & 15) >> 2)) * YP + (16 * (w & 1) + 4 * pp) * 2;       // second read: + 16 YP; block v: + 64 v
    int qb0 = qeoff(4 * (l >> 4) + ((l & 15) >> 2), 16 * (w >> 1) + 4 * pp);             // second read: + 2048 (sixteen rows on)
    // dq piece of this wave: elements 16 eq .. 16 eq + 15 of the 16-row half ih0 of the tile's queries
    const int eq = w >> 1, ih0 = w & 1;
    const int g16 = l >> 4, q16 = (l & 15) >> 2;                   // 16x16x32 transposed pattern: rows 8 g16 + q16 (+4), 4 columns at 4 pp
    int xa0 = ((8 * g16 + q16) * 64 + ((pp ^ ((4 * g16 + (q16 >> 1)) & 7)) << 3)) ^ (ih0 << 5);
    int xa1 = ((8 * g16 + q16 + 4) * 64 + ((pp ^ ((4 * g16 + 2 + (q16 >> 1)) & 7)) << 3)) ^ (ih0 << 5);
    const int kch = 2 * eq + (pp >> 1);
    const int ksw = ((g16 & 1) << 2) | (((q16 >> 1) & 1) << 1);
    int ka0 = (8 * g16 + q16) * ROWB + ((kch ^ ksw) << 4) + ((pp & 1) << 3);
    int ka1 = (8 * g16 + q16 + 4) * ROWB + ((kch ^ (ksw | 1)) << 4) + ((pp & 1) << 3);
    int yq0 = (16 * ih0 + (l & 15)) * YP + 16 * g16;         // Y row fragment (16x16x32 A): row = query
    int krow = kloc0 * ROWB + ((hh ^ sswz(kloc0)) << 4);           // K image row fragment of the lane's key (B operand of S), k-step ks: krow ^ (ks << 5)

    // ---- prologue: biases, K image, Y zeroed, V fragments (B operand of dP: lane = key, k = 16 ks + 8 hh + j)
    if (tid < 128) sBias[tid] = (tid < 64) ? p.rwb[h * 64 + tid] : p.rrb[h * 64 + tid - 64];
#pragma unroll
    for (int n = 0; n < 2048 / NT; n++) {
        const int c = tid + n * NT;
        const int row = c >> 3, ch = c & 7;
        const int srow = KBLK * bx_ + row;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (srow < p.Kc) val = *reinterpret_cast<const u32x4*>(kbase + (size_t)srow * p.kv_rs + ch * 8);
        *reinterpret_cast<u32x4*>(sK + soff(row, ch)) = val;
    }
    for (int i = tid; i < Y_BYTES / 16; i += NT) reinterpret_cast<u32x4*>(sY)[i] = u32x4{0u, 0u, 0u, 0u};
    bf16x8 vf[4];
    const bool kok = Pw + r < T;
    {
        const size_t srow = (size_t)(kok ? Pw + r - p0 : 0);
        const bf16_t* vp = vbase + srow * p.kv_rs;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const bf16x8 vv = *reinterpret_cast<const bf16x8*>(vp + 16 * ks + 8 * hh);
#pragma unroll
            for (int j = 0; j < 8; j++) vf[ks][j] = kok ? vv[j] : (short)0;
        }
    }

    // queries that can see any key of this workgroup: i in [P0, P0 + 255 + M - 1], clipped to [0, T)
    const int i_lo = max(P0, 0), i_hi = min(P0 + KBLK - 1 + M - 1, T - 1);
    const int it_lo = i_lo >> 5, it_hi = i_hi >> 5;

    // ---- staging: every thread owns one 16-byte chunk (row (tid & 255) >> 3, chunk tid & 7) of two of the four 4 KB images a tile
    // needs: waves 0-3 the Qw image (from the q rows) and the new Rd block, waves 4-7 the Qr image (from the same q rows) and the dO
    // image -- two 16-byte loads, one or two conversions and two 16-byte LDS stores per thread on every wave.  (Until round 5 waves 0-3
    // formed Qw AND Qr and waves 4-7 dO + Rd: the first four waves carried 2.5 x the conversion work at the top of every tile, with the
    // others waiting for them at barrier 1.)  Every global access of the tile loop goes through a buffer descriptor (wave-uniform base
    // in scalar registers) with a 32-bit lane offset and a scalar tile offset.
    const bool st_q = w < 4;                    // (wave-uniform: `w` is a scalar)
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)qbase, 0, -1, 0x00020000);
    // the second chunk: Rd rows (waves 0-3) or dO rows (waves 4-7) -- one descriptor, selected once with scalar selects (a
    // lane-dependent select of two descriptors makes hipcc wrap the load in a waterfall loop)
    const __amdgpu_buffer_rsrc_t rs_2 = __builtin_amdgcn_make_buffer_rsrc((void*)(st_q ? rbase : dobase), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lse = __builtin_amdgcn_make_buffer_rsrc((void*)(p.lse + ((size_t)b * p.H + h) * T), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.delta + ((size_t)b * p.H + h) * T), 0, -1, 0x00020000);
    u32x4 tq = {0u, 0u, 0u, 0u}, t2 = {0u, 0u, 0u, 0u};
    float tl = 0.f, tdl = 0.f;
    // The staging indices are recomputed from the thread id at every use (the id goes through an empty asm, so hipcc cannot hoist what
    // is derived from it): held across the tile loop, their registers were what the kernel spilled.
#define STG_IDS int tid_o = tid; asm volatile("" : "+v"(tid_o)); const int srow_ = (tid_o & 255) >> 3, sch_ = tid_o & 7;
    auto load_stage = [&](int it, int nblk) {   // the rows of tile `it` (T % 32 == 0: every row exists) and distance block `nblk`
        STG_IDS
        const int I = it * QT;
        tq = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_q, (srow_ * p.q_rs + sch_ * 8) * 2, I * p.q_rs * 2, 0));
        int d = 32 * nblk + srow_;              // (row index clamped: out-of-range cells are masked)
        d = d < 0 ? 0 : (d > M - 1 ? M - 1 : d);
        const int vo2 = st_q ? (d * p.rd_rs + sch_ * 8) * 2 : (srow_ * p.o_rs + sch_ * 8) * 2;
        const int so2 = st_q ? 0 : I * p.o_rs * 2;
        t2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_2, vo2, so2, 0));
        tl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_lse, (tid & 31) * 4, I * 4, 0));
        tdl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_dl, (tid & 31) * 4, I * 4, 0));
    };
    auto store_stage = [&](int buf, int slot) { // -> Q set `buf`, ring slot `slot`
        STG_IDS
        char* sQw = sQ + buf * QSET;
        char* sQr = sQw + QIMG;
        char* sDO = sQr + QIMG;
        float* sLse = reinterpret_cast<float*>(sDO + QIMG);
        {                                       // Qw (r_w_bias) or Qr (r_r_bias): operand scaling as in relattn_fwd.hip, so the recomputed
            const bf16_t* src = reinterpret_cast<const bf16_t*>(&tq);          // scores match its LSE bit for bit
            const float* bb = sBias + (st_q ? 0 : 64) + sch_ * 8;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bb), b1 = *reinterpret_cast<const f32x4*>(bb + 4);
            u32x4 wq;
            bf16_t* dq_ = reinterpret_cast<bf16_t*>(&wq);
#pragma unroll
            for (int j = 0; j < 8; j++) dq_[j] = f2bf((bf2f(src[j]) + (j < 4 ? b0[j & 3] : b1[j & 3])) * p.scale_log2e);
            *reinterpret_cast<u32x4*>((st_q ? sQw : sQr) + qoff(srow_, sch_)) = wq;
        }
        if (st_q) {
            *reinterpret_cast<u32x4*>(sR + slot * RBLK_BYTES + soff(srow_, sch_)) = t2;
        } else {
            const bf16_t* src = reinterpret_cast<const bf16_t*>(&t2);
            u32x4 wd;
            bf16_t* dd = reinterpret_cast<bf16_t*>(&wd);
#pragma unroll
            for (int j = 0; j < 8; j++) dd[j] = f2bf(bf2f(src[j]) * p.scale);
            *reinterpret_cast<u32x4*>(sDO + qoff(srow_, sch_)) = wd;
        }
        if (tid < 32) { sLse[tid] = -tl * LOG2E; sLse[QT + tid] = -p.scale * tdl; }
    };
    // ring slots: block n sits in slot (n + RING_OFF) mod RING_BLKS.  rs0 = slot of the first block n0 of the current tile's window,
    // advanced by one per tile; the slots of the window's other blocks by add / compare / select (no division in the loop)
    auto slot_add = [&](int s, int k) { const int x = s + k; return x >= RING_BLKS ? x - RING_BLKS : x; };
    // G^T block j of the window whose first block sits in ring slot s0 (lane = query, registers = distances) -> fp16 skew buffer
    auto gblock = [&](int j, int s0, const bf16x8 (&bq)[4]) {
        f32x16 g;
#pragma unroll
        for (int t = 0; t < 16; t++) g[t] = 0.f;
        const char* rb = sR + slot_add(s0, j) * RBLK_BYTES;
        bf16x8 a[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) a[ks] = *reinterpret_cast<const bf16x8*>(rb + (rows ^ (ks << 5)));
#pragma unroll
        for (int ks = 0; ks < 4; ks++) g = mfma32(a[ks], bq[ks], g);
        char* gw = sG + r * GP + (32 * j + 4 * hh) * 2;
#pragma unroll
        for (int grp = 0; grp < 4; grp++) {
            const f32x4v v4 = {g[4 * grp], g[4 * grp + 1], g[4 * grp + 2], g[4 * grp + 3]};
            *reinterpret_cast<f16x4*>(gw + 16 * grp) = __builtin_convertvector(v4, f16x4);
        }
    };

    int rs0;                                    // ring slot of the current tile's first distance block
    {
        const int n0 = (it_lo * QT - P0 - KBLK) >> 5;
        rs0 = (n0 + RING_OFF) % RING_BLKS;
        load_stage(it_lo, n0 + 8);
        __syncthreads();                        // biases in LDS
        store_stage(0, slot_add(rs0, 8));       // (block n0 + 8 is one of the nine loaded below: the same bytes twice)
        // the nine Rd blocks of the first window: 9 x 256 sixteen-byte chunks, loads first
        constexpr int NPR = (9 * 256 + NT - 1) / NT;
        u32x4 pr_[NPR];
#pragma unroll
        for (int n = 0; n < NPR; n++) {
            const int c = tid + n * NT;
            int d = 32 * n0 + (c >> 3);
            d = d < 0 ? 0 : (d > M - 1 ? M - 1 : d);
            pr_[n] = (c < 9 * 256) ? *reinterpret_cast<const u32x4*>(rbase + (size_t)d * p.rd_rs + (c & 7) * 8) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int n = 0; n < NPR; n++) {
            const int c = tid + n * NT;
            if (c < 9 * 256) {
                const int slot = slot_add(rs0, c >> 8);
                *reinterpret_cast<u32x4*>(sR + slot * RBLK_BYTES + soff((c >> 3) & 31, c & 7)) = pr_[n];
            }
        }
    }
    __syncthreads();
    {                                           // the nine G blocks of the first tile: wave w takes block w, wave 0 also block 8
        bf16x8 bq[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) bq[ks] = *reinterpret_cast<const bf16x8*>(sQ + QIMG + (rowq ^ (ks << 5)));
        gblock(w, rs0, bq);
        if (w == 0) gblock(8, rs0, bq);
    }
    load_stage(min(it_lo + 1, it_hi), ((it_lo * QT - P0 - KBLK) >> 5) + 9);                      // stored at the top of the first tile
    __syncthreads();

    f32x16 ak[2], av[2];                        // dK^T, dV^T : [e half][e][key]
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
        for (int j = 0; j < 16; j++) { ak[e][j] = 0.f; av[e][j] = 0.f; }
    // dRd: racc[v] = this wave's 16 x 16 slice of the fp32 sum of the distance block at window position v.  The window slides by one
    // block per tile, so the MFMA of position v writes the registers of position v - 1 (an MFMA's destination need not be its C
    // operand): no register moves; position 0 leaves the window with the tile and goes to the park buffer, position 8 starts at zero.
    // Until round 5 a whole block lived in ONE wave (32 registers) for the nine tiles it stayed in the window, and that wave alone
    // wrote it out -- thirty-two memory operations at the end of its phase B once per tile, with seven waves waiting at barrier 2.
    f32x4 racc[9];
#pragma unroll
    for (int v = 0; v < 9; v++) racc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cw = 0.f, cr = 0.f;                   // running column sums of this wave's dQw / dQr pieces (d r_w_bias / d r_r_bias)

    // dRd flush.  A block leaves the window once per tile, in the registers of ONE wave.  Thirty-two atomics from that wave stalled it
    // (issue stalls beyond 16 outstanding) and sat in its in-order vmcnt queue in front of the next tile's staged rows; parked in LDS
    // instead (32 ds_write_b32), every wave adds four rows of it in the next tile: the same four atomics per wave and tile on every path.
    // (num_records = this head's last element: an atomic whose lane offset lies beyond it is dropped by the address unit -- that is
    // where the blocks outside [0, M) go, so that every path still issues the same operations without adding zeros to memory:
    // until round 5 they went onto block 0, 0.4 GB of atomic traffic per layer onto one 8 KB target)
    const int ld4 = p.drd_ld * 4;
    const __amdgpu_buffer_rsrc_t rs_drd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.drd + h * 64), 0, (int)((32 * MB - 1) * ld4 + 256), 0x00020000);
    constexpr int DRD_OOB = 0x7ffff000;
    auto drd_park = [&](const f32x4& leave) {   // this wave's slice of the block that leaves the window -> sPark[distance][e]
#pragma unroll
        for (int t = 0; t < 4; t++) sPark[(16 * (w & 1) + 4 * (l >> 4) + t) * 64 + 16 * (w >> 1) + (l & 15)] = leave[t];
    };
    // the slice straight from the registers: one atomic instruction covers 4 distances x 16 elements = four 64-byte segments -- the
    // granule the memory-side atomic unit works on (MI355X_MICROARCH.md, Global float atomics: a 256-byte instruction leaves L2 as
    // four 64-byte requests) -- and every wave issues the same four per tile
    auto drd_add = [&](const f32x4& leave, int nL) {
        const bool ok = nL >= 0 && nL < MB;
        const float f = 1.f / p.scale_log2e;                 // the Qr image carries scale * log2(e)
        const int so = (ok ? 32 * nL : 0) + 16 * (w & 1);
        const int vo = ok ? 4 * (l >> 4) * ld4 + (16 * (w >> 1) + (l & 15)) * 4 : DRD_OOB;      // out of range: dropped
#pragma unroll
        for (int t = 0; t < 4; t++)
#ifndef MXL_ABL_NO_ATOMICS
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(leave[t] * f, rs_drd, vo, (so + t) * ld4, 0);
#else
            asm volatile("" :: "v"(leave[t]));
#endif
    };
    float pk[4];                                // this wave's four rows of the parked block (row 4 w + i, column l)
    auto park_read = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) pk[i] = sPark[(4 * w + i) * 64 + l];
    };
    auto park_add = [&](int nL) {               // block nL (out of range: dropped -- the operation count stays the same)
        const bool ok = nL >= 0 && nL < MB;
        const int so = ok ? 32 * nL * ld4 : 0;
        const float f = 1.f / p.scale_log2e;                 // the Qr image carries scale * log2(e)
#pragma unroll
        for (int i = 0; i < 4; i++)
#ifndef MXL_ABL_NO_ATOMICS
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(pk[i] * f, rs_drd, ok ? l * 4 : DRD_OOB, so + (4 * w + i) * ld4, 0);
#else
            asm volatile("" :: "v"(pk[i]));
#endif
    };
    auto drd_flush_direct = [&](int nfirst) {   // after the sweep: the eight blocks still in registers (positions 0 .. 7 = blocks nfirst ..)
#ifndef MXL_ABL_NO_ATOMICS
        const float f = 1.f / p.scale_log2e;
        const int vo = 4 * (l >> 4) * ld4 + (16 * (w >> 1) + (l & 15)) * 4;
#pragma unroll
        for (int v = 0; v < 8; v++) {
            const int nb = nfirst + v;
            if (nb >= 0 && nb < MB) {
#pragma unroll
                for (int t = 0; t < 4; t++)
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(racc[v][t] * f, rs_drd, vo, (32 * nb + 16 * (w & 1) + t) * ld4, 0);
            }
        }
#endif
    };
    // (the park buffer starts as zeros: the first tile adds them)
    if (FUSED_PARK) {
        for (int i = tid; i < PARK_BYTES / 4; i += NT) sPark[i] = 0.f;
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < VM_PER_TILE; i++)       // the loop is entered with as many operations behind the first staged rows as every later tile has
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(0.f, rs_drd, DRD_OOB, i * ld4, 0);

    int cur = 0;
    STAMP(15)
#pragma unroll 1
    for (int it = it_lo; it <= it_hi; it++) {
        const int I = it * QT;
        const bool more = it < it_hi;
        // The lane constants are "redefined" by an empty asm at the top of every tile: what hipcc derives from them (rowq ^ (ks << 5),
        // tkb ^ 64, ka0 + slot offsets ...) then cannot be hoisted out of the loop -- hoisted, each derived address was one more
        // register live across the whole tile, and the kernel spilled (the V fragments, the staged rows) to hold them.
        asm volatile("" : "+v"(rowq), "+v"(rows), "+v"(tkb), "+v"(xw), "+v"(yt0), "+v"(qb0), "+v"(xa0), "+v"(xa1), "+v"(ka0), "+v"(ka1), "+v"(yq0), "+v"(krow));
        const char* sQw = sQ + cur * QSET;
        const char* sQr = sQw + QIMG;
        const char* sDO = sQr + QIMG;
        const float* sLse = reinterpret_cast<const float*>(sDO + QIMG);
        const int n0 = (I - P0 - KBLK) >> 5;    // first distance block of this tile's window: column c = distance - 32 n0
        // The rows of tile it + 1 were requested a whole tile ago: their LDS images (into the Q set and the ring slot that phase
        // B of tile it - 1 was the last to read) go in now, and the rows of tile it + 2 are requested.
#ifndef MXL_ABL_NO_STAGING
#ifdef MXL_STAMP
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // (diagnostic build: the wait for the staged rows as a segment of its own)
        STAMP(14)
#endif
        // (unconditional: on the last tile the rows of tile it_hi go, once more, into the image and the ring slot nobody reads again)
        store_stage(cur ^ 1, slot_add(rs0, 9));
#ifdef MXL_ABL_HOT_LOADS
        load_stage(it_lo, n0 + 10);
#else
        load_stage(min(it + 2, it_hi), n0 + 10);
#endif
#endif
        if (FUSED_PARK) park_read();            // the block parked by the previous tile (read before barrier 1, rewritten behind it)
        STAMP(0)

        // =============================== phase A: scores, dS -> X / Y ===============================
        const int dmin_w = I - Pw - 31, dmax_w = I + 31 - Pw;
        const bool active = (dmax_w >= 0) && (dmin_w <= M - 1) && (Pw < T);
        const bool full = (dmin_w >= 0) && (dmax_w <= M - 1) && (Pw + 31 < T);
        uint32_t dsw[8], prw[8];                // dS and P as bf16 pairs (2m, 2m + 1): live across barrier 1 (dV / dK run behind it)
        bf16x8 tdo_[2][2], tqw_[2][2];          // transposed dO / Qw fragments (A operands of dV^T / dK^T)
        // dV^T += dO^T . P ; dK^T += Qw^T . dS   (A through transposed reads, accumulator-permuted k order)
        auto dvdk = [&]() {
#ifndef MXL_ABL_NO_DVDK
#pragma unroll
            for (int st = 0; st < 2; st++) {
                const u32x4 pw = {prw[4 * st], prw[4 * st + 1], prw[4 * st + 2], prw[4 * st + 3]};
                const u32x4 dw = {dsw[4 * st], dsw[4 * st + 1], dsw[4 * st + 2], dsw[4 * st + 3]};
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    av[e] = mfma32(tdo_[st][e], __builtin_bit_cast(bf16x8, pw), av[e]);
                    ak[e] = mfma32(tqw_[st][e], __builtin_bit_cast(bf16x8, dw), ak[e]);
                }
            }
#endif
        };
        auto tr_frags = [&]() {
#pragma unroll
            for (int st = 0; st < 2; st++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int lo = 2048 * st + (tkb ^ (e << 6)), hi = 2048 * st + 1024 + (tkb ^ 16 ^ (e << 6));
                    tdo_[st][e] = tr_pair(sDO + lo, sDO + hi);
                    tqw_[st][e] = tr_pair(sQw + lo, sQw + hi);
                }
        };
        if (active) {
            // every LDS read of the score phase first, then the chains
            // S chain first (operands: the Qw rows and the wave's K rows), then the dP chain (dO rows; V in registers): one chain's
            // operands at a time -- 32 registers instead of 48 beside the two accumulators -- and a single accumulation chain of this
            // MFMA runs at the issue rate (MI355X_MICROARCH.md, cycle constants).  The dO rows are requested ahead of the S MFMAs.
            bf16x8 aq[4], ad[4], kfl[4];
            f32x16 s, dp;
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {     // -lse of the tile's queries: the chain's start value
                const f32x4 cl = *reinterpret_cast<const f32x4*>(sLse + 8 * grp + 4 * hh);
#pragma unroll
                for (int t = 0; t < 4; t++) s[4 * grp + t] = cl[t];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                aq[ks] = *reinterpret_cast<const bf16x8*>(sQw + (rowq ^ (ks << 5)));
                // K as the B operand of S (lane = key, k = 16 ks + 8 hh ..): re-read per tile from the LDS image (no 16 registers to
                // hold it for the sweep), whose swizzle makes the row reads conflict-free
                kfl[ks] = *reinterpret_cast<const bf16x8*>(sK + (krow ^ (ks << 5)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {     // -scale * delta
                const f32x4 cd = *reinterpret_cast<const f32x4*>(sLse + QT + 8 * grp + 4 * hh);
#pragma unroll
                for (int t = 0; t < 4; t++) dp[4 * grp + t] = cd[t];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ks++) ad[ks] = *reinterpret_cast<const bf16x8*>(sDO + (rowq ^ (ks << 5)));
            PRIO_UP();
#pragma unroll
            for (int ks = 0; ks < 4; ks++) s = mfma32(aq[ks], kfl[ks], s);
            __builtin_amdgcn_sched_barrier(0);
#if FUSED_VF_RELOAD
            bf16x8 vfl[4];
            {
                const bf16_t* vp = vbase + (size_t)(kok ? Pw + r - p0 : 0) * p.kv_rs;
#pragma unroll
                for (int ks = 0; ks < 4; ks++) vfl[ks] = *reinterpret_cast<const bf16x8*>(vp + 16 * ks + 8 * hh);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ks++) dp = mfma32(ad[ks], vfl[ks], dp);
#else
#pragma unroll
            for (int ks = 0; ks < 4; ks++) dp = mfma32(ad[ks], vf[ks], dp);
#endif
            PRIO_DOWN();
            // the positional term of the wave's cells: sixteen 16-bit reads of the skew buffer, issued behind the chains' MFMAs (their
            // 256 cycles cover the reads; ahead of the chains' operand reads the sixteen result registers were live beside the 48 operand
            // registers -- the kernel's peak)
            uint32_t bdu[16];
            gskew_read16(gRb, bdu);
            STAMP(1)
            gskew_wait();
            STAMP(2)
            f32x16 pr;
            const bool fl = __builtin_amdgcn_readfirstlane((int)full) != 0;
            if (fl) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const float pv = __builtin_amdgcn_exp2f(add_f16(s[j], bdu[j]));
                    pr[j] = pv;
                    s[j] = pv * dp[j];
                }
            } else {
                const int pk = Pw + r;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int ii = (j & 3) + 8 * (j >> 2) + 4 * hh;
                    const int d = I + ii - pk;
                    const bool valid = (d >= 0) && (d <= M - 1) && kok;
                    // (the exponent is replaced, not the result: with `valid ? exp2(x) : 0` hipcc branches around every
                    // exponential -- sixteen exec-masked blocks on the tiles that cross the causal diagonal)
                    const float x = add_f16(s[j], bdu[j]);
                    const float pv = __builtin_amdgcn_exp2f(valid ? x : -1.0e30f);
                    pr[j] = pv;
                    s[j] = pv * (valid ? dp[j] : 0.f);
                }
            }
#pragma unroll
            for (int m = 0; m < 8; m++) { dsw[m] = pack2bf(s[2 * m], s[2 * m + 1]); prw[m] = pack2bf(pr[2 * m], pr[2 * m + 1]); }
            // the transposed dO / Qw fragments: requested ahead of the twenty LDS writes below, consumed behind barrier 1
            if (FUSED_TRF_EARLY || !FUSED_DVDK_LATE) tr_frags();
#pragma unroll
            for (int grp = 0; grp < 4; grp++)
                *reinterpret_cast<u32x2*>(sX + (xw ^ (grp << 4))) = u32x2{dsw[2 * grp], dsw[2 * grp + 1]};
            yskew_write16(yWb, dsw);
            if (!FUSED_DVDK_LATE) dvdk();
        } else {
            // no valid cell for this wave's keys in this tile: its rows of X and its cells of Y still have to read as zero
            const uint32_t z[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
            for (int grp = 0; grp < 4; grp++) *reinterpret_cast<u32x2*>(sX + (xw ^ (grp << 4))) = u32x2{0u, 0u};
            yskew_write16(yWb, z);
            if (FUSED_DVDK_LATE) {
                // "define" what the products behind barrier 1 read (they are skipped on this path): left undefined here, the 48
                // registers count as live around the whole loop.  No instruction is emitted.
#pragma unroll
                for (int m = 0; m < 8; m++) asm volatile("" : "=v"(dsw[m]), "=v"(prw[m]));
#pragma unroll
                for (int st = 0; st < 2; st++)
#pragma unroll
                    for (int e = 0; e < 2; e++) asm volatile("" : "=v"(tdo_[st][e]), "=v"(tqw_[st][e]));
            }
        }
        if (FUSED_PARK) park_add(n0 - 1);       // the block that left the window with the previous tile
        STAMP(3)
        lds_barrier();
        STAMP(4)

        // =============================== phase B: dV / dK, dq piece, dRd block, next tile's G ===============================
        {
            // unit n = one 16x16x32 MFMA of the dq piece with its two operand fragments: n < 8 key block n (A = X^T rows, B = K^T),
            // n >= 8 distance block n - 8 of the window (A = Y rows, B = Rd^T from ring slot rs0 + n - 8)
            bf16x8 ua[26], ub[17], qrb;
            int sl = rs0;
            auto rd_unit = [&](int n) {
                if (n < 8) {
                    ub[n] = tr_pair(sK + n * 4096 + ka0, sK + n * 4096 + ka1);
                    ua[n] = tr_pair(sX + n * 2048 + xa0, sX + n * 2048 + xa1);
                } else if (n < 17) {
                    const char* rb = sR + sl * RBLK_BYTES;
                    ub[n] = tr_pair(rb + ka0, rb + ka1);
                    ua[n] = *reinterpret_cast<const bf16x8*>(sY + yq0 + (n - 8) * 64);
                    sl = (sl == RING_BLKS - 1) ? 0 : sl + 1;
                } else {                        // n >= 17: dRd, window position n - 17 (A = Y^T fragment; B = Qr^T, shared)
                    if (n == 17) qrb = tr_pair(sQr + qb0, sQr + qb0 + 2048);
                    ua[n] = tr_pair(sY + yt0 + (n - 17) * 64, sY + yt0 + 16 * YP + (n - 17) * 64);
                }
            };
            if (FUSED_DVDK_LATE && !FUSED_TRF_EARLY && active) tr_frags();
#pragma unroll
            for (int n = 0; n < FUSED_PF; n++) rd_unit(n);
            UNIT_FENCE();
            PRIO_UP();
            if (FUSED_DVDK_LATE && active) dvdk();
            PRIO_DOWN();
            UNIT_FENCE();
            STAMP(5)
            f32x4 aw4[2], ar4[2];               // two independent accumulator chains per half of the contraction
#pragma unroll
            for (int c = 0; c < 2; c++) { aw4[c] = f32x4{0.f, 0.f, 0.f, 0.f}; ar4[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            f32x4 leave;
            PRIO_UP_B(2);
#pragma unroll
            for (int n = 0; n < 26; n++) {
                if (n < 8) aw4[n & 1] = mfma16(ua[n], ub[n], aw4[n & 1]);
                else if (n < 17) ar4[n & 1] = mfma16(ua[n], ub[n], ar4[n & 1]);
                else if (n == 17) leave = mfma16(ua[n], qrb, racc[0]);
                else racc[n - 18] = mfma16(ua[n], qrb, racc[n - 17]);
                if (n + FUSED_PF < 26) rd_unit(n + FUSED_PF);
                UNIT_FENCE();
            }
            PRIO_DOWN_B(2);
            racc[8] = f32x4{0.f, 0.f, 0.f, 0.f};
            STAMP(6)
            // the next tile's G operands (its Qr image was stored before barrier 1; its window starts one ring slot further)
            const char* sQrN = sQ + (cur ^ 1) * QSET + QIMG;
            const int rs1 = slot_add(rs0, 1);
            // (read on the last tile too, where nothing uses them: defined under `if (more)` and used under a second `if (more)` the
            // 32 registers count as live on the path around the definition, i.e. across the whole loop)
            bf16x8 gq_[4], ga[4];
            {
                const char* rb = sR + slot_add(rs1, w) * RBLK_BYTES;
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    gq_[ks] = *reinterpret_cast<const bf16x8*>(sQrN + (rowq ^ (ks << 5)));
                    ga[ks] = *reinterpret_cast<const bf16x8*>(rb + (rows ^ (ks << 5)));
                }
            }
            UNIT_FENCE();
            // this tile's partial dq -> slab of this key block (rows = queries 16 ih0 + 4 (l >> 4) + t, column = element 16 eq + (l & 15))
            {
#pragma unroll
                for (int t = 0; t < 4; t++) { aw4[0][t] += aw4[1][t]; ar4[0][t] += ar4[1][t]; }
                const int x = I - M - 254 - p0;
                const int kb_lo = x <= 0 ? 0 : (x + 255) >> 8;
                // The slabs are bf16 (half the bytes of the round trip through the finish kernel, which sums them in fp32): a lane owns one
                // column of four rows, so neighbouring lanes trade values (DPP quad swap) and the even lane stores the packed pair of
                // rows 0 / 1, the odd lane that of rows 2 / 3 -- two 4-byte stores per lane instead of four.
                const int rowb = p.H * 64 * (int)sizeof(slab_t);              // bytes per dq row of a slab
                slab_t* tile = p.slab + (size_t)(bx_ - kb_lo) * p.slab_stride + ((size_t)b * T + I + 16 * ih0) * (size_t)(p.H * 64) + h * 64 + 16 * eq;
                const __amdgpu_buffer_rsrc_t rs_sl = __builtin_amdgcn_make_buffer_rsrc((void*)tile, 0, -1, 0x00020000);
#if MXL_SLAB_BF16
                const bool odd = (l & 1) != 0;
                const int vo = (4 * g16 + (odd ? 2 : 0)) * rowb + ((l & 15) >> 1) * 4;
                float v[4], nb[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    v[t] = aw4[0][t] + ar4[0][t];
                    nb[t] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v[t]), 0xB1, 0xf, 0xf, true));   // lane ^ 1
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {       // even lanes: rows 0, 1 (own column first); odd lanes: rows 2, 3 (the neighbour's first)
                    const float lo = odd ? nb[2 + u] : v[u], hi = odd ? v[2 + u] : nb[u];
                    __builtin_amdgcn_raw_buffer_store_b32((int)pack2bf(lo, hi), rs_sl, vo, u * rowb, 0);
                }
#else
                const int vo = 4 * g16 * rowb + (l & 15) * 4;
#pragma unroll
                for (int t = 0; t < 4; t++)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, aw4[0][t] + ar4[0][t]), rs_sl, vo, t * rowb, 0);
#endif
                cw += (aw4[0][0] + aw4[0][1]) + (aw4[0][2] + aw4[0][3]);
                cr += (ar4[0][0] + ar4[0][1]) + (ar4[0][2] + ar4[0][3]);
            }
            STAMP(7)
            if (FUSED_PARK) drd_park(leave); else drd_add(leave, n0);
            STAMP(8)
            // the next tile's G block(s): wave w block w of its window, one wave also block 8
            if (more) {
                f32x16 g;
#pragma unroll
                for (int t = 0; t < 16; t++) g[t] = 0.f;
                PRIO_UP_B(4);
#pragma unroll
                for (int ks = 0; ks < 4; ks++) g = mfma32(ga[ks], gq_[ks], g);
                PRIO_DOWN_B(4);
                char* gw = sG + r * GP + (32 * w + 4 * hh) * 2;
#pragma unroll
                for (int grp = 0; grp < 4; grp++) {
                    const f32x4v v4 = {g[4 * grp], g[4 * grp + 1], g[4 * grp + 2], g[4 * grp + 3]};
                    *reinterpret_cast<f16x4*>(gw + 16 * grp) = __builtin_convertvector(v4, f16x4);
                }
                if (w == ((n0 + 4) & 7)) {      // the ninth block of the window, on the wave four places from the one that parks this tile
                    const char* rb = sR + slot_add(rs1, 8) * RBLK_BYTES;
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) ga[ks] = *reinterpret_cast<const bf16x8*>(rb + (rows ^ (ks << 5)));
                    UNIT_FENCE();
                    f32x16 g8;
#pragma unroll
                    for (int t = 0; t < 16; t++) g8[t] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) g8 = mfma32(ga[ks], gq_[ks], g8);
                    char* gw8 = sG + r * GP + (32 * 8 + 4 * hh) * 2;
#pragma unroll
                    for (int grp = 0; grp < 4; grp++) {
                        const f32x4v v4 = {g8[4 * grp], g8[4 * grp + 1], g8[4 * grp + 2], g8[4 * grp + 3]};
                        *reinterpret_cast<f16x4*>(gw8 + 16 * grp) = __builtin_convertvector(v4, f16x4);
                    }
                }
            }
            STAMP(9)
            STAMP(10)
        }
        lds_barrier();
        STAMP(12)
        cur ^= 1;
        rs0 = (rs0 == RING_BLKS - 1) ? 0 : rs0 + 1;
    }
    // the block parked by the last tile, then the blocks still in registers
    {
        const int n0_end = ((it_hi + 1) * QT - P0 - KBLK) >> 5;
        if (FUSED_PARK && it_lo <= it_hi) { park_read(); park_add(n0_end - 1); }
    }
    drd_flush_direct(((it_hi + 1) * QT - P0 - KBLK) >> 5);
    STAMP(13)
    STAMP_FLUSH

    // ---- epilogue: bias gradients (stored keys' part), dk, dv
    {
        float a = cw, c = cr;
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
        if (l < 16) {
            atomicAdd(p.d_rwb + h * 64 + 16 * eq + l, a);
            atomicAdd(p.d_rrb + h * 64 + 16 * eq + l, c);
        }
    }
    if (kok) {
        const float fk = 1.f / p.scale_log2e, fv = 1.f / p.scale;      // dK was accumulated against scale*log2(e)*Qw, dV against scale*dO
        const size_t srow = (size_t)(Pw + r - p0);
        bf16_t* dkp = p.dk + (size_t)b * p.dkv_bs + srow * p.dkv_rs + (size_t)h * 64;
        bf16_t* dvp = p.dv + (size_t)b * p.dkv_bs + srow * p.dkv_rs + (size_t)h * 64;
#pragma unroll
        for (int e = 0; e < 2; e++) {
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                const int e0 = 32 * e + 8 * grp + 4 * hh;
                const u32x2 wk = {pack2bf(ak[e][4 * grp] * fk, ak[e][4 * grp + 1] * fk), pack2bf(ak[e][4 * grp + 2] * fk, ak[e][4 * grp + 3] * fk)};
                const u32x2 wv = {pack2bf(av[e][4 * grp] * fv, av[e][4 * grp + 1] * fv), pack2bf(av[e][4 * grp + 2] * fv, av[e][4 * grp + 3] * fv)};
                *reinterpret_cast<u32x2*>(dkp + e0) = wk;
                *reinterpret_cast<u32x2*>(dvp + e0) = wv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dq[b,i,:] = bf16( sum over the key blocks that see query tile i/32 of their slab rows  +  phantom term ),
// phantom term (zero memories, oph != NULL):  -scale * delta[b,h,i] * 2^(mph[b,h,i] - lse[b,h,i] log2 e) * oph[b,i,h,:]
// -- the dQr of every key position below the first stored one, from the forward's value-sum (relattn_fwd.hip).  Elementwise and
// HBM-bound: a workgroup takes sixteen query rows (they share the slab range), a thread 8 consecutive elements of a row, four such
// items per pass; the (at most M/256 + 1) slab reads of an item are independent 16-byte loads (the slabs are bf16, the sum fp32).
// The column sums of the phantom term are those cells' part of d r_r_bias (d r_r_bias = sum_i dQr_i): kept in the registers of the
// thread that owns the column chunk, one float atomic per column and workgroup at the end (measured on the way: ds_add_f32 per
// element doubled the kernel's time, and so did a flush per row group -- 6 M float atomics on 3 KB of addresses).
// ---------------------------------------------------------------------------------------------------------------
struct FinP {
    const slab_t* slab; const bf16_t* oph; const float *mph, *lse, *delta;
    bf16_t* dq; float* d_rrb;
    int B, T, H, M, Kc;
    long long slab_stride, o_bs, dq_bs; int o_rs, dq_rs;
    float scale;
};
#ifndef FIN_K_
#define FIN_K_ 1
#endif
constexpr int FIN_K = FIN_K_;                   // rows per thread and pass
constexpr int FIN_ROWS = 16;                    // query rows per row group (half a tile: the rows share the slab range)
__global__ __launch_bounds__(256) void relattn_dq_finish_kernel(FinP p) {
    __shared__ float s_red[256 * 8];
    const int d = p.H * 64, nch = d >> 3;
    const int per_b = p.T / FIN_ROWS;
    const int p0 = p.T - p.Kc;
    const int nkb = (p.Kc + KBLK - 1) / KBLK;
    const bool colsum = p.oph != nullptr && p.d_rrb != nullptr;
    const slab_t* __restrict__ slab = p.slab;
    const int t = threadIdx.x;
    // A thread owns ONE 8-element column chunk for the whole launch (so the column sums of the phantom term stay in its registers)
    // and walks rows: cw = min(nch, 256) chunks side by side, kr = 256 / cw rows side by side (d = 768: 96 x 2, 192 threads busy);
    // a workgroup walks row groups with the grid's stride.  One float atomic per column and WORKGROUP at the end.
    for (int cb = 0; cb < nch; cb += 256) {
        const int cw = min(256, nch - cb), kr = 256 / cw;
        const int c = cb + t % cw, rsub = t / cw;
        const bool act = rsub < kr;
        float cs[8];
#pragma unroll
        for (int j = 0; j < 8; j++) cs[j] = 0.f;
        for (int rg = blockIdx.x; rg < p.B * per_b; rg += gridDim.x) {
            const int b = rg / per_b, i0 = (rg % per_b) * FIN_ROWS;
            const int I = i0 & ~31;
            const int x = I - p.M - 254 - p0;
            const int kb_lo = x <= 0 ? 0 : (x + 255) >> 8;
            const int kb_hi = min(nkb - 1, (I + 31 - p0) >> 8);
            const int nsl = kb_hi - kb_lo + 1;
            for (int rbase = 0; rbase < FIN_ROWS; rbase += FIN_K * kr) {
                float acc[FIN_K][8];
                int i_[FIN_K];
                bool ok[FIN_K];
#pragma unroll
                for (int k = 0; k < FIN_K; k++) {
                    const int rr = rbase + rsub + kr * k;
                    ok[k] = act && rr < FIN_ROWS;
                    i_[k] = i0 + (ok[k] ? rr : 0);           // (idle slots re-read the group's first row: no store below)
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[k][j] = 0.f;
                }
#pragma unroll 5
                for (int s = 0; s < nsl; s++) {
#if MXL_SLAB_BF16
                    bf16x8 a[FIN_K];
#pragma unroll
                    for (int k = 0; k < FIN_K; k++)
                        a[k] = *reinterpret_cast<const bf16x8*>(slab + (size_t)s * p.slab_stride + ((size_t)b * p.T + i_[k]) * (size_t)d + c * 8);
#pragma unroll
                    for (int k = 0; k < FIN_K; k++)
#pragma unroll
                        for (int j = 0; j < 8; j++) acc[k][j] += bf2f((bf16_t)a[k][j]);
#else
                    f32x4 a[FIN_K][2];
#pragma unroll
                    for (int k = 0; k < FIN_K; k++) {
                        const float* sp = slab + (size_t)s * p.slab_stride + ((size_t)b * p.T + i_[k]) * (size_t)d + c * 8;
                        a[k][0] = *reinterpret_cast<const f32x4*>(sp); a[k][1] = *reinterpret_cast<const f32x4*>(sp + 4);
                    }
#pragma unroll
                    for (int k = 0; k < FIN_K; k++)
#pragma unroll
                        for (int j = 0; j < 4; j++) { acc[k][j] += a[k][0][j]; acc[k][4 + j] += a[k][1][j]; }
#endif
                }
#pragma unroll
                for (int k = 0; k < FIN_K; k++) {
                    if (!ok[k]) continue;
                    const int i = i_[k];
                    if (p.oph) {
                        const size_t sidx = ((size_t)b * p.H + (c >> 3)) * p.T + i;
                        // (exponent clamped: a query with no phantom cell has oph = 0, mph = 0, and with lse below about -88 the
                        // factor would overflow to inf -- inf * 0 = NaN in dq and, through the column sums, in d r_r_bias)
                        const float f = -p.scale * p.delta[sidx] * __builtin_amdgcn_exp2f(fminf(p.mph[sidx] - p.lse[sidx] * LOG2E, 126.f));
                        const bf16x8 o = *reinterpret_cast<const bf16x8*>(p.oph + (size_t)b * p.o_bs + (size_t)i * p.o_rs + c * 8);
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const float v = f * bf2f((bf16_t)o[j]);
                            acc[k][j] += v;
                            cs[j] += v;
                        }
                    }
                    const u32x4 wq = {pack2bf(acc[k][0], acc[k][1]), pack2bf(acc[k][2], acc[k][3]), pack2bf(acc[k][4], acc[k][5]),
                                      pack2bf(acc[k][6], acc[k][7])};
                    *reinterpret_cast<u32x4*>(p.dq + (size_t)b * p.dq_bs + (size_t)i * p.dq_rs + c * 8) = wq;
                }
            }
        }
        if (colsum) {           // the kr row lanes of a column chunk meet in LDS; thread a = (chunk, element) in memory order adds their sum
            __syncthreads();
            if (act) {
#pragma unroll
                for (int j = 0; j < 8; j++) s_red[(rsub * 8 + j) * cw + (c - cb)] = cs[j];
            }
            __syncthreads();
            for (int a = t; a < cw * 8; a += blockDim.x) {
                float v = 0.f;
                for (int q = 0; q < kr; q++) v += s_red[(q * 8 + (a & 7)) * cw + (a >> 3)];
                atomicAdd(p.d_rrb + cb * 8 + a, v);
            }
        }
    }
}

}  // namespace

#ifdef MXL_STAMP
extern "C" int mxl_debug_fused_stamps(unsigned long long* host_out16) {
    hipError_t e = hipMemcpyFromSymbol(host_out16, HIP_SYMBOL(g_fused_stamps), sizeof(unsigned long long) * 16);
    if (e != hipSuccess) return (int)e;
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_fused_stamps), z, sizeof(z));
    return (int)e;
}
#endif

extern "C" size_t mxl_relattn_bwd_fused_ws_bytes(int B, int T, int H, int dh, int M) {
    if (B <= 0 || T <= 0 || H <= 0 || dh != 64 || M <= 0) return 0;
    return (size_t)((M + KBLK - 1) / KBLK + 1) * (size_t)B * T * H * 64 * sizeof(slab_t);     // a query tile's M + 31 keys touch at most that many key blocks
}

extern "C" int mxl_relattn_bwd_fused(const void* q, const void* k, const void* v, const void* rd, const float* r_w_bias,
                                     const float* r_r_bias, const void* out, const void* dout, const float* lse, float* delta,
                                     void* dq, void* dk, void* dv, float* d_rd, int drd_ld, float* d_r_w_bias, float* d_r_r_bias,
                                     const void* oph, const float* mph, void* ws, int B, int T, int H, int dh, int M, int Kc,
                                     long long q_bs, int q_rs, long long kv_bs, int kv_rs, int rd_rs, long long o_bs, int o_rs,
                                     long long dq_bs, int dq_rs, long long dkv_bs, int dkv_rs, float scale, int defer_finish,
                                     void* stream) {
    MXL_CHECK_ARG(q && k && v && rd && r_w_bias && r_r_bias && out && dout && lse && delta && dq && dk && dv && d_rd && d_r_w_bias &&
                  d_r_r_bias && ws);
    if (dh != 64) return MXL_EUNSUPPORTED;
    MXL_CHECK_ARG(B > 0 && T > 0 && H > 0 && M > 0 && Kc >= T && Kc <= M + T && drd_ld >= H * 64);
    if ((T % 32) != 0 || (M % 32) != 0 || (Kc % 32) != 0) return MXL_EUNSUPPORTED;
    // key positions below the first stored one are phantom distances: the caller must have their dq part in oph / mph (zero memories),
    // unless every visible key is stored (Kc == M + T)
    MXL_CHECK_ARG(Kc == M + T || (oph && mph));
    MXL_CHECK_ARG((q_rs % 8) == 0 && (kv_rs % 8) == 0 && (rd_rs % 8) == 0 && (o_rs % 8) == 0 && (dq_rs % 8) == 0 && (dkv_rs % 4) == 0);
    MXL_CHECK_ARG((q_bs % 8) == 0 && (kv_bs % 8) == 0 && (o_bs % 8) == 0 && (dq_bs % 8) == 0 && (dkv_bs % 4) == 0);
    MXL_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)rd % 16) == 0 &&
                  ((uintptr_t)out % 16) == 0 && ((uintptr_t)dout % 16) == 0 && ((uintptr_t)dq % 16) == 0 && ((uintptr_t)ws % 16) == 0 &&
                  (oph == nullptr || ((uintptr_t)oph % 16) == 0));
    hipStream_t s = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&relattn_bwd_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const bool delta_ready = (defer_finish & 2) != 0;       // the caller has filled `delta` (mxl_gemm_bf16_headdot)
    defer_finish &= 1;
    if (!delta_ready) {
        mxl_kt::Scope kt(MXL_KT_RELATTN_DELTA, s);
        hipLaunchKernelGGL(fused_delta_kernel, dim3((B * T + 3) / 4), dim3(256), 0, s, (const bf16_t*)out, (const bf16_t*)dout, delta, B, T,
                           H, o_bs, o_rs);
    }
    FusedP p;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.rd = (const bf16_t*)rd; p.dout = (const bf16_t*)dout;
    p.rwb = r_w_bias; p.rrb = r_r_bias; p.lse = lse; p.delta = delta;
    p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.slab = (slab_t*)ws; p.drd = d_rd; p.d_rwb = d_r_w_bias; p.d_rrb = d_r_r_bias;
    p.B = B; p.T = T; p.H = H; p.M = M; p.Kc = Kc;
    p.q_bs = q_bs; p.kv_bs = kv_bs; p.o_bs = o_bs; p.dkv_bs = dkv_bs; p.slab_stride = (long long)B * T * H * 64;
    p.q_rs = q_rs; p.kv_rs = kv_rs; p.rd_rs = rd_rs; p.o_rs = o_rs; p.dkv_rs = dkv_rs; p.drd_ld = drd_ld;
    p.scale = scale; p.scale_log2e = scale * LOG2E;
    {
        mxl_kt::Scope kt(MXL_KT_RELATTN_FUSED, s);
        hipLaunchKernelGGL(relattn_bwd_fused_kernel, dim3((Kc + KBLK - 1) / KBLK, H, B), dim3(512), SMEM, s, p);
    }
    MXL_LAUNCH_CHECK();
    if (defer_finish) return MXL_OK;
    return mxl_relattn_dq_finish(ws, oph, mph, lse, delta, dq, d_r_r_bias, B, T, H, dh, M, Kc, o_bs, o_rs, dq_bs, dq_rs, scale, stream);
}

extern "C" int mxl_relattn_dq_finish(const void* ws, const void* oph, const float* mph, const float* lse, const float* delta, void* dq,
                                     float* d_r_r_bias, int B, int T, int H, int dh, int M, int Kc, long long o_bs, int o_rs,
                                     long long dq_bs, int dq_rs, float scale, void* stream) {
    MXL_CHECK_ARG(ws && lse && delta && dq && B > 0 && T > 0 && H > 0 && M > 0 && Kc >= T && Kc <= M + T);
    if (dh != 64 || (T % 32) != 0 || (M % 32) != 0 || (Kc % 32) != 0) return MXL_EUNSUPPORTED;
    MXL_CHECK_ARG(Kc == M + T || (oph && mph));
    MXL_CHECK_ARG((o_rs % 8) == 0 && (o_bs % 8) == 0 && (dq_rs % 8) == 0 && (dq_bs % 8) == 0 && ((uintptr_t)dq % 16) == 0 &&
                  ((uintptr_t)ws % 16) == 0 && (oph == nullptr || ((uintptr_t)oph % 16) == 0));
    hipStream_t s = (hipStream_t)stream;
    FinP f;
    f.slab = (const slab_t*)ws; f.oph = (Kc < M + T) ? (const bf16_t*)oph : nullptr; f.mph = mph; f.lse = lse; f.delta = delta;
    f.dq = (bf16_t*)dq; f.d_rrb = d_r_r_bias;
    f.B = B; f.T = T; f.H = H; f.M = M; f.Kc = Kc;
    f.slab_stride = (long long)B * T * H * 64; f.o_bs = o_bs; f.dq_bs = dq_bs; f.o_rs = o_rs; f.dq_rs = dq_rs; f.scale = scale;
    {
        mxl_kt::Scope kt(MXL_KT_RELATTN_DQFIN, s);
        // threads: the busy ones only (d = 768: 2 rows x 96 chunks = 192 = three whole waves), ten such workgroups per CU
        const long long rgs = (long long)B * (T / FIN_ROWS);
        const int nch = H * 8, cw = nch < 256 ? nch : 256, busy = (256 / cw) * cw, nthr = (busy + 63) / 64 * 64;
        static const int per_cu = getenv("MXL_DQFIN_WG_PER_CU") ? atoi(getenv("MXL_DQFIN_WG_PER_CU")) : 2048 / nthr;
        const long long want = 256ll * per_cu;
        hipLaunchKernelGGL(relattn_dq_finish_kernel, dim3((unsigned)(rgs < want ? rgs : want)), dim3(nthr), 0, s, f);
    }
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}
