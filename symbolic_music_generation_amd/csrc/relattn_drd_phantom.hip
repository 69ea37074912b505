// The phantom cells' part of dRd (zero memories: the reference's training, musicnlp/models/transformer_xl.py:163-171 calls the model
// without mems, so upstream's init_mems supplies zeros).  Key positions below the first stored key have k = v = 0 and exist only as
// distances, so their score gradient needs no K / V:
//     dG[i, d] = -scale * delta_i * exp(scale * (q_i + r_r_bias) . Rd[d] - lse_i)          for every distance d > i - pz
//     dRd[d, h, :] += sum_{b,i} dG[i, d] * (q_i + r_r_bias)
// Two MFMA products per cell (G = Qr Rd^T and the contraction over the queries), one exponential, no HBM traffic beyond the Qr rows:
// the companion of mxl_relattn_bwd_fused, which owns the cells of the stored keys (relattn_bwd_fused.hip).
//
// Round 4, second form.  The first form (a mode of relattn_drd_kernel, relattn_bwd.hip) ran 0.80 ms per layer at the bench shape for
// 0.17 ms of MFMA work: hipcc put s_waitcnt vmcnt(0) in front of the first transposed LDS read of every step (it cannot tell the read
// from the LDS-DMA in flight), draining the prefetch ring each step, and laid the rest out as "four LDS reads, s_waitcnt lgkmcnt(0),
// four MFMAs" eight times over.  This form: 0.63 ms, parity unchanged; profiles/r04_phantom_drd_notes.txt has the stamps and what
// was tried.  Structure:
//   * workgroup = 256 distances of one head x a group of sequences, wave = 64 distances as two 32-distance blocks; the wave's Rd rows
//     (pre-multiplied by scale * log2 e) stay in registers for the whole pass, the fp32 dRd accumulators too (64 registers).
//   * 32x32x16 MFMAs: an MFMA holds the SIMD's vector issue for 8 of its 32 cycles (16x16x32: 8 of 16), which leaves the issue slots
//     the 32 exponentials + 32 multiplies + 16 conversions of a step need (MI355X_MICROARCH.md, per-instruction constants).
//   * G with the query on the rows: the accumulator then has the distance on the lane and sixteen queries in the registers, which IS
//     the A operand of the contraction (k order permuted; the Qr^T fragments are read in the same order by transposed LDS reads from
//     the same image the row reads use): no LDS round trip for dG.
//   * -lse * log2 e is the G accumulators' start value, read from LDS straight into them (mxl_relattn_drd_phantom_prep stores it per
//     tile beside delta); -scale is applied once, to the accumulators at the end.
//   * software pipeline over 32-distance units: G of unit u + 1 is issued before the exponentials of unit u, whose contraction MFMAs
//     follow; every LDS operand is requested a phase ahead of its use.  One workgroup barrier per two steps (a ring of sixteen
//     5 KB stages filled by LDS-DMA twelve steps ahead).
//   * blocks below the diagonal of a tile that straddles the first stored key hold no phantom cell and are skipped (the first form
//     computed and masked them): 28 of the 64 blocks of every 256 x 256 diagonal square.
// The cells' part of d r_r_bias -- the column sums of the phantom dQr term -- moved to relattn_dq_finish_kernel, which has that term
// in registers (the first form spent four MFMAs per step on column sums of dG for it).
#include "common.h"
#include "musicxl_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr float LOG2E = 1.4426950408889634f;

// one v_mul_f32, opaque to the SLP vectoriser: packed into v_pk_mul_f32 the sixteen multiplies of a unit came with a v_mov / v_pk_mov
// per pair to line the operands up (and packed f32 beside MFMAs costs more than two plain ones, MI355X_MICROARCH.md)
__device__ __forceinline__ float mul1(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a), __builtin_bit_cast(mfma_bf16x8, b), c, 0, 0, 0);
}
// Transposed fragment reads as inline asm: through the builtin hipcc puts s_waitcnt vmcnt(0) in front of the first transposed read of
// every step -- it cannot tell the read from the LDS-DMA writes in flight -- which drains the prefetch ring each step (that wait WAS
// the first form's 0.80 ms; gemm.hip's weight-gradient kernel met the same thing).  hipcc neither counts reads made by inline asm nor
// keeps copies of their result registers behind a later wait, so the step's eight reads and their s_waitcnt are ONE statement: its
// outputs are valid when it ends.  It stands right behind the four G MFMAs of the step, which run while the wave waits here.
// Fragment (e half et, k-step s): rows +0 / +8 (image offsets 0 / 1024, chunk bit 0 flipped), k-step 1: +2048.
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tr_read8(uint32_t a0, uint32_t a1, bf16x8 (&fb)[2][2]) {
    u64x2 v00, v01, v10, v11;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9 offset:1024\n\t"
        "ds_read_b64_tr_b16 %2, %8 offset:2048\n\tds_read_b64_tr_b16 %3, %9 offset:3072\n\t"
        "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %11 offset:1024\n\t"
        "ds_read_b64_tr_b16 %6, %10 offset:2048\n\tds_read_b64_tr_b16 %7, %11 offset:3072\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v00.x), "=&v"(v00.y), "=&v"(v01.x), "=&v"(v01.y), "=&v"(v10.x), "=&v"(v10.y), "=&v"(v11.x), "=&v"(v11.y)
        : "v"(a0), "v"(a0 ^ 16u), "v"(a1), "v"(a1 ^ 16u)
        : "memory");
    fb[0][0] = __builtin_bit_cast(bf16x8, v00); fb[0][1] = __builtin_bit_cast(bf16x8, v01);
    fb[1][0] = __builtin_bit_cast(bf16x8, v10); fb[1][1] = __builtin_bit_cast(bf16x8, v11);
}

// Per (sequence, head, 32-query tile) one 4352-byte record, what a step of the main kernel wants in LDS:
//   [32 rows x 128 B of qs = bf16((q + r_r_bias) * scale * log2 e), 16-byte chunk c of row r at c ^ ph_swz(r)]  [-lse * log2 e, 32 floats]
// (the scaled rows are the attention kernels' Qr operand, bit for bit: the rebuilt scores match the forward's).  A step is four
// contiguous 1 KB LDS-DMA pieces (one per wave) + 128 bytes of -lse2 + 128 bytes of delta (from the (B,H,T) array the fused pass fills).
// In training the forward writes the records as it goes (mxl_relattn_fwd_phantom2(..., ph_ws)); mxl_relattn_drd_phantom_prep
// fills them from q and lse for callers that did not ask it to.
constexpr int PH_REC = 4096 + 256;
__device__ __forceinline__ int ph_swz(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }

__global__ void phantom_prep_kernel(const bf16_t* q, long long q_bs, int q_rs, const float* rrb, char* rec, int B, int T, int H,
                                    float scale_log2e) {
    const int chunks = H * 8;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * T * chunks) return;
    const int c = (int)(gid % chunks);
    const long long row = gid / chunks;
    const int b = (int)(row / T), t = (int)(row % T);
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(q + b * q_bs + (long long)t * q_rs + c * 8);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = (bf2f((bf16_t)v[j]) + rrb[c * 8 + j]) * scale_log2e;
    const u32x4 w = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
    const int r = t & 31;
    char* dst = rec + (((size_t)b * H + (c >> 3)) * (T >> 5) + (t >> 5)) * PH_REC + r * 128 + (((c & 7) ^ ph_swz(r)) << 4);
    *reinterpret_cast<u32x4*>(dst) = w;
}
// the records' -lse2: thread = (b, h, t) in memory order of lse, so reads and the 128-byte runs written are whole lines
__global__ void phantom_prep_sc_kernel(const float* lse, char* rec, long long n, int T) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long bh = i / T;
    const int t = (int)(i % T);
    reinterpret_cast<float*>(rec + (bh * (T >> 5) + (t >> 5)) * PH_REC + 4096)[t & 31] = -lse[i] * LOG2E;
}

// In-kernel stamps (diagnostic builds only: scripts/ab_build.sh relattn_drd_phantom stamp -DMXL_STAMP; scripts/stamp_phantom.py)
#ifdef MXL_STAMP
__device__ unsigned long long g_ph_stamps[16];
#define STAMP_DECL unsigned long long st_last, st_acc[16] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}; \
    { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define STAMP(i) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); st_acc[i] += t_ - st_last; st_last = t_; }
#define STAMP_FLUSH if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 16; i_++) atomicAdd(&g_ph_stamps[i_], st_acc[i_]); }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

struct PhP {
    const char* rec; const float* delta; const bf16_t* rd; float* drd;
    int B, T, H, M, bgroup;
    int rd_rs, drd_ld;
    float scale; int pz;
    // distance block k takes mk[k] batch groups per workgroup (a power of two; the other workgroups of its column exit at once): a
    // block's steps per sequence grow with k (8 k + 8), so equal batch groups left the launch either unbalanced (few, long
    // workgroups) or paying a prologue and 64 KB of float atomics per short workgroup (many)
    int mk[32];
};
constexpr int PH_STAGE = PH_REC + 256;      // a record (Qr tile 4 KB, -lse2 [32] + 128 B slack) + delta [32] (+ the next tile's 32, unused)
#ifndef PH_NST_
#define PH_NST_ 16
#endif
constexpr int PH_NST = PH_NST_, PH_PA = PH_NST_ - 4;      // ring stages, prefetch distance (steps)
constexpr int PH_SMEM = PH_STAGE * PH_NST;  // 72 KB: two workgroups per CU

// Qr tile image (ph_swz above): 128-byte rows, 16-byte chunk c of row at c ^ s(row), s = (bit 1, bit 2, bit 3) of the row index as
// chunk bits (2, 1, 0).  Row reads (32x32x16 A operand, ds_read_b128 lane groups of 16 rows) then cover the 64 banks once, and so
// does a 32-lane half of the transposed reads (4 rows x 64 bytes: rows q, q + 2 land in different 64-byte halves, rows q, q + 1
// are 128 bytes apart).

__global__ __launch_bounds__(256, 2) void relattn_drd_phantom_kernel(PhP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, r = l & 31, hh = l >> 5;
    int bx_, h, bz_;
    xcd_block(bx_, h, bz_);
    STAMP_DECL
    // a distance block's work grows with its index (block k: 8 k + 8 query tiles per sequence): highest block first
    bx_ = gridDim.x - 1 - bx_;
    const int d0 = bx_ * 256;
    const int mk = p.mk[bx_ & 31];
    if (bz_ & (mk - 1)) return;
    const int b0 = bz_ * p.bgroup;
    const int nb = min(p.bgroup * mk, p.B - b0);
    const int spb = p.T >> 5;
    const int tph = (d0 + p.pz) >> 5;            // tiles below tph: every cell of the workgroup's 256 distances is a phantom cell
    const int nph = max(0, min(spb, tph + 8));   // tiles [tph, tph + 8) straddle the first stored key; above them there is no phantom cell
    const int SA = nb * nph;
    if (SA <= 0) return;
    const int tb0 = tph + 2 * wid;               // unit nt (distances d0 + 64 wid + 32 nt ..): tiles < tb0 + nt full, == diagonal, > none

    // ---- the wave's Rd rows: B operand of G (lane = distance, k = e); scale * log2 e rides on the query rows, as in the attention kernels
    bf16x8 rdf[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int dd = d0 + 64 * wid + 32 * nt + r;      // (M need not be a multiple of 256: the last block's rows past M - 1 are zeros
            const bf16x8 zz = {0, 0, 0, 0, 0, 0, 0, 0};       // here and are not written back in the epilogue)
            rdf[nt][ks] = dd < p.M ? *reinterpret_cast<const bf16x8*>(p.rd + (size_t)dd * p.rd_rs + h * 64 + 16 * ks + 8 * hh) : zz;
        }
    // ---- LDS addresses (bytes inside a stage)
    const int A0 = r * 128 + ((hh ^ ph_swz(r)) << 4);                    // row read, k-step ks: A0 ^ (ks << 5)
    const int q4 = (l & 15) >> 2, pp = l & 3, cg_ = (l >> 4) & 1;
    // transposed read, fragment (e half et, k-step s): rows 16 s + 8 jj + 4 hh + q4 (jj = 0, 1), columns 32 et + 16 cg_ + 4 pp ..
    const int B0 = (4 * hh + q4) * 128 + ((((2 * cg_ + (pp >> 1)) ^ (4 * ((q4 >> 1) & 1) + 2 * hh))) << 4) + ((pp & 1) << 3);
    const int sv0 = 4096 + 16 * hh;                                      // the lane's first -lse2 value; delta: + 256; group g4: + 32 g4

    // ---- LDS-DMA: wave w copies bytes [1024 w, 1024 w + 1024) of the record, wave 0 also its 256 bytes of scalars, wave 1 the
    // tile's delta (256 bytes = this tile's 32 values and the next one's, which nobody reads; past the array the descriptor returns 0)
    const int rec_b = p.H * spb * PH_REC;      // bytes of one sequence's records
    const __amdgpu_buffer_rsrc_t rs_dl = __builtin_amdgcn_make_buffer_rsrc((void*)p.delta, 0, p.B * p.H * p.T * 4, 0x00020000);
    int ib = b0, it = 0, ig = 0;               // next step to request: sequence, tile, running index
    auto issue = [&]() {
        char* st = smem + (ig & (PH_NST - 1)) * PH_STAGE;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.rec + (size_t)ib * rec_b), 0, -1, 0x00020000);
        const int so = (h * spb + it) * PH_REC;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(st + wid * 1024), 16, wid * 1024 + l * 16, so, 0, 0);
        if (wid == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(st + 4096), 4, 4096 + l * 4, so, 0, 0);
        if (wid == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dl, (lptr_t)(st + PH_REC), 4, l * 4, (((ib * p.H + h) * spb + it) * 32) * 4, 0, 0);
        ig++;
        if (++it == nph) { it = 0; ib++; }
    };
    // all but the (PA - 2) youngest steps' pieces landed: waves 0 and 1 have two pieces per step in flight, the others one
    auto wait_ring = [&]() {
        if (wid < 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (PH_PA - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PH_PA - 2) : "memory");
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int et = 0; et < 2; et++)
#pragma unroll
            for (int t = 0; t < 16; t++) acc[nt][et][t] = 0.f;

    auto ld16 = [&](const char* a) {            // sixteen floats of the lane's queries (4 hh + 8 g4 + 0..3): four 16-byte reads
        f32x16 c;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(a + 32 * g4);
#pragma unroll
            for (int j = 0; j < 4; j++) c[4 * g4 + j] = v[j];
        }
        return c;
    };
    auto ld_qa = [&](const char* st, bf16x8 (&qa)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) qa[ks] = *reinterpret_cast<const bf16x8*>(st + (A0 ^ (ks << 5)));
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto gprod = [&](const bf16x8 (&qa)[4], int nt, f32x16 c) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) c = mfma32(qa[ks], rdf[nt][ks], c);
        return c;
    };
    // exponentials of one unit -> the A fragments of its contraction (k-step s = registers 8 s .. 8 s + 7); on the diagonal block
    // the exponent is replaced (not the result: no branch per cell): valid <=> distance - query >= 1 inside the block
    auto pexp = [&](const f32x16& c, const f32x16& nd, bool diag, bf16x8 (&pa)[2]) {
        float pv[16];
        if (!diag) {
#pragma unroll
            for (int t = 0; t < 16; t++) pv[t] = mul1(__builtin_amdgcn_exp2f(c[t]), nd[t]);
        } else {
            const int tl = r - 4 * hh;
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const bool valid = tl >= (t & 3) + 8 * (t >> 2) + 1;
                pv[t] = mul1(__builtin_amdgcn_exp2f(valid ? c[t] : -1.0e30f), nd[t]);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const u32x4 w = {pack2bf(pv[8 * s], pv[8 * s + 1]), pack2bf(pv[8 * s + 2], pv[8 * s + 3]),
                             pack2bf(pv[8 * s + 4], pv[8 * s + 5]), pack2bf(pv[8 * s + 6], pv[8 * s + 7])};
            pa[s] = __builtin_bit_cast(bf16x8, w);
        }
    };
    auto contract = [&](const bf16x8 (&pa)[2], const bf16x8 (&fb)[2][2], int nt) {
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int et = 0; et < 2; et++) acc[nt][et] = mfma32(pa[s], fb[et][s], acc[nt][et]);
    };

    // ---- prologue: the first PA steps requested, steps 0 and 1 landed, G of unit (0, 0) under way
    for (int g = 0; g < min(PH_PA, SA); g++) issue();
    if (SA >= PH_PA) wait_ring();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 qa0[4], qa1[4];                       // the query rows of the even / odd step of a pair (no copies between steps)
    f32x16 cg0, cg1;                             // G of unit (step, 0), issued a step ahead
    int tc = 0;                                  // tile of the step being computed
    ld_qa(smem, qa0);
    cg0 = ld16(smem + sv0);
    if (tc <= tb0) cg0 = gprod(qa0, 0, cg0);
    STAMP(15)
    // One step.  `qa`, `cgA`: this step's rows and its unit-0 G (in flight); `qaN`, `cgN`: the next step's, filled here.  ODD: the
    // second step of a pair, which holds the pair's wait + barrier.
    auto step = [&](int g, bf16x8 (&qa)[4], f32x16& cgA, bf16x8 (&qaN)[4], f32x16& cgN, const bool odd) {
        STAMP(0)
        const char* st = smem + (g & (PH_NST - 1)) * PH_STAGE;
        const char* stN = smem + ((g + 1) & (PH_NST - 1)) * PH_STAGE;
        const bool more = g + 1 < SA;
        const int tcN = (tc + 1 == nph) ? 0 : tc + 1;
        // ---- G of unit (g, 1), then the contraction operands (one asm statement: reads + wait, under the four MFMAs just issued)
        const bool uA = tc <= tb0, uB = tc <= tb0 + 1;
        f32x16 cgB = ld16(st + sv0);
        const f32x16 nd = ld16(st + sv0 + 256);
        if (uB) cgB = gprod(qa, 1, cgB);
        STAMP(1)
        bf16x8 fb[2][2];
        {
            const uint32_t sb = lds0 + (g & (PH_NST - 1)) * PH_STAGE + B0;
            tr_read8(sb, sb ^ 64u, fb);
        }
        STAMP(2)
        if (odd) {
            // steps g + 1, g + 2 must have landed before anyone reads them (every wave waits for its own requests, then they meet);
            // the stages requested at the top of the next step were last read two barriers ago
            if (g + PH_PA < SA) wait_ring();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        STAMP(3)
        // ---- unit (g, 0): exponentials; the next step's rows and start values requested behind them (hipcc waits for every LDS
        // read it knows of before the first use of `nd`); contraction
        bf16x8 paA[2];
        if (uA) pexp(cgA, nd, tc == tb0, paA);
        STAMP(4)
        if (more) { ld_qa(stN, qaN); cgN = ld16(stN + sv0); }
        STAMP(5)
        if (uA) contract(paA, fb, 0);
        STAMP(6)
        // ---- G of unit (g + 1, 0) ahead of unit (g, 1)'s exponentials
        if (more && tcN <= tb0) cgN = gprod(qaN, 0, cgN);
        STAMP(7)
        if (uB) {
            bf16x8 pa[2];
            pexp(cgB, nd, tc == tb0 + 1, pa);
            STAMP(8)
            contract(pa, fb, 1);
        }
        STAMP(9)
        tc = tcN;
    };
#pragma unroll 1
    for (int g = 0; g < SA; g += 2) {
        if (g + PH_PA < SA) issue();
        if (g + PH_PA + 1 < SA) issue();
        step(g, qa0, cg0, qa1, cg1, false);
        if (g + 1 < SA) step(g + 1, qa1, cg1, qa0, cg0, true);
    }

    // ---- epilogue: acc[nt][et][t] = distance d0 + 64 wid + 32 nt + (t & 3) + 8 (t >> 2) + 4 hh, element 32 et + r: an atomic
    // instruction covers two rows x 128 contiguous bytes
    const float nsc = -1.f / LOG2E;            // dG = -scale delta P and the rows carry scale * log2 e: -scale / (scale log2 e)
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int dd = d0 + 64 * wid + 32 * nt + (t & 3) + 8 * (t >> 2) + 4 * hh;
            float* dst = p.drd + (size_t)dd * p.drd_ld + h * 64 + r;
            if (dd < p.M) {
#pragma unroll
                for (int et = 0; et < 2; et++) atomicAdd(dst + 32 * et, acc[nt][et][t] * nsc);
            }
        }
    STAMP(14)
    STAMP_FLUSH
}

}  // namespace

#ifdef MXL_STAMP
extern "C" int mxl_debug_phantom_occupancy() {
    int n = -1;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&relattn_drd_phantom_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PH_SMEM);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, relattn_drd_phantom_kernel, 256, PH_SMEM);
    return e == hipSuccess ? n : -(int)e;
}
extern "C" int mxl_debug_phantom_stamps(unsigned long long* host_out16) {
    hipError_t e = hipMemcpyFromSymbol(host_out16, HIP_SYMBOL(g_ph_stamps), sizeof(unsigned long long) * 16);
    if (e != hipSuccess) return (int)e;
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_ph_stamps), z, sizeof(z));
    return (int)e;
}
#endif

extern "C" size_t mxl_relattn_drd_phantom_ws_bytes(int B, int T, int H) {
    if (B <= 0 || T <= 0 || H <= 0 || (T % 32) != 0) return 0;
    return (size_t)B * H * (T / 32) * PH_REC;
}

extern "C" int mxl_relattn_drd_phantom_prep(const void* q, long long q_bs, int q_rs, const float* r_r_bias, const float* lse, void* ws,
                                            int B, int T, int H, int dh, float scale, void* stream) {
    MXL_CHECK_ARG(q && r_r_bias && lse && ws && B > 0 && T > 0 && H > 0);
    if (dh != 64 || (T % 32) != 0) return MXL_EUNSUPPORTED;
    MXL_CHECK_ARG((q_rs % 8) == 0 && (q_bs % 8) == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)ws % 16) == 0);
    const long long tot = (long long)B * T * (H * 8), nsc = (long long)B * H * T;
    {
        mxl_kt::Scope kt(MXL_KT_ROWBIAS, (hipStream_t)stream);
        hipLaunchKernelGGL(phantom_prep_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q,
                           q_bs, q_rs, r_r_bias, (char*)ws, B, T, H, scale * LOG2E);
        hipLaunchKernelGGL(phantom_prep_sc_kernel, dim3((unsigned)((nsc + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lse,
                           (char*)ws, nsc, T);
    }
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_relattn_drd_phantom(const void* ws, const float* delta, float* d_rd, int B, int T, int H, int dh, int M, int drd_ld,
                                       const void* rd, int rd_rs, int Kc, void* stream) {
    MXL_CHECK_ARG(ws && delta && d_rd && rd && B > 0 && T > 0 && H > 0 && M > 0 && Kc >= T && Kc <= M + T);
    if (dh != 64 || (T % 32) != 0 || (M % 32) != 0 || (M + 255) / 256 > 32) return MXL_EUNSUPPORTED;
    MXL_CHECK_ARG(((T - Kc) % 64) == 0);
    MXL_CHECK_ARG((rd_rs % 8) == 0 && drd_ld >= H * 64 && ((uintptr_t)ws % 16) == 0 && ((uintptr_t)rd % 16) == 0);
    // 32-bit byte offsets inside one sequence's records and inside delta
    MXL_CHECK_ARG((long long)H * (T / 32) * PH_REC < (1ll << 31) && (long long)B * H * T * 4 < (1ll << 31));
    if (Kc >= M + T) return MXL_OK;             // every visible key is stored: no phantom cell
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&relattn_drd_phantom_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, PH_SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    PhP p;
    p.rec = (const char*)ws; p.delta = delta; p.rd = (const bf16_t*)rd; p.drd = d_rd;
    p.B = B; p.T = T; p.H = H; p.M = M; p.rd_rs = rd_rs; p.drd_ld = drd_ld;
    p.scale = 0.f; p.pz = T - Kc;
    // Batch groups sized for the LONGEST distance block (steps per sequence: min(T / 32, (256 k + pz) / 32 + 8), 8 .. 64 at M = 2048),
    // `fac` times as many of them as resident workgroup slots (two per CU), the long blocks dispatched first; a shorter block takes a
    // power-of-two number of groups per workgroup so that every workgroup runs between half and all of the longest one's steps.
    const int nk = (M + 255) / 256, spb = T / 32;
    int nph[32], ref = 0;
    for (int k = 0; k < nk; k++) {
        const int v = (256 * k + p.pz) / 32 + 8;          // (256 k + pz is a multiple of 32; negative: no phantom cell at all)
        nph[k] = (256 * k + p.pz < -256) ? 0 : (v < 0 ? 0 : (v > spb ? spb : v));
        if (nph[k] > ref) ref = nph[k];
    }
    if (ref == 0) return MXL_OK;
    const int tiles = nk * H;
    static const int fac = getenv("MXL_DRD_PH_FACTOR") ? atoi(getenv("MXL_DRD_PH_FACTOR")) : 6;
    int groups = (fac * 512 + tiles - 1) / tiles;
    if (groups < 1) groups = 1;
    if (groups > B) groups = B;
    p.bgroup = (B + groups - 1) / groups;
    groups = (B + p.bgroup - 1) / p.bgroup;
    static const int balance = getenv("MXL_DRD_PH_BALANCE") ? atoi(getenv("MXL_DRD_PH_BALANCE")) : 0;
    for (int k = 0; k < 32; k++) {
        int m = 1;
        if (balance && k < nk && nph[k] > 0) while (2 * m * nph[k] <= ref && 2 * m <= groups) m *= 2;
        p.mk[k] = m;
    }
    {
        mxl_kt::Scope kt(MXL_KT_RELATTN_DRD, (hipStream_t)stream);
        hipLaunchKernelGGL(relattn_drd_phantom_kernel, dim3(nk, H, groups), dim3(256), PH_SMEM, (hipStream_t)stream, p);
    }
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}
