// Shared device/host helpers for libmusicxl (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MXL_OK 0
#define MXL_EINVAL (-1)
#define MXL_EUNSUPPORTED (-2)

#define MXL_CHECK_ARG(cond) do { if (!(cond)) return MXL_EINVAL; } while (0)
#define MXL_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

typedef uint16_t bf16_t;  // raw bf16 bits

using bf16x8 = __attribute__((ext_vector_type(8))) short;   // 8 bf16 = 4 VGPRs (MFMA A/B fragment)
using bf16x4 = __attribute__((ext_vector_type(4))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }

// round-to-nearest-even f32 -> bf16; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    f2_t v = {lo, hi};
    bf2_t r = __builtin_convertvector(v, bf2_t);
    return __builtin_bit_cast(uint32_t, r);
}

// Attention grids are (blocks of one (head, sequence), heads, sequences).  Workgroups are handed to the 8 XCDs round-robin in
// launch order, so by default the blocks of one (head, sequence) -- which share its K / V / Q tiles -- land on 8 different L2s.
// Launch slot L = (xcd = L % 8, idx = L / 8) is remapped so that an XCD works through whole groups:
//   groups % 8 == 0: group (idx / gx) * 8 + xcd, block idx % gx.  The eight XCDs then work on eight NEIGHBOURING groups at any
//     time (mostly the heads of one sequence): their K / V rows and dG rows are adjacent in HBM.  Measured at C3: forward
//     0.548 -> 0.475 ms, backward 2.17 -> 1.77 ms per layer.
//   otherwise: XCD x takes a contiguous run of the (group-major) logical order, bijective for any grid (whole groups apart from
//     one straddling each boundary).  Same L2 sharing, but the XCDs are then far apart in HBM: backward 2.05 ms in the C3 test.
__device__ __forceinline__ void xcd_block(int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y, groups = gridDim.y * gridDim.z;
#ifndef MXL_NO_XCD_REMAP
    const int L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int xcd = L & 7, idx = L >> 3;
    int grp;
    if ((groups & 7) == 0) {
        grp = (idx / gx) * 8 + xcd;
        bx = idx % gx;
    } else {
        const int n = gx * groups, q = n >> 3, r = n & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        grp = id / gx;
        bx = id - grp * gx;
    }
    by = grp % gy; bz = grp / gy;
#else
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
#endif
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based keep-mask for dropout.  Regenerated (not stored) in backward from (seed, site, index).
// lowbias32-style integer hash: cheap, stateless, identical in forward and backward.
__device__ __forceinline__ uint32_t mxl_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint32_t site, uint64_t idx, uint32_t thresh) {
    // thresh = p * 2^32 ; keep iff rnd >= thresh.  `mix` depends only on kernel arguments (hoisted out of every loop);
    // per element: two multiplies for the index spread + one lowbias32 round.
    const uint32_t mix = mxl_hash32((uint32_t)seed ^ (site * 0x9E3779B9U)) + (uint32_t)(seed >> 32);
    const uint32_t h = mxl_hash32(((uint32_t)idx * 0x9E3779B1U) ^ ((uint32_t)(idx >> 32) * 0x85EBCA77U) ^ mix);
    return h >= thresh;
}
// the same decision for an element index below 2^32 (the high-word term of dropout_keep is zero): one multiply less, no 64-bit math
__device__ __forceinline__ bool dropout_keep32(uint64_t seed, uint32_t site, uint32_t idx, uint32_t thresh) {
    const uint32_t mix = mxl_hash32((uint32_t)seed ^ (site * 0x9E3779B9U)) + (uint32_t)(seed >> 32);
    return mxl_hash32((idx * 0x9E3779B1U) ^ mix) >= thresh;
}
static inline uint32_t dropout_thresh(float p) {
    if (p <= 0.f) return 0u;
    double t = (double)p * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}
