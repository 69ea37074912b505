// L2 row-gather microbenchmark: does the ROW STRIDE of a K / V / Rd-like tile stream decide how fast an XCD's L2 serves it?
// The attention kernels read 64-row tiles of 128-byte rows (one head's slice of a token row) at a row stride of 3 * d_model * 2 =
// 4,608 bytes (K, V inside qkv) or 1,536 bytes (Rd): 36 and 12 cache lines -- both multiples of 4 lines, so if the L2 channel is
// picked from the low line-address bits, one head's rows live on 4 of the 16 channels, and which 4 depends on (head mod 4).
//
// Setup mirrors the forward kernel: 512 workgroups of 256 threads (two per CU); the 16 workgroups of a group share one 2,048-row
// region and sit on one XCD (blockIdx % 8); each workgroup sweeps the region in 64-row tiles (two 16-byte loads per thread per
// tile, pieces of 8 rows x 128 B per wave-instruction), `rounds` times.  Variants: row stride, and how the head offset (128 B x
// head) of the groups that share an XCD is chosen: hmul = 4 -> all congruent mod 4 (the forward's grouping today), 1 -> consecutive.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/l2_stride scripts/ubench/l2_stride.hip ; run: l2_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void sweep(const char* buf, size_t region_bytes, int stride, int hmul, int rounds, unsigned* sink) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;       // idx = 0..63 on this XCD: 4 groups of 16 at a time
    const int gi = idx >> 4;                                     // group slot on the XCD
    const int grp = gi * 8 + xcd;
    const int head = (gi * hmul + xcd) % 12;
    const char* base = buf + (size_t)grp * region_bytes + head * 128;
    const int t = threadIdx.x, row = t >> 3, ch = t & 7;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int r = 0; r < rounds; r++) {
        // the 16 workgroups of a group start at different tiles, like query blocks of different lengths
        for (int tile = 0; tile < 32; tile++) {
            const int tt = (tile + (idx & 15) * 2) & 31;
            const char* p = base + (size_t)(tt * 64 + row) * stride + ch * 16;
            const u32x4 a = *reinterpret_cast<const u32x4*>(p);
            const u32x4 b = *reinterpret_cast<const u32x4*>(p + (size_t)32 * stride);
            acc ^= a; acc ^= b;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main() {
    const int max_stride = 8192;
    const size_t region = (size_t)2048 * max_stride + 4096;
    const int groups = 32;
    char* buf; unsigned* sink;
    hipMalloc(&buf, region * groups); hipMalloc(&sink, 4);
    hipMemset(buf, 1, region * groups);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int strides[] = {128, 1536, 1664, 4608, 4736, 4096, 8192};
    const int rounds = 40;
    for (int hmul : {4, 1}) {
        for (int s : strides) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(sweep, dim3(512), dim3(256), 0, 0, buf, region, s, hmul, rounds, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) {
                    const double bytes = 512.0 * rounds * 32 * 64 * 128;
                    printf("hmul %d  row stride %5d B: %8.3f ms  %7.2f TB/s  (%5.1f GB/s per CU)\n", hmul, s, ms, bytes / ms * 1e-9,
                           bytes / ms * 1e-6 / 256);
                }
            }
        }
    }
    return 0;
}
