// Micro-benchmark: fp32 atomic-add throughput vs plain stores, for sizing a fused dRd / dk / dv accumulation.
//   hipcc --offload-arch=gfx950 -O3 -o atomics atomics.hip && ./atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// A: each wave stores `iters` x 1 KB (16 B per lane), streaming
__global__ void store_kernel(uint4* out, int iters) {
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint4 v = {1, 2, 3, 4};
    for (int i = 0; i < iters; i++) out[(w * iters + i) * 64 + (threadIdx.x & 63)] = v;
}
// B: dRd pattern. WG = (qblock 16, h 12, b 16); for each of 32 distance blocks every wave adds a 16 x 64 fp32 slab of
// dRd[h][block*64 + 16*wave .. +16][64]: 16 instructions of 64 lanes x 4 B.  start block staggered by qblock (as the band is)
__global__ void drd_atomic_kernel(float* drd, int M) {
    const int qb = blockIdx.x, h = blockIdx.y;
    const int wid = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int it = 0; it < 32; it++) {
        const int blk = (it + 2 * qb) & 31;
        float* base = drd + ((size_t)h * M + blk * 64 + 16 * wid) * 64;
        for (int r = 0; r < 16; r++) atomicAdd(base + r * 64 + l, 1.0f);
    }
}
// C: dk/dv pattern. WG = (qblock, h, b): for each of `nt` key tiles, each wave adds 2 x (16 keys x 64) fp32 into (b, key, h, :)
__global__ void dkv_atomic_kernel(float* dk, float* dv, int T, int H) {
    const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int wid = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int nt = 2 * qb + 2;   // causal: tiles 0 .. 2*qb+1
    for (int kt = 0; kt < nt; kt++) {
        for (int r = 0; r < 16; r++) {
            const size_t row = (size_t)b * T + kt * 64 + 16 * wid + r;
            atomicAdd(dk + (row * H + h) * 64 + l, 1.0f);
            atomicAdd(dv + (row * H + h) * 64 + l, 1.0f);
        }
    }
}
int main() {
    const int B = 16, H = 12, T = 2048, M = 2048;
    float *drd, *dk, *dv; uint4* big;
    CK(hipMalloc(&drd, (size_t)H * M * 64 * 4)); CK(hipMalloc(&dk, (size_t)B * T * H * 64 * 4)); CK(hipMalloc(&dv, (size_t)B * T * H * 64 * 4));
    const size_t big_bytes = (size_t)B * H * T * M * 2;   // 1.6 GB
    CK(hipMalloc(&big, big_bytes));
    CK(hipMemset(drd, 0, (size_t)H * M * 64 * 4)); CK(hipMemset(dk, 0, (size_t)B * T * H * 64 * 4)); CK(hipMemset(dv, 0, (size_t)B * T * H * 64 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        const int waves = 3072 * 4, iters = (int)(big_bytes / 1024 / waves);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(store_kernel, dim3(3072), dim3(256), 0, 0, big, iters); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("A stores   %.3f GB in %.3f ms = %.2f TB/s\n", (double)waves * iters * 1024 / 1e9, ms, (double)waves * iters * 1024 / ms / 1e9);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(drd_atomic_kernel, dim3(16, H, B), dim3(256), 0, 0, drd, M); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double nb = 3072.0 * 4 * 32 * 16 * 256;
        printf("B dRd atomics %.3f GB in %.3f ms = %.2f TB/s\n", nb / 1e9, ms, nb / ms / 1e9);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(dkv_atomic_kernel, dim3(16, H, B), dim3(256), 0, 0, dk, dv, T, H); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        double tiles = 0; for (int q = 0; q < 16; q++) tiles += 2 * q + 2;
        const double nc = tiles * H * B * 4 * 16 * 2 * 256;
        printf("C dk/dv atomics %.3f GB in %.3f ms = %.2f TB/s\n", nc / 1e9, ms, nc / ms / 1e9);
    }
    float chk; CK(hipMemcpy(&chk, drd, 4, hipMemcpyDeviceToHost)); printf("drd[0] = %.0f (expect %d)\n", chk, 3 * 16 * 16);
    return 0;
}
