// fp32 atomic-add of one 128 x 128 tile per workgroup (the split-K weight-gradient epilogue), by lane -> element mapping:
//   pattern 0: MFMA C layout with lane = (row l & 15, 4 consecutive columns): each of the 4 atomics of a quad is 16 rows x 4
//              separate dwords (what gemm_bf16_kernel does today)
//   pattern 1: lane = (column l & 15, 4 consecutive rows): an atomic instruction = 4 rows x 64 contiguous bytes
//   pattern 2: an atomic instruction = 1 row x 256 contiguous bytes (needs an LDS transpose in a real kernel)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/atomic_tiles scripts/ubench/atomic_tiles.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int PAT>
__global__ __launch_bounds__(256) void k(float* C, int ldc, int tiles_n, int ntiles, int rounds) {
    const int wid = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int wr = wid >> 1, wc = wid & 1;
    for (int r = 0; r < rounds; r++) {
        const int tile = (blockIdx.x * 7 + r * 13) % ntiles;          // split-K slices of one tile collide, as in the GEMM
        float* base = C + (size_t)(tile / tiles_n) * 128 * ldc + (tile % tiles_n) * 128;
        if (PAT == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        atomicAdd(base + (size_t)(wr * 64 + i * 16 + (l & 15)) * ldc + wc * 64 + j * 16 + (l >> 4) * 4 + e, 1.0f);
        } else if (PAT == 1) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        atomicAdd(base + (size_t)(wr * 64 + i * 16 + (l >> 4) * 4 + e) * ldc + wc * 64 + j * 16 + (l & 15), 1.0f);
        } else {
#pragma unroll
            for (int i = 0; i < 64; i++) atomicAdd(base + (size_t)(wr * 64 + i) * ldc + wc * 64 + l, 1.0f);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

int main() {
    const int M = 3072, N = 768;
    float* C;
    (void)hipMalloc(&C, (size_t)M * N * 4);
    (void)hipMemset(C, 0, (size_t)M * N * 4);
    const int tiles_n = N / 128, ntiles = (M / 128) * tiles_n;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int rounds = 32;
    for (int pat = 0; pat < 3; pat++)
        for (int g : {1, 64, 256, 512}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                (void)hipEventRecord(e0);
                if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(g), dim3(256), 0, 0, C, N, tiles_n, ntiles, rounds);
                else if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(g), dim3(256), 0, 0, C, N, tiles_n, ntiles, rounds);
                else hipLaunchKernelGGL(k<2>, dim3(g), dim3(256), 0, 0, C, N, tiles_n, ntiles, rounds);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double us = best * 1e3 / rounds;
            printf("pattern %d  workgroups %3d : %7.2f us per 128x128 fp32 tile per workgroup, %6.2f TB/s chip\n", pat, g, us,
                   65536.0 * g / us / 1e6);
        }
    return 0;
}
