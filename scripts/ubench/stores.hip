// Store-pattern microbenchmark: how fast can ONE workgroup per CU push a 256 x 256 bf16 tile (128 KB) to a row-major
// (M, N) matrix, as a function of the per-instruction footprint and of how many CUs do it at once?
//   pattern 0: MFMA C layout, 8-byte stores: one instruction = 16 rows x 32 B            (the GEMM epilogue today)
//   pattern 1: 16-byte stores, one instruction = 16 rows x 64 B
//   pattern 2: 16-byte stores, one instruction =  8 rows x 128 B (full lines)
//   pattern 3: 16-byte stores, one instruction =  2 rows x 512 B (whole tile rows)
//   pattern 4: 16-byte stores, one instruction = 32 rows x 32 B
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/stores scripts/ubench/stores.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* C, int ldc, int tiles_n, int rounds, int ntiles) {
    const int wid = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int wr = wid >> 2, wc = wid & 3;
    for (int r = 0; r < rounds; r++) {
        const int tile = (blockIdx.x + r * gridDim.x) % ntiles;
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
        unsigned short* base = C + (size_t)m0 * ldc + n0;
        const unsigned v = 0x3f803f80u + r;
        if (PAT == 0) {
            // wave tile 128 x 64: i = 0..7 row blocks of 16, j = 0..3 column blocks of 16; lane: row l & 15, 4 columns at (l >> 4) * 4
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    unsigned short* p = base + (size_t)(wr * 128 + i * 16 + (l & 15)) * ldc + wc * 64 + j * 16 + (l >> 4) * 4;
                    *reinterpret_cast<u32x2*>(p) = u32x2{v, v};
                }
        } else if (PAT == 1) {
            // wave tile 128 x 64: 16 instructions of 16 rows x 64 B... lane: row l >> 2 (16 rows), 8 columns at (l & 3) * 8
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    unsigned short* p = base + (size_t)(wr * 128 + i * 16 + (l >> 2)) * ldc + wc * 64 + j * 32 + (l & 3) * 8;
                    *reinterpret_cast<u32x4*>(p) = u32x4{v, v, v, v};
                }
        } else if (PAT == 2) {
            // wave tile 128 x 64 -> 8 rows x 128 B per instruction: lane row l >> 3, 8 columns at (l & 7) * 8; 16 instructions
#pragma unroll
            for (int i = 0; i < 16; i++) {
                unsigned short* p = base + (size_t)(wr * 128 + i * 8 + (l >> 3)) * ldc + wc * 64 + (l & 7) * 8;
                *reinterpret_cast<u32x4*>(p) = u32x4{v, v, v, v};
            }
        } else if (PAT == 4) {
            // 16-byte stores, one instruction = 32 rows x 32 B (two 16-row blocks, 2 lanes per row): lane row (l & 15) + 16 * ((l >> 4) & 1), 8 columns at (l >> 5) * 8
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    unsigned short* p = base + (size_t)(wr * 128 + i * 32 + (l & 15) + 16 * ((l >> 4) & 1)) * ldc + wc * 64 + j * 16 + (l >> 5) * 8;
                    *reinterpret_cast<u32x4*>(p) = u32x4{v, v, v, v};
                }
        } else {
            // wave owns 32 whole tile rows: 2 rows x 512 B per instruction, 16 instructions
#pragma unroll
            for (int i = 0; i < 16; i++) {
                unsigned short* p = base + (size_t)(wid * 32 + i * 2 + (l >> 5)) * ldc + (l & 31) * 8;
                *reinterpret_cast<u32x4*>(p) = u32x4{v, v, v, v};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

int main() {
    const int M = 32768, N = 2304;
    unsigned short* C;
    hipMalloc(&C, (size_t)M * N * 2);
    hipMemset(C, 0, (size_t)M * N * 2);
    const int tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 64;
    for (int pat = 0; pat < 5; pat++)
        for (int g : {1, 8, 32, 64, 128, 256}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                switch (pat) {
                    case 0: hipLaunchKernelGGL(store_kernel<0>, dim3(g), dim3(512), 0, 0, C, N, tiles_n, rounds, ntiles); break;
                    case 1: hipLaunchKernelGGL(store_kernel<1>, dim3(g), dim3(512), 0, 0, C, N, tiles_n, rounds, ntiles); break;
                    case 2: hipLaunchKernelGGL(store_kernel<2>, dim3(g), dim3(512), 0, 0, C, N, tiles_n, rounds, ntiles); break;
                    case 4: hipLaunchKernelGGL(store_kernel<4>, dim3(g), dim3(512), 0, 0, C, N, tiles_n, rounds, ntiles); break;
                    default: hipLaunchKernelGGL(store_kernel<3>, dim3(g), dim3(512), 0, 0, C, N, tiles_n, rounds, ntiles); break;
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double us_round = best * 1e3 / rounds;
            printf("pattern %d  workgroups %3d : %7.2f us per 128 KB tile per CU, %7.1f GB/s per CU, %6.2f TB/s chip\n", pat, g, us_round,
                   131072.0 / us_round / 1e3, 131072.0 * g / us_round / 1e6);
        }
    return 0;
}
