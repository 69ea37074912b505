// Micro-benchmark: what does a cross-workgroup barrier inside ONE launch cost on gfx950 (8 XCDs, one L2 each), and does data
// exchanged around it arrive intact without flushing the L2s?  Sizing a fused decode "layer tail" (o-proj -> LN -> FFN -> LN ->
// next qkv in one launch with grid barriers) against the 5-7 us each of those launches costs today.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
// Variants: (a) data through agent-scope (sc1) loads / stores + relaxed agent-scope atomics on the counter: no cache flush;
//           (b) plain stores / loads with __threadfence() on both sides (buffer_wbl2 / buffer_inv: what the compiler's
//               release / acquire fences cost).
// Every spin is bounded: a lost arrival ends the kernel with an error count instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <bool FENCE>
__global__ __launch_bounds__(512) void barrier_kernel(unsigned* ctr, unsigned* slots, unsigned* err, int iters, int payload) {
    const unsigned nwg = gridDim.x, wg = blockIdx.x;
    unsigned bad = 0;
    for (int it = 1; it <= iters; it++) {
        // every thread publishes `payload` words (the activations a phase hands to the next one)
        for (int i = threadIdx.x; i < payload; i += blockDim.x) {
            if (FENCE) slots[(size_t)wg * payload + i] = it * 7u + i;
            else st_sc1(slots + (size_t)wg * payload + i, it * 7u + i);
        }
        if (FENCE) __threadfence();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)it * nwg;
            unsigned spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 22))
                __builtin_amdgcn_s_sleep(1);
            if (spins >= (1u << 22)) bad += 1u << 16;
        }
        __syncthreads();
        if (FENCE) __threadfence();
        // read a far workgroup's payload
        const unsigned src = (wg + nwg / 2 + 1) % nwg;
        for (int i = threadIdx.x; i < payload; i += blockDim.x) {
            const unsigned v = FENCE ? slots[(size_t)src * payload + i] : ld_sc1(slots + (size_t)src * payload + i);
            if (v != it * 7u + i) bad++;
        }
        // a second barrier protects the slots from the next iteration's writes (as a real phase chain would)
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)it * nwg;
            unsigned spins = 0;
            while (__hip_atomic_load(ctr + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 22))
                __builtin_amdgcn_s_sleep(1);
            if (spins >= (1u << 22)) bad += 1u << 16;
        }
        __syncthreads();
    }
    if (bad) atomicAdd(err, bad);
}

__global__ void empty_kernel(unsigned* p) { if (threadIdx.x == 12345) p[0] = 1; }

int main() {
    unsigned *ctr, *slots, *err;
    const int payload = 512;             // 2 KB per workgroup per phase
    CK(hipMalloc(&ctr, 1024)); CK(hipMalloc(&slots, 256 * payload * 4)); CK(hipMalloc(&err, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int nwg : {48, 144, 192, 256}) {
        for (int variant = 0; variant < 2; variant++) {
            CK(hipMemset(ctr, 0, 1024)); CK(hipMemset(err, 0, 4)); CK(hipMemset(slots, 0, 256 * payload * 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL(barrier_kernel<false>, dim3(nwg), dim3(512), 0, 0, ctr, slots, err, iters, payload);
            else hipLaunchKernelGGL(barrier_kernel<true>, dim3(nwg), dim3(512), 0, 0, ctr, slots, err, iters, payload);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
            printf("%3d workgroups, %s: %.2f us per (publish 2 KB + barrier + read + barrier), i.e. %.2f us per barrier; errors %u\n", nwg,
                   variant == 0 ? "sc1 data path, relaxed atomics" : "plain data + __threadfence  ", 1e3f * ms / iters,
                   0.5e3f * ms / iters, h);
        }
    }
    // for scale: back-to-back empty launches on one stream
    CK(hipEventRecord(e0));
    for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(empty_kernel, dim3(192), dim3(512), 0, 0, ctr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty launch, 192 x 512 threads, back to back: %.2f us each\n", 1e3f * ms / 2000);
    return 0;
}
