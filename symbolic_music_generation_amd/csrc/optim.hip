// Optimiser step (K18): global grad-norm, clip, AdamW, bf16 working-copy refresh.  One flat fp32 parameter buffer,
// so the whole model is three launches.  Semantics = torch.optim.AdamW + torch.nn.utils.clip_grad_norm_ as driven by
// HF Trainer with the reference's TrainingArguments (musicnlp/trainer/train.py:165-190: beta 0.9/0.999, eps 1e-8,
// max_grad_norm 1, weight decay from the preset).
#include "common.h"
#include "musicxl_internal.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* x, long long n, float* out) {
    __shared__ float part[4];
    float s = 0.f;
    const long long n4 = n >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = x4[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[(n4 << 2) + threadIdx.x]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// p, g, m, v fp32 [n]; w16 = bf16 working copy.  g_eff = g * gscale * clip,  clip = min(1, max_norm / (norm + 1e-6)),
// norm = sqrt(sumsq) * gscale.  Decoupled weight decay on [0, n_decay) only (HF: no decay on biases / LayerNorm).
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, bf16_t* w16,
                                                    long long n, long long n_decay, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2_sqrt,
                                                    const float* sumsq, float max_norm, float gscale) {
    float coef = gscale;
    if (max_norm > 0.f && sumsq) {
        const float norm = sqrtf(*sumsq) * gscale;
        const float c = max_norm / (norm + 1e-6f);
        coef *= fminf(c, 1.f);
    }
    const float step_size = lr / bc1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * coef;
        float pi = p[i];
        if (i < n_decay) pi *= (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (w16) w16[i] = f2bf(pi);
    }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* x, bf16_t* y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = f2bf(x[i]);
}

// y = (float)x, or y += (float)x: the receive side of the bf16 gradient exchange (dist.GradSync)
__global__ __launch_bounds__(256) void widen_kernel(const bf16_t* x, float* y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = bf2f(x[i]);
}

inline int grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 2048) b = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int mxl_sumsq_f32(const float* x, long long n, float* out_accum, void* stream) {
    MXL_CHECK_ARG(x && out_accum && n > 0 && ((uintptr_t)x % 16) == 0);
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, n, out_accum);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_adamw_step(float* p, const float* g, float* m, float* v, void* w16, long long n, long long n_decay,
                              float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                              const float* sumsq, float max_norm, float grad_scale, void* stream) {
    MXL_CHECK_ARG(p && g && m && v && n > 0 && step >= 1 && n_decay >= 0 && n_decay <= n);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)w16, n,
                       n_decay, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), sumsq, max_norm,
                       grad_scale);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_cast_f32_bf16(const float* x, void* y, long long n, void* stream) {
    MXL_CHECK_ARG(x && y && n > 0);
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_cast_bf16_f32(const void* x, float* y, long long n, void* stream) {
    MXL_CHECK_ARG(x && y && n > 0);
    hipLaunchKernelGGL(widen_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, n);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}
