// Skinny-M GEMM for the decode step: C[M<=64, N] = A[M,K] . W[N,K]^T (+bias)(relu), bf16 in, fp32 accumulate.
// At M = 64 a GEMM is weight streaming: every W element is read once and the activations (64 x K, L2-resident) are
// re-read by everyone.  So: no LDS staging at all -- fragments go straight from global memory to the MFMA operands
// (16 bytes per lane per fragment) -- and the parallelism comes from N and K instead of M:
//   workgroup = 16 output columns x all 64 rows, 8 waves, wave w owns K-slice w (in-workgroup split-K, LDS reduction).
// N = 768 -> 48 workgroups x 8 waves; N = 2304 -> 144 x 8: enough loads in flight to stream the weights.
#include "common.h"
#include "musicxl_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

struct SkP {
    const bf16_t* A; const bf16_t* W; void* C; const float* bias;
    int M, N, K, lda, ldw, ldc, flags;
    // decode qkv projection (mxl_decode_qkv): N = 3 d; besides C, column block [0,d) also leaves as q + r_r_bias into qr (M, d),
    // [d,2d) / [2d,3d) are appended to the head-major K / V rings (B, H, Mring, dh) at slot *t_dev % Mring
    bf16_t* kc; bf16_t* vc; bf16_t* qr; const float* rrb; const int* t_dev; int d, dh, Mring;
    // K sliced over workgroups as well (grid.y = KS, mxl_gemm_skinny_partial): slice s writes its (64 x N) fp32 partial to slab s of ws
    float* ws; int KS;
};

__global__ __launch_bounds__(512) void gemm_skinny_kernel(SkP p) {
    __shared__ float red[8][64 * 16];   // [wave][m][n] partial sums, reduced in fixed order (deterministic)
    const int tid = threadIdx.x, wid = tid >> 6, l = tid & 63;
    const int n0 = blockIdx.x * 16;
    const int li = l & 15, kq = 8 * (l >> 4);
    // K slice of this wave (multiples of 32)
    const int ksteps_all = (p.K + 31) / 32;
    const int per_wg = (ksteps_all + p.KS - 1) / p.KS;                      // K-steps of this workgroup's slice (grid.y)
    const int kw0 = blockIdx.y * per_wg, kw1 = min(ksteps_all, kw0 + per_wg);
    const int per = (max(kw1 - kw0, 0) + 7) / 8;
    const int ks0 = kw0 + wid * per, ks1 = min(kw1, ks0 + per);
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int n = n0 + li;
    // epilogue operands requested up front (a thread's two outputs share one column): loaded after the reduction they would add one
    // more dependent memory round trip to a launch that is nothing but a dependent chain
    const int ecol = n0 + (tid & 15);
    const float ebias = ((p.flags & MXL_GEMM_BIAS) && ecol < p.N) ? p.bias[ecol] : 0.f;
    const int t_now = p.kc ? *p.t_dev : 0;
    const float errb = (p.kc && n0 < p.d) ? p.rrb[n0 + (tid & 15)] : 0.f;          // (q third only; its 16 columns are < d)
    const bf16_t* wrow = p.W + (size_t)(n < p.N ? n : 0) * p.ldw;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 2
    for (int ks = ks0; ks < ks1; ks++) {
        const int k = ks * 32 + kq;
        const bool kok = k < p.K;
        const bf16x8 fb = (kok && n < p.N) ? *reinterpret_cast<const bf16x8*>(wrow + k) : z;
        bf16x8 fa[4];
#pragma unroll
        for (int mf = 0; mf < 4; mf++) {
            const int m = mf * 16 + li;
            fa[mf] = (kok && m < p.M) ? *reinterpret_cast<const bf16x8*>(p.A + (size_t)m * p.lda + k) : z;
        }
#pragma unroll
        for (int mf = 0; mf < 4; mf++)
            acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, fb),
                                                              __builtin_bit_cast(mfma_bf16x8, fa[mf]), acc[mf], 0, 0, 0);
    }
    // acc[mf][r]: m = mf*16 + (l&15), n = n0 + (l>>4)*4 + r
#pragma unroll
    for (int mf = 0; mf < 4; mf++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[wid][(mf * 16 + li) * 16 + (l >> 4) * 4 + r] = acc[mf][r];
    __syncthreads();
    int kv_part = 0, kv_col0 = 0;
    size_t kv_off = 0, kv_bstride = 0;
    if (p.kc) {
        kv_part = n0 / p.d;
        kv_col0 = n0 - kv_part * p.d;
        const int hh = kv_col0 / p.dh, e0 = kv_col0 - hh * p.dh, slot = t_now % p.Mring;
        kv_bstride = (size_t)(p.d / p.dh) * p.Mring * p.dh;
        kv_off = ((size_t)hh * p.Mring + slot) * p.dh + e0;
    }
    if (p.ws) {      // partial product of this K slice; the consumer (mxl_ln_residual_fwd_partial) sums the slabs in slice order
        float* slab = p.ws + (size_t)blockIdx.y * 64 * p.N;
        for (int i = tid; i < 64 * 16; i += 512) {
            const int m = i >> 4, nn = n0 + (i & 15);
            if (m < p.M && nn < p.N)
                slab[(size_t)m * p.N + nn] =
                    ((red[0][i] + red[1][i]) + (red[2][i] + red[3][i])) + ((red[4][i] + red[5][i]) + (red[6][i] + red[7][i]));
        }
        return;
    }
    for (int i = tid; i < 64 * 16; i += 512) {
        const int m = i >> 4, nn = n0 + (i & 15);
        if (m < p.M && nn < p.N) {
            float v = ((red[0][i] + red[1][i]) + (red[2][i] + red[3][i])) + ((red[4][i] + red[5][i]) + (red[6][i] + red[7][i]));
            if (p.flags & MXL_GEMM_BIAS) v += ebias;
            if (p.flags & MXL_GEMM_RELU) v = fmaxf(v, 0.f);
            if (p.flags & MXL_GEMM_OUT_F32) reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + nn] = v;
            else reinterpret_cast<bf16_t*>(p.C)[(size_t)m * p.ldc + nn] = f2bf(v);
            if (p.kc) {                                        // workgroup-uniform: its 16 columns sit in one third and one head
                const bf16_t vb = f2bf(v);                     // the rings and qr see the value the qkv buffer holds
                if (kv_part == 0) p.qr[(size_t)m * p.d + kv_col0 + (i & 15)] = f2bf(bf2f(vb) + errb);
                else (kv_part == 1 ? p.kc : p.vc)[(size_t)m * kv_bstride + kv_off + (i & 15)] = vb;
            }
        }
    }
}

}  // namespace

extern "C" int mxl_gemm_skinny_bf16(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                                    int flags, const float* bias, void* stream) {
    MXL_CHECK_ARG(A && W && C && M > 0 && M <= 64 && N > 0 && K > 0);
    MXL_CHECK_ARG((K % 8) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && ldc >= N);
    MXL_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
    MXL_CHECK_ARG(!(flags & ~(MXL_GEMM_OUT_F32 | MXL_GEMM_BIAS | MXL_GEMM_RELU)));
    if (flags & MXL_GEMM_BIAS) MXL_CHECK_ARG(bias != nullptr);
    SkP p;
    p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = C; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.flags = flags;
    p.kc = nullptr; p.vc = nullptr; p.qr = nullptr; p.rrb = nullptr; p.t_dev = nullptr; p.d = p.dh = p.Mring = 0;
    p.ws = nullptr; p.KS = 1;
    hipLaunchKernelGGL(gemm_skinny_kernel, dim3((N + 15) / 16), dim3(512), 0, (hipStream_t)stream, p);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_decode_qkv(const void* x, const void* Wqkv, void* qkv, void* kcache, void* vcache, const int* t_dev,
                              const float* r_r_bias, void* qr_out, int B, int d, int dh, int Mring, void* stream) {
    MXL_CHECK_ARG(x && Wqkv && qkv && kcache && vcache && t_dev && r_r_bias && qr_out);
    MXL_CHECK_ARG(B > 0 && B <= 64 && d > 0 && (d % 16) == 0 && dh > 0 && (dh % 16) == 0 && (d % dh) == 0 && Mring > 0);
    MXL_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)Wqkv % 16) == 0);
    SkP p;
    p.A = (const bf16_t*)x; p.W = (const bf16_t*)Wqkv; p.C = qkv; p.bias = nullptr;
    p.M = B; p.N = 3 * d; p.K = d; p.lda = d; p.ldw = d; p.ldc = 3 * d; p.flags = 0;
    p.kc = (bf16_t*)kcache; p.vc = (bf16_t*)vcache; p.qr = (bf16_t*)qr_out; p.rrb = r_r_bias; p.t_dev = t_dev;
    p.d = d; p.dh = dh; p.Mring = Mring;
    p.ws = nullptr; p.KS = 1;
    hipLaunchKernelGGL(gemm_skinny_kernel, dim3((3 * d + 15) / 16), dim3(512), 0, (hipStream_t)stream, p);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}

extern "C" int mxl_gemm_skinny_partial(const void* A, const void* W, float* slabs, int M, int N, int K, int lda, int ldw, int KS,
                                       void* stream) {
    MXL_CHECK_ARG(A && W && slabs && M > 0 && M <= 64 && N > 0 && K > 0 && KS >= 1 && KS <= 16);
    MXL_CHECK_ARG((K % 8) == 0 && (lda % 8) == 0 && (ldw % 8) == 0);
    MXL_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
    SkP p;
    p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = nullptr; p.bias = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = N; p.flags = 0;
    p.kc = nullptr; p.vc = nullptr; p.qr = nullptr; p.rrb = nullptr; p.t_dev = nullptr; p.d = p.dh = p.Mring = 0;
    p.ws = slabs; p.KS = KS;
    hipLaunchKernelGGL(gemm_skinny_kernel, dim3((N + 15) / 16, KS), dim3(512), 0, (hipStream_t)stream, p);
    MXL_LAUNCH_CHECK();
    return MXL_OK;
}
