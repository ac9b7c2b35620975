// Products with a tiny inner extent for the style path: out[M, N] = alpha * (a[M, K] @ b[K, N]) + beta * bias[N], K <= 8.
// See gc_small_gemm_f32 in include/gancontrol_hip.h.
//
// The weight gradients of the mapping network, the 26 style modulations and the 18 demodulation sums of G (EqualLinear
// gan_model.py:171-202, ModulatedConv2d :284-293) are g[B, C]^T @ x[B, 512] with B = 2 .. 8 samples: ~75 calls per training iteration whose
// result is a rank-B update of a C x 512 matrix.  One lane per output element with its <= 8 products unrolled takes 2.7 us where the
// library GEMM takes 4 .. 4.4 (rocprofv3, tools/small_gemm_bench.py).  Measured and NOT kept: kernels of this kind for the skinny
// products of the same path ([B, 512] @ [512, C], forward and input gradient) -- one wave per output column with the lanes along k, and
// 16 waves splitting K with an LDS reduction -- ran in 5 .. 7 us against the library's 4.5 .. 5.5, and far worse at K = 8192.
#include "common.h"

namespace {

constexpr int MAXS = 8;          // the small extent

struct SgArgs {
    const float* a; const float* b; const float* bias; float* out;
    long long sa0, sa1, sb0, sb1;
    int M, K, N;
    float alpha, beta;
};

// K <= 8: one lane per output element
__global__ __launch_bounds__(256) void outer_kernel(SgArgs p) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)p.M * p.N) return;
    const int m = (int)(e / p.N), n = (int)(e - (long long)m * p.N);
    float av[MAXS], bv[MAXS];
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
        av[k] = k < p.K ? p.a[m * p.sa0 + k * p.sa1] : 0.f;
        bv[k] = k < p.K ? p.b[k * p.sb0 + n * p.sb1] : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXS; ++k) s = fmaf(av[k], bv[k], s);
    p.out[e] = fmaf(p.alpha, s, p.bias ? p.beta * p.bias[n] : 0.f);
}

}  // namespace

extern "C" int gc_small_gemm_ok(int M, int K, int N, int64_t sb0, int64_t sb1) {
    (void)sb0; (void)sb1;
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return K <= MAXS && (long long)M * N <= (1 << 19) ? 1 : 0;       // larger outputs: the library GEMM streams them faster
}

extern "C" int gc_small_gemm_f32(const float* a, int64_t sa0, int64_t sa1, const float* b, int64_t sb0, int64_t sb1,
                                 const float* bias, float beta, float alpha, float* out, int M, int K, int N, gc_stream_t stream) {
    if (!a || !b || !out) return gc::fail(GC_ERR_BAD_ARG, "gc_small_gemm_f32: null pointer");
    if (M <= 0 || K <= 0 || N <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_small_gemm_f32: non-positive extent");
    if (sa0 < 0 || sa1 < 0 || sb0 < 0 || sb1 < 0) return gc::fail(GC_ERR_BAD_ARG, "gc_small_gemm_f32: negative stride");
    if (K > MAXS) return gc::fail(GC_ERR_UNSUPPORTED, "gc_small_gemm_f32: inner extent %d > %d (gc_small_gemm_ok)", K, MAXS);
    const long long blocks = ((long long)M * N + 255) / 256;
    if (blocks > 2147483647LL) return gc::fail(GC_ERR_UNSUPPORTED, "gc_small_gemm_f32: output too large");
    const SgArgs p{a, b, bias, out, sa0, sa1, sb0, sb1, M, K, N, alpha, beta};
    hipLaunchKernelGGL(outer_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return gc::check_launch("gc_small_gemm_f32");
}
