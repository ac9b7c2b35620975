// Weight re-layout for K3/K4: one pass that scales, permutes and (optionally) mirrors the taps of a convolution weight.
// See gc_weight_layout_f32 in include/gancontrol_hip.h.
//
// The reference prepares weights with separate ATen passes on every call (gan_model.py:154 `weight * scale`,
// :284-306 scale, modulate, view, transpose; convolution_backward permutes again).  The kernels here want
// [taps, K, N] (N contiguous), the parameters are stored [N, K, taps] (or [K, N, taps] / [1, N, K, taps]), the input
// gradient needs [mirrored taps, N, K].  All of these are the same operation:
//
//     dst[t' * dt + k * dk + n * dn] = scale * src[t * st + k * sk + n * sn],   t' = flip ? T - 1 - t : t
//
// A workgroup moves a 32(k) x 32(n) x (<= 9 taps) tile through LDS: it is read in the memory order of `src` and written
// in the memory order of `dst`, so both sides are coalesced whatever the two layouts are.
#include <algorithm>

#include "common.h"

namespace {

constexpr int TILE = 32, TAPS = 9, PITCH = TILE + 1;

struct LayoutArgs {
    const float* src; float* dst;
    int T, K, N;
    long long st, sk, sn, dt, dk, dn;
    int r0, r1, r2;     // axis ids (0 = tap, 1 = k, 2 = n) of the read phase, slowest -> fastest in src memory
    int w0, w1, w2;     // same for the write phase / dst memory
    int flip; float scale;
};

__device__ __forceinline__ int pick(int which, int a, int b, int c) { return which == 0 ? a : (which == 1 ? b : c); }

__global__ __launch_bounds__(256) void weight_layout_kernel(LayoutArgs a) {
    __shared__ float tile[TAPS * TILE * PITCH];
    const int k0 = blockIdx.x * TILE, n0 = blockIdx.y * TILE, t0 = blockIdx.z * TAPS;
    const int ke = min(TILE, a.K - k0), ne = min(TILE, a.N - n0), te = min(TAPS, a.T - t0);
    const int total = te * ke * ne;
    {
        const int e1 = pick(a.r1, te, ke, ne), e2 = pick(a.r2, te, ke, ne);
        for (int e = threadIdx.x; e < total; e += 256) {
            const int c2 = e % e2, r = e / e2, c1 = r % e1, c0 = r / e1;
            // coordinate of axis X = the c_i whose r_i == X
            const int t = a.r0 == 0 ? c0 : (a.r1 == 0 ? c1 : c2);
            const int k = a.r0 == 1 ? c0 : (a.r1 == 1 ? c1 : c2);
            const int n = a.r0 == 2 ? c0 : (a.r1 == 2 ? c1 : c2);
            tile[(t * TILE + k) * PITCH + n] = a.src[(t0 + t) * a.st + (k0 + k) * a.sk + (n0 + n) * a.sn];
        }
    }
    __syncthreads();
    {
        const int e1 = pick(a.w1, te, ke, ne), e2 = pick(a.w2, te, ke, ne);
        for (int e = threadIdx.x; e < total; e += 256) {
            const int c2 = e % e2, r = e / e2, c1 = r % e1, c0 = r / e1;
            const int t = a.w0 == 0 ? c0 : (a.w1 == 0 ? c1 : c2);
            const int k = a.w0 == 1 ? c0 : (a.w1 == 1 ? c1 : c2);
            const int n = a.w0 == 2 ? c0 : (a.w1 == 2 ? c1 : c2);
            const int td = a.flip ? a.T - 1 - (t0 + t) : t0 + t;
            a.dst[td * a.dt + (k0 + k) * a.dk + (n0 + n) * a.dn] = a.scale * tile[(t * TILE + k) * PITCH + n];
        }
    }
}

// axis ids sorted by stride, largest first (ties: keep tap, k, n order -- extents of 1 make the stride irrelevant)
void memory_order(const int64_t s[3], int out[3]) {
    int idx[3] = {0, 1, 2};
    std::stable_sort(idx, idx + 3, [&](int x, int y) { return s[x] > s[y]; });
    out[0] = idx[0]; out[1] = idx[1]; out[2] = idx[2];
}

}  // namespace

extern "C" int gc_weight_layout_f32(const float* src, float* dst, int taps, int k, int n,
                                    const int64_t src_stride[3], const int64_t dst_stride[3],
                                    int flip_taps, float scale, gc_stream_t stream) {
    if (!src || !dst || !src_stride || !dst_stride) return gc::fail(GC_ERR_BAD_ARG, "gc_weight_layout_f32: null pointer");
    if (taps <= 0 || k <= 0 || n <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_weight_layout_f32: non-positive extent");
    for (int i = 0; i < 3; ++i)
        if (src_stride[i] < 0 || dst_stride[i] < 0) return gc::fail(GC_ERR_BAD_ARG, "gc_weight_layout_f32: negative stride");
    if (gc::ceil_div(n, TILE) > 65535 || gc::ceil_div(taps, TAPS) > 65535) return gc::fail(GC_ERR_UNSUPPORTED, "gc_weight_layout_f32: extent too large");
    LayoutArgs a;
    a.src = src; a.dst = dst; a.T = taps; a.K = k; a.N = n;
    a.st = src_stride[0]; a.sk = src_stride[1]; a.sn = src_stride[2];
    a.dt = dst_stride[0]; a.dk = dst_stride[1]; a.dn = dst_stride[2];
    int r[3], w[3];
    memory_order(src_stride, r);
    memory_order(dst_stride, w);
    a.r0 = r[0]; a.r1 = r[1]; a.r2 = r[2];
    a.w0 = w[0]; a.w1 = w[1]; a.w2 = w[2];
    a.flip = flip_taps ? 1 : 0; a.scale = scale;
    dim3 grid(gc::ceil_div(k, TILE), gc::ceil_div(n, TILE), gc::ceil_div(taps, TAPS));
    hipLaunchKernelGGL(weight_layout_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return gc::check_launch("gc_weight_layout_f32");
}
