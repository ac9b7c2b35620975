// K3/K4 for 1x1 convolutions with at most 4 channels on one side: ToRGB (C -> 3, gan_model.py:411-435), the discriminator's
// FromRGB ConvLayer (3 -> C, gan_model.py:955) and their gradients.  A matrix tile would be >= 87 % padding there and these
// layers are pure HBM streams (the wide side is a [B, C, 1024, 1024] tensor), so they run on the vector ALUs: one lane owns
// 4 consecutive pixels (16-byte loads / stores), the <= 4 x C weights come through the scalar cache.
//   pw_narrow_kernel   y[b,j,p] = A(so[b,j] * sum_k w[k,j] si[b,k] x[b,k,p] + ...)      j < NOUT <= 4   (read-bound)
//   pw_widen_kernel    y[b,n,p] = A(so[b,n] * sum_j w[j,n] si[b,j] x[b,j,p] + ...)      j < KIN <= 4    (write-bound)
//   pw_wgrad_kernel    dw[k,n]  = sum_{b,p} si[b,k] x[b,k,p] so[b,n] dy[b,n,p]          min(K, N) <= 4  (read-bound)
// All three keep the fused epilogue of gc_conv_epilogue; the weight gradient is reduced in a fixed order (no atomics).
#include <algorithm>

#include "conv_common.h"

namespace {

using namespace gcconv;

constexpr int MAXS = 4;          // channels on the thin side
#ifndef GC_PW_CHUNK
#define GC_PW_CHUNK 16
#endif
constexpr int CHUNK = GC_PW_CHUNK;        // wide-side channels whose loads are in flight together

struct PwArgs {
    ConvArgs c;
    long long plane;             // pixels per channel plane
    int vec;                     // planes are multiples of 4 pixels and 16-byte aligned
    // gc_pw_act_dgrad_f32: x is the gradient arriving at a fused bias + leaky-ReLU; it is multiplied by that activation's mask
    // (in_mask > 0 ? mpos : mneg, in_mask shaped like x) while it is loaded -- no separate activation-backward pass
    const float* in_mask; float mpos, mneg;
};

__device__ __forceinline__ float4 mask4(float4 v, float4 m, float pos, float neg) {
    return make_float4(v.x * (m.x > 0.f ? pos : neg), v.y * (m.y > 0.f ? pos : neg), v.z * (m.z > 0.f ? pos : neg), v.w * (m.w > 0.f ? pos : neg));
}

template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* p, long long i, long long n) {
    if (VEC) return *reinterpret_cast<const float4*>(p + i);
    return make_float4(i < n ? p[i] : 0.f, i + 1 < n ? p[i + 1] : 0.f, i + 2 < n ? p[i + 2] : 0.f, i + 3 < n ? p[i + 3] : 0.f);
}
// the wide operand of a narrow / weight-gradient launch (32 channels x 1024^2 per sample) is read exactly once
#ifndef GC_PW_NT_LOAD
#define GC_PW_NT_LOAD 0      // measured neutral on the whole step (75.78 vs 75.66 images/s, round 4): off
#endif
template <bool VEC>
__device__ __forceinline__ float4 ld4s(const float* p, long long i, long long n) {
#if GC_PW_NT_LOAD
    if (VEC) { const gc::f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const gc::f32x4_t*>(p + i)); return make_float4(t.x, t.y, t.z, t.w); }
#endif
    return ld4<VEC>(p, i, n);
}
template <bool VEC>
__device__ __forceinline__ void st4(float* p, long long i, long long n, float4 v) {
    if (VEC) { gc::stream_store4(p + i, v.x, v.y, v.z, v.w); return; }
    if (i < n) p[i] = v.x;
    if (i + 1 < n) p[i + 1] = v.y;
    if (i + 2 < n) p[i + 2] = v.z;
    if (i + 3 < n) p[i + 3] = v.w;
}
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 c) {
    return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}

// grid: (pixel groups, batch); w is [K, N] (1x1 taps)
template <bool VEC>
__global__ __launch_bounds__(256) void pw_narrow_kernel(PwArgs a) {
    const ConvArgs& p = a.c;
    const int b = blockIdx.y;
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= a.plane) return;
    const float* xb = p.x + (size_t)b * p.K * a.plane;
    float4 acc[MAXS];
#pragma unroll
    for (int j = 0; j < MAXS; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < p.K; k0 += CHUNK) {
        float4 v[CHUNK];
#pragma unroll
        for (int q = 0; q < CHUNK; ++q) v[q] = ld4s<VEC>(xb + (size_t)min(k0 + q, p.K - 1) * a.plane, i, a.plane);     // clamped: the scale below is 0 past K
        if (a.in_mask) {
            const float* mb = a.in_mask + (size_t)b * p.K * a.plane;
#pragma unroll
            for (int q = 0; q < CHUNK; ++q) v[q] = mask4(v[q], ld4s<VEC>(mb + (size_t)min(k0 + q, p.K - 1) * a.plane, i, a.plane), a.mpos, a.mneg);
        }
#pragma unroll
        for (int q = 0; q < CHUNK; ++q) {
            const int k = min(k0 + q, p.K - 1);
            const float s = (k0 + q < p.K) ? (p.si ? p.si[(size_t)b * p.K + k] : 1.f) : 0.f;
#pragma unroll
            for (int j = 0; j < MAXS; ++j)
                if (j < p.N) acc[j] = fma4(p.w[(size_t)k * p.N + j] * s, v[q], acc[j]);
        }
    }
    const EpilogueConsts ec = epilogue_consts(p);
    const float4 nz = p.noise ? ld4<VEC>(p.noise + (size_t)b * a.plane, i, a.plane) : make_float4(0.f, 0.f, 0.f, 0.f);
    float so[MAXS], bs[MAXS];
#pragma unroll
    for (int j = 0; j < MAXS; ++j) {
        so[j] = (p.so && j < p.N) ? p.so[(size_t)b * p.N + j] : 1.f;
        bs[j] = (p.bias && j < p.N) ? p.bias[j] : 0.f;
    }
    float* yb = p.y + (size_t)b * p.N * a.plane;
    float4 res[MAXS];
#pragma unroll
    for (int j = 0; j < MAXS; ++j)
        res[j] = (p.residual && j < p.N) ? ld4<VEC>(p.residual + ((size_t)b * p.N + j) * a.plane, i, a.plane) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < MAXS; ++j)
        if (j < p.N)
            st4<VEC>(yb + (size_t)j * a.plane, i, a.plane,
                make_float4(conv_epilogue(ec, acc[j].x, so[j], bs[j], nz.x) + res[j].x, conv_epilogue(ec, acc[j].y, so[j], bs[j], nz.y) + res[j].y,
                            conv_epilogue(ec, acc[j].z, so[j], bs[j], nz.z) + res[j].z, conv_epilogue(ec, acc[j].w, so[j], bs[j], nz.w) + res[j].w));
}

template <bool VEC>
__global__ __launch_bounds__(256) void pw_widen_kernel(PwArgs a) {
    const ConvArgs& p = a.c;
    const int b = blockIdx.y;
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= a.plane) return;
    const float* xb = p.x + (size_t)b * p.K * a.plane;
    float4 xs[MAXS];
#pragma unroll
    for (int j = 0; j < MAXS; ++j) {
        xs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < p.K) {
            const float4 v = ld4<VEC>(xb + (size_t)j * a.plane, i, a.plane);
            const float s = p.si ? p.si[(size_t)b * p.K + j] : 1.f;
            xs[j] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
        }
    }
    const EpilogueConsts ec = epilogue_consts(p);
    const float4 nz = p.noise ? ld4<VEC>(p.noise + (size_t)b * a.plane, i, a.plane) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* yb = p.y + (size_t)b * p.N * a.plane;
    // per-channel factors are uniform (scalar loads): they do not sit between the stores as vector loads would
    constexpr int RB = 8;             // residual rows fetched together, ahead of their stores
    for (int n0 = 0; n0 < p.N; n0 += RB) {
        float4 res[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q)
            res[q] = p.residual ? ld4<VEC>(p.residual + ((size_t)b * p.N + min(n0 + q, p.N - 1)) * a.plane, i, a.plane) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int n = n0 + q;
            if (n >= p.N) break;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < MAXS; ++j)
                if (j < p.K) acc = fma4(p.w[(size_t)j * p.N + n], xs[j], acc);
            const float so = p.so ? p.so[(size_t)b * p.N + n] : 1.f, bs = p.bias ? p.bias[n] : 0.f;
            st4<VEC>(yb + (size_t)n * a.plane, i, a.plane,
                make_float4(conv_epilogue(ec, acc.x, so, bs, nz.x) + res[q].x, conv_epilogue(ec, acc.y, so, bs, nz.y) + res[q].y,
                            conv_epilogue(ec, acc.z, so, bs, nz.z) + res[q].z, conv_epilogue(ec, acc.w, so, bs, nz.w) + res[q].w));
        }
    }
}

// Weight gradient: thin side S (<= 4 planes of `thin`), wide side L planes of `wide`; part[block][l][s] partial sums.
struct PwWgArgs {
    const float* thin; const float* wide; const float* st; const float* sw;     // tensors and their per-sample scales ([B, S] / [B, L])
    float* part;
    int B, S, L;
    long long plane; int vec; int groups_per_block; int thin_is_x;
    // gc_pw_act_wgrad_f32: the wide operand (dy) is multiplied by the activation mask of wide_mask while it is loaded, and thin channel
    // `ones` (= S - 1 when >= 0) is a plane of ones: its row of the result is the bias gradient sum_p dy_masked
    const float* wide_mask; float mpos, mneg; int ones;
};

template <bool VEC>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(PwWgArgs a) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long groups = (a.plane + 3) / 4;                       // 4-pixel groups per plane
    const long long g_begin = (long long)blockIdx.x * a.groups_per_block, g_end = min(groups, g_begin + a.groups_per_block);
    float* out = a.part + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * a.L * a.S;
    const int b = blockIdx.y;
    float ts[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) ts[s] = (a.st && s < a.S) ? a.st[(size_t)b * a.S + s] : 1.f;
    {       // one chunk of 16 wide-side channels per workgroup (blockIdx.z): a 256^2 plane has only 8 pixel blocks per sample
        const int l0 = blockIdx.z * CHUNK;
        float acc[CHUNK][MAXS];
#pragma unroll
        for (int q = 0; q < CHUNK; ++q)
#pragma unroll
            for (int s = 0; s < MAXS; ++s) acc[q][s] = 0.f;
        for (long long g = g_begin + threadIdx.x; g < g_end; g += 256) {
            const long long i = g * 4;
            float4 t[MAXS];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) {
                const int real = a.ones >= 0 ? a.S - 1 : a.S;          // thin planes that exist in memory
                if (a.ones >= 0 && s == a.ones)
                    t[s] = make_float4(i < a.plane ? 1.f : 0.f, i + 1 < a.plane ? 1.f : 0.f, i + 2 < a.plane ? 1.f : 0.f, i + 3 < a.plane ? 1.f : 0.f);
                else
                    t[s] = ld4<VEC>(a.thin + ((size_t)b * real + min(s, real - 1)) * a.plane, i, a.plane);   // clamped duplicates are never written
            }
            float4 v[CHUNK];
#pragma unroll
            for (int q = 0; q < CHUNK; ++q) v[q] = ld4s<VEC>(a.wide + ((size_t)b * a.L + min(l0 + q, a.L - 1)) * a.plane, i, a.plane);
            if (a.wide_mask) {
#pragma unroll
                for (int q = 0; q < CHUNK; ++q)
                    v[q] = mask4(v[q], ld4s<VEC>(a.wide_mask + ((size_t)b * a.L + min(l0 + q, a.L - 1)) * a.plane, i, a.plane), a.mpos, a.mneg);
            }
#pragma unroll
            for (int q = 0; q < CHUNK; ++q)
#pragma unroll
                for (int s = 0; s < MAXS; ++s)
                    acc[q][s] += v[q].x * t[s].x + v[q].y * t[s].y + v[q].z * t[s].z + v[q].w * t[s].w;
        }
        // fixed-order block reduction: lanes (shuffles), then the four waves through LDS
#pragma unroll
        for (int q = 0; q < CHUNK; ++q)
#pragma unroll
            for (int s = 0; s < MAXS; ++s) {
                float r = acc[q][s];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) r += __shfl_down(r, o, 64);
                if (lane == 0) red[wave][q * MAXS + s] = r;
            }
        __syncthreads();
        if (threadIdx.x < CHUNK * MAXS) {
            const int q = threadIdx.x / MAXS, s = threadIdx.x % MAXS;
            if (l0 + q < a.L && s < a.S) {
                const float sum = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
                const float sw = a.sw ? a.sw[(size_t)b * a.L + l0 + q] : 1.f;
                // dw order [k][n]: the thin side is k when it is x, n when it is dy
                out[a.thin_is_x ? (size_t)s * a.L + l0 + q : (size_t)(l0 + q) * a.S + s] = sum * ts[s] * sw;
            }
        }
        __syncthreads();
    }
}

// dw[e] = sum over parts (fixed order: lane i adds parts i, i + 64, ...; then a shuffle tree).  One wave per output element:
// a serial loop over ~10^2..10^3 parts in a single lane would be a chain of dependent loads.
__global__ __launch_bounds__(64) void pw_wgrad_finish_kernel(const float* __restrict__ part, float* __restrict__ dw, int parts, int count) {
    const int e = blockIdx.x;
    float s = 0.f;
    for (int pt = threadIdx.x; pt < parts; pt += 64) s += part[(size_t)pt * count + e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (threadIdx.x == 0) dw[e] = s;
}

// per-sample form: part is ordered [b][block]; samples[b][e] = sum over that sample's blocks, dw[e] = sum_b samples[b][e]
__global__ __launch_bounds__(64) void pw_wgrad_finish_samples_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ samples,
                                                                     int blocks, int batch, int count) {
    const int e = blockIdx.x;
    float tot = 0.f;
    for (int b = 0; b < batch; ++b) {
        float s = 0.f;
        for (int pt = threadIdx.x; pt < blocks; pt += 64) s += part[((size_t)b * blocks + pt) * count + e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (threadIdx.x == 0) samples[(size_t)b * count + e] = s;
        tot += s;
    }
    if (threadIdx.x == 0) dw[e] = tot;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int wgrad_blocks(long long plane) {
    const long long groups = (plane + 3) / 4;
    return (int)std::min<long long>(std::max<long long>(groups / 2048, 1), 128);      // >= 8 groups per lane, <= 128 blocks per sample
}

}  // namespace

// the weight gradient splits over 16-channel chunks as well (pw_wgrad_kernel), so it fills the chip on smaller planes than the forward kernels
bool gcconv::pointwise_thin_wgrad(const gc_conv_desc* d) {
    constexpr int lg = 14;       // pixels x batch; scan: 3 -> 128 @256^2 B = 2: 54 -> 32 us, 512 -> 3 @64^2: 30 -> 24 us, nothing below
    return d && d->kh == 1 && d->kw == 1 && d->up == 1 && d->down == 1 && d->pad_x == 0 && d->pad_y == 0 &&
           (d->in_ch <= MAXS || d->out_ch <= MAXS) && d->batch <= 65535 &&
           (long long)d->out_h * d->out_w * d->batch >= (1LL << lg);
}

bool gcconv::pointwise_thin(const gc_conv_desc* d) {
    // only where the launch is a bandwidth problem: one lane per 4 pixels must still fill the chip (>= 256 workgroups);
    // the low-resolution ToRGB layers (512 channels, <= 128 x 128) stay on the matrix kernels with their split over K
    return d && d->kh == 1 && d->kw == 1 && d->up == 1 && d->down == 1 && d->pad_x == 0 && d->pad_y == 0 &&
           (d->in_ch <= MAXS || d->out_ch <= MAXS) && d->batch <= 65535 &&
           (long long)d->out_h * d->out_w * d->batch >= (1LL << 18);
}

int gcconv::pointwise_conv(const gc_conv_desc* d, const float* x, const float* w, const float* in_scale, const float* out_scale,
                           const gc_conv_epilogue* ep, float* y, gc_stream_t stream) {
    PwArgs a;
    a.c = ConvArgs{x, w, in_scale, out_scale, y, d->batch, d->in_ch, d->out_ch, d->in_h, d->in_w, d->out_h, d->out_w, 0, 0, 0, 0};
    set_epilogue(a.c, ep);
    a.c.k_per_split = 0; a.c.part = nullptr;
    a.plane = (long long)d->out_h * d->out_w;
    a.vec = a.plane % 4 == 0 && aligned16(x) && aligned16(y) && (!a.c.noise || aligned16(a.c.noise)) && (!a.c.residual || aligned16(a.c.residual));
    a.in_mask = nullptr; a.mpos = a.mneg = 1.f;
    const long long groups = (a.plane + 3) / 4;
    dim3 grid((unsigned)((groups + 255) / 256), d->batch);
    hipStream_t s = (hipStream_t)stream;
    if (gc::probing()) return gc::probe_name("%s|up1,down1,k1", d->out_ch <= MAXS ? "pw_narrow_kernel" : "pw_widen_kernel");
    if (d->out_ch <= MAXS) {
        if (a.vec) hipLaunchKernelGGL(pw_narrow_kernel<true>, grid, dim3(256), 0, s, a);
        else       hipLaunchKernelGGL(pw_narrow_kernel<false>, grid, dim3(256), 0, s, a);
    } else {
        if (a.vec) hipLaunchKernelGGL(pw_widen_kernel<true>, grid, dim3(256), 0, s, a);
        else       hipLaunchKernelGGL(pw_widen_kernel<false>, grid, dim3(256), 0, s, a);
    }
    return gc::check_launch("gc_conv2d_f32(pointwise)");
}

size_t gcconv::pointwise_wgrad_workspace(const gc_conv_desc* d) {
    return (size_t)wgrad_blocks((long long)d->out_h * d->out_w) * d->batch * d->in_ch * d->out_ch * sizeof(float);
}

int gcconv::pointwise_wgrad(const gc_conv_desc* d, const float* x, const float* dy, const float* in_scale, const float* out_scale,
                            float* dw, float* dw_samples, void* workspace, gc_stream_t stream) {
    PwWgArgs a;
    const bool thin_is_x = d->in_ch <= MAXS;
    a.thin = thin_is_x ? x : dy; a.wide = thin_is_x ? dy : x;
    a.st = thin_is_x ? in_scale : out_scale; a.sw = thin_is_x ? out_scale : in_scale;
    a.S = thin_is_x ? d->in_ch : d->out_ch; a.L = thin_is_x ? d->out_ch : d->in_ch;
    a.B = d->batch; a.thin_is_x = thin_is_x ? 1 : 0;
    a.plane = (long long)d->out_h * d->out_w;
    a.vec = a.plane % 4 == 0 && aligned16(x) && aligned16(dy);
    a.wide_mask = nullptr; a.mpos = a.mneg = 1.f; a.ones = -1;
    a.part = static_cast<float*>(workspace);
    const int blocks = wgrad_blocks(a.plane);
    a.groups_per_block = (int)(((a.plane + 3) / 4 + blocks - 1) / blocks);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(blocks, d->batch, (a.L + CHUNK - 1) / CHUNK);
    if (a.vec) hipLaunchKernelGGL(pw_wgrad_kernel<true>, grid, dim3(256), 0, s, a);
    else       hipLaunchKernelGGL(pw_wgrad_kernel<false>, grid, dim3(256), 0, s, a);
    int rc = gc::check_launch("gc_conv2d_wgrad_f32(pointwise)");
    if (rc) return rc;
    const int kn = d->in_ch * d->out_ch;
    if (dw_samples) hipLaunchKernelGGL(pw_wgrad_finish_samples_kernel, dim3(kn), dim3(64), 0, s, a.part, dw, dw_samples, blocks, d->batch, kn);
    else            hipLaunchKernelGGL(pw_wgrad_finish_kernel, dim3(kn), dim3(64), 0, s, a.part, dw, blocks * d->batch, kn);
    return gc::check_launch("gc_conv2d_wgrad_f32(pointwise finish)");
}

// ---- the backward of a fused (3 -> C) 1x1 convolution + bias + leaky-ReLU without a separate activation-backward pass ----------------
// (the discriminator's FromRGB ConvLayer, gan_model.py:955, 844-890: its output is the largest activation of D, [B, 32, 1024, 1024])
extern "C" size_t gc_pw_act_wgrad_workspace(int batch, int k, int n, int64_t plane) {
    if (batch <= 0 || k <= 0 || k >= MAXS || n <= 0 || plane <= 0) return 0;
    return (size_t)wgrad_blocks(plane) * batch * (k + 1) * n * sizeof(float);
}

extern "C" int gc_pw_act_wgrad_f32(const float* x, const float* dy, const float* y_ref, float* dw_db, int batch, int k, int n, int64_t plane,
                                   float slope, float gain, void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    if (!x || !dy || !y_ref || !dw_db) return gc::fail(GC_ERR_BAD_ARG, "gc_pw_act_wgrad_f32: null pointer");
    if (batch <= 0 || batch > 65535 || k <= 0 || k >= MAXS || n <= 0 || plane <= 0)
        return gc::fail(GC_ERR_UNSUPPORTED, "gc_pw_act_wgrad_f32: batch %d, %d -> %d channels (at most %d input channels)", batch, k, n, MAXS - 1);
    const size_t need = gc_pw_act_wgrad_workspace(batch, k, n, plane);
    if (!workspace || workspace_bytes < need) return gc::fail(GC_ERR_WORKSPACE, "gc_pw_act_wgrad_f32: workspace %zu < %zu bytes", workspace_bytes, need);
    PwWgArgs a;
    a.thin = x; a.wide = dy; a.st = nullptr; a.sw = nullptr;
    a.S = k + 1; a.L = n; a.B = batch; a.thin_is_x = 1;
    a.plane = plane;
    a.vec = plane % 4 == 0 && aligned16(x) && aligned16(dy) && aligned16(y_ref);
    a.wide_mask = y_ref; a.mpos = gain; a.mneg = gain * slope; a.ones = k;
    a.part = static_cast<float*>(workspace);
    const int blocks = wgrad_blocks(plane);
    a.groups_per_block = (int)(((plane + 3) / 4 + blocks - 1) / blocks);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(blocks, batch, (a.L + CHUNK - 1) / CHUNK);
    if (a.vec) hipLaunchKernelGGL(pw_wgrad_kernel<true>, grid, dim3(256), 0, s, a);
    else       hipLaunchKernelGGL(pw_wgrad_kernel<false>, grid, dim3(256), 0, s, a);
    int rc = gc::check_launch("gc_pw_act_wgrad_f32");
    if (rc) return rc;
    const int count = (k + 1) * n;
    hipLaunchKernelGGL(pw_wgrad_finish_kernel, dim3(count), dim3(64), 0, s, a.part, dw_db, blocks * batch, count);
    return gc::check_launch("gc_pw_act_wgrad_f32(finish)");
}

extern "C" int gc_pw_act_dgrad_f32(const float* dy, const float* y_ref, const float* w, float* gx, int batch, int n, int k, int64_t plane,
                                   float slope, float gain, gc_stream_t stream) {
    if (!dy || !y_ref || !w || !gx) return gc::fail(GC_ERR_BAD_ARG, "gc_pw_act_dgrad_f32: null pointer");
    if (batch <= 0 || batch > 65535 || n <= 0 || k <= 0 || k > MAXS || plane <= 0)
        return gc::fail(GC_ERR_UNSUPPORTED, "gc_pw_act_dgrad_f32: batch %d, %d -> %d channels (at most %d output channels)", batch, n, k, MAXS);
    PwArgs a;
    a.c = ConvArgs{dy, w, nullptr, nullptr, gx, batch, n, k, 1, 1, 1, 1, 0, 0, 0, 0};
    set_epilogue(a.c, nullptr);
    a.c.k_per_split = 0; a.c.part = nullptr;
    a.plane = plane;
    a.vec = plane % 4 == 0 && aligned16(dy) && aligned16(gx) && aligned16(y_ref);
    a.in_mask = y_ref; a.mpos = gain; a.mneg = gain * slope;
    const long long groups = (plane + 3) / 4;
    dim3 grid((unsigned)((groups + 255) / 256), batch);
    if (a.vec) hipLaunchKernelGGL(pw_narrow_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else       hipLaunchKernelGGL(pw_narrow_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return gc::check_launch("gc_pw_act_dgrad_f32");
}
