// Grouped dense layers of the style path: many small `y = alpha * x @ W^T + beta * bias` products in ONE launch.
// See gc_grouped_linear_f32 / gc_grouped_linear_bwd_x_f32 / gc_grouped_linear_bwd_w_f32 in include/gancontrol_hip.h.
//
// A generator forward pass evaluates 26 style modulations (EqualLinear 512 -> IC of every ModulatedConv2d, gan_model.py:171-202, 281-283)
// and 18 demodulation sums (`sum_ic s^2 * sum_k W^2`, :284-293) on [B, 512] operands with B = 2 .. 8: 44 GEMM calls of 5 - 8 us each,
// three times that in the backward pass and again in the second-order pass of the path-length regulariser -- the device spends more time
// between these launches than in them.  Here all layers of one kind are one launch: every group g has its own operand pointers and
// extents (a table passed by value), the weights (18.6 MB for the 26 modulations of a 1024 x 1024 generator) are streamed exactly
// once with 16-byte loads, and the three kernels are each other's derivatives (forward, input gradient, weight gradient), so the
// autograd layer closes under differentiation without any other operation.  Every sum runs in a fixed order: results are bit-identical
// from run to run and from rank to rank.
#include <algorithm>
#include <cstdint>

#include "common.h"

namespace {

constexpr int MAXG = 32;         // groups per launch (the table travels in the kernel arguments)
constexpr int BT = 8;            // samples per register tile

struct Group {
    const float* x; const float* w; const float* bias; float* y;      // roles per kernel: see the C ABI comments
    int n, k;
    long long x_stride;
    float alpha, beta;
    int first_block;             // prefix sum of the blocks this group owns
};

struct GArgs {
    Group g[MAXG];
    int n_groups, batch;
};

__device__ __forceinline__ int group_of(const GArgs& a, int block) {
    int g = 0;
#pragma unroll 1
    for (int i = 1; i < a.n_groups; ++i) g = (block >= a.g[i].first_block) ? i : g;
    return g;
}

__device__ __forceinline__ float dot4(float4 a, float4 b, float s) {
    s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); return fmaf(a.w, b.w, s);
}

// ---- forward: y[b, j] = alpha * sum_k x[b, k] w[j, k] + beta * bias[j] -----------------------------------------------------------
// A block owns ROWS consecutive rows j of one group.  The lanes of a wave split a row along k (16-byte loads, consecutive lanes on
// consecutive addresses); when a row is shorter than a wave's 256 floats several rows share the wave.  x (<= BT samples at a time) sits
// in LDS; the partial sums of a row meet in a fixed-order butterfly over the lanes that share it.
constexpr int FWD_ROWS = 16;

__global__ __launch_bounds__(256) void style_glin_fwd_kernel(GArgs a) {
    extern __shared__ float xs[];                      // [BT][k]
    const int gi = group_of(a, blockIdx.x);
    const Group& G = a.g[gi];
    const int k = G.k, n = G.n;
    const int k4 = k >> 2;
    int lpr = 1;                                       // lanes per row: the power of two >= k / 4, at most 64
    while (lpr < k4 && lpr < 64) lpr <<= 1;
    const int rows_per_pass = 64 / lpr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & (lpr - 1), r_in_pass = lane / lpr;
    const int row0 = (blockIdx.x - G.first_block) * FWD_ROWS;
    for (int b0 = 0; b0 < a.batch; b0 += BT) {
        const int nb = min(BT, a.batch - b0);
        __syncthreads();
        for (int i = threadIdx.x; i < nb * k4; i += 256) {
            const int b = i / k4, q = i - b * k4;
            reinterpret_cast<float4*>(xs)[b * k4 + q] = *reinterpret_cast<const float4*>(G.x + (long long)(b0 + b) * G.x_stride + 4 * q);
        }
        __syncthreads();
        // the four waves take the passes of the block round-robin
        const int passes = (FWD_ROWS + rows_per_pass - 1) / rows_per_pass;
        for (int p = wave; p < passes; p += 4) {
            const int j = row0 + p * rows_per_pass + r_in_pass;
            const bool live = j < n && (p * rows_per_pass + r_in_pass) < FWD_ROWS;
            float acc[BT];
#pragma unroll
            for (int b = 0; b < BT; ++b) acc[b] = 0.f;
            if (live) {
                const float4* wrow = reinterpret_cast<const float4*>(G.w + (long long)j * k);
                for (int q = c; q < k4; q += lpr) {
                    const float4 w4 = wrow[q];
#pragma unroll
                    for (int b = 0; b < BT; ++b)
                        if (b < nb) acc[b] = dot4(w4, reinterpret_cast<const float4*>(xs)[b * k4 + q], acc[b]);
                }
            }
            // fixed-order butterfly over the lanes of a row
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                float v = acc[b];
                for (int off = lpr >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                acc[b] = v;
            }
            if (live && c == 0) {
                const float bj = G.bias ? G.beta * G.bias[j] : 0.f;
#pragma unroll
                for (int b = 0; b < BT; ++b)
                    if (b < nb) G.y[(long long)(b0 + b) * n + j] = fmaf(G.alpha, acc[b], bj);
            }
        }
    }
}

// ---- input gradient: gx[b, k] = alpha * sum_j gy[b, j] w[j, k] --------------------------------------------------------------------
// roles: y = gy (read), x = gx (written, rows x_stride apart).  A block owns a slice of BX_K = 32 consecutive k of one group and walks ALL
// rows j: 8 lanes cover the slice with 16-byte loads (one 128-byte line per row), 32 row-lanes take the rows round-robin, the 32 partial
// sums meet in LDS in a fixed order.
constexpr int BX_K = 32;

__global__ __launch_bounds__(256) void style_glin_bwd_x_kernel(GArgs a) {
    extern __shared__ float sm[];                      // gy tile [BT][n]  |  partials [32][BT][BX_K]
    const int gi = group_of(a, blockIdx.x);
    const Group& G = a.g[gi];
    const int k = G.k, n = G.n;
    const int kl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int k0 = (blockIdx.x - G.first_block) * BX_K + 4 * kl;
    float* gys = sm;
    float* part = sm + BT * n;
    for (int b0 = 0; b0 < a.batch; b0 += BT) {
        const int nb = min(BT, a.batch - b0);
        __syncthreads();
        for (int i = threadIdx.x; i < nb * n; i += 256) {
            const int b = i / n, j = i - b * n;
            gys[b * n + j] = G.y[(long long)(b0 + b) * n + j];
        }
        __syncthreads();
        float acc[BT][4];
#pragma unroll
        for (int b = 0; b < BT; ++b) acc[b][0] = acc[b][1] = acc[b][2] = acc[b][3] = 0.f;
        if (k0 < k) {
            for (int j = rl; j < n; j += 32) {
                const float4 w4 = *reinterpret_cast<const float4*>(G.w + (long long)j * k + k0);
#pragma unroll
                for (int b = 0; b < BT; ++b) {
                    if (b < nb) {
                        const float g = gys[b * n + j];
                        acc[b][0] = fmaf(g, w4.x, acc[b][0]); acc[b][1] = fmaf(g, w4.y, acc[b][1]);
                        acc[b][2] = fmaf(g, w4.z, acc[b][2]); acc[b][3] = fmaf(g, w4.w, acc[b][3]);
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < BT; ++b)
            *reinterpret_cast<float4*>(part + ((rl * BT + b) * BX_K + 4 * kl)) = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
        __syncthreads();
        // BT * BX_K = 256 outputs, one per thread: sum the 32 row-lane partials in order
        const int b = threadIdx.x / BX_K, kk = threadIdx.x - b * BX_K;
        const int kg = (blockIdx.x - G.first_block) * BX_K + kk;
        if (b < nb && kg < k) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) s += part[(r * BT + b) * BX_K + kk];
            const_cast<float*>(G.x)[(long long)(b0 + b) * G.x_stride + kg] = G.alpha * s;
        }
    }
}

// ---- weight gradient: gw[j, k] = alpha * sum_b gy[b, j] x[b, k] ;  gbias[j] = beta * sum_b gy[b, j] -------------------------------
// roles: w = gw (written), bias = gbias (written, may be null), y = gy (read), x (read).  One lane per four consecutive k of a row: a
// rank-`batch` update streamed out with 16-byte stores.
__global__ __launch_bounds__(256) void style_glin_bwd_w_kernel(GArgs a) {
    const int gi = group_of(a, blockIdx.x);
    const Group& G = a.g[gi];
    const int k4 = G.k >> 2;
    const long long e = (long long)(blockIdx.x - G.first_block) * 256 + threadIdx.x;
    if (e >= (long long)G.n * k4) return;
    const int j = (int)(e / k4), q = (int)(e - (long long)j * k4);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float sb = 0.f;
    for (int b = 0; b < a.batch; ++b) {
        const float g = G.y[(long long)b * G.n + j];
        const float4 x4 = *reinterpret_cast<const float4*>(G.x + (long long)b * G.x_stride + 4 * q);
        s.x = fmaf(g, x4.x, s.x); s.y = fmaf(g, x4.y, s.y); s.z = fmaf(g, x4.z, s.z); s.w = fmaf(g, x4.w, s.w);
        sb += g;
    }
    *reinterpret_cast<float4*>(const_cast<float*>(G.w) + (long long)j * G.k + 4 * q) = make_float4(G.alpha * s.x, G.alpha * s.y, G.alpha * s.z, G.alpha * s.w);
    if (q == 0 && G.bias) const_cast<float*>(G.bias)[j] = G.beta * sb;
}

// ---- sum over the taps of W^2 (the weight half of the demodulation coefficient) for many weight tensors, and its gradient ----------
constexpr int MAXWS = 48;
struct WsqGroup { const float* w; const float* g; float* out; int rows, taps, first; };
struct WsqArgs { WsqGroup g[MAXWS]; int n_groups; };

template <bool BWD>
__global__ __launch_bounds__(256) void style_wsq_kernel(WsqArgs a) {
    int gi = 0;
#pragma unroll 1
    for (int i = 1; i < a.n_groups; ++i) gi = ((int)blockIdx.x >= a.g[i].first) ? i : gi;
    const WsqGroup& G = a.g[gi];
    const int r = (blockIdx.x - G.first) * 256 + threadIdx.x;
    if (r >= G.rows) return;
    const float* w = G.w + (long long)r * G.taps;
    if (!BWD) {
        float s = 0.f;
        for (int t = 0; t < G.taps; ++t) s = fmaf(w[t], w[t], s);
        G.out[r] = s;
    } else {
        const float g2 = 2.f * G.g[r];
        float* o = G.out + (long long)r * G.taps;
        for (int t = 0; t < G.taps; ++t) o[t] = g2 * w[t];
    }
}

int launch_wsq(const gc_wsq_group* groups, int n_groups, gc_stream_t stream, bool bwd, const char* what) {
    if (n_groups < 0 || (n_groups > 0 && !groups)) return gc::fail(GC_ERR_BAD_ARG, "%s: bad group table", what);
    for (int first = 0; first < n_groups; first += MAXWS) {
        WsqArgs a;
        a.n_groups = std::min(MAXWS, n_groups - first);
        long long blocks = 0;
        for (int i = 0; i < a.n_groups; ++i) {
            const gc_wsq_group& g = groups[first + i];
            if (!g.w || !g.out || (bwd && !g.g)) return gc::fail(GC_ERR_BAD_ARG, "%s: group %d has a null operand", what, first + i);
            if (g.rows <= 0 || g.taps <= 0) return gc::fail(GC_ERR_BAD_ARG, "%s: group %d has extents rows = %d, taps = %d", what, first + i, g.rows, g.taps);
            a.g[i] = WsqGroup{g.w, g.g, g.out, g.rows, g.taps, (int)blocks};
            blocks += gc::ceil_div(g.rows, 256);
            if (blocks > 2147483647LL) return gc::fail(GC_ERR_UNSUPPORTED, "%s: too many blocks", what);
        }
        if (!blocks) continue;
        if (bwd) hipLaunchKernelGGL(style_wsq_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
        else     hipLaunchKernelGGL(style_wsq_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    }
    return gc::check_launch(what);
}

enum { K_FWD = 0, K_BWD_X = 1, K_BWD_W = 2 };

int launch(const gc_glin_group* groups, int n_groups, int batch, gc_stream_t stream, int kind, const char* what) {
    if (n_groups < 0 || batch <= 0) return gc::fail(GC_ERR_BAD_ARG, "%s: n_groups %d, batch %d", what, n_groups, batch);
    if (n_groups > 0 && !groups) return gc::fail(GC_ERR_BAD_ARG, "%s: null group table", what);
    for (int i = 0; i < n_groups; ++i) {
        const gc_glin_group& g = groups[i];
        if (!g.x || !g.w || !g.y) return gc::fail(GC_ERR_BAD_ARG, "%s: group %d has a null operand", what, i);
        if (g.n <= 0 || g.k <= 0) return gc::fail(GC_ERR_BAD_ARG, "%s: group %d has extents n = %d, k = %d", what, i, g.n, g.k);
        if (g.k % 4 != 0 || g.x_stride % 4 != 0 || g.x_stride < g.k)
            return gc::fail(GC_ERR_UNSUPPORTED, "%s: group %d: k = %d and the row stride %lld must be multiples of 4 (16-byte loads), stride >= k", what, i, g.k, (long long)g.x_stride);
        if ((reinterpret_cast<uintptr_t>(g.x) | reinterpret_cast<uintptr_t>(g.w)) & 15)
            return gc::fail(GC_ERR_UNSUPPORTED, "%s: group %d: x and w must be 16-byte aligned", what, i);
        if (kind == K_BWD_X && (size_t)(BT * g.n + 32 * BT * BX_K) * 4 > 160 * 1024)
            return gc::fail(GC_ERR_UNSUPPORTED, "%s: group %d: n = %d does not fit the LDS tile", what, i, g.n);
        if (kind == K_FWD && (size_t)BT * g.k * 4 > 160 * 1024)
            return gc::fail(GC_ERR_UNSUPPORTED, "%s: group %d: k = %d does not fit the LDS tile", what, i, g.k);
    }
    for (int first = 0; first < n_groups; first += MAXG) {
        GArgs a;
        a.n_groups = n_groups - first < MAXG ? n_groups - first : MAXG;
        a.batch = batch;
        long long blocks = 0;
        size_t lds = 0;
        for (int i = 0; i < a.n_groups; ++i) {
            const gc_glin_group& g = groups[first + i];
            a.g[i] = Group{g.x, g.w, g.bias, g.y, g.n, g.k, (long long)g.x_stride, g.alpha, g.beta, (int)blocks};
            if (kind == K_FWD) { blocks += gc::ceil_div(g.n, FWD_ROWS); lds = std::max(lds, (size_t)BT * g.k * 4); }
            else if (kind == K_BWD_X) { blocks += gc::ceil_div(g.k, BX_K); lds = std::max(lds, (size_t)(BT * g.n + 32 * BT * BX_K) * 4); }
            else blocks += gc::ceil_div64((long long)g.n * (g.k / 4), 256);
            if (blocks > 2147483647LL) return gc::fail(GC_ERR_UNSUPPORTED, "%s: too many blocks", what);
        }
        if (blocks == 0) continue;
        hipStream_t s = (hipStream_t)stream;
        if (kind == K_FWD) {
            if (lds > 64 * 1024) {
                static bool done[16] = {false};
                if (int rc = gc::allow_dynamic_lds(reinterpret_cast<const void*>(style_glin_fwd_kernel), 160 * 1024, done, what)) return rc;
            }
            hipLaunchKernelGGL(style_glin_fwd_kernel, dim3((unsigned)blocks), dim3(256), lds, s, a);
        } else if (kind == K_BWD_X) {
            if (lds > 64 * 1024) {
                static bool done[16] = {false};
                if (int rc = gc::allow_dynamic_lds(reinterpret_cast<const void*>(style_glin_bwd_x_kernel), 160 * 1024, done, what)) return rc;
            }
            hipLaunchKernelGGL(style_glin_bwd_x_kernel, dim3((unsigned)blocks), dim3(256), lds, s, a);
        } else {
            hipLaunchKernelGGL(style_glin_bwd_w_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
        }
        int rc = gc::check_launch(what);
        if (rc != GC_OK) return rc;
    }
    return GC_OK;
}

}  // namespace

extern "C" int gc_grouped_linear_f32(const gc_glin_group* groups, int n_groups, int batch, gc_stream_t stream) {
    return launch(groups, n_groups, batch, stream, K_FWD, "gc_grouped_linear_f32");
}

extern "C" int gc_grouped_linear_bwd_x_f32(const gc_glin_group* groups, int n_groups, int batch, gc_stream_t stream) {
    return launch(groups, n_groups, batch, stream, K_BWD_X, "gc_grouped_linear_bwd_x_f32");
}

extern "C" int gc_grouped_linear_bwd_w_f32(const gc_glin_group* groups, int n_groups, int batch, gc_stream_t stream) {
    return launch(groups, n_groups, batch, stream, K_BWD_W, "gc_grouped_linear_bwd_w_f32");
}

extern "C" int gc_weight_sq_grouped_f32(const gc_wsq_group* groups, int n_groups, gc_stream_t stream) {
    return launch_wsq(groups, n_groups, stream, false, "gc_weight_sq_grouped_f32");
}

extern "C" int gc_weight_sq_bwd_grouped_f32(const gc_wsq_group* groups, int n_groups, gc_stream_t stream) {
    return launch_wsq(groups, n_groups, stream, true, "gc_weight_sq_bwd_grouped_f32");
}
