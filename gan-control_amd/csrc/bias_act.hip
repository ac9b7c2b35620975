// K2  fused (noise +) bias + leaky-ReLU * gain, its masked adjoint, and the per-channel sum that
// yields the bias gradient.  HBM-bound: one read + one write per element (plus the noise plane,
// which is re-read once per channel from L2).  See include/gancontrol_hip.h for the contract.
#include "common.h"

namespace {

__device__ __forceinline__ float lrelu_gain(float v, float slope, float gain) {
    return (v > 0.f ? v : v * slope) * gain;
}

// One (b, c) plane per blockIdx.y; float4 path when `inner` is a multiple of 4 and bases are aligned.
template <bool VEC, bool NOISE>
__global__ __launch_bounds__(256) void bias_act_plane_kernel(
    const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ noise,
    const float* __restrict__ noise_w, float* __restrict__ y, int channels, int64_t inner, float slope, float gain) {
    const int plane = blockIdx.y;
    const int c = plane % channels, b = plane / channels;
    const float bv = bias ? bias[c] : 0.f;
    const float nw = NOISE ? noise_w[0] : 0.f;
    const float* xp = x + (size_t)plane * inner;
    float* yp = y + (size_t)plane * inner;
    const float* np = NOISE ? noise + (size_t)b * inner : nullptr;
    if (VEC) {
        const int64_t n4 = inner >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const gc::f32x4_t xv = gc::stream_load4(xp + 4 * i);
            float4 v = make_float4(xv.x, xv.y, xv.z, xv.w);
            if (NOISE) {
                const float4 nz = reinterpret_cast<const float4*>(np)[i];
                v.x = fmaf(nw, nz.x, v.x); v.y = fmaf(nw, nz.y, v.y); v.z = fmaf(nw, nz.z, v.z); v.w = fmaf(nw, nz.w, v.w);
            }
            v.x = lrelu_gain(v.x + bv, slope, gain); v.y = lrelu_gain(v.y + bv, slope, gain);
            v.z = lrelu_gain(v.z + bv, slope, gain); v.w = lrelu_gain(v.w + bv, slope, gain);
            gc::stream_store4(yp + 4 * i, v.x, v.y, v.z, v.w);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < inner; i += (int64_t)gridDim.x * 256) {
            float v = xp[i];
            if (NOISE) v = fmaf(nw, np[i], v);
            yp[i] = lrelu_gain(v + bv, slope, gain);
        }
    }
}

// Flat variant for small `inner` (mapping network / D head: inner = 1, and the 4x4..16x16 layers).
template <bool NOISE>
__global__ __launch_bounds__(256) void bias_act_flat_kernel(
    const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ noise,
    const float* __restrict__ noise_w, float* __restrict__ y, int channels, int inner, int64_t total, float slope, float gain) {
    const float nw = NOISE ? noise_w[0] : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t plane = i / inner;
        const int p = (int)(i - plane * inner);
        const int c = (int)(plane % channels);
        float v = x[i];
        if (NOISE) v = fmaf(nw, noise[(plane / channels) * inner + p], v);
        y[i] = lrelu_gain(v + (bias ? bias[c] : 0.f), slope, gain);
    }
}

// Every block owns ONE contiguous run of BWD_RUN float4 (16 KB per operand): the blocks resident together then stream neighbouring
// memory (a grid-stride loop has each of them hop through the tensor in 32 MB steps: 4.7 vs 5.5 TB/s on [4, 32, 1024, 1024]; runs of
// 64 KB 5.4-5.6 TB/s, of 16 KB 5.75-5.85, of 8 KB 5.7-5.85: same-box A/B at the end of round 3).
#ifndef GC_BWD_RUN
#define GC_BWD_RUN 1024
#endif
constexpr int BWD_RUN = GC_BWD_RUN;
template <bool VEC>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ yref, float* __restrict__ dx, int64_t count, float pos, float neg) {
    if (VEC) {
        const int64_t n4 = count >> 2;
        const int64_t lo = (int64_t)blockIdx.x * BWD_RUN, hi = min(n4, lo + BWD_RUN);
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
            const gc::f32x4_t g = gc::stream_load4(dy + 4 * i);
            const gc::f32x4_t r = gc::stream_load4(yref + 4 * i);
            float4 o;
            o.x = g.x * (r.x > 0.f ? pos : neg); o.y = g.y * (r.y > 0.f ? pos : neg);
            o.z = g.z * (r.z > 0.f ? pos : neg); o.w = g.w * (r.w > 0.f ? pos : neg);
            gc::stream_store4(dx + 4 * i, o.x, o.y, o.z, o.w);
        }
        if (blockIdx.x == 0)
            for (int64_t i = (n4 << 2) + threadIdx.x; i < count; i += 256) dx[i] = dy[i] * (yref[i] > 0.f ? pos : neg);
    } else {
        const int64_t lo = (int64_t)blockIdx.x * (4 * BWD_RUN), hi = min(count, lo + 4 * BWD_RUN);
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) dx[i] = dy[i] * (yref[i] > 0.f ? pos : neg);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ float block_sum(float v, float* lds) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return r;
}

// Backward through the activation fused with the two reductions its caller needs:
//   dx = dy * mask(y_ref);  psum[plane][chunk] = sum_chunk dx;  pdot[plane][chunk] = sum_chunk dx * noise[b, :]
// (bias gradient = sum of psum over batch and chunks; noise-strength gradient = sum of pdot).  One read of dy and
// y_ref, one write of dx -- the separate channel-sum / product passes over dx disappear.
// SELF adds pself[plane][chunk] = sum_chunk dx * x_pre, where x_pre = act^-1(y_ref) - bias[c] - noise_w * noise[b, :] is
// the activation's INPUT rebuilt from its output (leaky-ReLU is invertible): the out_scale gradient of a convolution whose
// activation ran in the epilogue (the pre-activation tensor was never written), with no pass of its own.
template <bool NOISE, bool SELF>
__global__ __launch_bounds__(256) void bias_act_bwd_reduce_kernel(
    const float* __restrict__ dy, const float* __restrict__ yref, const float* __restrict__ noise, const float* __restrict__ bias,
    const float* __restrict__ noise_w, float* __restrict__ dx, float* __restrict__ psum, float* __restrict__ pdot, float* __restrict__ pself,
    int channels, int64_t inner, int chunks, int64_t chunk_len, float pos, float neg) {
    __shared__ float lds[4];
    const int plane = blockIdx.x / chunks, j = blockIdx.x - plane * chunks;      // 1-D grid, plane-major: no 65535 limit on planes
    const int b = plane / channels, c = plane % channels;
    const size_t base = (size_t)plane * inner;
    const float* np = NOISE ? noise + (size_t)b * inner : nullptr;
    const float bv = (SELF && bias) ? bias[c] : 0.f, nw = (SELF && NOISE) ? noise_w[0] : 0.f;
    const float ipos = 1.f / pos, ineg = 1.f / neg;
    const int64_t lo = (int64_t)j * chunk_len, hi = min(inner, lo + chunk_len);
    float s = 0.f, d = 0.f, q = 0.f;
    auto one = [&](float yv, float dyv, float nz) {
        const float g = dyv * (yv > 0.f ? pos : neg);
        s += g;
        if (NOISE) d = fmaf(g, nz, d);
        if (SELF) q = fmaf(g, yv * (yv > 0.f ? ipos : ineg) - bv - nw * nz, q);
        return g;
    };
    // 16-byte accesses (4-byte alignment is enough on gfx9: planes of odd size start anywhere), then the <= 3 leftovers
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const int64_t n4 = (hi - lo) >> 2;
    for (int64_t v = threadIdx.x; v < n4; v += 256) {
        const int64_t i = lo + 4 * v;
        const f4u y4 = gc::stream_load4u(yref + base + i), g4 = gc::stream_load4u(dy + base + i);
        f4u z4 = {0.f, 0.f, 0.f, 0.f};
        if (NOISE) z4 = *reinterpret_cast<const f4u*>(np + i);
        const f4u o4 = {one(y4.x, g4.x, z4.x), one(y4.y, g4.y, z4.y), one(y4.z, g4.z, z4.z), one(y4.w, g4.w, z4.w)};
        gc::stream_store4u(dx + base + i, o4.x, o4.y, o4.z, o4.w);
    }
    for (int64_t i = lo + 4 * n4 + threadIdx.x; i < hi; i += 256)
        dx[base + i] = one(yref[base + i], dy[base + i], NOISE ? np[i] : 0.f);
    const float rs = block_sum(s, lds);
    if (threadIdx.x == 0) psum[(size_t)plane * chunks + j] = rs;
    if (NOISE) {
        const float rd = block_sum(d, lds);
        if (threadIdx.x == 0) pdot[(size_t)plane * chunks + j] = rd;
    }
    if (SELF) {
        const float rq = block_sum(q, lds);
        if (threadIdx.x == 0) pself[(size_t)plane * chunks + j] = rq;
    }
}

// Adjoint of the pass above (see gc_bias_act_bwd_reduce_adjoint_f32): every cotangent of a chunk reduction is one scalar per
// (plane, chunk), so the whole second-order formula is a single elementwise pass with two small reductions.
__global__ __launch_bounds__(256) void bias_act_bwd_reduce_adjoint_kernel(
    const float* __restrict__ ggx, const float* __restrict__ cs, const float* __restrict__ cd, const float* __restrict__ cw,
    const float* __restrict__ yref, const float* __restrict__ dx, const float* __restrict__ noise, const float* __restrict__ bias,
    const float* __restrict__ noise_w, float* __restrict__ g_dy, float* __restrict__ g_yref, float* __restrict__ pgb, float* __restrict__ pgn,
    int channels, int64_t inner, int chunks, int64_t chunk_len, float pos, float neg) {
    __shared__ float lds[4];
    const int plane = blockIdx.x / chunks, j = blockIdx.x - plane * chunks;      // 1-D grid, plane-major: no 65535 limit on planes
    const int b = plane / channels, c = plane % channels;
    const size_t base = (size_t)plane * inner, pj = (size_t)plane * chunks + j;
    const float* np = noise ? noise + (size_t)b * inner : nullptr;
    const float bv = bias ? bias[c] : 0.f, nw = (noise && noise_w) ? noise_w[0] : 0.f;
    const float ipos = 1.f / pos, ineg = 1.f / neg;
    const float vs = cs ? cs[pj] : 0.f, vd = (cd && noise) ? cd[pj] : 0.f, vw = cw ? cw[pj] : 0.f;
    const int64_t lo = (int64_t)j * chunk_len, hi = min(inner, lo + chunk_len);
    float sb = 0.f, sn = 0.f;
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    auto one = [&](float yv, float gg, float gxv, float nz, float* gyref) {
        const bool up = yv > 0.f;
        const float xpre = yv * (up ? ipos : ineg) - bv - nw * nz;
        const float total = gg + vs + vd * nz + vw * xpre;
        const float wg = vw * gxv;
        sb += wg;
        sn = fmaf(wg, nz, sn);
        *gyref = wg * (up ? ipos : ineg);
        return total * (up ? pos : neg);
    };
    const int64_t n4 = (hi - lo) >> 2;
    for (int64_t v = threadIdx.x; v < n4; v += 256) {
        const int64_t i = lo + 4 * v;
        const f4u y4 = *reinterpret_cast<const f4u*>(yref + base + i);
        f4u g4 = {0.f, 0.f, 0.f, 0.f}, x4 = g4, z4 = g4;
        if (ggx) g4 = *reinterpret_cast<const f4u*>(ggx + base + i);
        if (cw) x4 = *reinterpret_cast<const f4u*>(dx + base + i);
        if (np) z4 = *reinterpret_cast<const f4u*>(np + i);
        float r0, r1, r2, r3;
        const f4u o4 = {one(y4.x, g4.x, x4.x, z4.x, &r0), one(y4.y, g4.y, x4.y, z4.y, &r1), one(y4.z, g4.z, x4.z, z4.z, &r2), one(y4.w, g4.w, x4.w, z4.w, &r3)};
        gc::stream_store4u(g_dy + base + i, o4.x, o4.y, o4.z, o4.w);
        if (g_yref) gc::stream_store4u(g_yref + base + i, r0, r1, r2, r3);
    }
    for (int64_t i = lo + 4 * n4 + threadIdx.x; i < hi; i += 256) {
        float r;
        g_dy[base + i] = one(yref[base + i], ggx ? ggx[base + i] : 0.f, cw ? dx[base + i] : 0.f, np ? np[i] : 0.f, &r);
        if (g_yref) g_yref[base + i] = r;
    }
    if (pgb) {
        const float rb = block_sum(sb, lds);
        if (threadIdx.x == 0) pgb[pj] = rb;
    }
    if (pgn) {
        const float rn = block_sum(sn, lds);
        if (threadIdx.x == 0) pgn[pj] = rn;
    }
}

// partial[plane][chunk] = sum over the chunk of a * b  (per-sample modulation / demodulation gradients)
__global__ __launch_bounds__(256) void plane_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ partial,
                                                        int64_t inner, int chunks, int64_t chunk_len) {
    __shared__ float lds[4];
    const int plane = blockIdx.x / chunks, j = blockIdx.x - plane * chunks;      // 1-D grid, plane-major: no 65535 limit on planes
    const size_t base = (size_t)plane * inner;
    const int64_t lo = (int64_t)j * chunk_len, hi = min(inner, lo + chunk_len);
    float acc = 0.f;
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const int64_t n4 = (hi - lo) >> 2;
    for (int64_t v = threadIdx.x; v < n4; v += 256) {
        const int64_t i = lo + 4 * v;
        const f4u a4 = gc::stream_load4u(a + base + i), b4 = gc::stream_load4u(b + base + i);
        acc = fmaf(a4.x, b4.x, acc); acc = fmaf(a4.y, b4.y, acc); acc = fmaf(a4.z, b4.z, acc); acc = fmaf(a4.w, b4.w, acc);
    }
    for (int64_t i = lo + 4 * n4 + threadIdx.x; i < hi; i += 256) acc = fmaf(a[base + i], b[base + i], acc);
    const float r = block_sum(acc, lds);
    if (threadIdx.x == 0) partial[(size_t)plane * chunks + j] = r;
}

// out[r] = (sum_j partial[r][j]) / den[r], a zero denominator counting as one (a measure-zero modulation factor: the gradient it
// scales is zero as well).  Second stage of the plane reductions and the division by the modulation / demodulation factor in ONE launch
// (it was a reduction + compare + select + divide, four launches of a few microseconds each, per factor per layer per backward pass).
__global__ __launch_bounds__(256) void rows_sum_div_kernel(const float* __restrict__ partial, const float* __restrict__ den, float* __restrict__ out,
                                                           int rows, int chunks) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float acc = 0.f;
    for (int j = 0; j < chunks; ++j) acc += partial[(size_t)r * chunks + j];      // fixed order: deterministic
    if (den) { const float d = den[r]; acc = acc / (d == 0.f ? 1.f : d); }
    out[r] = acc;
}

// plane_dot over row-pitched planes: chunk j of a plane covers rows [j * rpc, (j + 1) * rpc)
__global__ __launch_bounds__(256) void plane_dot_pitched_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ partial,
                                                                int rows, int width, int a_pitch, int b_pitch, int chunks, int rpc) {
    __shared__ float lds[4];
    const int plane = blockIdx.x / chunks, j = blockIdx.x - plane * chunks;
    const float* ap = a + (size_t)plane * rows * a_pitch;
    const float* bp = b + (size_t)plane * rows * b_pitch;
    const int r0 = j * rpc, r1 = min(rows, r0 + rpc);
    // four waves take rows r0 + wave, r0 + wave + 4, ...; a lane walks its row in 16-byte steps (rows of a dense (2H + 1)-wide operand are
    // only 4-byte aligned, which a 16-byte load accepts on gfx9)
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w4 = width >> 2;
    float acc0 = 0.f, acc1 = 0.f;
    for (int r = r0 + wave; r < r1; r += 4) {
        const float* ar = ap + (size_t)r * a_pitch;
        const float* br = bp + (size_t)r * b_pitch;
        int v = lane;
        for (; v + 64 < w4; v += 128) {
            const f4u a0 = *reinterpret_cast<const f4u*>(ar + 4 * v), b0 = *reinterpret_cast<const f4u*>(br + 4 * v);
            const f4u a1 = *reinterpret_cast<const f4u*>(ar + 4 * (v + 64)), b1 = *reinterpret_cast<const f4u*>(br + 4 * (v + 64));
            acc0 = fmaf(a0.x, b0.x, acc0); acc0 = fmaf(a0.y, b0.y, acc0); acc0 = fmaf(a0.z, b0.z, acc0); acc0 = fmaf(a0.w, b0.w, acc0);
            acc1 = fmaf(a1.x, b1.x, acc1); acc1 = fmaf(a1.y, b1.y, acc1); acc1 = fmaf(a1.z, b1.z, acc1); acc1 = fmaf(a1.w, b1.w, acc1);
        }
        for (; v < w4; v += 64) {
            const f4u a0 = *reinterpret_cast<const f4u*>(ar + 4 * v), b0 = *reinterpret_cast<const f4u*>(br + 4 * v);
            acc0 = fmaf(a0.x, b0.x, acc0); acc0 = fmaf(a0.y, b0.y, acc0); acc0 = fmaf(a0.z, b0.z, acc0); acc0 = fmaf(a0.w, b0.w, acc0);
        }
        for (int c = 4 * w4 + lane; c < width; c += 64) acc1 = fmaf(ar[c], br[c], acc1);
    }
    const float s = block_sum(acc0 + acc1, lds);
    if (threadIdx.x == 0) partial[(size_t)plane * chunks + j] = s;
}

// stage 1: partial[c][b * chunks + j] = sum over chunk j of plane (b, c); fixed summation order
__global__ __launch_bounds__(256) void channel_sum_stage1(const float* __restrict__ x, float* __restrict__ partial,
                                                          int batch, int channels, int64_t inner, int chunks, int64_t chunk_len) {
    __shared__ float lds[4];
    const int plane = blockIdx.x / chunks, j = blockIdx.x - plane * chunks;      // 1-D grid, plane-major: no 65535 limit on planes
    const int c = plane % channels, b = plane / channels;
    const float* xp = x + (size_t)plane * inner;
    const int64_t lo = (int64_t)j * chunk_len, hi = min(inner, lo + chunk_len);
    float acc = 0.f;
    if (((inner | lo | hi) & 3) == 0 && (reinterpret_cast<uintptr_t>(xp) & 15) == 0) {
        // 16-byte loads, four independent partial sums per lane (a single dependent chain of dword loads reached 3.7 TB/s; fixed order)
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
        for (int64_t i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
            const gc::f32x4_t v = gc::stream_load4(reinterpret_cast<const float*>(x4 + i));
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
        acc = (a0 + a1) + (a2 + a3);
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) acc += xp[i];
    }
    const float r = block_sum(acc, lds);
    if (threadIdx.x == 0) partial[((size_t)c * batch + b) * chunks + j] = r;
}

// stage 2: out[c] = sum of partial[c][0..n)
__global__ __launch_bounds__(256) void channel_sum_stage2(const float* __restrict__ partial, float* __restrict__ out, int n) {
    __shared__ float lds[4];
    const int c = blockIdx.x;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[(size_t)c * n + i];
    const float r = block_sum(acc, lds);
    if (threadIdx.x == 0) out[c] = r;
}

inline void channel_sum_plan(int64_t inner, int* chunks, int64_t* chunk_len) {
    // 16K elements per block keeps the grid >> 256 CUs on the large layers (end of round 3: 4K-element chunks run the activation-backward
    // reductions 6-8 % faster in isolation, 5.27 -> 5.63 TB/s on [4, 32, 1024, 1024], and change nothing in the training step: kept at 16K)
#ifndef GC_SUM_LEN
#define GC_SUM_LEN 16384
#endif
    int64_t len = GC_SUM_LEN;
    int n = (int)gc::ceil_div64(inner, len);
    if (n < 1) n = 1;
    *chunks = n;
    *chunk_len = len;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int gc_bias_act_f32(const float* x, const float* bias, const float* noise, const float* noise_w,
                               float* y, int batch, int channels, int64_t inner, float slope, float gain, gc_stream_t stream) {
    if (!x || !y) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_f32: null pointer");
    if ((noise == nullptr) != (noise_w == nullptr)) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_f32: noise and noise_w must both be set or both be null");
    if (batch < 0 || channels <= 0 || inner <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_f32: bad extents");
    if (batch == 0) return GC_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t planes = (int64_t)batch * channels;
    if (inner >= 1024 && planes <= 65535) {
        const bool vec = (inner % 4 == 0) && aligned16(x) && aligned16(y) && (!noise || aligned16(noise));
        const int64_t work = vec ? inner / 4 : inner;
        dim3 grid((unsigned)std::min<int64_t>(gc::ceil_div64(work, 256 * 4), 1024), (unsigned)planes);
#define GC_LAUNCH(V, N) hipLaunchKernelGGL((bias_act_plane_kernel<V, N>), grid, dim3(256), 0, s, x, bias, noise, noise_w, y, channels, inner, slope, gain)
        if (vec) { if (noise) GC_LAUNCH(true, true); else GC_LAUNCH(true, false); }
        else     { if (noise) GC_LAUNCH(false, true); else GC_LAUNCH(false, false); }
#undef GC_LAUNCH
        return gc::check_launch("gc_bias_act_f32(plane)");
    }
    if (inner > INT32_MAX) return gc::fail(GC_ERR_UNSUPPORTED, "gc_bias_act_f32: inner too large for the flat path");
    const int64_t total = planes * inner;
    const int blocks = (int)std::min<int64_t>(gc::ceil_div64(total, 256), 4096);
    if (noise)
        hipLaunchKernelGGL(bias_act_flat_kernel<true>, dim3(blocks), dim3(256), 0, s, x, bias, noise, noise_w, y, channels, (int)inner, total, slope, gain);
    else
        hipLaunchKernelGGL(bias_act_flat_kernel<false>, dim3(blocks), dim3(256), 0, s, x, bias, noise, noise_w, y, channels, (int)inner, total, slope, gain);
    return gc::check_launch("gc_bias_act_f32(flat)");
}

extern "C" int gc_bias_act_bwd_f32(const float* dy, const float* y_ref, float* dx, int64_t count,
                                   float slope, float gain, gc_stream_t stream) {
    if (!dy || !y_ref || !dx) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_bwd_f32: null pointer");
    if (count < 0) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_bwd_f32: negative count");
    if (count == 0) return GC_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = aligned16(dy) && aligned16(y_ref) && aligned16(dx);
    const int64_t nblocks = gc::ceil_div64(count, 4 * (int64_t)BWD_RUN);
    if (nblocks > INT32_MAX) return gc::fail(GC_ERR_UNSUPPORTED, "gc_bias_act_bwd_f32: more than 2^31 blocks");
    const int blocks = (int)nblocks;
    if (vec)
        hipLaunchKernelGGL(bias_act_bwd_kernel<true>, dim3(blocks), dim3(256), 0, s, dy, y_ref, dx, count, gain, gain * slope);
    else
        hipLaunchKernelGGL(bias_act_bwd_kernel<false>, dim3(blocks), dim3(256), 0, s, dy, y_ref, dx, count, gain, gain * slope);
    return gc::check_launch("gc_bias_act_bwd_f32");
}

extern "C" size_t gc_channel_sum_workspace(int batch, int channels, int64_t inner) {
    if (batch <= 0 || channels <= 0 || inner <= 0) return 0;
    int chunks; int64_t len;
    channel_sum_plan(inner, &chunks, &len);
    return (size_t)batch * channels * chunks * sizeof(float);
}

extern "C" int gc_channel_sum_f32(const float* x, float* out, int batch, int channels, int64_t inner,
                                  void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    if (!x || !out) return gc::fail(GC_ERR_BAD_ARG, "gc_channel_sum_f32: null pointer");
    if (batch <= 0 || channels <= 0 || inner <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_channel_sum_f32: bad extents");
    if ((int64_t)batch * channels * gc_bias_act_bwd_chunks(inner) > INT32_MAX) return gc::fail(GC_ERR_UNSUPPORTED, "gc_channel_sum_f32: more than 2^31 blocks");
    const size_t need = gc_channel_sum_workspace(batch, channels, inner);
    if (!workspace || workspace_bytes < need) return gc::fail(GC_ERR_WORKSPACE, "gc_channel_sum_f32: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    int chunks; int64_t len;
    channel_sum_plan(inner, &chunks, &len);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(channel_sum_stage1, dim3((unsigned)chunks * (unsigned)(batch * channels)), dim3(256), 0, s, x, partial, batch, channels, inner, chunks, len);
    int rc = gc::check_launch("gc_channel_sum_f32(stage1)");
    if (rc) return rc;
    hipLaunchKernelGGL(channel_sum_stage2, dim3(channels), dim3(256), 0, s, partial, out, batch * chunks);
    return gc::check_launch("gc_channel_sum_f32(stage2)");
}

extern "C" int gc_bias_act_bwd_chunks(int64_t inner) {
    if (inner <= 0) return 0;
    int chunks; int64_t len;
    channel_sum_plan(inner, &chunks, &len);
    return chunks;
}

extern "C" int gc_bias_act_bwd_reduce_self_f32(const float* dy, const float* y_ref, const float* noise, const float* bias, const float* noise_w,
                                               float* dx, float* psum, float* pdot, float* pself, int batch, int channels, int64_t inner,
                                               float slope, float gain, gc_stream_t stream) {
    if (!dy || !y_ref || !dx || !psum) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_bwd_reduce_f32: null pointer");
    if ((noise == nullptr) != (pdot == nullptr)) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_bwd_reduce_f32: noise and pdot go together");
    if (pself && noise && !noise_w) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_bwd_reduce_f32: pself with noise needs noise_w");
    if (pself && (slope == 0.f || gain == 0.f)) return gc::fail(GC_ERR_UNSUPPORTED, "gc_bias_act_bwd_reduce_f32: pself needs an invertible activation (slope, gain != 0)");
    if (batch <= 0 || channels <= 0 || inner <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_bwd_reduce_f32: bad extents");
    hipStream_t s = (hipStream_t)stream;
    int chunks; int64_t len;
    channel_sum_plan(inner, &chunks, &len);
    if ((int64_t)batch * channels * chunks > INT32_MAX) return gc::fail(GC_ERR_UNSUPPORTED, "gc_bias_act_bwd_reduce_f32: more than 2^31 blocks");
    dim3 grid((unsigned)chunks * (unsigned)(batch * channels));
#define GC_LAUNCH(N, S) hipLaunchKernelGGL((bias_act_bwd_reduce_kernel<N, S>), grid, dim3(256), 0, s, dy, y_ref, noise, bias, noise_w, dx, psum, pdot, pself, \
                                           channels, inner, chunks, len, gain, gain * slope)
    if (noise) { if (pself) GC_LAUNCH(true, true); else GC_LAUNCH(true, false); }
    else       { if (pself) GC_LAUNCH(false, true); else GC_LAUNCH(false, false); }
#undef GC_LAUNCH
    return gc::check_launch("gc_bias_act_bwd_reduce_f32");
}

extern "C" int gc_bias_act_bwd_reduce_adjoint_f32(const float* ggx, const float* cs, const float* cd, const float* cw, const float* y_ref, const float* dx,
                                                  const float* noise, const float* bias, const float* noise_w, float* g_dy, float* g_yref, float* pgb, float* pgn,
                                                  int batch, int channels, int64_t inner, float slope, float gain, gc_stream_t stream) {
    if (!y_ref || !g_dy) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_bwd_reduce_adjoint_f32: null pointer");
    if (cw && !dx) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_bwd_reduce_adjoint_f32: cw needs dx");
    if (cw && noise && !noise_w) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_bwd_reduce_adjoint_f32: cw with noise needs noise_w");
    if (cd && !noise) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_bwd_reduce_adjoint_f32: cd needs noise");
    if (slope == 0.f || gain == 0.f) return gc::fail(GC_ERR_UNSUPPORTED, "gc_bias_act_bwd_reduce_adjoint_f32: needs an invertible activation (slope, gain != 0)");
    if (batch <= 0 || channels <= 0 || inner <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_bias_act_bwd_reduce_adjoint_f32: bad extents");
    int chunks; int64_t len;
    channel_sum_plan(inner, &chunks, &len);
    if ((int64_t)batch * channels * chunks > INT32_MAX) return gc::fail(GC_ERR_UNSUPPORTED, "gc_bias_act_bwd_reduce_adjoint_f32: more than 2^31 blocks");
    hipLaunchKernelGGL(bias_act_bwd_reduce_adjoint_kernel, dim3((unsigned)chunks * (unsigned)(batch * channels)), dim3(256), 0, (hipStream_t)stream, ggx, cs, cd, cw, y_ref, dx,
                       noise, bias, noise_w, g_dy, g_yref, pgb, pgn, channels, inner, chunks, len, gain, gain * slope);
    return gc::check_launch("gc_bias_act_bwd_reduce_adjoint_f32");
}

extern "C" int gc_bias_act_bwd_reduce_f32(const float* dy, const float* y_ref, const float* noise, float* dx,
                                          float* psum, float* pdot, int batch, int channels, int64_t inner,
                                          float slope, float gain, gc_stream_t stream) {
    return gc_bias_act_bwd_reduce_self_f32(dy, y_ref, noise, nullptr, nullptr, dx, psum, pdot, nullptr, batch, channels, inner, slope, gain, stream);
}

static void plane_dot_pitched_plan(int rows, int* chunks, int* rpc) {
    *rpc = rows <= 64 ? rows : 32;               // 32 rows per block: >= 8 K elements at the widths that are ever pitched (>= 129)
    *chunks = (rows + *rpc - 1) / *rpc;
}

extern "C" int gc_plane_dot_pitched_chunks(int rows) {
    if (rows <= 0) return 0;
    int chunks, rpc;
    plane_dot_pitched_plan(rows, &chunks, &rpc);
    return chunks;
}

extern "C" int gc_plane_dot_pitched_f32(const float* a, const float* b, float* partial, int planes, int rows, int width, int a_pitch, int b_pitch,
                                        gc_stream_t stream) {
    if (!a || !b || !partial) return gc::fail(GC_ERR_BAD_ARG, "gc_plane_dot_pitched_f32: null pointer");
    if (planes <= 0 || rows <= 0 || width <= 0 || a_pitch < width || b_pitch < width) return gc::fail(GC_ERR_BAD_ARG, "gc_plane_dot_pitched_f32: bad extents");
    int chunks, rpc;
    plane_dot_pitched_plan(rows, &chunks, &rpc);
    if ((int64_t)planes * chunks > INT32_MAX) return gc::fail(GC_ERR_UNSUPPORTED, "gc_plane_dot_pitched_f32: more than 2^31 blocks");
    hipLaunchKernelGGL(plane_dot_pitched_kernel, dim3((unsigned)planes * (unsigned)chunks), dim3(256), 0, (hipStream_t)stream,
                       a, b, partial, rows, width, a_pitch, b_pitch, chunks, rpc);
    return gc::check_launch("gc_plane_dot_pitched_f32");
}

extern "C" int gc_rows_sum_div_f32(const float* partial, const float* den, float* out, int rows, int chunks, gc_stream_t stream) {
    if (!partial || !out) return gc::fail(GC_ERR_BAD_ARG, "gc_rows_sum_div_f32: null pointer");
    if (rows <= 0 || chunks <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_rows_sum_div_f32: bad extents");
    hipLaunchKernelGGL(rows_sum_div_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial, den, out, rows, chunks);
    return gc::check_launch("gc_rows_sum_div_f32");
}

extern "C" int gc_plane_dot_f32(const float* a, const float* b, float* partial, int planes, int64_t inner, gc_stream_t stream) {
    if (!a || !b || !partial) return gc::fail(GC_ERR_BAD_ARG, "gc_plane_dot_f32: null pointer");
    if (planes <= 0 || inner <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_plane_dot_f32: bad extents");
    int chunks; int64_t len;
    channel_sum_plan(inner, &chunks, &len);
    if ((int64_t)planes * chunks > INT32_MAX) return gc::fail(GC_ERR_UNSUPPORTED, "gc_plane_dot_f32: more than 2^31 blocks");
    hipLaunchKernelGGL(plane_dot_kernel, dim3((unsigned)chunks * (unsigned)planes), dim3(256), 0, (hipStream_t)stream, a, b, partial, inner, chunks, len);
    return gc::check_launch("gc_plane_dot_f32");
}
