// f-2: the forward pass of the FID feature network (fid_utils/inception.py:17-165 over overwrite_inception.py) --
// inference only, fp32.  Four primitives cover its 94 convolutions and its pooling / resizing steps:
//   gc_conv2d_bn_relu_f32     any kh x kw (1x1, 3x3, 5x5, 1x7, 7x1, 1x3, 3x1), stride 1 / 2, zero padding, the folded BatchNorm scale / shift and
//                             the ReLU in the epilogue, output written at a CHANNEL OFFSET of a wider tensor (the branches of an Inception block
//                             write straight into the concatenated result: no torch.cat pass)
//   gc_pool2d_f32             3x3 max / average pooling (average without the padded zeros: the FID patch), same channel-offset output
//   gc_global_avgpool_f32     AdaptiveAvgPool2d((1, 1))
//   gc_resize_bilinear_f32    F.interpolate(mode='bilinear', align_corners=False) fused with the 2x - 1 input normalisation
#include "common.h"

namespace {

constexpr int OCT = 64, TILE = 8, KC = 8;     // 64 output channels x (8 x 8) pixels per workgroup, 8 input channels per LDS stage

struct DirectArgs {
    const float* x; const float* w; const float* scale; const float* shift; float* y;
    int B, K, N, in_h, in_w, out_h, out_w, kh, kw, stride, pad_y, pad_x, relu;
    int out_channels, chan_off;          // the output tensor has out_channels planes per sample; this launch writes [chan_off, chan_off + N)
    int tiles_x;
};

// Direct convolution on the vector ALUs: a thread owns 4 output channels x 4 consecutive pixels of a row (16 accumulators); per chunk of
// 8 input channels the workgroup stages the halo'd input patch and the [chunk][tap][64 oc] weight slab in LDS.  The network is ~11 GFLOP
// per 299 x 299 image; this kernel is not on the training hot path (FID is evaluated every 10 000 iterations, tracker.py:322-341).
__global__ __launch_bounds__(256) void conv_direct_kernel(DirectArgs a) {
    extern __shared__ float lds[];
    const int taps = a.kh * a.kw;
    const int ph = (TILE - 1) * a.stride + a.kh, pw = (TILE - 1) * a.stride + a.kw;
    float* patch = lds;                                   // [KC][ph][pw]
    float* wl = lds + KC * ph * pw;                       // [KC][taps][OCT]
    const int tid = threadIdx.x;
    const int og = tid >> 4, pg = tid & 15;               // 16 channel groups of 4, 16 pixel groups of 4
    const int prow = pg >> 1, pcol = (pg & 1) * 4;
    const int tile = blockIdx.x, ty0 = (tile / a.tiles_x) * TILE, tx0 = (tile % a.tiles_x) * TILE;
    const int n0 = blockIdx.y * OCT, b = blockIdx.z;
    const int iy0 = ty0 * a.stride - a.pad_y, ix0 = tx0 * a.stride - a.pad_x;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const float* xb = a.x + (size_t)b * a.K * a.in_h * a.in_w;
    for (int k0 = 0; k0 < a.K; k0 += KC) {
        __syncthreads();
        for (int i = tid; i < KC * ph * pw; i += 256) {
            const int c = i / (ph * pw), r = (i / pw) % ph, q = i % pw;
            const int iy = iy0 + r, ix = ix0 + q, k = k0 + c;
            patch[i] = (k < a.K && iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w) ? xb[((size_t)k * a.in_h + iy) * a.in_w + ix] : 0.f;
        }
        for (int i = tid; i < KC * taps * OCT; i += 256) {
            const int oc = i % OCT, t = (i / OCT) % taps, c = i / (OCT * taps);
            const int n = n0 + oc, k = k0 + c;
            wl[i] = (n < a.N && k < a.K) ? a.w[((size_t)n * a.K + k) * taps + t] : 0.f;      // reference layout [N, K, kh, kw]
        }
        __syncthreads();
        for (int c = 0; c < KC; ++c) {
            const float* pc = patch + c * ph * pw + (prow * a.stride) * pw + pcol * a.stride;
            const float* wc = wl + c * taps * OCT + og * 4;
            for (int ty = 0; ty < a.kh; ++ty) {
                for (int tx = 0; tx < a.kw; ++tx) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wc + (ty * a.kw + tx) * OCT);
                    const float* px = pc + ty * pw + tx;
                    const float x0 = px[0], x1 = px[a.stride], x2 = px[2 * a.stride], x3 = px[3 * a.stride];
                    acc[0][0] = fmaf(w4.x, x0, acc[0][0]); acc[0][1] = fmaf(w4.x, x1, acc[0][1]); acc[0][2] = fmaf(w4.x, x2, acc[0][2]); acc[0][3] = fmaf(w4.x, x3, acc[0][3]);
                    acc[1][0] = fmaf(w4.y, x0, acc[1][0]); acc[1][1] = fmaf(w4.y, x1, acc[1][1]); acc[1][2] = fmaf(w4.y, x2, acc[1][2]); acc[1][3] = fmaf(w4.y, x3, acc[1][3]);
                    acc[2][0] = fmaf(w4.z, x0, acc[2][0]); acc[2][1] = fmaf(w4.z, x1, acc[2][1]); acc[2][2] = fmaf(w4.z, x2, acc[2][2]); acc[2][3] = fmaf(w4.z, x3, acc[2][3]);
                    acc[3][0] = fmaf(w4.w, x0, acc[3][0]); acc[3][1] = fmaf(w4.w, x1, acc[3][1]); acc[3][2] = fmaf(w4.w, x2, acc[3][2]); acc[3][3] = fmaf(w4.w, x3, acc[3][3]);
                }
            }
        }
    }
    const int oy = ty0 + prow;
    if (oy >= a.out_h) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + og * 4 + i;
        if (n >= a.N) continue;
        const float sc = a.scale ? a.scale[n] : 1.f, sh = a.shift ? a.shift[n] : 0.f;
        float* yp = a.y + (((size_t)b * a.out_channels + a.chan_off + n) * a.out_h + oy) * a.out_w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = tx0 + pcol + j;
            if (ox >= a.out_w) continue;
            float v = fmaf(acc[i][j], sc, sh);
            if (a.relu) v = fmaxf(v, 0.f);
            yp[ox] = v;
        }
    }
}

// mode 0: max, 1: average over the taps inside the image (count_include_pad = False)
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int in_h, int in_w, int out_h, int out_w,
                                                   int k, int stride, int pad, int mode, int out_channels, int chan_off) {
    const size_t total = (size_t)B * C * out_h * out_w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % out_w), oy = (int)((i / out_w) % out_h);
        const int c = (int)((i / ((size_t)out_w * out_h)) % C), b = (int)(i / ((size_t)out_w * out_h * C));
        const float* xp = x + ((size_t)b * C + c) * in_h * in_w;
        float acc = mode == 0 ? -INFINITY : 0.f;
        int n = 0;
        for (int ty = 0; ty < k; ++ty) {
            const int iy = oy * stride - pad + ty;
            if (iy < 0 || iy >= in_h) continue;
            for (int tx = 0; tx < k; ++tx) {
                const int ix = ox * stride - pad + tx;
                if (ix < 0 || ix >= in_w) continue;
                const float v = xp[(size_t)iy * in_w + ix];
                acc = mode == 0 ? fmaxf(acc, v) : acc + v;
                ++n;
            }
        }
        y[(((size_t)b * out_channels + chan_off + c) * out_h + oy) * out_w + ox] = mode == 0 ? acc : acc / (float)max(n, 1);
    }
}

__global__ __launch_bounds__(64) void global_avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int inner) {
    const float* xp = x + (size_t)blockIdx.x * inner;
    float acc = 0.f;
    for (int i = threadIdx.x; i < inner; i += 64) acc += xp[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (threadIdx.x == 0) y[blockIdx.x] = acc / (float)inner;
}

// aten::upsample_bilinear2d, align_corners = False: src = max((dst + 0.5) * in / out - 0.5, 0); y = mul * v + add
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int in_h, int in_w, int out_h, int out_w,
                                                              float mul, float add) {
    const size_t total = (size_t)planes * out_h * out_w;
    const float sy = (float)in_h / (float)out_h, sx = (float)in_w / (float)out_w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % out_w), oy = (int)((i / out_w) % out_h);
        const size_t p = i / ((size_t)out_w * out_h);
        const float fy = fmaxf(((float)oy + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)ox + 0.5f) * sx - 0.5f, 0.f);
        const int y0 = min((int)fy, in_h - 1), x0 = min((int)fx, in_w - 1);
        const int y1 = min(y0 + 1, in_h - 1), x1 = min(x0 + 1, in_w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* xp = x + p * in_h * in_w;
        const float top = xp[(size_t)y0 * in_w + x0] * (1.f - lx) + xp[(size_t)y0 * in_w + x1] * lx;
        const float bot = xp[(size_t)y1 * in_w + x0] * (1.f - lx) + xp[(size_t)y1 * in_w + x1] * lx;
        y[i] = fmaf(mul, top * (1.f - ly) + bot * ly, add);
    }
}

}  // namespace

extern "C" int gc_conv2d_bn_relu_f32(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                     int batch, int in_ch, int out_ch, int in_h, int in_w, int kh, int kw, int stride, int pad_y, int pad_x,
                                     int relu, int out_channels, int chan_off, gc_stream_t stream) {
    if (!x || !w || !y) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_bn_relu_f32: null pointer");
    if (batch < 0 || in_ch <= 0 || out_ch <= 0 || in_h <= 0 || in_w <= 0 || kh <= 0 || kw <= 0 || kh > 7 || kw > 7 || (stride != 1 && stride != 2) ||
        pad_y < 0 || pad_x < 0 || chan_off < 0 || chan_off + out_ch > out_channels)
        return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_bn_relu_f32: bad geometry (taps up to 7 x 7, stride 1 or 2, channel window inside the output)");
    const int out_h = (in_h + 2 * pad_y - kh) / stride + 1, out_w = (in_w + 2 * pad_x - kw) / stride + 1;
    if (out_h <= 0 || out_w <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_bn_relu_f32: empty output");
    if (batch == 0) return GC_OK;
    if (batch > 65535) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_bn_relu_f32: batch > 65535");
    DirectArgs a{x, w, scale, shift, y, batch, in_ch, out_ch, in_h, in_w, out_h, out_w, kh, kw, stride, pad_y, pad_x, relu, out_channels, chan_off,
                 gc::ceil_div(out_w, TILE)};
    const int ph = (TILE - 1) * stride + kh, pw = (TILE - 1) * stride + kw;
    const size_t smem = (size_t)(KC * ph * pw + KC * kh * kw * OCT) * sizeof(float);
    if (smem > 64 * 1024) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_bn_relu_f32: %d x %d taps need %zu bytes of LDS (64 KiB available to a dynamic allocation)", kh, kw, smem);
    dim3 grid((unsigned)(a.tiles_x * gc::ceil_div(out_h, TILE)), (unsigned)gc::ceil_div(out_ch, OCT), (unsigned)batch);
    hipLaunchKernelGGL(conv_direct_kernel, grid, dim3(256), smem, (hipStream_t)stream, a);
    return gc::check_launch("gc_conv2d_bn_relu_f32");
}

extern "C" int gc_pool2d_f32(const float* x, float* y, int batch, int channels, int in_h, int in_w, int k, int stride, int pad, int mode,
                             int out_channels, int chan_off, gc_stream_t stream) {
    if (!x || !y) return gc::fail(GC_ERR_BAD_ARG, "gc_pool2d_f32: null pointer");
    if (batch < 0 || channels <= 0 || in_h <= 0 || in_w <= 0 || k <= 0 || stride <= 0 || pad < 0 || pad >= k || (mode != 0 && mode != 1) ||
        chan_off < 0 || chan_off + channels > out_channels)
        return gc::fail(GC_ERR_BAD_ARG, "gc_pool2d_f32: bad geometry");
    const int out_h = (in_h + 2 * pad - k) / stride + 1, out_w = (in_w + 2 * pad - k) / stride + 1;
    if (out_h <= 0 || out_w <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_pool2d_f32: empty output");
    if (batch == 0) return GC_OK;
    const size_t total = (size_t)batch * channels * out_h * out_w;
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, (hipStream_t)stream,
                       x, y, batch, channels, in_h, in_w, out_h, out_w, k, stride, pad, mode, out_channels, chan_off);
    return gc::check_launch("gc_pool2d_f32");
}

extern "C" int gc_global_avgpool_f32(const float* x, float* y, int planes, int inner, gc_stream_t stream) {
    if (!x || !y) return gc::fail(GC_ERR_BAD_ARG, "gc_global_avgpool_f32: null pointer");
    if (planes < 0 || inner <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_global_avgpool_f32: bad extents");
    if (planes == 0) return GC_OK;
    hipLaunchKernelGGL(global_avgpool_kernel, dim3((unsigned)planes), dim3(64), 0, (hipStream_t)stream, x, y, inner);
    return gc::check_launch("gc_global_avgpool_f32");
}

extern "C" int gc_resize_bilinear_f32(const float* x, float* y, int planes, int in_h, int in_w, int out_h, int out_w, float mul, float add, gc_stream_t stream) {
    if (!x || !y) return gc::fail(GC_ERR_BAD_ARG, "gc_resize_bilinear_f32: null pointer");
    if (planes < 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_resize_bilinear_f32: bad extents");
    if (planes == 0) return GC_OK;
    const size_t total = (size_t)planes * out_h * out_w;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, (hipStream_t)stream,
                       x, y, planes, in_h, in_w, out_h, out_w, mul, add);
    return gc::check_launch("gc_resize_bilinear_f32");
}
