// Image-space pieces of the ADA augmentation (non_leaking.py:316-371) that are not FIR passes:
//   * affine_warp_kernel      bilinear resampling under a per-sample affine map (the reference's make_grid -> affine_grid ->
//                             F.grid_sample(bilinear, zeros, align_corners=False) chain is an affine function of the output pixel
//                             index; the host folds it into one 2 x 3 matrix in input-pixel units) and its adjoint (scatter-add)
//   * reflect_pad_kernel      F.pad(mode='reflect') and its adjoint (fold-back)
// Both are linear maps of the image, so forward and adjoint are each other's derivatives (closure for higher orders).
// HBM-bound, [B, 3, H, W] tensors: one lane per output (forward) / input (adjoint) element, coalesced along x.
#include <algorithm>

#include "common.h"

namespace {

struct WarpArgs {
    const float* x; float* y; const float* mat;      // mat [B, 6]: sx = m0*ox + m1*oy + m2, sy = m3*ox + m4*oy + m5 (input pixel units)
    int batch, channels, in_h, in_w, out_h, out_w;
};

// forward: y[b,c,oy,ox] = sum of the 4 neighbours of (sx, sy) weighted bilinearly, zeros outside
// adjoint: x'[b,c,iy,ix] += w * g[b,c,oy,ox]   (x = g here: [B,C,out_h,out_w] -> y = [B,C,in_h,in_w], pre-zeroed by the caller)
template <bool ADJOINT>
__global__ __launch_bounds__(256) void affine_warp_kernel(WarpArgs a) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (ox >= a.out_w || oy >= a.out_h) return;
    const float* m = a.mat + b * 6;
    const float sx = fmaf(m[0], (float)ox, fmaf(m[1], (float)oy, m[2]));
    const float sy = fmaf(m[3], (float)ox, fmaf(m[4], (float)oy, m[5]));
    const float fx = floorf(sx), fy = floorf(sy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = sx - fx, ty = sy - fy;
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    const bool vx0 = x0 >= 0 && x0 < a.in_w, vx1 = x0 + 1 >= 0 && x0 + 1 < a.in_w;
    const bool vy0 = y0 >= 0 && y0 < a.in_h, vy1 = y0 + 1 >= 0 && y0 + 1 < a.in_h;
    const size_t in_plane = (size_t)a.in_h * a.in_w, out_plane = (size_t)a.out_h * a.out_w;
    for (int c = 0; c < a.channels; ++c) {
        const size_t pc = (size_t)b * a.channels + c;
        if (!ADJOINT) {
            const float* xp = a.x + pc * in_plane;
            float v = 0.f;
            if (vy0 && vx0) v = fmaf(w00, xp[(size_t)y0 * a.in_w + x0], v);
            if (vy0 && vx1) v = fmaf(w01, xp[(size_t)y0 * a.in_w + x0 + 1], v);
            if (vy1 && vx0) v = fmaf(w10, xp[(size_t)(y0 + 1) * a.in_w + x0], v);
            if (vy1 && vx1) v = fmaf(w11, xp[(size_t)(y0 + 1) * a.in_w + x0 + 1], v);
            a.y[pc * out_plane + (size_t)oy * a.out_w + ox] = v;
        } else {
            const float g = a.x[pc * out_plane + (size_t)oy * a.out_w + ox];
            float* yp = a.y + pc * in_plane;
            if (vy0 && vx0) atomicAdd(yp + (size_t)y0 * a.in_w + x0, w00 * g);
            if (vy0 && vx1) atomicAdd(yp + (size_t)y0 * a.in_w + x0 + 1, w01 * g);
            if (vy1 && vx0) atomicAdd(yp + (size_t)(y0 + 1) * a.in_w + x0, w10 * g);
            if (vy1 && vx1) atomicAdd(yp + (size_t)(y0 + 1) * a.in_w + x0 + 1, w11 * g);
        }
    }
}

__device__ __forceinline__ int reflect_index(int i, int n) {      // torch 'reflect': no edge repeat, pads < n
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// forward: y[p, oy, ox] = x[p, reflect(oy - top), reflect(ox - left)]
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int in_h, int in_w,
                                                          int out_h, int out_w, int left, int top) {
    const size_t total = (size_t)planes * out_h * out_w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % out_w);
        const size_t r = i / out_w;
        const int oy = (int)(r % out_h);
        const size_t p = r / out_h;
        y[i] = x[(p * in_h + reflect_index(oy - top, in_h)) * in_w + reflect_index(ox - left, in_w)];
    }
}

// adjoint, gather form (deterministic): gx[p, iy, ix] = sum of g over the <= 3 x 3 padded positions that read (iy, ix)
__global__ __launch_bounds__(256) void reflect_pad_adjoint_kernel(const float* __restrict__ g, float* __restrict__ gx, int planes, int in_h, int in_w,
                                                                  int out_h, int out_w, int left, int top) {
    const size_t total = (size_t)planes * in_h * in_w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ix = (int)(i % in_w);
        const size_t r = i / in_w;
        const int iy = (int)(r % in_h);
        const size_t p = r / in_h;
        // padded coordinates that map to ix: ix + left (interior), left - ix (left mirror, ix >= 1), left + 2 (in_w - 1) - ix (right mirror, ix <= in_w - 2)
        int xs[3], ys[3], nx = 0, ny = 0;
        xs[nx++] = ix + left;
        if (ix >= 1 && left - ix >= 0) xs[nx++] = left - ix;
        if (ix <= in_w - 2 && left + 2 * (in_w - 1) - ix < out_w) xs[nx++] = left + 2 * (in_w - 1) - ix;
        ys[ny++] = iy + top;
        if (iy >= 1 && top - iy >= 0) ys[ny++] = top - iy;
        if (iy <= in_h - 2 && top + 2 * (in_h - 1) - iy < out_h) ys[ny++] = top + 2 * (in_h - 1) - iy;
        const float* gp = g + p * (size_t)out_h * out_w;
        float acc = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int c = 0; c < nx; ++c) acc += gp[(size_t)ys[a] * out_w + xs[c]];
        gx[i] = acc;
    }
}

}  // namespace

extern "C" int gc_affine_warp_bilinear_f32(const float* x, const float* mat, float* y, int batch, int channels,
                                           int in_h, int in_w, int out_h, int out_w, int adjoint, gc_stream_t stream) {
    if (!x || !mat || !y) return gc::fail(GC_ERR_BAD_ARG, "gc_affine_warp_bilinear_f32: null pointer");
    if (batch < 0 || channels <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_affine_warp_bilinear_f32: bad extents");
    if (batch == 0) return GC_OK;
    if (batch > 65535) return gc::fail(GC_ERR_UNSUPPORTED, "gc_affine_warp_bilinear_f32: batch > 65535");
    hipStream_t s = (hipStream_t)stream;
    WarpArgs a{x, y, mat, batch, channels, in_h, in_w, out_h, out_w};
    dim3 grid(gc::ceil_div(out_w, 64), gc::ceil_div(out_h, 4), batch);
    if (adjoint) {
        // x is the gradient of the [B, C, out_h, out_w] output; y receives the gradient of the [B, C, in_h, in_w] input
        hipError_t e = hipMemsetAsync(y, 0, (size_t)batch * channels * in_h * in_w * sizeof(float), s);
        if (e != hipSuccess) return gc::fail(GC_ERR_HIP, "gc_affine_warp_bilinear_f32: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(affine_warp_kernel<true>, grid, dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(affine_warp_kernel<false>, grid, dim3(256), 0, s, a);
    }
    return gc::check_launch("gc_affine_warp_bilinear_f32");
}

extern "C" int gc_reflect_pad_f32(const float* x, float* y, int planes, int in_h, int in_w, int left, int right, int top, int bottom,
                                  int adjoint, gc_stream_t stream) {
    if (!x || !y) return gc::fail(GC_ERR_BAD_ARG, "gc_reflect_pad_f32: null pointer");
    if (planes < 0 || in_h <= 0 || in_w <= 0 || left < 0 || right < 0 || top < 0 || bottom < 0) return gc::fail(GC_ERR_BAD_ARG, "gc_reflect_pad_f32: bad extents");
    if (left >= in_w || right >= in_w || top >= in_h || bottom >= in_h)
        return gc::fail(GC_ERR_BAD_ARG, "gc_reflect_pad_f32: padding (%d, %d, %d, %d) must be smaller than the image (%d x %d)", left, right, top, bottom, in_h, in_w);
    if (planes == 0) return GC_OK;
    const int out_h = in_h + top + bottom, out_w = in_w + left + right;
    hipStream_t s = (hipStream_t)stream;
    if (adjoint) {
        const size_t total = (size_t)planes * in_h * in_w;
        hipLaunchKernelGGL(reflect_pad_adjoint_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65536)), dim3(256), 0, s, x, y, planes, in_h, in_w, out_h, out_w, left, top);
    } else {
        const size_t total = (size_t)planes * out_h * out_w;
        hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65536)), dim3(256), 0, s, x, y, planes, in_h, in_w, out_h, out_w, left, top);
    }
    return gc::check_launch("gc_reflect_pad_f32");
}
