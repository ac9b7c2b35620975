// Helpers shared by the fp32 (conv.hip) and split-bf16 (conv_bf16x3.hip) convolution kernels.
#pragma once
#include "common.h"

namespace gcconv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* x; const float* w; const float* si; const float* so; float* y;
    int B, K, N, in_h, in_w, out_h, out_w, pad_y, pad_x;
    int tiles_x, tiles_y;     // pixel tiles per phase sub-grid (sized for phase 0, the largest)
    // fused epilogue (gc_conv_epilogue): y = act(so * acc + noise_w * noise[b, pixel] + bias[n])
    const float* bias; const float* noise; const float* noise_w;
    float slope, gain; int act;
    const float* residual;    // added after the activation (gc_conv_epilogue)
    // grid-level split over the input channels (conv_mfma_kernel, small planes): slice z handles channels
    // [z * k_per_split, min(K, (z + 1) * k_per_split)) and writes its raw partial sums to part + z * B*N*out_h*out_w
    int k_per_split; float* part;
};

inline void set_epilogue(ConvArgs& a, const gc_conv_epilogue* ep) {
    a.bias = ep ? ep->bias : nullptr;
    a.noise = ep ? ep->noise : nullptr;
    a.noise_w = ep ? ep->noise_w : nullptr;
    a.slope = ep ? ep->slope : 1.f;
    a.gain = ep ? ep->gain : 1.f;
    a.act = ep ? ep->activate : 0;
    a.residual = ep ? ep->residual : nullptr;
}

inline int validate_epilogue(const gc_conv_epilogue* ep, const char* who) {
    if (ep && (ep->noise == nullptr) != (ep->noise_w == nullptr))
        return gc::fail(GC_ERR_BAD_ARG, "%s: epilogue noise and noise_w must both be set or both be null", who);
    return GC_OK;
}

// Same arithmetic, in the same order, as bias_act_plane_kernel (bias_act.hip): the fused and the two-pass results are
// bit-identical.  Everything is a VALUE fetched before the store loop and the function is branch-free:
//  * so / bias: 1 / 0 when absent; nw = nz = 0 without noise; slope = gain = 1 without activation -- all exact no-ops;
//  * a vector load between two stores would make the compiler wait for vmcnt(0), i.e. for every store issued so far
//    (one memory round trip per output row), and a uniform branch per element keeps it from batching the stores.
struct EpilogueConsts { float nw, slope, gain; };
__device__ __forceinline__ EpilogueConsts epilogue_consts(const ConvArgs& p) {
    EpilogueConsts e;
    e.nw = p.noise ? p.noise_w[0] : 0.f;
    e.slope = p.act ? p.slope : 1.f;
    e.gain = p.act ? p.gain : 1.f;
    return e;
}
__device__ __forceinline__ float conv_epilogue(const EpilogueConsts& e, float acc, float so, float bias, float nz) {
#pragma clang fp contract(off)      // the products below are rounded on their own (as in bias_act.hip); a residual the caller adds must not fuse into them
    float v = acc * so;
    v = fmaf(e.nw, nz, v);
    v += bias;
    return (v > 0.f ? v : v * e.slope) * e.gain;
}

// Taps of one output phase along one axis: tap index t0 + j*up, source offset d0 + j, j < n.
struct AxisTaps { int t0, n, d0; };
template <int UP, int KS>
__device__ __forceinline__ AxisTaps axis_taps(int phase, int pad) {
    AxisTaps a;
    if (UP == 1) { a.t0 = 0; a.n = KS; a.d0 = -pad; return a; }
    a.t0 = gc::pos_mod(pad - phase, UP);
    a.n = a.t0 < KS ? (KS - a.t0 + UP - 1) / UP : 0;
    a.d0 = gc::floor_div(phase + a.t0 - pad, UP);
    return a;
}

// The scalar twin: keeps loop-invariant address arithmetic of a per-tile epilogue from being hoisted out of the tile loop
// (where it would sit in dozens of registers across the MFMA phases and spill).
__device__ __forceinline__ int opaque_s(int v) {
    asm volatile("" : "+s"(v));
    return v;
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int patch_pitch(int width, int tpw) {
    // distinct LDS banks for the (32/tpw) rows one MFMA column block touches: pitch == tpw (mod 32)
    if (tpw == 32) return width | 1;
    int pp = width;
    while (pp % 32 != tpw) ++pp;
    return pp;
}

// Fresh, optimiser-opaque copy of a lane value: staging index arithmetic written in terms of it is
// recomputed where it is used (a few dozen VALU ops per chunk) instead of being hoisted out of the
// K loop into dozens of live registers, which would cost a wave of occupancy.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// true when the caller asked for dense output rows (what every launch except the fused transposed convolution writes)
inline bool dense_output(const gc_conv_desc* d) { return d->out_pitch == 0 || d->out_pitch == d->out_w; }

inline int validate(const gc_conv_desc* d, const char* who, bool wgrad) {
    if (!d) return gc::fail(GC_ERR_BAD_ARG, "%s: null descriptor", who);
    if (d->batch < 0 || d->in_ch <= 0 || d->out_ch <= 0 || d->in_h <= 0 || d->in_w <= 0 || d->out_h <= 0 || d->out_w <= 0)
        return gc::fail(GC_ERR_BAD_ARG, "%s: non-positive extent", who);
    if (d->kh != d->kw || (d->kh != 1 && d->kh != 3)) return gc::fail(GC_ERR_UNSUPPORTED, "%s: taps %dx%d (1x1 and 3x3 only)", who, d->kh, d->kw);
    const bool ok = (d->up == 1 && (d->down == 1 || d->down == 2)) || (d->up == 2 && d->down == 1);
    if (!ok) return gc::fail(GC_ERR_UNSUPPORTED, "%s: up=%d down=%d", who, d->up, d->down);
    if (wgrad && d->up != 1) return gc::fail(GC_ERR_UNSUPPORTED, "%s: up must be 1 (swap the operands for a transposed conv)", who);
    if (d->in_pitch != 0 && d->in_pitch < d->in_w) return gc::fail(GC_ERR_BAD_ARG, "%s: in_pitch %d < in_w %d", who, d->in_pitch, d->in_w);
    if (d->out_pitch != 0 && d->out_pitch < d->out_w) return gc::fail(GC_ERR_BAD_ARG, "%s: out_pitch %d < out_w %d", who, d->out_pitch, d->out_w);
    const long long lim = 2147483647LL;
    if ((long long)d->in_ch * d->in_h * d->in_w > lim || (long long)d->out_ch * d->out_h * d->out_w > lim ||
        (long long)d->kh * d->kw * d->in_ch * d->out_ch > lim)
        return gc::fail(GC_ERR_UNSUPPORTED, "%s: a per-sample plane set exceeds 2^31 elements", who);
    // buffer-descriptor addressing: one sample's planes (and the weight slab) must stay below 2 GiB
    const long long blim = (1LL << 31) - 1;
    if ((long long)d->in_ch * d->in_h * d->in_w * 4 > blim || (long long)d->out_ch * d->out_h * d->out_w * 4 > blim ||
        (long long)d->kh * d->kw * (d->in_ch + 7) / 8 * d->out_ch * 16 > blim)
        return gc::fail(GC_ERR_UNSUPPORTED, "%s: a per-sample tensor exceeds 2 GiB", who);
    return GC_OK;
}


// Buffer-descriptor loads (cdna guide T8): address = descriptor base + scalar byte offset + 32-bit lane byte
// offset -> no 64-bit VALU address arithmetic, and the hardware returns 0 for lane offsets >= the buffer size, so
// out-of-image elements need no exec-mask branch: they are given the offset OOB (beyond any buffer used here).
constexpr unsigned OOB = 0x80000000u;
// The descriptor inputs go through readfirstlane so the compiler can PROVE them wave-uniform; otherwise it wraps
// every buffer instruction in a waterfall loop (cdna guide T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, (int)nb, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)__builtin_amdgcn_readfirstlane(soff), 0));
}
__device__ __forceinline__ uint4 buf_load_u128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)__builtin_amdgcn_readfirstlane(soff), 0));
}

// Retire every staged load on ALL paths before the next prefetch is issued.  If a loaded register is consumed only
// under a branch, the compiler's wait-count model keeps it "pending" across the loop back-edge and plants
// s_waitcnt vmcnt(N) in the middle of the NEXT prefetch; counted in hardware, that waits for the fresh loads and
// serialises the pipeline.  simm16 = vmcnt(0) with expcnt / lgkmcnt left at their maxima.
__device__ __forceinline__ void wait_staged_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }


// defined in pointwise.hip: 1x1 convolutions with <= 4 channels on one side (ToRGB / FromRGB) on the vector ALUs
bool pointwise_thin(const gc_conv_desc* d);
bool pointwise_thin_wgrad(const gc_conv_desc* d);
int pointwise_conv(const gc_conv_desc* d, const float* x, const float* w, const float* in_scale, const float* out_scale,
                   const gc_conv_epilogue* ep, float* y, gc_stream_t stream);
size_t pointwise_wgrad_workspace(const gc_conv_desc* d);
int pointwise_wgrad(const gc_conv_desc* d, const float* x, const float* dy, const float* in_scale, const float* out_scale,
                    float* dw, float* dw_samples, void* workspace, gc_stream_t stream);      // dw_samples: optional [B][K * N] per-sample shares

// defined in conv.hip: the fp32 convolution with an optional workspace (split-K over the input channels on small planes)
size_t conv2d_f32_workspace(const gc_conv_desc* d);
int conv2d_f32_ws(const gc_conv_desc* d, const float* x, const float* w, const float* in_scale, const float* out_scale,
                  const gc_conv_epilogue* ep, float* y, void* workspace, size_t workspace_bytes, gc_stream_t stream);

// defined in conv.hip: 3 x 3 weight gradients onto planes <= 8 x 8 that gc_conv2d_wgrad_f32 computes in one launch in exact fp32 (every arithmetic mode sends them there)
bool wgrad_small_eligible(const gc_conv_desc* d);

// defined in conv.hip: y = epilogue(sum over the K slices fin.part[z * per_slice + i]) in fixed order (split-K finish pass)
int launch_splitk_finish(const ConvArgs& fin, int slices, long long per_slice, hipStream_t s);

// defined in conv.hip: dw[i] = sum_s ws[s][i] in fixed order (deterministic split reduction)
int launch_wgrad_reduce(const float* ws, float* dw, size_t count, int parts, hipStream_t s);
// the same for partial sums grouped by sample, ws[b][j][i] (j < per_sample): samples[b][i] = sum_j, dw[i] = sum_b samples[b][i]
int launch_wgrad_reduce_samples(const float* ws, float* dw, float* samples, size_t count, int batch, int per_sample, hipStream_t s);

}  // namespace gcconv
