// K3/K4  generalised 2-D convolution and its weight gradient on the gfx950 matrix cores.
// Contract and reference lines: include/gancontrol_hip.h.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- fp32 operands, fp32 accumulate, bit-identical to an fmaf
// chain, so the fp32 parity mode needs no separate VALU path.  Peak 157.3 TFLOP/s (MI355X guide).
//
// conv_mfma_kernel (forward conv, transposed conv, every input gradient)
//   GEMM view per sample and output phase:   Y^T[oc][px] = sum_{tap,k} W[tap][k][oc] * X[k][px + tap]
//   A operand = weights  (lane l holds W[k0 + l/32][oc0 + l%32]),
//   B operand = the input patch (lane l holds X[k0 + l/32][pixel l%32 shifted by the tap]),
//   so the 32 lanes of an accumulator column are 32 consecutive output pixels of one row and the
//   epilogue stores 128-byte runs.  A workgroup (4 waves) stages, per chunk of KC = 8 input
//   channels, the halo'd input patch ONCE in LDS -- all taps re-read it at shifted addresses, no
//   im2col duplication -- plus the [taps][KC][OCT] weight slab.  Per-sample modulation is applied
//   while staging (x * in_scale[b,k]) and in the epilogue (* out_scale[b,oc]); the per-sample
//   [B*OC, IC, k, k] weight tensor of the reference is never formed.
//   up = 2 (transposed conv) is decomposed into its up*up output phases: each workgroup handles
//   one phase, whose taps are the subset {ty : (phase + ty - pad) % up == 0} -- no multiplies by
//   stuffed zeros.
//
// wgrad_mfma_kernel (weight gradient)
//   dW[tap][k][n] = sum_px X[k][px + tap] * dY[n][px]: A = input patch (lane -> channel k),
//   B = dY tile (lane -> channel n), reduction over pixels.  Each wave owns a 32x32 (k, n) block
//   for ALL taps (<= 9 accumulators = 144 registers), so dY is read from LDS once per pixel pair.
//   The pixel space is split across workgroups; partial sums are written to a workspace and
//   reduced in fixed order by wgrad_reduce_kernel (deterministic, no atomics).
#include "common.h"

#include <mutex>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 8;  // input channels staged per LDS chunk (forward kernel)

struct ConvArgs {
    const float* x; const float* w; const float* si; const float* so; float* y;
    int B, K, N, in_h, in_w, out_h, out_w, kh, kw, up, down, pad_y, pad_x;
    int tiles_x, tiles_y;     // pixel tiles per phase sub-grid (sized for phase 0, the largest)
    int pp;                   // patch row pitch (floats)
    int ph_max;               // patch rows allocated
};

// Taps of one output phase along one axis: tap index t0 + j*up, source offset d0 + j, j < n.
struct AxisTaps { int t0, n, d0; };
__device__ __forceinline__ AxisTaps axis_taps(int phase, int k, int up, int pad) {
    AxisTaps a;
    a.t0 = gc::pos_mod(pad - phase, up);
    a.n = a.t0 < k ? (k - a.t0 + up - 1) / up : 0;
    a.d0 = gc::floor_div(phase + a.t0 - pad, up);
    return a;
}

template <int WG_OC, int WG_PX, int WOC, int WPX, int TPW>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs p) {
    constexpr int OCT = WG_OC * WOC * 32;      // output channels per workgroup
    constexpr int RPB = 32 / TPW;              // tile rows covered by one 32-pixel MFMA column block
    constexpr int TPH = WG_PX * WPX * RPB;     // tile rows per workgroup
    static_assert(WG_OC * WG_PX == 4, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wave_oc = wave / WG_PX, wave_px = wave % WG_PX;

    // ---- decode the workgroup: (tile_x, tile_y, phase, sample) x n-tile ----
    int bid = blockIdx.x;
    const int tile_x = bid % p.tiles_x; bid /= p.tiles_x;
    const int tile_y = bid % p.tiles_y; bid /= p.tiles_y;
    const int nph = p.up * p.up;
    const int phase = bid % nph;
    const int b = bid / nph;
    const int phy = phase / p.up, phx = phase % p.up;
    const int n0 = blockIdx.y * OCT;
    const int qh = (p.out_h - phy + p.up - 1) / p.up, qw = (p.out_w - phx + p.up - 1) / p.up;
    const int qy0 = tile_y * TPH, qx0 = tile_x * TPW;
    if (qy0 >= qh || qx0 >= qw) return;

    const AxisTaps ay = axis_taps(phy, p.kh, p.up, p.pad_y), ax = axis_taps(phx, p.kw, p.up, p.pad_x);
    const int ntaps = ay.n * ax.n;
    const int PH = (TPH - 1) * p.down + (ay.n > 0 ? ay.n : 1);
    const int PWd = (TPW - 1) * p.down + (ax.n > 0 ? ax.n : 1);
    const int PP = p.pp;
    const int iy0 = qy0 * p.down + ay.d0, ix0 = qx0 * p.down + ax.d0;

    float* wl = smem;                                   // [ntaps][KC][OCT]
    float* patch = smem + p.kh * p.kw * KC * OCT;       // [KC][PH][PP]
    const int plane = p.ph_max * PP;

    f32x16 acc[WOC][WPX];
#pragma unroll
    for (int i = 0; i < WOC; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-lane patch offsets of this wave's pixel blocks (tap (0,0), channel 0)
    int boff[WPX];
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        const int row = (wave_px * WPX + j) * RPB + l31 / TPW, col = l31 % TPW;
        boff[j] = hi * plane + row * p.down * PP + col * p.down;
    }
    const int aoff = hi * OCT + wave_oc * WOC * 32 + l31;

    const float* xb = p.x + (size_t)b * p.K * p.in_h * p.in_w;
    const float* sib = p.si ? p.si + (size_t)b * p.K : nullptr;
    const bool wvec = (p.N % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.w) & 15) == 0);

    for (int k0 = 0; k0 < p.K && ntaps > 0; k0 += KC) {
        // ---- stage weights: wl[t][kk][nn] = w[ty][tx][k0+kk][n0+nn] ----
        if (wvec) {
            const int total4 = ntaps * KC * (OCT / 4);
            for (int idx = tid; idx < total4; idx += 256) {
                const int nn4 = idx % (OCT / 4);
                const int rest = idx / (OCT / 4);
                const int kk = rest % KC, t = rest / KC;
                const int ty = ay.t0 + (t / ax.n) * p.up, tx = ax.t0 + (t % ax.n) * p.up;
                const int k = k0 + kk, n = n0 + nn4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K && n < p.N) v = *reinterpret_cast<const float4*>(p.w + ((size_t)(ty * p.kw + tx) * p.K + k) * p.N + n);
                *reinterpret_cast<float4*>(wl + (t * KC + kk) * OCT + nn4 * 4) = v;
            }
        } else {
            const int total = ntaps * KC * OCT;
            for (int idx = tid; idx < total; idx += 256) {
                const int nn = idx % OCT;
                const int rest = idx / OCT;
                const int kk = rest % KC, t = rest / KC;
                const int ty = ay.t0 + (t / ax.n) * p.up, tx = ax.t0 + (t % ax.n) * p.up;
                const int k = k0 + kk, n = n0 + nn;
                float v = 0.f;
                if (k < p.K && n < p.N) v = p.w[((size_t)(ty * p.kw + tx) * p.K + k) * p.N + n];
                wl[idx] = v;
            }
        }
        // ---- stage the input patch (zero outside the image), modulated by in_scale ----
        for (int rowid = wave; rowid < KC * PH; rowid += 4) {
            const int kk = rowid / PH, r = rowid - kk * PH;
            const int k = k0 + kk, iy = iy0 + r;
            const bool rowok = k < p.K && iy >= 0 && iy < p.in_h;
            const float sc = (rowok && sib) ? sib[k] : 1.f;
            const float* src = xb + ((size_t)k * p.in_h + iy) * p.in_w;
            float* dst = patch + kk * plane + r * PP;
            for (int c = lane; c < PWd; c += 64) {
                const int ix = ix0 + c;
                float v = 0.f;
                if (rowok && ix >= 0 && ix < p.in_w) v = src[ix] * sc;
                dst[c] = v;
            }
        }
        __syncthreads();
        // ---- MFMA over taps x channel pairs ----
        for (int jy = 0; jy < ay.n; ++jy) {
            for (int jx = 0; jx < ax.n; ++jx) {
                const float* wt = wl + (jy * ax.n + jx) * KC * OCT + aoff;
                const float* pt = patch + jy * PP + jx;
#pragma unroll
                for (int kp = 0; kp < KC / 2; ++kp) {
                    float a[WOC], bv[WPX];
#pragma unroll
                    for (int i = 0; i < WOC; ++i) a[i] = wt[kp * 2 * OCT + i * 32];
#pragma unroll
                    for (int j = 0; j < WPX; ++j) bv[j] = pt[kp * 2 * plane + boff[j]];
#pragma unroll
                    for (int i = 0; i < WOC; ++i)
#pragma unroll
                        for (int j = 0; j < WPX; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: demodulate and store; lanes 0..31 of a register are consecutive pixels ----
    const float* sob = p.so ? p.so + (size_t)b * p.N : nullptr;
    float* yb = p.y + (size_t)b * p.N * p.out_h * p.out_w;
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        const int qy = qy0 + (wave_px * WPX + j) * RPB + l31 / TPW, qx = qx0 + l31 % TPW;
        if (qy >= qh || qx >= qw) continue;
        const int oy = qy * p.up + phy, ox = qx * p.up + phx;
#pragma unroll
        for (int i = 0; i < WOC; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oc = n0 + (wave_oc * WOC + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (oc < p.N) {
                    float v = acc[i][j][r];
                    if (sob) v *= sob[oc];
                    yb[((size_t)oc * p.out_h + oy) * p.out_w + ox] = v;
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x; const float* dy; const float* si; const float* so; float* ws;
    int B, K, N, in_h, in_w, out_h, out_w, kh, kw, down, pad_y, pad_x;
    int tiles_x, tiles_y;     // pixel tiles per sample
    int splits;               // workgroups along the pixel axis
    int tiles_per_split;
    int pp, csx, csy;         // patch row pitch, channel strides of the two LDS tiles (odd: conflict-free)
};

template <int WK, int WN, int WP, int TR, int NT>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradArgs p) {
    constexpr int KT = WK * 32, NTL = WN * 32;   // channel tiles of the workgroup
    constexpr int TPW = 32;                      // pixel tile: TR rows x 32 columns
    static_assert(WK * WN * WP == 4, "4 waves per workgroup");
    static_assert(TR % WP == 0, "rows split evenly over the pixel waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                        // [KT][PH][PP]  (channel stride csx)
    float* ds = smem + KT * p.csx;           // [NTL][TR*32]  (channel stride csy)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wp = wave % WP, wn = (wave / WP) % WN, wk = wave / (WP * WN);
    const int k0 = blockIdx.x * KT, n0 = blockIdx.y * NTL, split = blockIdx.z;
    const int PH = (TR - 1) * p.down + p.kh, PWd = (TPW - 1) * p.down + p.kw;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int tiles_per_sample = p.tiles_x * p.tiles_y;
    const int total_tiles = tiles_per_sample * p.B;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(total_tiles, t_begin + p.tiles_per_split);

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int b = tile / tiles_per_sample;
        const int rem = tile - b * tiles_per_sample;
        const int oy0 = (rem / p.tiles_x) * TR, ox0 = (rem % p.tiles_x) * TPW;
        const int iy0 = oy0 * p.down - p.pad_y, ix0 = ox0 * p.down - p.pad_x;
        // ---- stage x patch: xs[kk][r][c] = x[b, k0+kk, iy0+r, ix0+c] * si[b,k] ----
        for (int rowid = wave; rowid < KT * PH; rowid += 4) {
            const int kk = rowid / PH, r = rowid - kk * PH;
            const int k = k0 + kk, iy = iy0 + r;
            const bool rowok = k < p.K && iy >= 0 && iy < p.in_h;
            const float sc = (rowok && p.si) ? p.si[(size_t)b * p.K + k] : 1.f;
            const float* src = p.x + (((size_t)b * p.K + k) * p.in_h + iy) * p.in_w;
            float* dst = xs + kk * p.csx + r * p.pp;
            for (int c = lane; c < PWd; c += 64) {
                const int ix = ix0 + c;
                float v = 0.f;
                if (rowok && ix >= 0 && ix < p.in_w) v = src[ix] * sc;
                dst[c] = v;
            }
        }
        // ---- stage dy tile: ds[nn][r*32 + c] = dy[b, n0+nn, oy0+r, ox0+c] * so[b,n] ----
        for (int rowid = wave; rowid < NTL * TR; rowid += 4) {
            const int nn = rowid / TR, r = rowid - nn * TR;
            const int n = n0 + nn, oy = oy0 + r;
            const bool rowok = n < p.N && oy < p.out_h;
            const float sc = (rowok && p.so) ? p.so[(size_t)b * p.N + n] : 1.f;
            const float* src = p.dy + (((size_t)b * p.N + n) * p.out_h + oy) * p.out_w;
            float* dst = ds + nn * p.csy + r * TPW;
            if (lane < TPW) {
                const int ox = ox0 + lane;
                dst[lane] = (rowok && ox < p.out_w) ? src[ox] * sc : 0.f;
            }
        }
        __syncthreads();
        // ---- MFMA: reduction over the pixels of this tile; this wave takes rows r = wp, wp+WP, ... ----
        const float* xa = xs + (wk * 32 + l31) * p.csx;
        const float* db = ds + (wn * 32 + l31) * p.csy;
        for (int r = wp; r < TR; r += WP) {
#pragma unroll 4
            for (int c = 0; c < TPW; c += 2) {
                const int col = c + hi;
                const float bv = db[r * TPW + col];
                const float* xr = xa + r * p.down * p.pp + col * p.down;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int ty = t / (NT == 9 ? 3 : 1), tx = t % (NT == 9 ? 3 : 1);
                    const float av = xr[ty * p.pp + tx];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- partial result: ws[split*WP + wp][tap][k][n]; lanes 0..31 = consecutive n ----
    float* out = p.ws + (size_t)(split * WP + wp) * p.kh * p.kw * p.K * p.N;
    const int n = n0 + wn * 32 + l31;
    if (n < p.N) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (k < p.K) out[((size_t)t * p.K + k) * p.N + n] = acc[t][r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, size_t count, int parts) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        float acc = 0.f;
        for (int s = 0; s < parts; ++s) acc += ws[(size_t)s * count + i];
        dw[i] = acc;
    }
}

// --------------------------------------------------------------------------------------------
int validate(const gc_conv_desc* d, const char* who, bool wgrad) {
    if (!d) return gc::fail(GC_ERR_BAD_ARG, "%s: null descriptor", who);
    if (d->batch < 0 || d->in_ch <= 0 || d->out_ch <= 0 || d->in_h <= 0 || d->in_w <= 0 || d->out_h <= 0 || d->out_w <= 0)
        return gc::fail(GC_ERR_BAD_ARG, "%s: non-positive extent", who);
    if (d->kh != d->kw || (d->kh != 1 && d->kh != 3)) return gc::fail(GC_ERR_UNSUPPORTED, "%s: taps %dx%d (1x1 and 3x3 only)", who, d->kh, d->kw);
    const bool ok = (d->up == 1 && (d->down == 1 || d->down == 2)) || (d->up == 2 && d->down == 1);
    if (!ok) return gc::fail(GC_ERR_UNSUPPORTED, "%s: up=%d down=%d", who, d->up, d->down);
    if (wgrad && d->up != 1) return gc::fail(GC_ERR_UNSUPPORTED, "%s: up must be 1 (swap the operands for a transposed conv)", who);
    return GC_OK;
}

int patch_pitch(int width, int tpw) {
    // distinct LDS banks for the (32/tpw) rows one MFMA column block touches: pitch == tpw (mod 32)
    if (tpw == 32) return width | 1;
    int pp = width;
    while (pp % 32 != tpw) ++pp;
    return pp;
}

template <typename K>
int ensure_lds(K kernel, size_t bytes, const char* who) {
    if (bytes > 160 * 1024) return gc::fail(GC_ERR_UNSUPPORTED, "%s: needs %zu bytes of LDS", who, bytes);
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return gc::fail(GC_ERR_HIP, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
    }
    return GC_OK;
}

template <int WG_OC, int WG_PX, int WOC, int WPX, int TPW>
int launch_conv(ConvArgs a, hipStream_t s) {
    constexpr int OCT = WG_OC * WOC * 32, RPB = 32 / TPW, TPH = WG_PX * WPX * RPB;
    const int qh = gc::ceil_div(a.out_h, a.up), qw = gc::ceil_div(a.out_w, a.up);
    a.tiles_y = gc::ceil_div(qh, TPH);
    a.tiles_x = gc::ceil_div(qw, TPW);
    const int ntx_max = a.up == 1 ? a.kw : gc::ceil_div(a.kw, a.up), nty_max = a.up == 1 ? a.kh : gc::ceil_div(a.kh, a.up);
    a.ph_max = (TPH - 1) * a.down + nty_max;
    a.pp = patch_pitch((TPW - 1) * a.down + ntx_max, TPW);
    const size_t lds = ((size_t)a.kh * a.kw * KC * OCT + (size_t)KC * a.ph_max * a.pp) * sizeof(float);
    auto kern = conv_mfma_kernel<WG_OC, WG_PX, WOC, WPX, TPW>;
    int rc = ensure_lds(kern, lds, "gc_conv2d_f32");
    if (rc) return rc;
    const long long gx = (long long)a.tiles_x * a.tiles_y * a.up * a.up * a.B;
    if (gx > 2147483647LL) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_f32: grid too large");
    dim3 grid((unsigned)gx, gc::ceil_div(a.N, OCT));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return gc::check_launch("gc_conv2d_f32");
}

struct WgradPlan { int wk, wn, wp, tr, splits, tiles_per_split, tiles_x, tiles_y, parts; };

WgradPlan plan_wgrad(const gc_conv_desc* d) {
    WgradPlan pl;
    const bool small = d->in_ch <= 32 && d->out_ch <= 32;
    pl.wk = small ? 1 : 2; pl.wn = small ? 1 : 2; pl.wp = small ? 4 : 1;
    pl.tr = small ? 4 : 2;
    pl.tiles_x = gc::ceil_div(d->out_w, 32);
    pl.tiles_y = gc::ceil_div(d->out_h, pl.tr);
    const int total = pl.tiles_x * pl.tiles_y * d->batch;
    const int ctiles = gc::ceil_div(d->in_ch, pl.wk * 32) * gc::ceil_div(d->out_ch, pl.wn * 32);
    int want = gc::ceil_div(1024, ctiles);          // ~4 workgroups per CU over the whole grid
    if (want > total) want = total;
    if (want < 1) want = 1;
    pl.tiles_per_split = gc::ceil_div(total, want);
    pl.splits = gc::ceil_div(total, pl.tiles_per_split);
    pl.parts = pl.splits * pl.wp;
    return pl;
}

}  // namespace

extern "C" int gc_conv2d_f32(const gc_conv_desc* d, const float* x, const float* w,
                             const float* in_scale, const float* out_scale, float* y, gc_stream_t stream) {
    int rc = validate(d, "gc_conv2d_f32", false);
    if (rc) return rc;
    if (!x || !w || !y) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_f32: null pointer");
    if (d->batch == 0) return GC_OK;
    ConvArgs a{x, w, in_scale, out_scale, y, d->batch, d->in_ch, d->out_ch, d->in_h, d->in_w, d->out_h, d->out_w,
               d->kh, d->kw, d->up, d->down, d->pad_y, d->pad_x, 0, 0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    const int qw = gc::ceil_div(d->out_w, d->up);
    if (qw <= 4) return launch_conv<4, 1, 1, 1, 4>(a, s);
    if (qw <= 8) return launch_conv<4, 1, 1, 1, 8>(a, s);
    if (qw <= 16) return launch_conv<2, 2, 2, 2, 16>(a, s);
    if (d->out_ch <= 32) return launch_conv<1, 4, 1, 4, 32>(a, s);
    if (d->out_ch <= 64) return launch_conv<1, 4, 2, 2, 32>(a, s);
    return launch_conv<2, 2, 2, 2, 32>(a, s);
}

extern "C" size_t gc_conv2d_wgrad_workspace(const gc_conv_desc* d) {
    if (!d || d->batch <= 0 || d->in_ch <= 0 || d->out_ch <= 0 || d->out_h <= 0 || d->out_w <= 0) return 0;
    const WgradPlan pl = plan_wgrad(d);
    return (size_t)pl.parts * d->kh * d->kw * d->in_ch * d->out_ch * sizeof(float);
}

extern "C" int gc_conv2d_wgrad_f32(const gc_conv_desc* d, const float* x, const float* dy,
                                   const float* in_scale, const float* out_scale, float* dw,
                                   void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    int rc = validate(d, "gc_conv2d_wgrad_f32", true);
    if (rc) return rc;
    if (!x || !dy || !dw) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_wgrad_f32: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t count = (size_t)d->kh * d->kw * d->in_ch * d->out_ch;
    if (d->batch == 0) {
        hipError_t e = hipMemsetAsync(dw, 0, count * sizeof(float), s);
        return e == hipSuccess ? GC_OK : gc::fail(GC_ERR_HIP, "gc_conv2d_wgrad_f32: memset: %s", hipGetErrorString(e));
    }
    const WgradPlan pl = plan_wgrad(d);
    const size_t need = gc_conv2d_wgrad_workspace(d);
    if (!workspace || workspace_bytes < need) return gc::fail(GC_ERR_WORKSPACE, "gc_conv2d_wgrad_f32: workspace %zu < %zu bytes", workspace_bytes, need);
    WgradArgs a{x, dy, in_scale, out_scale, static_cast<float*>(workspace), d->batch, d->in_ch, d->out_ch, d->in_h, d->in_w,
                d->out_h, d->out_w, d->kh, d->kw, d->down, d->pad_y, d->pad_x, pl.tiles_x, pl.tiles_y, pl.splits, pl.tiles_per_split, 0, 0, 0};
    const int PH = (pl.tr - 1) * d->down + d->kh, PWd = 31 * d->down + d->kw;
    a.pp = PWd;
    a.csx = (PH * a.pp) | 1;
    a.csy = (pl.tr * 32) | 1;
    const int KT = pl.wk * 32, NTL = pl.wn * 32;
    const size_t lds = ((size_t)KT * a.csx + (size_t)NTL * a.csy) * sizeof(float);
    dim3 grid(gc::ceil_div(d->in_ch, KT), gc::ceil_div(d->out_ch, NTL), pl.splits);
#define GC_WGRAD(WK, WN, WP, TR, NT)                                                          \
    do {                                                                                      \
        auto kern = wgrad_mfma_kernel<WK, WN, WP, TR, NT>;                                    \
        rc = ensure_lds(kern, lds, "gc_conv2d_wgrad_f32");                                    \
        if (rc) return rc;                                                                    \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);                                 \
    } while (0)
    const bool small = pl.wp == 4;
    if (d->kh == 3) { if (small) GC_WGRAD(1, 1, 4, 4, 9); else GC_WGRAD(2, 2, 1, 2, 9); }
    else            { if (small) GC_WGRAD(1, 1, 4, 4, 1); else GC_WGRAD(2, 2, 1, 2, 1); }
#undef GC_WGRAD
    rc = gc::check_launch("gc_conv2d_wgrad_f32(mfma)");
    if (rc) return rc;
    const int blocks = (int)std::min<size_t>((count + 255) / 256, 2048);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const float*>(workspace), dw, count, pl.parts);
    return gc::check_launch("gc_conv2d_wgrad_f32(reduce)");
}
