// K1  upfirdn2d for gfx950.  See include/gancontrol_hip.h for the contract and the reference lines.
//
// Two kernels:
//  * fir44_tile_kernel   -- the hot case (up = down = 1, 4x4 taps: every Blur in G and D and their
//    adjoints).  A 256-thread workgroup produces a 32x128 output tile of one plane: the (35 x 131)
//    input patch is staged in LDS with coalesced dword loads (input rows are 1025/513/... floats
//    wide, so wider global loads cannot be aligned), then each lane produces a 4x4 micro-tile from
//    a 7x7 register window read with one ds_read_b128 + one ds_read_b96 per row -- 14 LDS
//    instructions and 256 FMAs per 16 outputs -- and stores four float4 rows.  HBM-bound:
//    algorithmic bytes = (in + out) * 4.
//  * fir44_small_kernel  -- 4x4 taps, up = down = 1, planes up to 33 x 33 staged whole in LDS
//  * generic_kernel      -- any up/down/taps (ToRGB skip up-sampling, Downsample, the 12x12 ADA
//    filters, tiny planes): one output per lane, taps cached in LDS.
#include "common.h"

namespace {

constexpr int TH = 32, TW = 128;          // output tile of the fast kernel
constexpr int PH = TH + 3, PW = TW + 3;   // input patch (4x4 taps)
constexpr int PITCH = 132;                // patch row pitch in floats (multiple of 4: 16-B aligned rows)

// Optional epilogue of the up = down = 1 kernel (gc_upfirdn2d_act_f32): the NoiseInjection + FusedLeakyReLU that follow
// the Blur of an up-sampling StyledConv (gan_model.py:402-408), in the arithmetic order of bias_act_plane_kernel.
struct FirEpilogue {
    const float* bias; const float* noise; const float* noise_w;
    float slope, gain; int channels;
    // EPI = 2 (gc_upfirdn2d_mask_f32): y = FIR(x) * (mask_ref > 0 ? mpos : mneg), mask_ref shaped like the output (dense rows) -- the Blur
    // adjoint followed by the activation backward of the layer whose output the Blur read (ResBlock conv1, gan_model.py:893-922)
    const float* mask_ref; float mpos, mneg;
    // PRO (gc_upfirdn2d_actbwd_f32): the INPUT is multiplied by the activation mask of pro_ref while it is staged -- x is the gradient that
    // arrives at a fused Blur + noise + bias + leaky-ReLU (StyledConv's up-sampling branch), the kernel computes the Blur adjoint of the
    // pre-activation gradient and, per (plane, tile), the sums the bias / noise-strength gradients need (each input element is owned by
    // exactly one tile: the first TH x TW elements of its patch)
    const float* pro_ref; const float* pro_noise; float* pro_psum; float* pro_pdot; float ppos, pneg;
};

// GC_FIR_NT vertically consecutive tiles per workgroup in one software pipeline (the 16-byte loads of tile t + 1 issued before the
// arithmetic and the stores of tile t, committed to LDS after them).  Measured with 4 tiles per workgroup (round 3, same-box A/B,
// tools/kbench.py): SLOWER -- [4, 32, 1025^2] 212 -> 252 us (5.07 -> 4.26 TB/s), [4, 64, 513^2] 111 -> 128 us, [4, 128, 257^2] 54 -> 64 us: the
// eight workgroups a CU holds already overlap each other's load / compute / store phases, and a quarter of the workgroups with two more
// barriers per tile only lengthens the tail.  One tile per workgroup stays the default.
#ifndef GC_FIR_NT
#define GC_FIR_NT 1
#endif

#ifndef GC_FIR_OCC
#define GC_FIR_OCC 0          // dev knob: > 0 = waves per SIMD the tile kernel is compiled for (register cap 512 / OCC)
#endif
#ifndef GC_FIR_NT_LOAD
#define GC_FIR_NT_LOAD 0      // 1 = non-temporal loads of the input patch: measured SLOWER (whole step 75.78 -> 75.10 images/s, in-step FIR 4.70 -> 4.35 TB/s):
                              // the halo rows are re-read by the neighbouring tiles out of L2
#endif
#if GC_FIR_OCC > 0
#define GC_FIR_BOUNDS __launch_bounds__(256, GC_FIR_OCC)
#else
#define GC_FIR_BOUNDS __launch_bounds__(256)
#endif
template <bool VEC, int EPI, bool PRO = false>
__global__ GC_FIR_BOUNDS void fir44_tile_kernel(
    const float* __restrict__ x, const float* __restrict__ taps, float* __restrict__ y,
    int in_h, int in_w, int out_h, int out_w, int pad_x0, int pad_y0, int flip, FirEpilogue ep, int in_pitch, int out_pitch) {
    __shared__ __attribute__((aligned(16))) float patch[PH * PITCH];
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * TW;
    const size_t plane = blockIdx.z;
    const float* xp = x + plane * (size_t)in_h * in_pitch;      // in_pitch floats between input rows (in_w when dense)
    float* yp = y + plane * (size_t)out_h * out_pitch;          // out_pitch floats between output rows (out_w when dense)

    // taps -> registers (uniform address: scalar loads); T[a][b] multiplies U[oy + a - pad][ox + b - pad]
    float T[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) T[a][b] = flip ? taps[(3 - a) * 4 + (3 - b)] : taps[a * 4 + b];

    // stage the patch: a lane fetches 4 consecutive floats of one input row with one 16-byte load (rows are only 4-byte
    // aligned for 1025-wide planes: fine for gfx9) -- 5 loads per lane instead of 19 dword loads; groups that touch an
    // image border fall back to guarded scalar loads.  All loads are issued before the first LDS write.
    const int ix0 = ox0 - pad_x0;
    constexpr int GPR = PITCH / 4;                           // 16-byte groups per patch row
    constexpr int NLD = (PH * GPR + 255) / 256;
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    float4 stage[NLD];
    // guarded 16-byte load of four consecutive floats of one row (zeros outside the image)
    auto ld4 = [&](const float* rowp, int ix, int width) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* src = rowp + ix;
        if (ix >= 0 && ix + 3 < width) {
#if GC_FIR_NT_LOAD
            const f4u t = __builtin_nontemporal_load(reinterpret_cast<const f4u*>(src));
#else
            const f4u t = *reinterpret_cast<const f4u*>(src);
#endif
            v = make_float4(t.x, t.y, t.z, t.w);
        } else {
            if (ix >= 0 && ix < width) v.x = src[0];
            if (ix + 1 >= 0 && ix + 1 < width) v.y = src[1];
            if (ix + 2 >= 0 && ix + 2 < width) v.z = src[2];
            if (ix + 3 >= 0 && ix + 3 < width) v.w = src[3];
        }
        return v;
    };
    static_assert(!PRO || GC_FIR_NT == 1, "the prologue sums are written once per workgroup");
    float ps = 0.f, pd = 0.f;                                // PRO: this lane's share of the tile's sums
    const float* refp = PRO ? ep.pro_ref + plane * (size_t)in_h * in_w : nullptr;
    const float* nzp = (PRO && ep.pro_noise) ? ep.pro_noise + (plane / ep.channels) * (size_t)in_h * in_w : nullptr;
    auto fetch = [&](int oy0) {
        const int iy0 = oy0 - pad_y0;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx / GPR, c = (idx - r * GPR) * 4;
            const int iy = iy0 + r, ix = ix0 + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < PH * GPR && iy >= 0 && iy < in_h) {
                v = ld4(xp + (size_t)iy * in_pitch, ix, in_w);
                if (PRO) {
                    const float4 m = ld4(refp + (size_t)iy * in_w, ix, in_w);
                    v.x *= m.x > 0.f ? ep.ppos : ep.pneg; v.y *= m.y > 0.f ? ep.ppos : ep.pneg;
                    v.z *= m.z > 0.f ? ep.ppos : ep.pneg; v.w *= m.w > 0.f ? ep.ppos : ep.pneg;
                    // owned by this tile: the first TH x TW elements of the patch, plus what lies beyond them in the last tile row / column
                    // (there is no next tile to own it); elements outside the image are zero
                    if ((r < TH || blockIdx.y == gridDim.y - 1) && (c < TW || blockIdx.x == gridDim.x - 1)) {
                        ps += (v.x + v.y) + (v.z + v.w);
                        if (nzp) {
                            const float4 z = ld4(nzp + (size_t)iy * in_w, ix, in_w);
                            pd += (v.x * z.x + v.y * z.y) + (v.z * z.z + v.w * z.w);
                        }
                    }
                }
            }
            stage[j] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int idx = tid + 256 * j;
            if (idx < PH * GPR) *reinterpret_cast<float4*>(&patch[idx * 4]) = stage[j];
        }
    };

    const int cg = tid & 31, rg = tid >> 5;   // 32 column groups x 8 row groups of 4x4 outputs
    const int ox = ox0 + cg * 4;
    const int ty0 = blockIdx.y * GC_FIR_NT;
    const int tiles_y = (out_h + TH - 1) / TH;
    const int nt = min(GC_FIR_NT, tiles_y - ty0);
    fetch(ty0 * TH);
    __shared__ float red[8];
    if (PRO) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { ps += __shfl_xor(ps, off, 64); pd += __shfl_xor(pd, off, 64); }
        if ((tid & 63) == 0) { red[(tid >> 6) * 2] = ps; red[(tid >> 6) * 2 + 1] = pd; }
    }
    commit();
    __syncthreads();
    if (PRO && tid == 0) {
        const size_t slot = plane * ((size_t)gridDim.x * gridDim.y) + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        ep.pro_psum[slot] = (red[0] + red[2]) + (red[4] + red[6]);
        if (ep.pro_pdot) ep.pro_pdot[slot] = (red[1] + red[3]) + (red[5] + red[7]);
    }
    for (int t = 0; t < nt; ++t) {
        const int oy0 = (ty0 + t) * TH;
        const bool more = t + 1 < nt;
        if (more) fetch(oy0 + TH);            // in flight during the arithmetic and the stores below

        float acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[r][j] = 0.f;

#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const float* row = &patch[(rg * 4 + i) * PITCH + cg * 4];
            const float4 lo = *reinterpret_cast<const float4*>(row);
            const float v4 = row[4], v5 = row[5], v6 = row[6];
            const float v[7] = {lo.x, lo.y, lo.z, lo.w, v4, v5, v6};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = i - r;
                if (a < 0 || a > 3) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[r][j] = fmaf(T[a][b], v[j + b], acc[r][j]);
            }
        }

        if (EPI == 2) {
            // the activation output the mask comes from: all 16 values of the micro-tile are fetched before the first store
            const float* ref = ep.mask_ref + plane * (size_t)out_h * out_w;
            float m[4][4];
            const bool vec_ref = VEC && (out_w & 3) == 0 && ((reinterpret_cast<uintptr_t>(ref) & 15) == 0);      // dense rows of the activation output: 16-byte loads
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oy = oy0 + rg * 4 + r;
                if (vec_ref && oy < out_h && ox + 3 < out_w) {
                    const float4 t = *reinterpret_cast<const float4*>(ref + (size_t)oy * out_w + ox);
                    m[r][0] = t.x; m[r][1] = t.y; m[r][2] = t.z; m[r][3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) m[r][j] = (oy < out_h && ox + j < out_w) ? ref[(size_t)oy * out_w + ox + j] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][j] *= m[r][j] > 0.f ? ep.mpos : ep.mneg;
        }
        if (EPI == 1) {
            // every noise value of the micro-tile is fetched before the first store (a load between stores costs a full
            // memory round trip: vmcnt counts the stores too)
            const int c = (int)(plane % ep.channels);
            const size_t b = plane / ep.channels;
            const float bv = ep.bias ? ep.bias[c] : 0.f, nw = ep.noise ? ep.noise_w[0] : 0.f;
            float nz[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int oy = oy0 + rg * 4 + r;
                    nz[r][j] = (ep.noise && oy < out_h && ox + j < out_w) ? ep.noise[(b * out_h + oy) * out_w + ox + j] : 0.f;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = fmaf(nw, nz[r][j], acc[r][j]) + bv;
                    acc[r][j] = (v > 0.f ? v : v * ep.slope) * ep.gain;
                }
        }
        if (more) {
            __syncthreads();              // every lane has read its window of this tile
            commit();                     // waits for the loads issued above; no store is outstanding yet
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = oy0 + rg * 4 + r;
            if (oy >= out_h) break;
            float* dst = yp + (size_t)oy * out_pitch + ox;
            if (VEC) {
                if (ox + 3 < out_w) gc::stream_store4(dst, acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (ox + j < out_w) dst[j] = acc[r][j];          // the ragged end of a pitched odd-width row
                }
            } else if (ox + 3 < out_w) {
                // odd widths (1025, 513, ...): rows are only 4-byte aligned, which a 16-byte store accepts on gfx9 (as the
                // 16-byte loads of the weight-gradient kernels do): one store instruction instead of four
                gc::stream_store4u(dst, acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (ox + j < out_w) dst[j] = acc[r][j];
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// 4x4 taps, down = 2 (the decimating FIR of the ResBlock skip path, Downsample, and the adjoint of every up = 2 FIR).
// A workgroup produces 16 x 128 outputs from a (34 x 258) input patch; each lane owns a 2 x 4 micro-tile fed from a
// 6 x 10 register window.  Traffic: reads the input once, writes a quarter of it.
constexpr int D_TH = 16, D_TW = 128;
constexpr int D_PH = 2 * D_TH + 2, D_PW = 2 * D_TW + 2, D_PITCH = 260;

template <bool VEC>
__global__ __launch_bounds__(256) void fir44_down2_kernel(
    const float* __restrict__ x, const float* __restrict__ taps, float* __restrict__ y,
    int in_h, int in_w, int out_h, int out_w, int pad_x0, int pad_y0, int flip) {
    __shared__ __attribute__((aligned(16))) float patch[D_PH * D_PITCH];
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * D_TW, oy0 = blockIdx.y * D_TH;
    const size_t plane = blockIdx.z;
    const float* xp = x + plane * (size_t)in_h * in_w;
    float* yp = y + plane * (size_t)out_h * out_w;

    float T[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) T[a][b] = flip ? taps[(3 - a) * 4 + (3 - b)] : taps[a * 4 + b];

    const int iy0 = 2 * oy0 - pad_y0, ix0 = 2 * ox0 - pad_x0;
    constexpr int GPR = D_PITCH / 4;                         // 16-byte groups per patch row (see fir44_tile_kernel)
    constexpr int NLD = (D_PH * GPR + 255) / 256;
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    float4 stage[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int idx = tid + 256 * j;
        const int r = idx / GPR, c = (idx - r * GPR) * 4;
        const int iy = iy0 + r, ix = ix0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < D_PH * GPR && iy >= 0 && iy < in_h) {
            const float* src = xp + (size_t)iy * in_w + ix;
            if (ix >= 0 && ix + 3 < in_w) {
                const f4u t = *reinterpret_cast<const f4u*>(src);
                v = make_float4(t.x, t.y, t.z, t.w);
            } else {
                if (ix >= 0 && ix < in_w) v.x = src[0];
                if (ix + 1 >= 0 && ix + 1 < in_w) v.y = src[1];
                if (ix + 2 >= 0 && ix + 2 < in_w) v.z = src[2];
                if (ix + 3 >= 0 && ix + 3 < in_w) v.w = src[3];
            }
        }
        stage[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int idx = tid + 256 * j;
        if (idx < D_PH * GPR) *reinterpret_cast<float4*>(&patch[idx * 4]) = stage[j];
    }
    __syncthreads();

    const int cg = tid & 31, rg = tid >> 5;   // 32 column groups of 4 outputs x 8 row groups of 2 outputs
    float acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float* row = &patch[(rg * 4 + i) * D_PITCH + cg * 8];
        const float4 q0 = *reinterpret_cast<const float4*>(row), q1 = *reinterpret_cast<const float4*>(row + 4);
        const float2 q2 = *reinterpret_cast<const float2*>(row + 8);
        const float v[10] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int a = i - 2 * r;
            if (a < 0 || a > 3) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[r][j] = fmaf(T[a][b], v[2 * j + b], acc[r][j]);
        }
    }
    const int ox = ox0 + cg * 4;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oy = oy0 + rg * 2 + r;
        if (oy >= out_h) break;
        float* dst = yp + (size_t)oy * out_w + ox;
        if (VEC) {
            if (ox + 3 < out_w) gc::stream_store4(dst, acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
        } else if (ox + 3 < out_w) {
            gc::stream_store4u(dst, acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ox + j < out_w) dst[j] = acc[r][j];
        }
    }
}

// 4x4 taps, up = 2 (Upsample of the RGB skip, and the adjoint of every down = 2 FIR).  An output sees only the 2 x 2
// taps whose zero-stuffed coordinate is even, so a lane's 4 x 4 micro-tile needs a 4 x 4 input window.  With
// P = (first output coordinate of the lane) - pad, the window starts at floor(P / 2) and the tap / window indices depend
// only on the parity of P, which is uniform over the launch: EY / EX are template parameters.
constexpr int U_TH = 32, U_TW = 128;
constexpr int U_PH = U_TH / 2 + 3, U_PW = U_TW / 2 + 3, U_PITCH = 68;

template <int EY, int EX, bool VEC>
__global__ __launch_bounds__(256) void fir44_up2_kernel(
    const float* __restrict__ x, const float* __restrict__ taps, float* __restrict__ y,
    int in_h, int in_w, int out_h, int out_w, int pad_x0, int pad_y0, int flip) {
    __shared__ __attribute__((aligned(16))) float patch[U_PH * U_PITCH];
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * U_TW, oy0 = blockIdx.y * U_TH;
    const size_t plane = blockIdx.z;
    const float* xp = x + plane * (size_t)in_h * in_w;
    float* yp = y + plane * (size_t)out_h * out_w;

    float T[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) T[a][b] = flip ? taps[(3 - a) * 4 + (3 - b)] : taps[a * 4 + b];

    const int iy0 = (oy0 - pad_y0) >> 1, ix0 = (ox0 - pad_x0) >> 1;    // floor: tile origins are even, parity = pad parity
    constexpr int NLD = (U_PH * U_PITCH + 255) / 256;
    float stage[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int idx = tid + 256 * j;
        const int r = idx / U_PITCH, c = idx - r * U_PITCH;
        const int iy = iy0 + r, ix = ix0 + c;
        float v = 0.f;
        if (idx < U_PH * U_PITCH && c < U_PW && iy >= 0 && iy < in_h && ix >= 0 && ix < in_w) v = xp[(size_t)iy * in_w + ix];
        stage[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int idx = tid + 256 * j;
        if (idx < U_PH * U_PITCH) patch[idx] = stage[j];
    }
    __syncthreads();

    const int cg = tid & 31, rg = tid >> 5;   // 4 x 4 outputs per lane, window origin (rg * 2, cg * 2) in the patch
    float w[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float* row = &patch[(rg * 2 + i) * U_PITCH + cg * 2];
        const float2 lo = *reinterpret_cast<const float2*>(row), hi = *reinterpret_cast<const float2*>(row + 2);
        w[i][0] = lo.x; w[i][1] = lo.y; w[i][2] = hi.x; w[i][3] = hi.y;
    }
    const int ox = ox0 + cg * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int a0 = (EY + r) & 1, qy = (EY + r + a0) >> 1;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b0 = (EX + j) & 1, qx = (EX + j + b0) >> 1;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) s = fmaf(T[a0 + 2 * i][b0 + 2 * jj], w[qy + i][qx + jj], s);
            o[j] = s;
        }
        const int oy = oy0 + rg * 4 + r;
        if (oy < out_h) {
            float* dst = yp + (size_t)oy * out_w + ox;
            if (VEC) {
                if (ox + 3 < out_w) gc::stream_store4(dst, o[0], o[1], o[2], o[3]);
            } else if (ox + 3 < out_w) {
                gc::stream_store4u(dst, o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (ox + j < out_w) dst[j] = o[j];
            }
        }
    }
}

// 4x4 taps, up = down = 1 on SMALL planes (output <= 33 x 33: the 4^2 .. 32^2 layers, 512 channels): a workgroup stages whole
// zero-padded planes in LDS and every lane produces outputs from there -- one coalesced pass over x and y.  The generic kernel
// reads its 16 taps' inputs through L1 per output and reaches 350 GB/s on these planes.  Same tap order and fmaf chain as
// generic_kernel, so the two agree bit for bit.
constexpr int SMALL_LDS_FLOATS = 4096, SMALL_PLANE_MAX = 36 * 36;

__global__ __launch_bounds__(256) void fir44_small_kernel(
    const float* __restrict__ x, const float* __restrict__ taps, float* __restrict__ y,
    int planes, int in_h, int in_w, int out_h, int out_w, int pad_x0, int pad_y0, int flip, int ppw) {
    __shared__ float L[SMALL_LDS_FLOATS];
    __shared__ float T[16];
    const int tid = threadIdx.x;
    if (tid < 16) T[tid] = flip ? taps[15 - tid] : taps[tid];
    const int lh = out_h + 3, lw = out_w + 3, lp = lh * lw;
    const int p0 = blockIdx.x * ppw, np = min(ppw, planes - p0);
    const float* xb = x + (size_t)p0 * in_h * in_w;
    for (int i = tid; i < np * lp; i += 256) {
        const int pl = i / lp, rem = i - pl * lp;
        const int r = rem / lw, c = rem - r * lw;
        const int iy = r - pad_y0, ix = c - pad_x0;
        L[i] = (iy >= 0 && iy < in_h && ix >= 0 && ix < in_w) ? xb[((size_t)pl * in_h + iy) * in_w + ix] : 0.f;
    }
    __syncthreads();
    const int op = out_h * out_w;
    float* yb = y + (size_t)p0 * op;
    for (int i = tid; i < np * op; i += 256) {
        const int pl = i / op, rem = i - pl * op;
        const int oy = rem / out_w, ox = rem - oy * out_w;
        const float* l = L + pl * lp + oy * lw + ox;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc = fmaf(T[a * 4 + b], l[a * lw + b], acc);
        yb[i] = acc;
    }
}

constexpr int MAX_GENERIC_TAPS = 1024;

__global__ __launch_bounds__(256) void generic_kernel(
    const float* __restrict__ x, const float* __restrict__ taps, float* __restrict__ y,
    int planes, int in_h, int in_w, int out_h, int out_w, int kh, int kw,
    int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_y0, int flip) {
    __shared__ float T[MAX_GENERIC_TAPS];
    const int nt = kh * kw;
    for (int i = threadIdx.x; i < nt; i += 256) T[i] = flip ? taps[nt - 1 - i] : taps[i];
    __syncthreads();
    const size_t total = (size_t)planes * out_h * out_w;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int ox = (int)(idx % out_w);
        const size_t t = idx / out_w;
        const int oy = (int)(t % out_h);
        const size_t plane = t / out_h;
        const float* xp = x + plane * (size_t)in_h * in_w;
        // first tap whose zero-stuffed coordinate is a multiple of `up`, then step by `up`
        const int uy0 = oy * down_y - pad_y0, ux0 = ox * down_x - pad_x0;
        const int a0 = gc::pos_mod(-uy0, up_y), b0 = gc::pos_mod(-ux0, up_x);
        float acc = 0.f;
        for (int a = a0; a < kh; a += up_y) {
            const int uy = uy0 + a;
            if (uy < 0) continue;
            const int iy = uy / up_y;
            if (iy >= in_h) break;
            for (int b = b0; b < kw; b += up_x) {
                const int ux = ux0 + b;
                if (ux < 0) continue;
                const int ix = ux / up_x;
                if (ix >= in_w) break;
                acc = fmaf(T[a * kw + b], xp[(size_t)iy * in_w + ix], acc);
            }
        }
        y[idx] = acc;
    }
}

// Large square filters with up = 2 or down = 2 (the 12 x 12 sym6 anti-aliasing passes of the ADA warp, non_leaking.py:321-359,
// and their adjoints): a 256-thread workgroup makes a 64 x 16 output tile of one plane from an input patch staged ONCE in LDS
// (zero-filled outside the image), taps in LDS in multiplication order; a lane owns 4 consecutive outputs of a row and walks
// only the taps that meet a non-zero sample of the zero-stuffed input (K*K/UP^2 of them).  Replaces generic_kernel's one
// output per lane with up to K*K global reads each.
template <int K, int UP, int DOWN>
struct FirKCfg {
    static constexpr int TW = 64, TH = 16;
    static constexpr int PW = ((TW - 1) * DOWN + K - 1) / UP + 2, PH = ((TH - 1) * DOWN + K - 1) / UP + 2;
    static constexpr int PITCH = PW | 1;            // odd pitch: the stride-DOWN column walks of a wave spread over the banks
};

template <int K, int UP, int DOWN>
__global__ __launch_bounds__(256) void firK_tile_kernel(
    const float* __restrict__ x, const float* __restrict__ taps, float* __restrict__ y,
    int in_h, int in_w, int out_h, int out_w, int pad_x0, int pad_y0, int flip) {
    using C = FirKCfg<K, UP, DOWN>;
    __shared__ float patch[C::PH * C::PITCH];
    __shared__ float T[K * K];
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * C::TW, oy0 = blockIdx.y * C::TH;
    const size_t plane = blockIdx.z;
    const float* xp = x + plane * (size_t)in_h * in_w;
    float* yp = y + plane * (size_t)out_h * out_w;
    for (int i = tid; i < K * K; i += 256) T[i] = flip ? taps[K * K - 1 - i] : taps[i];
    // zero-stuffed coordinate of tap (0, 0) at the tile origin, and the input sample at / below it
    const int ux_base = ox0 * DOWN - pad_x0, uy_base = oy0 * DOWN - pad_y0;
    const int ixb = gc::floor_div(ux_base, UP), iyb = gc::floor_div(uy_base, UP);
    for (int i = tid; i < C::PH * C::PW; i += 256) {
        const int r = i / C::PW, c = i - r * C::PW;
        const int iy = iyb + r, ix = ixb + c;
        patch[r * C::PITCH + c] = (iy >= 0 && iy < in_h && ix >= 0 && ix < in_w) ? xp[(size_t)iy * in_w + ix] : 0.f;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = (tid & 15) * 4;
    const int oy = oy0 + ty;
    if (oy >= out_h) return;
    const int uy0 = oy * DOWN - pad_y0;
    const int a0 = gc::pos_mod(-uy0, UP);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int b0[4], cx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ux0 = (ox0 + tx + i) * DOWN - pad_x0;
        b0[i] = gc::pos_mod(-ux0, UP);
        cx[i] = (ux0 + b0[i] - ixb * UP) / UP;          // patch column of this output's first live tap
    }
    constexpr int NA = (K + UP - 1) / UP;
#pragma unroll 2
    for (int aa = 0; aa < NA; ++aa) {
        const int a = a0 + aa * UP;
        if (a >= K) break;
        const float* prow = patch + ((uy0 + a - iyb * UP) / UP) * C::PITCH;
        const float* trow = T + a * K;
#pragma unroll
        for (int bb = 0; bb < NA; ++bb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = b0[i] + bb * UP;
                if (UP == 1 || b < K) acc[i] = fmaf(trow[b], prow[cx[i] + bb], acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (ox0 + tx + i < out_w) yp[(size_t)oy * out_w + ox0 + tx + i] = acc[i];
}

template <int K>
int launch_firK(const float* x, const float* taps, float* y, int planes, int in_h, int in_w, int out_h, int out_w,
                int up, int down, int pad_x0, int pad_y0, int flip, hipStream_t s) {
    using C = FirKCfg<K, 1, 1>;
    dim3 grid(gc::ceil_div(out_w, C::TW), gc::ceil_div(out_h, C::TH), planes);
    if (up == 2 && down == 1)
        hipLaunchKernelGGL((firK_tile_kernel<K, 2, 1>), grid, dim3(256), 0, s, x, taps, y, in_h, in_w, out_h, out_w, pad_x0, pad_y0, flip);
    else if (up == 1 && down == 2)
        hipLaunchKernelGGL((firK_tile_kernel<K, 1, 2>), grid, dim3(256), 0, s, x, taps, y, in_h, in_w, out_h, out_w, pad_x0, pad_y0, flip);
    else
        hipLaunchKernelGGL((firK_tile_kernel<K, 1, 1>), grid, dim3(256), 0, s, x, taps, y, in_h, in_w, out_h, out_w, pad_x0, pad_y0, flip);
    return gc::check_launch("gc_upfirdn2d_f32(firK_tile)");
}

}  // namespace

namespace {

int upfirdn2d_impl(const float* x, const float* taps, float* y,
                   int planes, int in_h, int in_w, int out_h, int out_w,
                   int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                   int pad_x0, int pad_y0, int flip_taps, const FirEpilogue* ep, gc_stream_t stream, int in_pitch = 0, int out_pitch = 0) {
    if (in_pitch == 0) in_pitch = in_w;
    if (out_pitch == 0) out_pitch = out_w;
    if (out_pitch < out_w) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_f32: output pitch %d < width %d", out_pitch, out_w);
    if (in_pitch < in_w) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_f32: input pitch %d < width %d", in_pitch, in_w);
    if (!x || !taps || !y) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_f32: null pointer");
    if (planes < 0 || in_h <= 0 || in_w <= 0 || kh <= 0 || kw <= 0 || up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0)
        return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_f32: non-positive extent or factor");
    if (out_h <= 0 || out_w <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_f32: empty output (%d x %d)", out_h, out_w);
    if (planes == 0) return GC_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool fast = up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh == 4 && kw == 4 &&
                      out_w >= 64 && out_h >= 16 && planes <= 65535;
    if (ep && !fast) return gc::fail(GC_ERR_UNSUPPORTED, "gc_upfirdn2d_act_f32: the fused epilogue needs the 4x4, up = down = 1 tile kernel (planes >= 64 x 16)");
    if ((in_pitch != in_w || out_pitch != out_w) && !fast) return gc::fail(GC_ERR_UNSUPPORTED, "gc_upfirdn2d_pitched_f32: a pitched input needs the 4x4, up = down = 1 tile kernel (planes >= 64 x 16)");
    if (fast) {
        dim3 grid(gc::ceil_div(out_w, TW), gc::ceil_div(gc::ceil_div(out_h, TH), GC_FIR_NT), planes);
        const bool vec = (out_pitch % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);      // rows start on 16-byte boundaries
        const FirEpilogue none{nullptr, nullptr, nullptr, 1.f, 1.f, 1, nullptr, 1.f, 1.f, nullptr, nullptr, nullptr, nullptr, 1.f, 1.f};
        if (ep && ep->pro_ref) {
            if (in_pitch != in_w) return gc::fail(GC_ERR_UNSUPPORTED, "gc_upfirdn2d_actbwd_f32: the gradient must have dense rows");
            if (vec) hipLaunchKernelGGL((fir44_tile_kernel<true, 0, true>), grid, dim3(256), 0, s, x, taps, y, in_h, in_w, out_h, out_w, pad_x0, pad_y0, flip_taps, *ep, in_pitch, out_pitch);
            else     hipLaunchKernelGGL((fir44_tile_kernel<false, 0, true>), grid, dim3(256), 0, s, x, taps, y, in_h, in_w, out_h, out_w, pad_x0, pad_y0, flip_taps, *ep, in_pitch, out_pitch);
            return gc::check_launch("gc_upfirdn2d_actbwd_f32");
        }
#define GC_FIR(V, E) hipLaunchKernelGGL((fir44_tile_kernel<V, E>), grid, dim3(256), 0, s, x, taps, y, in_h, in_w, out_h, out_w, pad_x0, pad_y0, flip_taps, ep ? *ep : none, in_pitch, out_pitch)
        if (ep && ep->mask_ref) { if (vec) GC_FIR(true, 2); else GC_FIR(false, 2); }
        else if (ep) { if (vec) GC_FIR(true, 1); else GC_FIR(false, 1); }
        else    { if (vec) GC_FIR(true, 0); else GC_FIR(false, 0); }
#undef GC_FIR
        return gc::check_launch("gc_upfirdn2d_f32(fir44_tile)");
    }
    if (kh == 4 && kw == 4 && up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && (out_h + 3) * (out_w + 3) <= SMALL_PLANE_MAX) {
        const int ppw = std::max(1, std::min(SMALL_LDS_FLOATS / ((out_h + 3) * (out_w + 3)), 1024 / std::max(1, out_h * out_w) + 1));
        hipLaunchKernelGGL(fir44_small_kernel, dim3(gc::ceil_div(planes, ppw)), dim3(256), 0, s, x, taps, y, planes, in_h, in_w, out_h, out_w,
                           pad_x0, pad_y0, flip_taps, ppw);
        return gc::check_launch("gc_upfirdn2d_f32(fir44_small)");
    }
    const bool square44 = kh == 4 && kw == 4 && up_x == up_y && down_x == down_y && planes <= 65535 && out_w >= 32 && out_h >= 8;
    const bool vec_ok = (out_w % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    if (square44 && up_x == 1 && down_x == 2) {
        dim3 grid(gc::ceil_div(out_w, D_TW), gc::ceil_div(out_h, D_TH), planes);
        if (vec_ok)
            hipLaunchKernelGGL(fir44_down2_kernel<true>, grid, dim3(256), 0, s, x, taps, y, in_h, in_w, out_h, out_w, pad_x0, pad_y0, flip_taps);
        else
            hipLaunchKernelGGL(fir44_down2_kernel<false>, grid, dim3(256), 0, s, x, taps, y, in_h, in_w, out_h, out_w, pad_x0, pad_y0, flip_taps);
        return gc::check_launch("gc_upfirdn2d_f32(fir44_down2)");
    }
    if (square44 && up_x == 2 && down_x == 1) {
        dim3 grid(gc::ceil_div(out_w, U_TW), gc::ceil_div(out_h, U_TH), planes);
        const int ey = pad_y0 & 1, ex = pad_x0 & 1;     // parity of (tile origin - pad); tile origins are multiples of 4
#define GC_UP2(EY, EX)                                                                                                             \
        if (ey == EY && ex == EX) {                                                                                                \
            if (vec_ok) hipLaunchKernelGGL((fir44_up2_kernel<EY, EX, true>), grid, dim3(256), 0, s, x, taps, y, in_h, in_w, out_h, out_w, pad_x0, pad_y0, flip_taps); \
            else hipLaunchKernelGGL((fir44_up2_kernel<EY, EX, false>), grid, dim3(256), 0, s, x, taps, y, in_h, in_w, out_h, out_w, pad_x0, pad_y0, flip_taps);       \
        }
        GC_UP2(0, 0) GC_UP2(0, 1) GC_UP2(1, 0) GC_UP2(1, 1)
#undef GC_UP2
        return gc::check_launch("gc_upfirdn2d_f32(fir44_up2)");
    }
    // 12 x 12 taps (sym6 x sym6) with (up, down) in {(2, 1), (1, 2), (1, 1)} on planes worth tiling: the ADA anti-aliasing passes
    if (kh == 12 && kw == 12 && up_x == up_y && down_x == down_y && up_x * down_x <= 2 && planes <= 65535 && out_w >= 32 && out_h >= 8)
        return launch_firK<12>(x, taps, y, planes, in_h, in_w, out_h, out_w, up_x, down_x, pad_x0, pad_y0, flip_taps, s);
    if (kh * kw > MAX_GENERIC_TAPS) return gc::fail(GC_ERR_UNSUPPORTED, "gc_upfirdn2d_f32: %d x %d taps exceed %d", kh, kw, MAX_GENERIC_TAPS);
    const size_t total = (size_t)planes * out_h * out_w;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(generic_kernel, dim3(blocks), dim3(256), 0, s, x, taps, y, planes, in_h, in_w, out_h, out_w, kh, kw,
                       up_x, up_y, down_x, down_y, pad_x0, pad_y0, flip_taps);
    return gc::check_launch("gc_upfirdn2d_f32(generic)");
}

}  // namespace

extern "C" int gc_upfirdn2d_f32(const float* x, const float* taps, float* y,
                                int planes, int in_h, int in_w, int out_h, int out_w,
                                int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                                int pad_x0, int pad_y0, int flip_taps, gc_stream_t stream) {
    return upfirdn2d_impl(x, taps, y, planes, in_h, in_w, out_h, out_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, flip_taps, nullptr, stream);
}

extern "C" int gc_upfirdn2d_pitched_f32(const float* x, const float* taps, float* y, int batch, int channels, int in_h, int in_w, int in_pitch,
                                        int out_h, int out_w, int out_pitch, int kh, int kw, int pad_x0, int pad_y0, int flip_taps, int activate,
                                        const float* bias, const float* noise, const float* noise_w, float slope, float gain, gc_stream_t stream) {
    if ((noise == nullptr) != (noise_w == nullptr)) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_pitched_f32: noise and noise_w must both be set or both be null");
    if (batch < 0 || channels <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_pitched_f32: bad batch / channels");
    const bool epi = activate || bias || noise;
    const FirEpilogue ep{bias, noise, noise_w, activate ? slope : 1.f, activate ? gain : 1.f, channels, nullptr, 1.f, 1.f, nullptr, nullptr, nullptr, nullptr, 1.f, 1.f};
    return upfirdn2d_impl(x, taps, y, batch * channels, in_h, in_w, out_h, out_w, kh, kw, 1, 1, 1, 1, pad_x0, pad_y0, flip_taps, epi ? &ep : nullptr, stream, in_pitch, out_pitch);
}

extern "C" int gc_upfirdn2d_actbwd_tiles(int out_h, int out_w) {
    if (out_h <= 0 || out_w <= 0) return 0;
    return gc::ceil_div(out_w, TW) * gc::ceil_div(gc::ceil_div(out_h, TH), GC_FIR_NT);
}

extern "C" int gc_upfirdn2d_actbwd_f32(const float* gy, const float* y_ref, const float* noise, const float* taps, float* gx, float* psum, float* pdot,
                                       int batch, int channels, int in_h, int in_w, int out_h, int out_w, int out_pitch, int kh, int kw,
                                       int pad_x0, int pad_y0, int flip_taps, float slope, float gain, gc_stream_t stream) {
    if (!y_ref || !psum) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_actbwd_f32: null pointer");
    if ((noise == nullptr) != (pdot == nullptr)) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_actbwd_f32: noise and pdot go together");
    if (batch < 0 || channels <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_actbwd_f32: bad batch / channels");
    const FirEpilogue ep{nullptr, nullptr, nullptr, 1.f, 1.f, channels, nullptr, 1.f, 1.f, y_ref, noise, psum, pdot, gain, gain * slope};
    return upfirdn2d_impl(gy, taps, gx, batch * channels, in_h, in_w, out_h, out_w, kh, kw, 1, 1, 1, 1, pad_x0, pad_y0, flip_taps, &ep, stream, 0, out_pitch);
}

extern "C" int gc_upfirdn2d_mask_f32(const float* x, const float* taps, float* y, int batch, int channels, int in_h, int in_w, int in_pitch,
                                     int out_h, int out_w, int kh, int kw, int pad_x0, int pad_y0, int flip_taps,
                                     const float* mask_ref, float slope, float gain, gc_stream_t stream) {
    if (!mask_ref) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_mask_f32: null mask reference");
    if (batch < 0 || channels <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_mask_f32: bad batch / channels");
    const FirEpilogue ep{nullptr, nullptr, nullptr, 1.f, 1.f, channels, mask_ref, gain, gain * slope, nullptr, nullptr, nullptr, nullptr, 1.f, 1.f};
    return upfirdn2d_impl(x, taps, y, batch * channels, in_h, in_w, out_h, out_w, kh, kw, 1, 1, 1, 1, pad_x0, pad_y0, flip_taps, &ep, stream, in_pitch, 0);
}

extern "C" int gc_upfirdn2d_act_f32(const float* x, const float* taps, float* y,
                                    int batch, int channels, int in_h, int in_w, int out_h, int out_w,
                                    int kh, int kw, int pad_x0, int pad_y0, int flip_taps,
                                    const float* bias, const float* noise, const float* noise_w, float slope, float gain,
                                    gc_stream_t stream) {
    if ((noise == nullptr) != (noise_w == nullptr)) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_act_f32: noise and noise_w must both be set or both be null");
    if (batch < 0 || channels <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_upfirdn2d_act_f32: bad batch / channels");
    const FirEpilogue ep{bias, noise, noise_w, slope, gain, channels, nullptr, 1.f, 1.f, nullptr, nullptr, nullptr, nullptr, 1.f, 1.f};
    return upfirdn2d_impl(x, taps, y, batch * channels, in_h, in_w, out_h, out_w, kh, kw, 1, 1, 1, 1, pad_x0, pad_y0, flip_taps, &ep, stream);
}
