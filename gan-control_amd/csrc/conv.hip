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
//   Software pipeline: the global loads of chunk c+1 are issued into registers BEFORE the MFMA
//   block of chunk c and written to LDS after it, so HBM/L2 latency hides under the matrix work
//   (64 cycles per MFMA) even with one workgroup per CU.  Geometry (taps, up, down) is a template
//   parameter: every staging index is a compile-time division.
//   up = 2 (transposed conv) is decomposed into its up*up output phases: each workgroup handles
//   one phase, whose taps are the subset {ty : (phase + ty - pad) % up == 0} -- no multiplies by
//   stuffed zeros.
//   Small planes (<= 16 px wide) use KSPLIT = 4: one 32oc x 32px tile per workgroup, the four
//   waves split the channel pairs of every chunk and are summed through LDS at the end.
//
// wgrad_mfma_kernel (weight gradient)
//   dW[tap][k][n] = sum_px X[k][px + tap] * dY[n][px]: A = input patch (lane -> channel k),
//   B = dY tile (lane -> channel n), reduction over pixels.  Each wave owns a 32x32 (k, n) block
//   for ALL taps (<= 9 accumulators = 144 registers), so dY is read from LDS once per pixel pair.
//   Same register-prefetch pipeline over pixel tiles.  The pixel space is split across
//   workgroups; partial sums go to a workspace and are reduced in fixed order by
//   wgrad_reduce_kernel (deterministic, no atomics).
#include "conv_common.h"

namespace {

using namespace gcconv;

template <int WG_OC, int WG_PX, int KSPLIT, int KCT, int WOC, int WPX, int TPW, int UP, int DOWN, int KS>
struct ConvCfg {
    static constexpr int OCT = WG_OC * WOC * 32;           // output channels per workgroup
    static constexpr int RPB = 32 / TPW;                   // tile rows covered by one 32-pixel MFMA column block
    static constexpr int TPH = WG_PX * WPX * RPB;          // tile rows per workgroup
    static constexpr int NT1 = UP == 1 ? KS : (KS + UP - 1) / UP;   // max taps per axis in one phase
    static constexpr int PH = (TPH - 1) * DOWN + NT1;      // patch rows
    static constexpr int PWD = (TPW - 1) * DOWN + NT1;     // patch columns
    static constexpr int PP = patch_pitch(PWD, TPW);       // patch row pitch
    static constexpr int PLANE = PH * PP;
    static constexpr int WL = NT1 * NT1 * KCT * OCT;        // weight slab floats
    static constexpr int PATCH = KCT * PLANE;
    static constexpr int RED = KSPLIT > 1 ? 4 * WOC * WPX * 16 * 64 : 0;
    static constexpr int SMEM = cmax(WL + PATCH, RED);
    static constexpr int NPE = (KCT * PH * PWD + 255) / 256;   // patch elements prefetched per thread
    static constexpr int F4 = OCT / 4;                        // float4 per weight row
    static constexpr int RPI = 256 / F4;                      // weight rows per staging iteration
    static constexpr int NWI = (NT1 * NT1 * KCT + RPI - 1) / RPI;
};

template <int WG_OC, int WG_PX, int KSPLIT, int KCT, int WOC, int WPX, int TPW, int UP, int DOWN, int KS>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvArgs p) {
    using C = ConvCfg<WG_OC, WG_PX, KSPLIT, KCT, WOC, WPX, TPW, UP, DOWN, KS>;
    constexpr int KC = KCT;   // input channels staged per LDS chunk
    static_assert(WG_OC * WG_PX * KSPLIT == 4, "4 waves per workgroup");
    static_assert(UP == 1 || DOWN == 1, "up and down are exclusive");
    constexpr int OCT = C::OCT, RPB = C::RPB, TPH = C::TPH, PH = C::PH, PWD = C::PWD, PP = C::PP, PLANE = C::PLANE;
    __shared__ __attribute__((aligned(16))) float smem[C::SMEM];
    float* wl = smem;                 // [tap][KC][OCT]
    float* patch = smem + C::WL;      // [KC][PH][PP]
    __shared__ float s_so[OCT], s_bias[OCT];   // out_scale / bias of this workgroup's channels (see conv_epilogue)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wave_k = wave % KSPLIT, wave_px = (wave / KSPLIT) % WG_PX, wave_oc = wave / (KSPLIT * WG_PX);

    // ---- decode the workgroup: (tile_x, tile_y, phase, sample) x n-tile ----
    int bid = blockIdx.x;
    const int tile_x = bid % p.tiles_x; bid /= p.tiles_x;
    const int tile_y = bid % p.tiles_y; bid /= p.tiles_y;
    const int phase = bid % (UP * UP);
    const int b = bid / (UP * UP);
    const int phy = phase / UP, phx = phase % UP;
    const int n0 = blockIdx.y * OCT;
    const int qh = (p.out_h - phy + UP - 1) / UP, qw = (p.out_w - phx + UP - 1) / UP;
    const int qy0 = tile_y * TPH, qx0 = tile_x * TPW;
    if (qy0 >= qh || qx0 >= qw) return;

    const AxisTaps ay = axis_taps<UP, KS>(phy, p.pad_y), ax = axis_taps<UP, KS>(phx, p.pad_x);
    const int ntaps = ay.n * ax.n;
    const int iy0 = qy0 * DOWN + ay.d0, ix0 = qx0 * DOWN + ax.d0;

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
        boff[j] = hi * PLANE + row * DOWN * PP + col * DOWN;
    }
    const int aoff = hi * OCT + wave_oc * WOC * 32 + l31;

    const float* xb = p.x + (size_t)b * p.K * p.in_h * p.in_w;
    const float* sib = p.si ? p.si + (size_t)b * p.K : nullptr;
    for (int o = tid; o < OCT; o += 256) {     // read only in the epilogue
        const int oc = min(n0 + o, p.N - 1);
        s_so[o] = p.so ? p.so[(size_t)b * p.N + oc] : 1.f;
        s_bias[o] = p.bias ? p.bias[oc] : 0.f;
    }
    __syncthreads();
    const bool wvec = (p.N % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.w) & 15) == 0);
    const int chan = p.in_h * p.in_w;          // host guarantees K * H * W < 2^31

    float4 wreg[C::NWI];
    float preg[C::NPE], sreg[C::NPE];   // raw patch values and their in_scale factors (multiplied at commit time)

    // Buffer-descriptor loads (conv_common.h): 32-bit lane byte offsets, hardware zero-fill for everything outside
    // the image / the weight slab, results untouched until commit() so the loads span the whole MFMA block.
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)p.K * chan * 4u);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (unsigned)(KS * KS) * p.K * p.N * 4u);
    auto prefetch = [&](int k0) {
        const int t_ = opaque(tid);
        const int wcol = (t_ % C::F4) * 4, wrow0 = t_ / C::F4;
        // weights: row = (tap, kk), F4 float4 per row
#pragma unroll
        for (int j = 0; j < C::NWI; ++j) {
            const int row = wrow0 + C::RPI * j;
            const int t = row / KC, kk = row % KC;
            const int jy = UP == 1 ? t / KS : (ax.n == 2 ? t >> 1 : t), jx = UP == 1 ? t % KS : (ax.n == 2 ? t & 1 : 0);
            const int ty = ay.t0 + jy * UP, tx = ax.t0 + jx * UP;
            const int k = k0 + kk, n = n0 + wcol;
            const bool ok = t < ntaps && k < p.K;
            const unsigned base = (unsigned)(((ty * KS + tx) * p.K + k) * p.N + n) * 4u;
            if (wvec) {
                wreg[j] = __builtin_bit_cast(float4, buf_load_u128(rw, (ok && n < p.N) ? base : OOB, 0));
            } else {
                wreg[j].x = buf_load_f32(rw, (ok && n < p.N) ? base : OOB, 0);
                wreg[j].y = buf_load_f32(rw, (ok && n + 1 < p.N) ? base + 4u : OOB, 0);
                wreg[j].z = buf_load_f32(rw, (ok && n + 2 < p.N) ? base + 8u : OOB, 0);
                wreg[j].w = buf_load_f32(rw, (ok && n + 3 < p.N) ? base + 12u : OOB, 0);
            }
        }
        // input patch: element e = tid + 256 j -> (kk, r, c), zero outside the image
#pragma unroll
        for (int j = 0; j < C::NPE; ++j) {
            const int e = t_ + 256 * j;
            const int kk = e / (PH * PWD), pos = e % (PH * PWD);
            const int r = pos / PWD, c = pos % PWD;
            const int k = k0 + kk, iy = iy0 + r, ix = ix0 + c;
            const bool ok = e < KC * PH * PWD && k < p.K && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
            preg[j] = buf_load_f32(rx, ok ? (unsigned)(k * chan + iy * p.in_w + ix) * 4u : OOB, 0);
            sreg[j] = sib ? sib[min(k, p.K - 1)] : 1.f;
        }
    };
    auto commit = [&]() {
        wait_staged_loads();
        const int t_ = opaque(tid);
        const int wcol = (t_ % C::F4) * 4, wrow0 = t_ / C::F4;
#pragma unroll
        for (int j = 0; j < C::NWI; ++j) {
            const int row = wrow0 + C::RPI * j;
            if (row < C::NT1 * C::NT1 * KC) *reinterpret_cast<float4*>(wl + row * OCT + wcol) = wreg[j];
        }
#pragma unroll
        for (int j = 0; j < C::NPE; ++j) {
            const int e = t_ + 256 * j;
            const int kk = e / (PH * PWD), pos = e % (PH * PWD);
            const int r = pos / PWD, c = pos % PWD;
            if (e < KC * PH * PWD) patch[kk * PLANE + r * PP + c] = preg[j] * sreg[j];
        }
    };

    // channel range of this workgroup (the whole of K unless the launch splits it over blockIdx.z)
    const int kz0 = p.k_per_split ? blockIdx.z * p.k_per_split : 0;
    const int kz1 = p.k_per_split ? min(p.K, kz0 + p.k_per_split) : p.K;
    if (ntaps > 0) {
        prefetch(kz0);
        commit();
        __syncthreads();
        for (int k0 = kz0; k0 < kz1; k0 += KC) {
            wait_staged_loads();    // no-op in hardware (commit retired them); clears the compiler's pending-load model at the loop header
            const bool more = k0 + KC < kz1;
            if (more) prefetch(k0 + KC);
            // ---- MFMA over taps x channel pairs ----
            const int nty = UP == 1 ? KS : ay.n, ntx = UP == 1 ? KS : ax.n;
            for (int jy = 0; jy < nty; ++jy) {
                for (int jx = 0; jx < ntx; ++jx) {
                    const float* wt = wl + (jy * ntx + jx) * KC * OCT + aoff;
                    const float* pt = patch + jy * PP + jx;
#pragma unroll
                    for (int kp = wave_k; kp < KC / 2; kp += KSPLIT) {
                        float a[WOC], bv[WPX];
#pragma unroll
                        for (int i = 0; i < WOC; ++i) a[i] = wt[kp * 2 * OCT + i * 32];
#pragma unroll
                        for (int j = 0; j < WPX; ++j) bv[j] = pt[kp * 2 * PLANE + boff[j]];
#pragma unroll
                        for (int i = 0; i < WOC; ++i)
#pragma unroll
                            for (int j = 0; j < WPX; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
            if (!more) break;       // leave here: no path may reach the loop header with staged loads in flight
            {
                commit();
                __syncthreads();
            }
        }
    }

    if (KSPLIT > 1) {
        // sum the channel-pair partitions of the four waves through LDS
        float* red = smem;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < WOC; ++i)
#pragma unroll
            for (int j = 0; j < WPX; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wave * WOC * WPX + i * WPX + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        __syncthreads();
        if (wave_k != 0) return;
#pragma unroll
        for (int i = 0; i < WOC; ++i)
#pragma unroll
            for (int j = 0; j < WPX; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = acc[i][j][r];
                    for (int o = 1; o < KSPLIT; ++o) s += red[(((wave + o) * WOC * WPX + i * WPX + j) * 16 + r) * 64 + lane];
                    acc[i][j][r] = s;
                }
    }

    // ---- epilogue: demodulate and store; lanes 0..31 of a register are consecutive pixels ----
    float* yb = (p.k_per_split ? p.part + (size_t)blockIdx.z * p.B * p.N * p.out_h * p.out_w : p.y) + (size_t)b * p.N * p.out_h * p.out_w;
    const EpilogueConsts ec = epilogue_consts(p);        // identity for a split launch (the host clears the epilogue fields)
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        const int qy = qy0 + (wave_px * WPX + j) * RPB + l31 / TPW, qx = qx0 + l31 % TPW;
        if (qy >= qh || qx >= qw) continue;
        const int oy = qy * UP + phy, ox = qx * UP + phx;
        const float nz = p.noise ? p.noise[((size_t)b * p.out_h + oy) * p.out_w + ox] : 0.f;
        float res[WOC][16];
        if (p.residual) {           // fetched before the first store of this pixel row
#pragma unroll
            for (int i = 0; i < WOC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int oc = min(n0 + (wave_oc * WOC + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, p.N - 1);
                    res[i][r] = p.residual[(((size_t)b * p.N + oc) * p.out_h + oy) * p.out_w + ox];
                }
        }
#pragma unroll
        for (int i = 0; i < WOC; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ocl = (wave_oc * WOC + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (n0 + ocl < p.N) {
                    float v = conv_epilogue(ec, acc[i][j][r], s_so[ocl], s_bias[ocl], nz);
                    if (p.residual) v += res[i][r];
                    yb[((size_t)(n0 + ocl) * p.out_h + oy) * p.out_w + ox] = v;
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x; const float* dy; const float* si; const float* so; float* ws;
    int B, K, N, in_h, in_w, out_h, out_w, pad_y, pad_x;
    int tiles_x, tiles_y;     // pixel tiles per sample
    int tiles_per_split;
};

template <int WK, int WN, int WP, int TR, int DOWN, int KS>
struct WgradCfg {
    static constexpr int KT = WK * 32, NTL = WN * 32;    // channel tiles of the workgroup
    static constexpr int TPW = 32;                       // pixel tile: TR rows x 32 columns
    static constexpr int PH = (TR - 1) * DOWN + KS, PWD = (TPW - 1) * DOWN + KS;
    static constexpr int PP = PWD;                       // patch row pitch
    static constexpr int MAINW = 32 * DOWN;              // power-of-two part of a patch row (shift/mask staging indices)
    static constexpr int TAILW = PWD > MAINW ? PWD - MAINW : 0;   // the 1-2 halo columns beyond it
    static constexpr int CSX = (PH * PP) | 1;            // odd channel strides: conflict-free lane -> channel reads
    static constexpr int CSY = (TR * 32) | 1;
    static constexpr int RX = KT * PH;                   // patch rows staged per tile
    static constexpr int NPXM = RX * MAINW / 256, NPXT = (RX * TAILW + 255) / 256, NPY = NTL * TR * 32 / 256;
    static constexpr int RED = (WP - 1) * WK * WN * 16 * 64;
    static constexpr int SMEM = cmax(KT * CSX + NTL * CSY, RED);
    static constexpr int NT = KS * KS;
};

template <int WK, int WN, int WP, int TR, int DOWN, int KS>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WgradArgs p) {
    using C = WgradCfg<WK, WN, WP, TR, DOWN, KS>;
    constexpr int KT = C::KT, NTL = C::NTL, TPW = C::TPW, PH = C::PH, PWD = C::PWD, PP = C::PP, NT = C::NT;
    constexpr int MAINW = C::MAINW, TAILW = C::TAILW;
    static_assert(WK * WN * WP == 4, "4 waves per workgroup");
    static_assert(TR % WP == 0, "rows split evenly over the pixel waves");
    static_assert((C::RX * MAINW) % 256 == 0 && (NTL * TR * 32) % 256 == 0, "staging loops are exact");
    __shared__ __attribute__((aligned(16))) float smem[C::SMEM];
    float* xs = smem;                        // [KT][PH][PP]   (channel stride CSX)
    float* ds = smem + KT * C::CSX;          // [NTL][TR][32]  (channel stride CSY)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wp = wave % WP, wn = (wave / WP) % WN, wk = wave / (WP * WN);
    const int k0 = blockIdx.x * KT, n0 = blockIdx.y * NTL, split = blockIdx.z;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int tiles_per_sample = p.tiles_x * p.tiles_y;
    const int total_tiles = tiles_per_sample * p.B;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(total_tiles, t_begin + p.tiles_per_split);
    const int xchan = p.in_h * p.in_w, ychan = p.out_h * p.out_w;   // host guarantees C * H * W < 2^31

    float xreg[C::NPXM], treg[C::NPXT > 0 ? C::NPXT : 1], yreg[C::NPY];
    // per-sample channel factors of the tile in flight: this lane's A rows are channel k0 + wk*32 + l31 and its
    // B columns channel n0 + wn*32 + l31, so modulation is one multiply per fragment (loaded with the prefetch)
    float sx_cur = 1.f, sy_cur = 1.f, sx_next = 1.f, sy_next = 1.f;
    const unsigned xbytes = (unsigned)p.K * xchan * 4u, ybytes = (unsigned)p.N * ychan * 4u;
    auto load_x = [&](__amdgpu_buffer_rsrc_t rx, int row, int c, int iy0, int ix0) -> float {
        const int r = row % PH, kk = row / PH;
        const int k = k0 + kk, iy = iy0 + r, ix = ix0 + c;
        const bool ok = c < PWD && row < C::RX && k < p.K && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
        return buf_load_f32(rx, ok ? (unsigned)(k * xchan + iy * p.in_w + ix) * 4u : OOB, 0);
    };
    auto prefetch = [&](int tile) {
        const int t_ = opaque(tid);
        const int b = tile / tiles_per_sample;
        const int rem = tile - b * tiles_per_sample;
        const int oy0 = (rem / p.tiles_x) * TR, ox0 = (rem % p.tiles_x) * TPW;
        const int iy0 = oy0 * DOWN - p.pad_y, ix0 = ox0 * DOWN - p.pad_x;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)b * p.K * xchan, xbytes);
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.dy + (size_t)b * p.N * ychan, ybytes);
        {
            const int kl = k0 + wk * 32 + l31, nl = n0 + wn * 32 + l31;
            sx_next = (p.si && kl < p.K) ? p.si[(size_t)b * p.K + kl] : 1.f;
            sy_next = (p.so && nl < p.N) ? p.so[(size_t)b * p.N + nl] : 1.f;
        }
#pragma unroll
        for (int j = 0; j < C::NPXM; ++j) {
            const int e = t_ + 256 * j;
            xreg[j] = load_x(rx, e / MAINW, e % MAINW, iy0, ix0);
        }
        if (TAILW > 0) {
#pragma unroll
            for (int j = 0; j < C::NPXT; ++j) {
                const int e = t_ + 256 * j;
                treg[j] = load_x(rx, e / (TAILW > 0 ? TAILW : 1), MAINW + e % (TAILW > 0 ? TAILW : 1), iy0, ix0);
            }
        }
#pragma unroll
        for (int j = 0; j < C::NPY; ++j) {
            const int e = t_ + 256 * j;
            const int c = e % 32, row = e / 32;
            const int r = row % TR, nn = row / TR;
            const int n = n0 + nn, oy = oy0 + r, ox = ox0 + c;
            const bool ok = n < p.N && oy < p.out_h && ox < p.out_w;
            yreg[j] = buf_load_f32(ry, ok ? (unsigned)(n * ychan + oy * p.out_w + ox) * 4u : OOB, 0);
        }
    };
    auto commit = [&]() {
        wait_staged_loads();
        const int t_ = opaque(tid);
#pragma unroll
        for (int j = 0; j < C::NPXM; ++j) {
            const int e = t_ + 256 * j;
            const int c = e % MAINW, row = e / MAINW;
            if (c < PWD) xs[(row / PH) * C::CSX + (row % PH) * PP + c] = xreg[j];
        }
        if (TAILW > 0) {
#pragma unroll
            for (int j = 0; j < C::NPXT; ++j) {
                const int e = t_ + 256 * j;
                const int c = MAINW + e % (TAILW > 0 ? TAILW : 1), row = e / (TAILW > 0 ? TAILW : 1);
                if (row < C::RX) xs[(row / PH) * C::CSX + (row % PH) * PP + c] = treg[j];
            }
        }
#pragma unroll
        for (int j = 0; j < C::NPY; ++j) {
            const int e = t_ + 256 * j;
            ds[(e / (TR * 32)) * C::CSY + e % (TR * 32)] = yreg[j];
        }
    };

    if (t_begin < t_end) {
        prefetch(t_begin);
        commit();
        sx_cur = sx_next; sy_cur = sy_next;
        __syncthreads();
        const float* xa = xs + (wk * 32 + l31) * C::CSX;
        const float* db = ds + (wn * 32 + l31) * C::CSY;
        for (int tile = t_begin; tile < t_end; ++tile) {
            wait_staged_loads();    // no-op in hardware (commit retired them); clears the compiler's pending-load model at the loop header
            const bool more = tile + 1 < t_end;
            if (more) prefetch(tile + 1);
            // ---- MFMA: reduction over the pixels of this tile; this wave takes rows r = wp, wp+WP, ... ----
            for (int r = wp; r < TR; r += WP) {
#pragma unroll 2
                for (int c = 0; c < TPW; c += 2) {
                    const int col = c + hi;
                    const float bv = db[r * 32 + col] * sy_cur;
                    const float* xr = xa + r * DOWN * PP + col * DOWN;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float av = xr[(t / KS) * PP + (t % KS)] * sx_cur;
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
            if (!more) break;       // leave here: no path may reach the loop header with staged loads in flight
            {
                commit();
                sx_cur = sx_next; sy_cur = sy_next;
                __syncthreads();
            }
        }
    }

    if (WP > 1) {
        // the WP pixel-waves of a (wk, wn) group hold partial sums of the SAME (k, n) block: add them through LDS
        float* red = smem + (wk * WN + wn) * (WP - 1) * 16 * 64;
        for (int t = 0; t < NT; ++t) {
            __syncthreads();
            if (wp > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wp - 1) * 16 + r) * 64 + lane] = acc[t][r];
            }
            __syncthreads();
            if (wp == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[t][r];
                    for (int o = 0; o < WP - 1; ++o) v += red[(o * 16 + r) * 64 + lane];
                    acc[t][r] = v;
                }
            }
        }
        if (wp != 0) return;
    }

    // ---- partial result: ws[split][tap][k][n] (or dw itself when there is a single split); lanes 0..31 = consecutive n ----
    float* out = p.ws + (size_t)split * NT * p.K * p.N;
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

// dw[i] = sum_s ws[s][i] in a fixed order; count is a multiple of 4 when VEC (16-byte loads, 8 in flight per lane).
// JG > 1: a small weight tensor split over many pixel splits (32 x 32 x 9 floats x 512 splits at 1024^2) would be a handful of workgroups
// each walking hundreds of dependent-latency loads; there 256 / JG lanes own an element group and JG lane groups each sum every JG-th
// split, added through LDS in group order.
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int JG>
__device__ __forceinline__ float4 sum_parts4(const float* __restrict__ base, size_t count, size_t i, int parts, int jg) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = jg;
    for (; s + 7 * JG < parts; s += 8 * JG) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(base + (size_t)(s + u * JG) * count)[i];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = add4(acc, v[u]);
    }
    for (; s < parts; s += JG) acc = add4(acc, reinterpret_cast<const float4*>(base + (size_t)s * count)[i]);
    return acc;
}

// the JG partial sums of one element group, added in group order by the lanes of group 0 (all 256 lanes call this)
template <int JG>
__device__ __forceinline__ float4 join_groups(float4 acc, float4* red, int il, int jg) {
    if (JG == 1) return acc;
    __syncthreads();
    red[jg * (256 / JG) + il] = acc;
    __syncthreads();
    float4 tot = red[il];
#pragma unroll
    for (int g = 1; g < JG; ++g) tot = add4(tot, red[g * (256 / JG) + il]);
    return tot;
}

template <bool VEC, int JG>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, size_t count, int parts) {
    if (VEC) {
        __shared__ float4 red[JG == 1 ? 1 : 256];
        constexpr int IL = 256 / JG;
        const int il = threadIdx.x % IL, jg = threadIdx.x / IL;
        const size_t n4 = count / 4;
        for (size_t i0 = (size_t)blockIdx.x * IL; i0 < n4; i0 += (size_t)gridDim.x * IL) {       // uniform trip count: the barriers inside are safe
            const size_t i = i0 + il;
            const float4 acc = join_groups<JG>(i < n4 ? sum_parts4<JG>(ws, count, i, parts, jg) : make_float4(0.f, 0.f, 0.f, 0.f), red, il, jg);
            if (jg == 0 && i < n4) reinterpret_cast<float4*>(dw)[i] = acc;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
            float acc = 0.f;
            for (int s = 0; s < parts; ++s) acc += ws[(size_t)s * count + i];
            dw[i] = acc;
        }
    }
}

// Partial sums grouped by sample, ws[b][j][i]: samples[b][i] = sum_j ws[b][j][i] and dw[i] = sum_b samples[b][i], both in fixed order.
template <bool VEC, int JG>
__global__ __launch_bounds__(256) void wgrad_reduce_samples_kernel(const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ samples,
                                                                   size_t count, int batch, int per_sample) {
    if (VEC) {
        __shared__ float4 red[JG == 1 ? 1 : 256];
        constexpr int IL = 256 / JG;
        const int il = threadIdx.x % IL, jg = threadIdx.x / IL;
        const size_t n4 = count / 4;
        for (size_t i0 = (size_t)blockIdx.x * IL; i0 < n4; i0 += (size_t)gridDim.x * IL) {
            const size_t i = i0 + il;
            float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int b = 0; b < batch; ++b) {
                const float* base = ws + (size_t)b * per_sample * count;
                const float4 acc = join_groups<JG>(i < n4 ? sum_parts4<JG>(base, count, i, per_sample, jg) : make_float4(0.f, 0.f, 0.f, 0.f), red, il, jg);
                if (jg == 0 && i < n4) reinterpret_cast<float4*>(samples + (size_t)b * count)[i] = acc;
                tot = add4(tot, acc);
            }
            if (jg == 0 && i < n4) reinterpret_cast<float4*>(dw)[i] = tot;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
            float tot = 0.f;
            for (int b = 0; b < batch; ++b) {
                float acc = 0.f;
                for (int j = 0; j < per_sample; ++j) acc += ws[((size_t)b * per_sample + j) * count + i];
                samples[(size_t)b * count + i] = acc;
                tot += acc;
            }
            dw[i] = tot;
        }
    }
}

// gc_wgrad_samples_contract_f32, first pass: one workgroup per (a, b); lane c sums w * s over the taps, writes colpart[b][a][c] when the
// C-side result is wanted, and the workgroup adds its lanes in a fixed tree: g_a[b][a] = sum_{t,c} w[t][a][c] s[b][t][a][c] / sa[b][a].
__global__ __launch_bounds__(256) void wgrad_contract_rows_kernel(const float* __restrict__ s, const float* __restrict__ w, const float* __restrict__ sa,
                                                                   float* __restrict__ g_a, float* __restrict__ colpart, int taps, int A, int C) {
    __shared__ float red[4];
    const int a = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t count = (size_t)taps * A * C;
    const float* sb = s + (size_t)b * count;
    float row = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float p = 0.f;
        for (int t = 0; t < taps; ++t) {
            const size_t e = ((size_t)t * A + a) * C + c;
            p = fmaf(w[e], sb[e], p);
        }
        if (colpart) colpart[((size_t)b * A + a) * C + c] = p;
        row += p;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) row += __shfl_down(row, o, 64);
    if (lane == 0) red[wave] = row;
    __syncthreads();
    if (threadIdx.x == 0 && g_a) {
        const float sum = (red[0] + red[1]) + (red[2] + red[3]);
        const float den = sa ? sa[(size_t)b * A + a] : 1.f;
        g_a[(size_t)b * A + a] = sum / (den == 0.f ? 1.f : den);        // zero-safe like rows_sum_div_kernel
    }
}

// second pass: g_c[b][c] = sum_a colpart[b][a][c] / sc[b][c]; 64 columns x 4 interleaved row groups per workgroup, groups added in fixed order
__global__ __launch_bounds__(256) void wgrad_contract_cols_kernel(const float* __restrict__ colpart, const float* __restrict__ sc, float* __restrict__ g_c, int A, int C) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6, b = blockIdx.y;
    const int c = blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (c < C)
        for (int a = part; a < A; a += 4) acc += colpart[((size_t)b * A + a) * C + c];
    red[part][lane] = acc;
    __syncthreads();
    if (part == 0 && c < C) {
        const float sum = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        const float den = sc ? sc[(size_t)b * C + c] : 1.f;
        g_c[(size_t)b * C + c] = sum / (den == 0.f ? 1.f : den);
    }
}

// --------------------------------------------------------------------------------------------
template <int WG_OC, int WG_PX, int KSPLIT, int KCT, int WOC, int WPX, int TPW, int UP, int DOWN, int KS>
int launch_conv(ConvArgs a, hipStream_t s) {
    using C = ConvCfg<WG_OC, WG_PX, KSPLIT, KCT, WOC, WPX, TPW, UP, DOWN, KS>;
    const int qh = gc::ceil_div(a.out_h, UP), qw = gc::ceil_div(a.out_w, UP);
    a.tiles_y = gc::ceil_div(qh, C::TPH);
    a.tiles_x = gc::ceil_div(qw, TPW);
    const long long gx = (long long)a.tiles_x * a.tiles_y * UP * UP * a.B;
    if (gx > 2147483647LL) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_f32: grid too large");
    const int slices = a.k_per_split ? gc::ceil_div(a.K, a.k_per_split) : 1;
    if (gc::probing()) return gc::probe_name("conv_mfma_kernel<%d,%d,%d,%d,%d,%d,%d>|up%d,down%d,k%d", WG_OC, WG_PX, KSPLIT, KCT, WOC, WPX, TPW, UP, DOWN, KS);
    dim3 grid((unsigned)gx, gc::ceil_div(a.N, C::OCT), slices);
    hipLaunchKernelGGL((conv_mfma_kernel<WG_OC, WG_PX, KSPLIT, KCT, WOC, WPX, TPW, UP, DOWN, KS>), grid, dim3(256), 0, s, a);
    return gc::check_launch("gc_conv2d_f32");
}

// Planes up to 16 pixels wide (the 4x4 .. 16x16 layers, 512 channels): a launch has only B * N / 32 workgroups, each
// walking all of K chunk by chunk with nothing to hide the load latency behind -- 70..120 us for a few GFLOP.  Splitting K
// over blockIdx.z fills the chip and shortens the dependent chain; the slices are summed in a fixed order by
// splitk_finish_kernel, which also applies out_scale and the fused epilogue.
struct SplitPlan { int slices, k_per_split; };
SplitPlan plan_splitk(const gc_conv_desc* d) {
    SplitPlan sp{1, 0};
    const int qw = gc::ceil_div(d->out_w, d->up), qh = gc::ceil_div(d->out_h, d->up);
    if (qw > 16 || d->in_ch < 128) return sp;
    const int tpw = qw <= 4 ? 4 : (qw <= 8 ? 8 : 16), rows = 32 / tpw;
    const long long wgs = (long long)gc::ceil_div(qw, tpw) * gc::ceil_div(qh, rows) * d->up * d->up * d->batch * gc::ceil_div(d->out_ch, 32);
    int want = (int)std::min<long long>(std::max<long long>(512 / std::max<long long>(wgs, 1), 1), d->in_ch / 64);
    if (want <= 1) return sp;
    sp.k_per_split = gc::ceil_div(gc::ceil_div(d->in_ch, want), 32) * 32;
    sp.slices = gc::ceil_div(d->in_ch, sp.k_per_split);
    if (sp.slices <= 1) { sp.slices = 1; sp.k_per_split = 0; }
    return sp;
}

__global__ __launch_bounds__(256) void splitk_finish_kernel(ConvArgs p, int slices, long long per_slice) {
    const EpilogueConsts ec = epilogue_consts(p);
    const long long plane = (long long)p.out_h * p.out_w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_slice; i += (long long)gridDim.x * 256) {
        float acc = 0.f;
        int z = 0;
        for (; z + 8 <= slices; z += 8) {        // eight loads in flight, added in slice order (the launch is latency-bound: <= 2 blocks per CU)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p.part[(z + u) * per_slice + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; z < slices; ++z) acc += p.part[z * per_slice + i];
        const long long bn = i / plane, o = i - bn * plane;
        const int n = (int)(bn % p.N), b = (int)(bn / p.N);
        const float so = p.so ? p.so[(size_t)b * p.N + n] : 1.f, bias = p.bias ? p.bias[n] : 0.f;
        const float nz = p.noise ? p.noise[(size_t)b * plane + o] : 0.f;
        p.y[i] = conv_epilogue(ec, acc, so, bias, nz) + (p.residual ? p.residual[i] : 0.f);
    }
}

// tile configuration: by phase sub-grid width, output channels and how many workgroups result
template <int UP, int DOWN, int KS>
int dispatch_conv(const ConvArgs& a, hipStream_t s) {
    const int qw = gc::ceil_div(a.out_w, UP), qh = gc::ceil_div(a.out_h, UP);
    if (qw <= 4) return launch_conv<1, 1, 4, 32, 1, 1, 4, UP, DOWN, KS>(a, s);
    if (qw <= 8) return launch_conv<1, 1, 4, 32, 1, 1, 8, UP, DOWN, KS>(a, s);
    if (qw <= 16) return launch_conv<1, 1, 4, 32, 1, 1, 16, UP, DOWN, KS>(a, s);
    if constexpr (DOWN == 2) {
        if (a.N <= 64) return launch_conv<2, 2, 1, 8, 1, 1, 32, UP, DOWN, KS>(a, s);
    } else {
        if (a.N <= 32) return launch_conv<1, 4, 1, 8, 1, 4, 32, UP, DOWN, KS>(a, s);
        if (a.N <= 64) return launch_conv<1, 4, 1, 8, 2, 2, 32, UP, DOWN, KS>(a, s);
    }
    // 128oc x (4 rows x 32 px) tiles unless that leaves most CUs idle
    const long long big = (long long)gc::ceil_div(qw, 32) * gc::ceil_div(qh, 4) * UP * UP * a.B * gc::ceil_div(a.N, 128);
    if (big < 512) return launch_conv<2, 2, 1, 8, 1, 1, 32, UP, DOWN, KS>(a, s);
    return launch_conv<2, 2, 1, 8, 2, 2, 32, UP, DOWN, KS>(a, s);
}

struct WgradPlan { int cfg, kt, nt, tr, splits, tiles_per_split, tiles_x, tiles_y, parts; };

WgradPlan plan_wgrad(const gc_conv_desc* d) {
    WgradPlan pl;
    // cfg 0: 64k x 64n per workgroup; 1: 32k x 64n, 2: 64k x 32n (two pixel-waves); 3: 32k x 32n (four pixel-waves)
    const bool ksmall = d->in_ch <= 32, nsmall = d->out_ch <= 32;
    pl.cfg = (ksmall && nsmall) ? 3 : (ksmall ? 1 : (nsmall ? 2 : 0));
    pl.kt = (pl.cfg == 1 || pl.cfg == 3) ? 32 : 64;
    pl.nt = (pl.cfg == 2 || pl.cfg == 3) ? 32 : 64;
    const bool narrow = d->down == 2 || d->kh == 3;      // register budget: 144 accumulators + prefetch in 256 VGPRs
    pl.tr = pl.cfg == 0 ? (narrow ? 1 : 2) : (pl.cfg == 3 ? (d->down == 2 ? 4 : 4) : 2);
    pl.tiles_x = gc::ceil_div(d->out_w, 32);
    pl.tiles_y = gc::ceil_div(d->out_h, pl.tr);
    const int total = pl.tiles_x * pl.tiles_y * d->batch;
    const int ctiles = gc::ceil_div(d->in_ch, pl.kt) * gc::ceil_div(d->out_ch, pl.nt);
    int want = gc::ceil_div(512, ctiles);           // ~2 workgroups per CU (2 waves/SIMD at ~200 VGPRs) over the whole grid
    if (want > total) want = total;
    if (want < 1) want = 1;
    pl.tiles_per_split = gc::ceil_div(total, want);
    pl.splits = gc::ceil_div(total, pl.tiles_per_split);
    pl.parts = pl.splits;                            // pixel-waves are summed in LDS inside the kernel
    return pl;
}

// --------------------------------------------------------------------------------------------
// Output planes of <= 8 x 8 pixels (the 4^2 / 8^2 layers of G and D, D's 17 -> 8 and 9 -> 4 down-sampling convolutions, 513 -> 512 of D's
// last block: 512 channels, 9.4 MB of weights for 0.3 .. 2.4 GFLOP): the launch is bound by reading the weights once, by the partial
// sums and by latency, not by arithmetic -- so these layers stay in EXACT fp32 in every mode (v_mfma_f32_32x32x2_f32; their sums run
// over few pixels, where rounding errors do not average out).  One workgroup = ONE weight slab applied to every pixel of every sample,
// either 16 input channels x 64 output channels <WOC 2, NCH 1> or 32 x 32 <WOC 1, NCH 2>: 256 workgroups for a 512 -> 512 layer both
// ways, the second with half as many K slices to write and add up again (it needs twice the patch in LDS and is chosen when the partial
// sums, not the weights, are the larger traffic).  The zero-haloed planes of ALL samples are staged once, scaled by in_scale; eight
// waves take (32 pixels x 32 output channels) work items; each workgroup writes raw partial sums of its channels to part[slice], and
// splitk_finish_kernel adds the slices in fixed order and applies out_scale + the fused epilogue.
// conv_mfma_kernel with its split over K took 27 .. 65 us on these shapes (92 without a workspace); this one 10 .. 30 us.
struct SmallArgs { ConvArgs c; float* part; long long per_slice; int bgroup; };     // bgroup: samples per workgroup (blockIdx.z walks the groups)
constexpr int SMALL_THREADS = 512;       // eight waves: two per SIMD hide the staging latency
constexpr int SMALL_KC = 16;             // input channels per chunk

// UP = 2 (round 5: the 4^2 -> 9^2 transposed convolutions -- G's first up-sampling layer, the input gradient of D's 9 -> 4 convolution -- ran on
// conv_mfma_kernel at 65 / 113 us for B = 4 / 8): the staged plane is the ZERO-STUFFED input with its halo, (out + KS - 1)^2 positions, and the
// product loop is the stride-1 one; three quarters of its MFMAs multiply zeros, which costs nothing where the launch is bound by latency.
template <int KS, int WOC, int NCH, int DOWN, int UP = 1>
__global__ __launch_bounds__(SMALL_THREADS) void conv_f32_small_kernel(SmallArgs a) {
    constexpr int NTAP = KS * KS, OCT = 32 * WOC, NW = SMALL_THREADS / 64, KC = SMALL_KC;
    constexpr int WFLOATS = NCH * NTAP * KC * OCT;               // [chunk][tap][k][oc]
    static_assert(UP == 1 || DOWN == 1, "up and down are exclusive");
    extern __shared__ float small_smem[];
    const ConvArgs& p = a.c;
    const int halo = p.pad_y;                                    // = pad_x: KS / 2 at stride 1, 0 at stride 2
    const int ph = UP == 2 ? p.out_h + KS - 1 : p.in_h + 2 * halo, pw = UP == 2 ? p.out_w + KS - 1 : p.in_w + 2 * halo, plane = ph * pw;      // zero-haloed (UP = 2: zero-stuffed) plane of one sample
    const int b0 = blockIdx.z * a.bgroup, nb = min(a.bgroup, p.B - b0);       // this workgroup's samples
    const int per_k = nb * plane + 1;                            // + one zero for the lanes past the last pixel
    float* wl = small_smem;
    float* pl = wl + WFLOATS;                                    // [chunk * 16 + k][sample][ph][pw] (+ zero)
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k0 = blockIdx.x * KC * NCH, n0 = blockIdx.y * OCT;
    // weights: w[tap][k][n], rows of OCT consecutive output channels; zeros past K / N (513 channels: a ragged last chunk)
    constexpr int NWL = (WFLOATS + SMALL_THREADS - 1) / SMALL_THREADS;
    float wreg[NWL];
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
        const int e = tid + SMALL_THREADS * j;
        const int oc = e % OCT, row = e / OCT;                   // row = (chunk * NTAP + tap) * KC + k
        const int kk = row % KC, t = row / KC % NTAP, ch = row / (KC * NTAP);
        const int k = k0 + ch * KC + kk, n = n0 + oc;
        const bool ok = e < WFLOATS && k < p.K && n < p.N;
        wreg[j] = ok ? p.w[((size_t)t * p.K + k) * p.N + n] : 0.f;
    }
    // patch: every sample's plane with a zero halo, times in_scale.  One item = one position of the haloed planes x 8 channels (one
    // decode of the position, eight loads in flight, eight conflict-free LDS stores: consecutive lanes hold consecutive positions)
    const int chan = p.in_h * p.in_w, npos = nb * plane, nitems = npos * (NCH * KC / 8);
    for (int it = tid; it < nitems; it += SMALL_THREADS) {
        const int kg = it / npos, pos = it - kg * npos;
        const int b = pos / plane, q0 = pos - b * plane, yy = q0 / pw - halo, xx = q0 % pw - halo;
        const bool inside = UP == 2 ? (yy >= 0 && xx >= 0 && ((yy | xx) & 1) == 0 && (yy >> 1) < p.in_h && (xx >> 1) < p.in_w)
                                    : (yy >= 0 && yy < p.in_h && xx >= 0 && xx < p.in_w);
        const int kb = k0 + kg * 8;
        const int at = UP == 2 ? (yy >> 1) * p.in_w + (xx >> 1) : yy * p.in_w + xx;
        const float* src = p.x + (inside ? (size_t)(b0 + b) * p.K * chan + at : (size_t)0);
        const float* ssrc = p.si + (size_t)(b0 + b) * p.K;
        float v[8], sc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = inside ? min(kb + q, p.K - 1) : 0;      // branch-free: positions in the halo load the tensor's first value and drop it
            v[q] = src[(size_t)k * chan];
            sc[q] = p.si ? ssrc[min(kb + q, p.K - 1)] : 1.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) pl[(kg * 8 + q) * per_k + pos] = (inside && kb + q < p.K) ? v[q] * sc[q] : 0.f;
    }
    if (tid < NCH * KC) pl[tid * per_k + npos] = 0.f;          // the zero the lanes past the last pixel read
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
        const int e = tid + SMALL_THREADS * j;
        if (SMALL_THREADS * (j + 1) <= WFLOATS || e < WFLOATS) wl[e] = wreg[j];
    }
    __syncthreads();
    // one work item = 32 pixels x 32 output channels; the waves take them round-robin
    const int oplane = p.out_h * p.out_w, pixels = nb * oplane;
    const int items = (pixels + 31) / 32 * WOC;
    float* out = a.part + (size_t)blockIdx.x * a.per_slice + (size_t)b0 * p.N * oplane;
    for (int item = wave; item < items; item += NW) {
        const int cb = item / WOC, i = item % WOC;
        const int pix = cb * 32 + l31;
        const bool live = pix < pixels;
        const int b = live ? pix / oplane : 0, o = live ? pix - b * oplane : 0;
        const int oy = o / p.out_w, ox = o - oy * p.out_w;
        // this lane's pixel under tap (0, 0); the lanes past the last pixel read the zero at the end of their channel's row
        const int base = hi * per_k + (live ? b * plane + oy * DOWN * pw + ox * DOWN : nb * plane);
        const int wbase = hi * OCT + i * 32 + l31;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int ty = t / KS, tx = t % KS;
                const float* pb = pl + base + (size_t)ch * KC * per_k + (live ? ty * pw + tx : 0);
                const float* wa = wl + (ch * NTAP + t) * KC * OCT + wbase;
#pragma unroll
                for (int kp = 0; kp < KC / 2; ++kp)       // lanes 0..31 take channel 2 kp, lanes 32..63 channel 2 kp + 1
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[2 * kp * OCT], pb[2 * kp * per_k], acc, 0, 0, 0);
            }
        if (live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (n < p.N) out[((size_t)b * p.N + n) * oplane + o] = acc[r];
            }
        }
    }
}

// LDS of the small-plane kernel with `nch` channel chunks per workgroup and `bg` samples: the weight slab + the zero-haloed planes of those samples
inline size_t small_plane(const gc_conv_desc* d) {
    if (d->up == 2) return (size_t)(d->out_h + d->kh - 1) * (d->out_w + d->kw - 1);
    return (size_t)(d->in_h + 2 * d->pad_y) * (d->in_w + 2 * d->pad_x);
}
inline size_t small_lds_bytes(const gc_conv_desc* d, int nch, int bg) {
    return ((size_t)d->kh * d->kw * SMALL_KC * 64 + (size_t)nch * SMALL_KC * ((size_t)bg * small_plane(d) + 1)) * sizeof(float);
}
// what one workgroup may use: the device's per-block limit (160 KiB on gfx950; a part with less sends these shapes to conv_mfma_kernel's split over K)
inline size_t small_lds_max() { return std::min<size_t>(160 * 1024, gc::device_lds_limit()); }
constexpr int SMALL_MAX_PIXELS = 512;    // output pixels of one workgroup's samples (16 work items per wave and output-channel block)

// Samples per workgroup: all of them when their planes fit the LDS and SMALL_MAX_PIXELS, else halved until they do (round 5: D's 17 -> 8
// convolution at B = 8 needs 148 KB of planes -- it ran on conv_mfma_kernel at 138 us next to 34 us at B = 4).  0: not even one sample fits.
inline int small_bgroup(const gc_conv_desc* d) {
    int bg = d->batch;
    while (bg > 1 && (small_lds_bytes(d, 1, bg) > small_lds_max() || (long long)bg * d->out_h * d->out_w > SMALL_MAX_PIXELS)) bg = (bg + 1) / 2;
    if (bg < 1 || small_lds_bytes(d, 1, bg) > small_lds_max() || (long long)bg * d->out_h * d->out_w > SMALL_MAX_PIXELS) return 0;
    return bg;
}

// shapes the small-plane kernel takes: 1x1 / 3x3 taps at stride 1 with "same" padding or at stride 2 without padding, output planes <= 8 x 8;
// 3x3 taps with up = 2 onto planes <= 10 x 10 (4^2 -> 9^2); >= 64 input and output channels (any count), dense rows, and a patch that fits the
// LDS next to the weight slab (many samples of tiny planes are split into sample groups: their halo is most of the patch)
inline bool small_eligible(const gc_conv_desc* d) {
#ifdef GC_NO_SMALL
    return false;
#endif
    if (d->kh != d->kw || (d->kh != 1 && d->kh != 3) || d->pad_y != d->pad_x || d->batch < 1) return false;
    if (d->up == 2) {
#ifdef GC_NO_SMALL_UP
        return false;
#endif
        if (d->down != 1 || d->kh != 3 || d->pad_y < 0 || d->out_w > 10 || d->out_h > 10) return false;
    } else if (d->up == 1) {
        if (d->down == 1) { if (d->pad_y != d->kh / 2 || d->out_h != d->in_h || d->out_w != d->in_w) return false; }
        else if (d->down == 2) { if (d->pad_y != 0 || d->in_h < d->kh || d->in_w < d->kw || d->out_h != (d->in_h - d->kh) / 2 + 1 || d->out_w != (d->in_w - d->kw) / 2 + 1) return false; }
        else return false;
        if (d->out_w > 8 || d->out_h > 8) return false;
    } else return false;
    if ((d->in_pitch != 0 && d->in_pitch != d->in_w) || !dense_output(d)) return false;
    if (d->in_ch < 64 || d->out_ch < 64) return false;
    return small_bgroup(d) >= 1;
}

// channel chunks per workgroup: two (32 channels x 32 output channels) when the partial sums of 16-channel slices would outweigh the weights
inline int small_chunks(const gc_conv_desc* d) {
    const int bg = small_bgroup(d);
    if (d->in_ch % (2 * SMALL_KC) != 0 || small_lds_bytes(d, 2, bg) > small_lds_max()) return 1;
#ifdef GC_SMALL_NCH
    return GC_SMALL_NCH;
#endif
    const long long pixels = (long long)d->batch * d->out_h * d->out_w;
    return pixels > 16 * d->kh * d->kw ? 2 : 1;     // slices * pixels * N * 4 bytes  vs  taps * K * N * 4 bytes
}
inline int small_slices(const gc_conv_desc* d) { return gc::ceil_div(d->in_ch, SMALL_KC * small_chunks(d)); }

template <int KS, int DOWN, int UP = 1>
int launch_small(const gc_conv_desc* d, const SmallArgs& sa, hipStream_t s) {
    const size_t lds = small_lds_bytes(d, small_chunks(d), sa.bgroup);
    const unsigned groups = (unsigned)gc::ceil_div(d->batch, sa.bgroup);
    if (small_chunks(d) == 2) {
        static bool done[16] = {false};
        if (int rc = gc::allow_dynamic_lds(reinterpret_cast<const void*>(&conv_f32_small_kernel<KS, 1, 2, DOWN, UP>), small_lds_max(), done, "gc_conv2d_f32(small planes)")) return rc;
        hipLaunchKernelGGL((conv_f32_small_kernel<KS, 1, 2, DOWN, UP>), dim3(small_slices(d), gc::ceil_div(d->out_ch, 32), groups), dim3(SMALL_THREADS), lds, s, sa);
    } else {
        static bool done[16] = {false};
        if (int rc = gc::allow_dynamic_lds(reinterpret_cast<const void*>(&conv_f32_small_kernel<KS, 2, 1, DOWN, UP>), small_lds_max(), done, "gc_conv2d_f32(small planes)")) return rc;
        hipLaunchKernelGGL((conv_f32_small_kernel<KS, 2, 1, DOWN, UP>), dim3(small_slices(d), gc::ceil_div(d->out_ch, 64), groups), dim3(SMALL_THREADS), lds, s, sa);
    }
    return gc::check_launch("gc_conv2d_f32(small planes)");
}

// --------------------------------------------------------------------------------------------
// Weight gradient of a 3 x 3 convolution onto planes <= 8 x 8 (round 5; the 4^2 / 8^2 layers of D and its 9 -> 4 / 17 -> 8 down-sampling
// convolutions): the pixel-tile kernels fill 4 .. 8 of a tile's 32 columns, cut the pixels into eight splits to have workgroups at all and add
// the splits up again in a second launch -- 40 .. 60 us for 0.3 .. 2.4 GFLOP.  Here ONE workgroup owns a (32 k x 32 n) block of all nine taps
// and the pixels of ALL samples are the contraction index of v_mfma_f32_32x32x2_f32 (exact fp32, like the forward kernel of these planes):
// the zero-haloed input planes of its 32 input channels (x in_scale) and the gradient planes of its 32 output channels (x out_scale) are
// staged once per group of samples that fits the LDS, wave t multiplies tap t (nine waves), and the block is written once -- no workspace,
// no second launch, sums in a fixed order.
struct WgSmallArgs { WgradArgs a; int bgroup, xrow, yrow; };     // samples per staged group; floats per channel row of the two LDS arrays (odd: conflict-free columns)
constexpr int WGS_THREADS = 576;

// J: 64-lane chunks of one zero-haloed input plane (1, 2 or 5: planes of <= 64, 128, 320 positions)
template <int DOWN, int J>
__global__ __launch_bounds__(WGS_THREADS) void wgrad_f32_small_kernel(WgSmallArgs q) {
    constexpr int KS = 3, NWV = WGS_THREADS / 64, KW = (32 + NWV - 1) / NWV, BU = 4;      // KW channel rows per wave, BU samples per staging round
    extern __shared__ float wgs_smem[];
    const WgradArgs& p = q.a;
    const int pad = p.pad_y;                                              // 1 at stride 1, 0 at stride 2 (= pad_x)
    const int ph = p.in_h + 2 * pad, pw = p.in_w + 2 * pad, plane = ph * pw, oplane = p.out_h * p.out_w;
    float* xs = wgs_smem;                                                 // [32 k][bgroup * plane | 2 pw + 3 zeros (what a padding pixel reads under any tap) | pad to xrow]
    float* ys = xs + 32 * q.xrow;                                         // [32 n][bgroup * oplane rounded up to 16 pixels with zeros | pad to yrow]
    int* postab = reinterpret_cast<int*>(ys + 32 * q.yrow);               // pixel of the group -> its position under tap (0, 0)
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // = the tap
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int ty = wave / KS, tx = wave % KS, tapoff = ty * pw + tx;
    const int chan = p.in_h * p.in_w;
    // this lane's positions of a haloed plane -> offset inside the input plane (-1: halo or past the plane); the only divisions of the staging
    int xo[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int r0 = lane + 64 * j, yy = r0 / pw - pad, xx = r0 % pw - pad;
        xo[j] = (r0 < plane && yy >= 0 && yy < p.in_h && xx >= 0 && xx < p.in_w) ? yy * p.in_w + xx : -1;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int b0 = 0; b0 < p.B; b0 += q.bgroup) {
        const int nb = min(q.bgroup, p.B - b0), npos = nb * plane, npix = nb * oplane, npix2 = (npix + 15) & ~15;
        if (b0 > 0) __syncthreads();                                      // every wave has read the previous group
        // Staging: wave w owns channel rows w, w + 9, ... of BOTH arrays; one round = those rows x four samples, every load of the round in flight
        // before its first LDS store (a load per trip of a flat loop was one exposed memory latency -- and five integer divisions -- per element)
        for (int bb = 0; bb < nb; bb += BU) {
            float xv[KW][BU][J], yv[KW][BU], sx[KW][BU], sy[KW][BU];
#pragma unroll
            for (int i = 0; i < KW; ++i) {
                const int c = wave + NWV * i;
#pragma unroll
                for (int u = 0; u < BU; ++u) {
                    const int b = bb + u;
                    const bool okx = c < 32 && b < nb && k0 + c < p.K, oky = c < 32 && b < nb && n0 + c < p.N;
                    const size_t rowx = okx ? (size_t)(b0 + b) * p.K + k0 + c : 0, rowy = oky ? (size_t)(b0 + b) * p.N + n0 + c : 0;
#pragma unroll
                    for (int j = 0; j < J; ++j) xv[i][u][j] = (okx && xo[j] >= 0) ? p.x[rowx * chan + xo[j]] : 0.f;
                    yv[i][u] = (oky && lane < oplane) ? p.dy[rowy * oplane + lane] : 0.f;
                    sx[i][u] = (okx && p.si) ? p.si[rowx] : 1.f;
                    sy[i][u] = (oky && p.so) ? p.so[rowy] : 1.f;
                }
            }
#pragma unroll
            for (int i = 0; i < KW; ++i) {
                const int c = wave + NWV * i;
#pragma unroll
                for (int u = 0; u < BU; ++u) {
                    const int b = bb + u;
                    if (c < 32 && b < nb) {
#pragma unroll
                        for (int j = 0; j < J; ++j)
                            if (lane + 64 * j < plane) xs[c * q.xrow + b * plane + lane + 64 * j] = xv[i][u][j] * sx[i][u];
                        if (lane < oplane) ys[c * q.yrow + b * oplane + lane] = yv[i][u] * sy[i][u];
                    }
                }
            }
        }
        for (int e = tid; e < 32 * (2 * pw + 3); e += WGS_THREADS) xs[(e & 31) * q.xrow + npos + (e >> 5)] = 0.f;       // what the padding pixels read
        for (int e = tid; e < 32 * (npix2 - npix); e += WGS_THREADS) ys[(e & 31) * q.yrow + npix + (e >> 5)] = 0.f;
        for (int px = tid; px < npix2; px += WGS_THREADS) {
            int pos = npos;
            if (px < npix) {
                const int b = px / oplane, o = px - b * oplane, oy = o / p.out_w, ox = o - oy * p.out_w;
                pos = b * plane + oy * DOWN * pw + ox * DOWN;
            }
            postab[px] = pos;
        }
        __syncthreads();
        const float* xa = xs + l31 * q.xrow + tapoff;
        const float* yb = ys + l31 * q.yrow;
        // eight products per trip, their LDS reads issued together (position table, then the values): as a plain loop every product waited for
        // three dependent LDS round trips.  lanes 0..31 take pixel 2 j, lanes 32..63 pixel 2 j + 1
        for (int j0 = 0; j0 < npix2 / 2; j0 += 8) {
            int pos[8];
            float av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) pos[u] = postab[2 * (j0 + u) + hi];
#pragma unroll
            for (int u = 0; u < 8; ++u) { av[u] = xa[pos[u]]; bv[u] = yb[2 * (j0 + u) + hi]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
    }
    float* out = p.ws + ((size_t)wave * p.K + k0) * p.N + n0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (k0 + k < p.K && n0 + l31 < p.N) out[(size_t)k * p.N + l31] = acc[r];
    }
}

inline int wgs_odd(int v) { return v | 1; }
inline size_t wgs_lds_bytes(const gc_conv_desc* d, int bg) {
    const int pad = d->pad_y, plane = (d->in_h + 2 * pad) * (d->in_w + 2 * pad), oplane = d->out_h * d->out_w;
    const int npix2 = (bg * oplane + 15) & ~15, zeros = 2 * (d->in_w + 2 * pad) + 3;
    return ((size_t)32 * wgs_odd(bg * plane + zeros) + (size_t)32 * wgs_odd(npix2) + npix2) * sizeof(float);
}
inline int wgs_bgroup(const gc_conv_desc* d) {
    int bg = d->batch;
    while (bg > 1 && wgs_lds_bytes(d, bg) > small_lds_max()) bg = (bg + 1) / 2;
    return wgs_lds_bytes(d, bg) <= small_lds_max() ? bg : 0;
}

int launch_wgrad_small(const gc_conv_desc* d, const float* x, const float* dy, const float* in_scale, const float* out_scale, float* dw, hipStream_t s) {
    const int bg = wgs_bgroup(d), pad = d->pad_y, plane = (d->in_h + 2 * pad) * (d->in_w + 2 * pad), oplane = d->out_h * d->out_w;
    WgSmallArgs q{{x, dy, in_scale, out_scale, dw, d->batch, d->in_ch, d->out_ch, d->in_h, d->in_w, d->out_h, d->out_w, d->pad_y, d->pad_x, 0, 0, 0},
                  bg, wgs_odd(bg * plane + 2 * (d->in_w + 2 * pad) + 3), wgs_odd((bg * oplane + 15) & ~15)};
    const size_t lds = wgs_lds_bytes(d, bg);
    const dim3 grid(gc::ceil_div(d->in_ch, 32), gc::ceil_div(d->out_ch, 32));
    const int chunks = plane <= 64 ? 1 : (plane <= 128 ? 2 : 5);
    static bool done[6][16] = {{false}};
#define GC_WGS_LAUNCH(DOWN_, J_, SLOT_)                                                                                                                   \
    do {                                                                                                                                                \
        if (int rc = gc::allow_dynamic_lds(reinterpret_cast<const void*>(&wgrad_f32_small_kernel<DOWN_, J_>), small_lds_max(), done[SLOT_], "gc_conv2d_wgrad_f32(small planes)")) return rc; \
        hipLaunchKernelGGL((wgrad_f32_small_kernel<DOWN_, J_>), grid, dim3(WGS_THREADS), lds, s, q);                                                    \
    } while (0)
    if (d->down == 2) { if (chunks == 1) GC_WGS_LAUNCH(2, 1, 0); else if (chunks == 2) GC_WGS_LAUNCH(2, 2, 1); else GC_WGS_LAUNCH(2, 5, 2); }
    else              { if (chunks == 1) GC_WGS_LAUNCH(1, 1, 3); else if (chunks == 2) GC_WGS_LAUNCH(1, 2, 4); else GC_WGS_LAUNCH(1, 5, 5); }
#undef GC_WGS_LAUNCH
    return gc::check_launch("gc_conv2d_wgrad_f32(small planes)");
}

template <int DOWN, int KS>
int dispatch_wgrad(const WgradArgs& a, const WgradPlan& pl, hipStream_t s) {
    dim3 grid(gc::ceil_div(a.K, pl.kt), gc::ceil_div(a.N, pl.nt), pl.splits);
    constexpr bool narrow = DOWN == 2 || KS == 3;
    switch (pl.cfg) {
        case 0: hipLaunchKernelGGL((wgrad_mfma_kernel<2, 2, 1, (narrow ? 1 : 2), DOWN, KS>), grid, dim3(256), 0, s, a); break;
        case 1: hipLaunchKernelGGL((wgrad_mfma_kernel<1, 2, 2, 2, DOWN, KS>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((wgrad_mfma_kernel<2, 1, 2, 2, DOWN, KS>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((wgrad_mfma_kernel<1, 1, 4, 4, DOWN, KS>), grid, dim3(256), 0, s, a); break;
    }
    return gc::check_launch("gc_conv2d_wgrad_f32(mfma)");
}

}  // namespace

// 3 x 3 weight gradients the small-plane kernel takes: stride 1 with "same" padding or stride 2 without padding onto planes <= 8 x 8, dense rows,
// >= 64 channels on both sides (fewer leave the launch a handful of workgroups), at most 2048 pixels over the batch, the batch in at most two LDS-sized
// groups of samples.  Measured (512 -> 512, same box, pixel-tile kernels + reduce -> this kernel, profiles/tail_wg_ab_r05.log): @4^2 B = 2 / 4 / 8
// 32 / 41 / 47 -> 12 / 15 / 24 us; @8^2 40 / 46 / 56 -> 19 / 27 / 47; 9 -> 4: 39 / 42 / 48 -> 13 / 16 / 26; 17 -> 8: 42 / 47 -> 22 / 37 (B = 8 stays: 4 groups)
bool gcconv::wgrad_small_eligible(const gc_conv_desc* d) {
#ifdef GC_NO_SMALL_WGRAD
    return false;
#endif
    if (d->up != 1 || d->kh != 3 || d->kw != 3 || d->pad_y != d->pad_x || d->batch < 1) return false;
    if (d->down == 1) { if (d->pad_y != 1 || d->out_h != d->in_h || d->out_w != d->in_w) return false; }
    else if (d->down == 2) { if (d->pad_y != 0 || d->in_h < 3 || d->in_w < 3 || d->out_h != (d->in_h - 3) / 2 + 1 || d->out_w != (d->in_w - 3) / 2 + 1) return false; }
    else return false;
    if (d->in_pitch != 0 && d->in_pitch != d->in_w) return false;
    if (d->in_ch < 64 || d->out_ch < 64 || d->out_w > 8 || d->out_h > 8) return false;
    if ((long long)d->batch * d->out_h * d->out_w > 2048 || (d->in_h + 2 * d->pad_y) * (d->in_w + 2 * d->pad_x) > 320) return false;
    const int bg = wgs_bgroup(d);
    return bg >= 1 && gc::ceil_div(d->batch, bg) <= 2;       // each staged group costs ~12 us of latencies: with more than two the pixel-tile kernels win (512 -> 512, 17 -> 8, B = 8: 68 vs 59 us)
}

size_t gcconv::conv2d_f32_workspace(const gc_conv_desc* d) {
    if (!d || d->batch <= 0 || d->in_ch <= 0 || d->out_ch <= 0 || d->out_h <= 0 || d->out_w <= 0 || d->up <= 0) return 0;
    if (small_eligible(d)) return (size_t)small_slices(d) * d->batch * d->out_ch * d->out_h * d->out_w * sizeof(float);      // one slice per workgroup row
    const SplitPlan sp = plan_splitk(d);
    return sp.slices > 1 ? (size_t)sp.slices * d->batch * d->out_ch * d->out_h * d->out_w * sizeof(float) : 0;
}

int gcconv::conv2d_f32_ws(const gc_conv_desc* d, const float* x, const float* w, const float* in_scale, const float* out_scale,
                          const gc_conv_epilogue* ep, float* y, void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    int rc = validate(d, "gc_conv2d_f32", false);
    if (rc) return rc;
    if (!dense_output(d)) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_f32: out_pitch %d: the fp32 kernels write dense rows (gc_conv2d_out_pitch)", d->out_pitch);
    if (d->in_pitch != 0 && d->in_pitch != d->in_w) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_f32: in_pitch %d: the fp32 kernels read dense rows", d->in_pitch);
    if (!x || !w || !y) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_f32: null pointer");
    if ((rc = validate_epilogue(ep, "gc_conv2d_f32"))) return rc;
    if (d->batch == 0) return GC_OK;
    if (pointwise_thin(d)) return pointwise_conv(d, x, w, in_scale, out_scale, ep, y, stream);
    ConvArgs a{x, w, in_scale, out_scale, y, d->batch, d->in_ch, d->out_ch, d->in_h, d->in_w, d->out_h, d->out_w,
               d->pad_y, d->pad_x, 0, 0};
    set_epilogue(a, ep);
    a.k_per_split = 0; a.part = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const size_t need = conv2d_f32_workspace(d);
    if (small_eligible(d) && (gc::probing() || (workspace && workspace_bytes >= need))) {
        if (gc::probing()) return gc::probe_name("conv_f32_small_kernel<%d,%d,%d,%d>|up%d,down%d,k%d", d->kh, small_chunks(d) == 2 ? 1 : 2, small_chunks(d), d->down, d->up, d->down, d->kh);
        SmallArgs sa{a, static_cast<float*>(workspace), (long long)d->batch * d->out_ch * d->out_h * d->out_w, small_bgroup(d)};
        if (d->up == 2)        rc = launch_small<3, 1, 2>(d, sa, s);
        else if (d->down == 2) rc = d->kh == 3 ? launch_small<3, 2>(d, sa, s) : launch_small<1, 2>(d, sa, s);
        else                   rc = d->kh == 3 ? launch_small<3, 1>(d, sa, s) : launch_small<1, 1>(d, sa, s);
        if (rc) return rc;
        ConvArgs fin_s = a;
        fin_s.part = sa.part;
        return launch_splitk_finish(fin_s, small_slices(d), sa.per_slice, s);
    }
    const SplitPlan sp = plan_splitk(d);
    const bool split = sp.slices > 1 && workspace && workspace_bytes >= need;
    ConvArgs fin = a;
    if (split) {        // raw partial sums now, out_scale + epilogue in the finish pass
        a.so = nullptr;
        set_epilogue(a, nullptr);
        a.k_per_split = sp.k_per_split;
        a.part = static_cast<float*>(workspace);
    }
    if (d->kh == 3) {
        if (d->up == 2) rc = dispatch_conv<2, 1, 3>(a, s);
        else rc = d->down == 2 ? dispatch_conv<1, 2, 3>(a, s) : dispatch_conv<1, 1, 3>(a, s);
    } else {
        if (d->up == 2) rc = dispatch_conv<2, 1, 1>(a, s);
        else rc = d->down == 2 ? dispatch_conv<1, 2, 1>(a, s) : dispatch_conv<1, 1, 1>(a, s);
    }
    if (rc || !split) return rc;
    fin.part = static_cast<float*>(workspace);
    return launch_splitk_finish(fin, sp.slices, (long long)d->batch * d->out_ch * d->out_h * d->out_w, s);
}

int gcconv::launch_splitk_finish(const ConvArgs& fin, int slices, long long per_slice, hipStream_t s) {
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)std::min<long long>((per_slice + 255) / 256, 2048)), dim3(256), 0, s, fin, slices, per_slice);
    return gc::check_launch("gc_conv2d_f32(split-K finish)");
}

extern "C" int gc_conv2d_fused_f32(const gc_conv_desc* d, const float* x, const float* w,
                                   const float* in_scale, const float* out_scale, const gc_conv_epilogue* ep,
                                   float* y, gc_stream_t stream) {
    return gcconv::conv2d_f32_ws(d, x, w, in_scale, out_scale, ep, y, nullptr, 0, stream);
}

extern "C" size_t gc_conv2d_f32_workspace(const gc_conv_desc* d) { return gcconv::conv2d_f32_workspace(d); }

extern "C" int gc_conv2d_fused_f32_ws(const gc_conv_desc* d, const float* x, const float* w,
                                      const float* in_scale, const float* out_scale, const gc_conv_epilogue* ep,
                                      float* y, void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    return gcconv::conv2d_f32_ws(d, x, w, in_scale, out_scale, ep, y, workspace, workspace_bytes, stream);
}

extern "C" int gc_conv2d_f32(const gc_conv_desc* d, const float* x, const float* w,
                             const float* in_scale, const float* out_scale, float* y, gc_stream_t stream) {
    return gc_conv2d_fused_f32(d, x, w, in_scale, out_scale, nullptr, y, stream);
}

extern "C" size_t gc_conv2d_wgrad_workspace(const gc_conv_desc* d) {
    if (!d || d->batch <= 0 || d->in_ch <= 0 || d->out_ch <= 0 || d->out_h <= 0 || d->out_w <= 0) return 0;
    if (pointwise_thin_wgrad(d)) return pointwise_wgrad_workspace(d);
    const WgradPlan pl = plan_wgrad(d);
    return (size_t)pl.parts * d->kh * d->kw * d->in_ch * d->out_ch * sizeof(float);
}

extern "C" int gc_conv2d_wgrad_f32(const gc_conv_desc* d, const float* x, const float* dy,
                                   const float* in_scale, const float* out_scale, float* dw,
                                   void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    int rc = validate(d, "gc_conv2d_wgrad_f32", true);
    if (rc) return rc;
    if (d->in_pitch != 0 && d->in_pitch != d->in_w) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_wgrad_f32: in_pitch %d: the fp32 kernels read dense rows", d->in_pitch);
    if (!x || !dy || !dw) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_wgrad_f32: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t count = (size_t)d->kh * d->kw * d->in_ch * d->out_ch;
    if (d->batch == 0) {
        hipError_t e = hipMemsetAsync(dw, 0, count * sizeof(float), s);
        return e == hipSuccess ? GC_OK : gc::fail(GC_ERR_HIP, "gc_conv2d_wgrad_f32: memset: %s", hipGetErrorString(e));
    }
    if (wgrad_small_eligible(d)) return launch_wgrad_small(d, x, dy, in_scale, out_scale, dw, s);        // one launch, no workspace
    const WgradPlan pl = plan_wgrad(d);
    const size_t need = gc_conv2d_wgrad_workspace(d);
    if (!workspace || workspace_bytes < need) return gc::fail(GC_ERR_WORKSPACE, "gc_conv2d_wgrad_f32: workspace %zu < %zu bytes", workspace_bytes, need);
    if (pointwise_thin_wgrad(d)) return pointwise_wgrad(d, x, dy, in_scale, out_scale, dw, nullptr, workspace, stream);
    WgradArgs a{x, dy, in_scale, out_scale, pl.parts == 1 ? dw : static_cast<float*>(workspace), d->batch, d->in_ch, d->out_ch, d->in_h, d->in_w,
                d->out_h, d->out_w, d->pad_y, d->pad_x, pl.tiles_x, pl.tiles_y, pl.tiles_per_split};
    if (d->kh == 3) rc = d->down == 2 ? dispatch_wgrad<2, 3>(a, pl, s) : dispatch_wgrad<1, 3>(a, pl, s);
    else            rc = d->down == 2 ? dispatch_wgrad<2, 1>(a, pl, s) : dispatch_wgrad<1, 1>(a, pl, s);
    if (rc) return rc;
    if (pl.parts == 1) return GC_OK;
    return launch_wgrad_reduce(static_cast<const float*>(workspace), dw, count, pl.parts, s);
}

// lane groups per element group (see wgrad_reduce_kernel): only where the tensor alone cannot fill the chip and there are splits to share out
static int reduce_groups(size_t n4, int parts) {
#ifdef GC_REDUCE_FLAT       // A/B build: one lane per element group everywhere (the round-2 reduce)
    return 1;
#endif
    if (n4 >= 256 * 256 || parts < 16) return 1;
    return parts >= 64 ? 16 : 4;
}

int gcconv::launch_wgrad_reduce(const float* ws, float* dw, size_t count, int parts, hipStream_t s) {
    const bool vec = count % 4 == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0;
    const int jg = vec ? reduce_groups(count / 4, parts) : 1;
    const size_t per_block = vec ? 256 / jg : 256;
    const int blocks = (int)std::min<size_t>(((vec ? count / 4 : count) + per_block - 1) / per_block, 2048);
    if (!vec)         hipLaunchKernelGGL((wgrad_reduce_kernel<false, 1>), dim3(blocks), dim3(256), 0, s, ws, dw, count, parts);
    else if (jg == 1) hipLaunchKernelGGL((wgrad_reduce_kernel<true, 1>), dim3(blocks), dim3(256), 0, s, ws, dw, count, parts);
    else if (jg == 4) hipLaunchKernelGGL((wgrad_reduce_kernel<true, 4>), dim3(blocks), dim3(256), 0, s, ws, dw, count, parts);
    else              hipLaunchKernelGGL((wgrad_reduce_kernel<true, 16>), dim3(blocks), dim3(256), 0, s, ws, dw, count, parts);
    return gc::check_launch("gc_conv2d_wgrad(reduce)");
}

int gcconv::launch_wgrad_reduce_samples(const float* ws, float* dw, float* samples, size_t count, int batch, int per_sample, hipStream_t s) {
    const bool vec = count % 4 == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(samples) & 15) == 0;
    const int jg = vec ? reduce_groups(count / 4, per_sample) : 1;
    const size_t per_block = vec ? 256 / jg : 256;
    const int blocks = (int)std::min<size_t>(((vec ? count / 4 : count) + per_block - 1) / per_block, 2048);
    if (!vec)         hipLaunchKernelGGL((wgrad_reduce_samples_kernel<false, 1>), dim3(blocks), dim3(256), 0, s, ws, dw, samples, count, batch, per_sample);
    else if (jg == 1) hipLaunchKernelGGL((wgrad_reduce_samples_kernel<true, 1>), dim3(blocks), dim3(256), 0, s, ws, dw, samples, count, batch, per_sample);
    else if (jg == 4) hipLaunchKernelGGL((wgrad_reduce_samples_kernel<true, 4>), dim3(blocks), dim3(256), 0, s, ws, dw, samples, count, batch, per_sample);
    else              hipLaunchKernelGGL((wgrad_reduce_samples_kernel<true, 16>), dim3(blocks), dim3(256), 0, s, ws, dw, samples, count, batch, per_sample);
    return gc::check_launch("gc_conv2d_wgrad_samples(reduce)");
}

// fp32 arithmetic: only the thin 1x1 shapes (ToRGB / FromRGB class) have a per-sample form; gc_conv2d_wgrad_samples_workspace(d, 0) says which
extern "C" int gc_conv2d_wgrad_samples_f32(const gc_conv_desc* d, const float* x, const float* dy, const float* in_scale, const float* out_scale,
                                           float* dw, float* dw_samples, void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    int rc = validate(d, "gc_conv2d_wgrad_samples_f32", true);
    if (rc) return rc;
    if (!x || !dy || !dw || !dw_samples) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_wgrad_samples_f32: null pointer");
    if (d->batch <= 0 || !pointwise_thin_wgrad(d) || (d->in_pitch != 0 && d->in_pitch != d->in_w))
        return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_wgrad_samples_f32: only dense 1x1 shapes with <= 4 channels on one side (gc_conv2d_wgrad_samples_workspace() == 0 otherwise)");
    const size_t need = pointwise_wgrad_workspace(d);
    if (!workspace || workspace_bytes < need) return gc::fail(GC_ERR_WORKSPACE, "gc_conv2d_wgrad_samples_f32: workspace %zu < %zu bytes", workspace_bytes, need);
    return pointwise_wgrad(d, x, dy, in_scale, out_scale, dw, dw_samples, workspace, stream);
}

extern "C" size_t gc_wgrad_samples_contract_workspace(int batch, int a, int c) {
    return batch > 0 && a > 0 && c > 0 ? (size_t)batch * a * c * sizeof(float) : 0;
}

extern "C" int gc_wgrad_samples_contract_f32(const float* dw_samples, const float* w, const float* scale_a, const float* scale_c, float* g_a, float* g_c,
                                             int batch, int taps, int a, int c, void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    if (!dw_samples || !w || (!g_a && !g_c)) return gc::fail(GC_ERR_BAD_ARG, "gc_wgrad_samples_contract_f32: null pointer");
    if (batch < 0 || batch > 65535 || taps <= 0 || a <= 0 || c <= 0) return gc::fail(GC_ERR_BAD_ARG, "gc_wgrad_samples_contract_f32: batch %d, taps %d, %d x %d", batch, taps, a, c);
    if (batch == 0) return GC_OK;
    const size_t need = g_c ? gc_wgrad_samples_contract_workspace(batch, a, c) : 0;
    if (g_c && (!workspace || workspace_bytes < need)) return gc::fail(GC_ERR_WORKSPACE, "gc_wgrad_samples_contract_f32: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    float* colpart = g_c ? static_cast<float*>(workspace) : nullptr;
    hipLaunchKernelGGL(wgrad_contract_rows_kernel, dim3(a, batch), dim3(256), 0, s, dw_samples, w, scale_a, g_a, colpart, taps, a, c);
    int rc = gc::check_launch("gc_wgrad_samples_contract_f32(rows)");
    if (rc || !g_c) return rc;
    hipLaunchKernelGGL(wgrad_contract_cols_kernel, dim3(gc::ceil_div(c, 64), batch), dim3(256), 0, s, colpart, scale_c, g_c, a, c);
    return gc::check_launch("gc_wgrad_samples_contract_f32(columns)");
}
