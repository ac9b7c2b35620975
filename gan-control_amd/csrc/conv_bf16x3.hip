// K3/K4 fast path: the generalised convolution of include/gancontrol_hip.h on the bf16 matrix cores
// with SPLIT-bf16 ("bf16x3") arithmetic:
//
//   a = a_hi + a_lo,  a_hi = bf16(a), a_lo = bf16(a - a_hi)        (16 mantissa bits kept)
//   a * b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi                    (fp32 accumulate in the MFMA)
//
// Three v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) replace eight v_mfma_f32_32x32x2_f32
// (64 cycles, K = 2): 5.3x the fp32-MFMA rate.  Measured error vs fp64 on a 512-channel 3x3 layer:
// 5e-6 relative (fp32: 3e-7, plain bf16: 3e-3) -- two orders inside the 1e-3 parity bound.
// fp32 in HBM on both sides: the split happens while staging into LDS (activations, after the
// per-sample in_scale multiply) and in a pre-pass over the weights (pack_weights_kernel).
//
// Structure mirrors conv_mfma_kernel (conv.hip): per chunk of 16 input channels the workgroup stages
// the halo'd input patch once, channel-LAST in LDS -- one 16-byte unit = 8 consecutive channels of one
// pixel = exactly one lane's MFMA B fragment, so every tap reads it with a single conflict-free
// ds_read_b128 at a shifted unit index -- plus the [tap][2][OCT] weight units (A fragments).
// Register-prefetch pipeline over chunks, compile-time geometry, phase decomposition for up = 2.
#include "conv_common.h"
#include "conv_bf16x3_shared.h"

namespace {

using namespace gcconv;

// wp[t][kg][n] = 8 x bf16 of w[t][kg*8 + q][n], q = 0..7 (zero beyond K); hi and lo parts
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, uint4* __restrict__ wh, uint4* __restrict__ wl,
                                                           int taps, int K, int N, int kgroups) {
    const size_t total = (size_t)taps * kgroups * N;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (size_t)gridDim.x * 256) {
        const int n = (int)(u % N);
        const size_t rest = u / N;
        const int kg = (int)(rest % kgroups), t = (int)(rest / kgroups);
        bf16x8 h, l;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = kg * 8 + q;
            const float v = k < K ? w[((size_t)t * K + k) * N + n] : 0.f;
            const __bf16 hh = (__bf16)v;
            h[q] = hh;
            l[q] = (__bf16)(v - (float)hh);
        }
        wh[u] = *reinterpret_cast<uint4*>(&h);
        wl[u] = *reinterpret_cast<uint4*>(&l);
    }
}

// The same split for MANY weight tensors in one launch (gc_conv2d_pack_weights_bf16x3_grouped): a block finds its tensor in a table.
constexpr int MAXPG = 48;
struct PackGroup { const float* w; uint4* wh; uint4* wl; int taps, K, N, kgroups, first; };
struct PackGroupArgs { PackGroup g[MAXPG]; int n_groups; };

__global__ __launch_bounds__(256) void pack_weights_grouped_kernel(PackGroupArgs a) {
    int gi = 0;
#pragma unroll 1
    for (int i = 1; i < a.n_groups; ++i) gi = ((int)blockIdx.x >= a.g[i].first) ? i : gi;
    const PackGroup& G = a.g[gi];
    const size_t total = (size_t)G.taps * G.kgroups * G.N;
    const size_t u = (size_t)(blockIdx.x - G.first) * 256 + threadIdx.x;
    if (u >= total) return;
    const int n = (int)(u % G.N);
    const size_t rest = u / G.N;
    const int kg = (int)(rest % G.kgroups), t = (int)(rest / G.kgroups);
    bf16x8 h, l;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = kg * 8 + q;
        const float v = k < G.K ? G.w[((size_t)t * G.K + k) * G.N + n] : 0.f;
        const __bf16 hh = (__bf16)v;
        h[q] = hh;
        l[q] = (__bf16)(v - (float)hh);
    }
    G.wh[u] = *reinterpret_cast<uint4*>(&h);
    G.wl[u] = *reinterpret_cast<uint4*>(&l);
}

template <int WG_OC, int WG_PX, int WOC, int WPX, int UP, int DOWN, int KS, int CB = 1>
struct BCfg {
    static constexpr int OCT = WG_OC * WOC * 32;
    static constexpr int TPH = WG_PX * WPX / CB;            // tile rows (each 32-pixel MFMA column block is one row segment); CB column blocks side by side
    static constexpr int NT1 = UP == 1 ? KS : (KS + UP - 1) / UP;
    static constexpr int PH = (TPH - 1) * DOWN + NT1, PWD = (32 * CB - 1) * DOWN + NT1;
    // Column order of a patch row in LDS.  Stride 2: de-interleaved (even columns, then odd), so that lane l's fragment
    // read of column 2 l + tap is a run of consecutive 16-byte units -- row-major order spent 41 % of the LDS cycles in
    // bank conflicts there (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE), de-interleaved 15 %, kernel -3 %.  Stride 1 stays
    // row-major: its reads are conflict-free, and none of the de-interleaved variants measured (by 2 / by 4, pitches
    // 17..20 / 9..12) moved the kernel time by more than the noise although they change the conflict count by 2.4x:
    // the LDS is not what bounds that kernel.  Column c sits at unit (c % DI) * Q + c / DI.
    static constexpr int DI = DOWN == 1 ? 1 : 2;
    static constexpr int Q = DOWN == 1 ? PWD : (PWD + 1) / 2;
    static constexpr int RP = DI * Q;                       // row pitch in units
    static constexpr int PLANE = PH * RP;                   // units per channel group
    __device__ static __forceinline__ int ucol(int c) { return (c % DI) * Q + c / DI; }
    static constexpr int WUNITS = NT1 * NT1 * KG * OCT;     // weight units per chunk (per hi / lo)
    static constexpr int PUNITS = KG * PLANE;
    static constexpr int SMEM_UNITS = 2 * (WUNITS + PUNITS);
    static constexpr int NWU = (WUNITS + 255) / 256;        // weight units prefetched per thread (x2: hi, lo)
    // Patch staging.  The texture-address unit spends ~16 cycles per wave-level load whatever its width, and at <= 64 input
    // channels that -- not HBM, not the MFMAs -- bounds the kernel (measured: 37 % of a pipeline step issuing dword loads,
    // 30 % waiting for them).  So a lane fetches FOUR consecutive pixels of a channel with one 16-byte load, eight channels
    // = eight loads, and transposes them in registers into four channel-last units: 4x fewer load instructions.
    // A patch row starts `lead` floats before a 128-byte boundary (lead = pad for 32-pixel tiles): its SEG_M aligned 32-float
    // segments are cut into 16-byte groups ("main" tasks, all four pixels used), the EDGE columns left and right of them
    // are "edge" tasks (same load, first pixel used).  Out-of-range dwords of a buffer load read as zero one by one
    // (tools/micro/buf_oob.hip), so a group may straddle the end of the tensor; pixels past the end of an image ROW are
    // masked at commit.
    static constexpr int SEG_M = PWD % 32 == 0 ? PWD / 32 : (PWD - 1) / 32;
    static constexpr int EDGE = PWD - 32 * SEG_M;
    static constexpr int MAIN_T = PH * 8 * SEG_M, EDGE_T = PH * EDGE, TASKS = MAIN_T + EDGE_T;   // per 8-channel group
    static constexpr int NT = (TASKS + 127) / 128;          // tasks per thread (128 threads per channel group)
    struct Task { int row, col, used; };                    // patch row, first patch column, pixels used (0: no task)
    __device__ static __forceinline__ Task task_of(int t, int lead) {
        Task k;
        if (t < MAIN_T) { k.row = t / (8 * SEG_M); k.col = lead + 4 * (t % (8 * SEG_M)); k.used = 4; return k; }
        const int e = t - MAIN_T, ce = e % cmax(EDGE, 1);
        k.row = e / cmax(EDGE, 1);
        k.col = ce < lead ? ce : 32 * SEG_M + ce;
        k.used = t < TASKS ? 1 : 0;
        return k;
    }
};

#if defined(GC_EXP) && GC_EXP == 2      // experiment: three workgroups per CU for the 32-channel tiles (more loads in flight on the HBM-bound layers)
#define GC_CONV_OCC(WOC, WPX, DOWN) ((WOC) * (WPX) <= 2 && (DOWN) == 1 ? 3 : 2)
#else
#define GC_CONV_OCC(WOC, WPX, DOWN) 2
#endif
template <int WG_OC, int WG_PX, int WOC, int WPX, int UP, int DOWN, int KS>
__global__ __launch_bounds__(256, GC_CONV_OCC(WOC, WPX, DOWN)) void conv_bf16x3_kernel(Bf16Args a) {
    using C = BCfg<WG_OC, WG_PX, WOC, WPX, UP, DOWN, KS>;
    static_assert(WG_OC * WG_PX == 4, "4 waves per workgroup");
    constexpr int OCT = C::OCT, TPH = C::TPH, PWD = C::PWD, PLANE = C::PLANE;
    const ConvArgs& p = a.c;
    __shared__ uint4 smem[C::SMEM_UNITS];
    uint4* wl_h = smem;                         // [tap][kg][OCT]
    uint4* wl_l = wl_h + C::WUNITS;
    uint4* p_h = wl_l + C::WUNITS;              // [kg][PH][PWD]
    uint4* p_l = p_h + C::PUNITS;
    __shared__ __attribute__((aligned(16))) float s_si[MAX_K_BF16X3 + KCB];     // in_scale of this sample, zero-padded past K
    // out_scale and bias of this workgroup's channels.  They must NOT be fetched between the stores of the epilogue: a
    // vector load there forces s_waitcnt vmcnt(0), which also waits for every store issued before it -- one full memory
    // round trip per output row.
    __shared__ __attribute__((aligned(16))) float s_so[OCT], s_bias[OCT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wave_px = wave % WG_PX, wave_oc = wave / WG_PX;
    const int wave_oc_u = __builtin_amdgcn_readfirstlane(wave_oc);

    // A workgroup owns `tpb` consecutive pixel tiles of one (sample, phase, oc-block) and runs ONE software pipeline over all
    // their channel chunks: the first chunk of the next tile is in flight during the last MFMA phase and the stores of the
    // current tile, so neither the cold-start load latency nor the store drain is paid per tile (they dominate at K <= 64).
    int bid = blockIdx.x;
    const int grp = bid % a.groups; bid /= a.groups;
    const int phase = bid % (UP * UP);
    const int b = bid / (UP * UP);
    const int phy = phase / UP, phx = phase % UP;
    const int n0 = blockIdx.y * OCT, n0_blk = n0;
    const int qh = (p.out_h - phy + UP - 1) / UP, qw = (p.out_w - phx + UP - 1) / UP;
    // (GC_CONV_STRIDED: the tiles of a workgroup are `groups` apart, so that the workgroups resident together read neighbouring tiles)
    const int tstep = GC_CONV_STRIDED ? a.groups : 1;
    const int tile_begin = GC_CONV_STRIDED ? grp : grp * a.tpb;
    const int tile_end = GC_CONV_STRIDED ? p.tiles_x * p.tiles_y : min(p.tiles_x * p.tiles_y, tile_begin + a.tpb);
    const int kz0 = a.k_per_split ? (int)blockIdx.z * a.k_per_split : 0;
    const int kz1 = a.k_per_split ? min(p.K, kz0 + a.k_per_split) : p.K;
    if (UP > 1) {      // tpb == 1; phases other than 0 have a smaller sub-grid
        if ((tile_begin / p.tiles_x) * TPH >= qh || (tile_begin % p.tiles_x) * 32 >= qw) return;
    }

    const AxisTaps ay = axis_taps<UP, KS>(phy, p.pad_y), ax = axis_taps<UP, KS>(phx, p.pad_x);
    const int ntaps = ay.n * ax.n;
    // patch rows start at 32 * DOWN * tile_x + ax.d0: `lead` floats before an aligned 32-float segment
    const int lead = gc::pos_mod(-ax.d0, 32);      // <= C::EDGE (checked on the host)

    f32x16 acc[WOC][WPX];
#pragma unroll
    for (int i = 0; i < WOC; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int boff[WPX];
#pragma unroll
    for (int j = 0; j < WPX; ++j) boff[j] = hi * PLANE + (wave_px * WPX + j) * DOWN * C::RP;      // row of this lane's pixels (tap row 0)

    const int aoff = hi * OCT + wave_oc * WOC * 32 + l31;

    const float* xb = p.x + (size_t)b * p.K * p.in_h * a.in_pitch;
    const float* sib = p.si ? p.si + (size_t)b * p.K : nullptr;
    const int chan = p.in_h * a.in_pitch;          // a.in_pitch floats between input rows (the pitched output of a Blur; in_w when dense)

    uint4 wreg_h[C::NWU], wreg_l[C::NWU];
    uint4 preg[C::NT][8];                       // [task][channel] = 4 consecutive pixels

    // Buffer-descriptor loads: scalar channel offset + 32-bit lane offset, hardware zero-fill outside the image, and
    // nothing touches the results until commit(), so every load stays in flight across the MFMA block.
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)p.K * chan * 4u);
    const unsigned wbytes = (unsigned)(KS * KS) * a.kgroups * p.N * 16u;
    const __amdgpu_buffer_rsrc_t rwh = make_rsrc(a.wh, wbytes), rwl = make_rsrc(a.wl, wbytes);
    auto prefetch = [&](int tile, int k0) {
        const int t_ = tid;
        const int iy0 = (tile / p.tiles_x) * TPH * DOWN + ay.d0, ix0 = (tile % p.tiles_x) * 32 * DOWN + ax.d0;
        // weights: unit u -> (tap, kg, oc); plain 16-byte copies of the pre-split slab
#pragma unroll
        for (int j = 0; j < C::NWU; ++j) {
            const int u = t_ + 256 * j;
            const int oc = u % OCT, rest = u / OCT;
            const int kgl = rest % KG, t = rest / KG;
            const int jy = UP == 1 ? t / KS : (ax.n == 2 ? t >> 1 : t), jx = UP == 1 ? t % KS : (ax.n == 2 ? t & 1 : 0);
            const int tap = (ay.t0 + jy * UP) * KS + ax.t0 + jx * UP;
            const int kg = k0 / 8 + kgl, n = n0 + oc;
            const bool ok = u < C::WUNITS && t < ntaps && kg < a.kgroups && n < p.N;
            const unsigned gb = ok ? (unsigned)((tap * a.kgroups + kg) * p.N + n) * 16u : OOB;
            wreg_h[j] = buf_load_u128(rwh, gb, 0);
            wreg_l[j] = buf_load_u128(rwl, gb, 0);
        }
        // patch: waves 0,1 take channel group 0, waves 2,3 group 1
        const int kgl = __builtin_amdgcn_readfirstlane(t_ >> 7), tb = t_ & 127;
#pragma unroll
        for (int j = 0; j < C::NT; ++j) {
            const typename C::Task tk = C::task_of(tb + 128 * j, lead);
            const int iy = iy0 + tk.row, ix = ix0 + tk.col;
            const bool ok = tk.used > 0 && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
            const unsigned boff = ok ? (unsigned)(iy * a.in_pitch + ix) * 4u : OOB;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = min(k0 + kgl * 8 + q, p.K - 1);          // wave-uniform -> scalar offset
                if (!(DOWN == 2 && GC_S2_ABL == 2)) preg[j][q] = buf_load_u128(rx, boff, (unsigned)k * chan * 4u);
            }
        }
    };
    auto commit = [&](int tile, int k0) {
        const int t_ = tid;
#pragma unroll
        for (int j = 0; j < C::NWU; ++j) {
            const int u = t_ + 256 * j;
            if (u < C::WUNITS) { wl_h[u] = wreg_h[j]; GC_LO(wl_l[u] = wreg_l[j];) }
        }
        const int kgl = __builtin_amdgcn_readfirstlane(t_ >> 7), tb = t_ & 127;
        const int ix0 = (tile % p.tiles_x) * 32 * DOWN + ax.d0;
        // per-sample input scales of this chunk (zero beyond K: a ragged last chunk contributes nothing)
        const float4 sa = *reinterpret_cast<const float4*>(&s_si[k0 + kgl * 8]), sb = *reinterpret_cast<const float4*>(&s_si[k0 + kgl * 8 + 4]);
        const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
        // 16-byte groups start at multiples of four pixels (ix0 + lead is a multiple of 32), so in a row whose width is a multiple of
        // four a group lies entirely inside the image or entirely outside (and was then fetched as zeros): the per-pixel row-end mask --
        // 32 selects per chunk -- is only needed for the odd widths (the 1025-wide planes of the stride-2 convolutions).
        const bool ragged_rows = ((p.in_w & 3) != 0 || a.in_pitch != p.in_w) && ix0 + lead + 32 * C::SEG_M + 4 > p.in_w;      // ... and there only in the tiles that reach the row end
        auto convert = [&](auto masked, auto scaled) {
#pragma unroll
            for (int j = 0; j < C::NT; ++j) {
                const typename C::Task tk = C::task_of(tb + 128 * j, lead);
                const int inrow = p.in_w - (ix0 + tk.col);               // pixels of this group that are still inside the image row
                const int rbase = kgl * PLANE + tk.row * C::RP;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const unsigned raw = i == 0 ? preg[j][q].x : (i == 1 ? preg[j][q].y : (i == 2 ? preg[j][q].z : preg[j][q].w));
                        v[q] = (!decltype(masked)::value || i < inrow) ? __uint_as_float(raw) : 0.f;
                    }
                    uint4 h, l;
                    split8s<decltype(scaled)::value>(v, sc, &h, &l);        // plain (un-packed) multiplies and subtractions: see split8s
                    if (i < tk.used) {
                        const int u = rbase + C::ucol(tk.col + i);
                        p_h[u] = h;
                        GC_LO(p_l[u] = l;)
                    }
                }
            }
        };
        if (DOWN == 2 && GC_S2_ABL == 1) {
            // the 8 x 16 bytes of a task are 4 hi + 4 lo units' worth: stored without touching them
#pragma unroll
            for (int j = 0; j < C::NT; ++j) {
                const typename C::Task tk = C::task_of(tb + 128 * j, lead);
                const int rbase = kgl * PLANE + tk.row * C::RP;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < tk.used) { const int u = rbase + C::ucol(tk.col + i); p_h[u] = preg[j][i]; p_l[u] = preg[j][4 + i]; }
            }
        } else if (DOWN == 2 && GC_S2_ABL == 2) {
        } else if (p.si) {      // without modulation (every layer of D) the multiply by one is not issued
            if (ragged_rows) convert(std::true_type{}, std::true_type{}); else convert(std::false_type{}, std::true_type{});
        } else {
            if (ragged_rows) convert(std::true_type{}, std::false_type{}); else convert(std::false_type{}, std::false_type{});
        }
    };
    auto mfma_phase = [&]() {
#if GC_FRAG_PIPE
        if constexpr (UP == 1) {
            // Fragment double buffer, as in the wave-specialised kernel: the LDS reads of tap t + 1 are issued BEFORE the MFMAs of tap t (the
            // scheduling barriers pin that order).  Left alone the compiler issues a tap's reads and waits for them on the spot -- the
            // disassembly of the stride-2 variant showed `ds_read x4, s_waitcnt lgkmcnt, v_mfma` with one or two MFMAs between two waits.
            bf16x8 fa[2][2 * WOC], fb[2][2 * WPX];
            auto load_tap = [&](int t, int set) {
                const int jy = t / KS, jx = t % KS;
                const int wbase = t * KG * OCT + aoff;
                const int pbase = jy * C::RP + C::ucol(l31 * DOWN + jx);
#pragma unroll
                for (int i = 0; i < WOC; ++i) {
                    const uint4 uh = wl_h[wbase + i * 32];
                    fa[set][i] = *reinterpret_cast<const bf16x8*>(&uh);
                    GC_LO(const uint4 ul = wl_l[wbase + i * 32]; fa[set][WOC + i] = *reinterpret_cast<const bf16x8*>(&ul);)
                }
#pragma unroll
                for (int j = 0; j < WPX; ++j) {
                    const uint4 uh = p_h[pbase + boff[j]];
                    fb[set][j] = *reinterpret_cast<const bf16x8*>(&uh);
                    GC_LO(const uint4 ul = p_l[pbase + boff[j]]; fb[set][WPX + j] = *reinterpret_cast<const bf16x8*>(&ul);)
                }
            };
            load_tap(0, 0);
#pragma unroll
            for (int t = 0; t < KS * KS; ++t) {
                if (t + 1 < KS * KS) load_tap(t + 1, (t + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < WOC; ++i)
#pragma unroll
                    for (int j = 0; j < WPX; ++j) { GC_MFMA3(acc[i][j], fa[t & 1][i], fa[t & 1][WOC + i], fb[t & 1][j], fb[t & 1][WPX + j]); }
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#endif
        const int nty = UP == 1 ? KS : ay.n, ntx = UP == 1 ? KS : ax.n;
        for (int jy = 0; jy < nty; ++jy) {
            for (int jx = 0; jx < ntx; ++jx) {
                const int wbase = (jy * ntx + jx) * KG * OCT + aoff;
                const int pbase = jy * C::RP + C::ucol(l31 * DOWN + jx);      // LDS column of this lane's pixel under tap jx
                bf16x8 ah[WOC], al[WOC], bh[WPX], bl[WPX];
#pragma unroll
                for (int i = 0; i < WOC; ++i) {
                    const uint4 uh = wl_h[wbase + i * 32], ul = wl_l[wbase + i * 32];
                    ah[i] = *reinterpret_cast<const bf16x8*>(&uh);
                    al[i] = *reinterpret_cast<const bf16x8*>(&ul);
                }
#pragma unroll
                for (int j = 0; j < WPX; ++j) {
                    const uint4 uh = p_h[pbase + boff[j]], ul = p_l[pbase + boff[j]];
                    bh[j] = *reinterpret_cast<const bf16x8*>(&uh);
                    bl[j] = *reinterpret_cast<const bf16x8*>(&ul);
                }
#pragma unroll
                for (int i = 0; i < WOC; ++i)
#pragma unroll
                    for (int j = 0; j < WPX; ++j) { GC_MFMA3(acc[i][j], ah[i], al[i], bh[j], bl[j]); }
            }
        }
    };
    const float* sob = p.so ? p.so + (size_t)b * p.N : nullptr;
    // Output through a buffer descriptor as well: lane offset = pixel (+ the hi half's 4 channels), scalar offset = channel
    // plane; channels >= N and pixels outside the plane fall beyond num_records and are dropped by the hardware.
    const unsigned oplane = (unsigned)(p.out_h * p.out_w) * 4u;
    float* const ybase = a.k_per_split ? a.part + (size_t)blockIdx.z * a.per_slice : p.y;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(ybase + (size_t)b * p.N * p.out_h * p.out_w, (unsigned)p.N * oplane);
    // store one finished tile (demodulation + fused epilogue) and clear the accumulators for the next
    const EpilogueConsts ec = epilogue_consts(p);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.residual ? p.residual + (size_t)b * p.N * p.out_h * p.out_w : p.y, p.residual ? (unsigned)p.N * oplane : 0u);
    auto finish_tile = [&](int tile) {
        const int qy0 = (tile / p.tiles_x) * TPH, qx0 = (tile % p.tiles_x) * 32;
        const int n0 = opaque_s(n0_blk);          // recompute the channel offsets here rather than carry 64 of them across the loop
#pragma unroll
        for (int j = 0; j < WPX; ++j) {
            const int qy = qy0 + wave_px * WPX + j, qx = qx0 + l31;
            const int oy = qy * UP + phy, ox = qx * UP + phx;
            const bool inside = qy < qh && qx < qw;
            const unsigned voff = inside ? (unsigned)(oy * p.out_w + ox) * 4u + (unsigned)(4 * hi) * oplane : OOB;
            const float nz = (p.noise && inside) ? p.noise[((size_t)b * p.out_h + oy) * p.out_w + ox] : 0.f;
            float res[WOC][16];
            if (p.residual) {       // all residual values of this pixel row are fetched before its first store (a load between stores waits for them)
#pragma unroll
                for (int i = 0; i < WOC; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        res[i][r] = buf_load_f32(rres, voff, (unsigned)(n0 + (wave_oc_u * WOC + i) * 32 + (r & 3) + 8 * (r >> 2)) * oplane);
            }
#pragma unroll
            for (int i = 0; i < WOC; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // a register row holds channel ocs in lanes 0..31 and ocs + 4 in lanes 32..63
                    const int ocl = (wave_oc * WOC + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const int ocs = n0 + (wave_oc_u * WOC + i) * 32 + (r & 3) + 8 * (r >> 2);
                    float v = conv_epilogue(ec, acc[i][j][r], s_so[ocl], s_bias[ocl], nz);
                    if (p.residual) v += res[i][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, (int)voff, (int)((unsigned)ocs * oplane), GC_CONV_ST_AUX);
                    acc[i][j][r] = 0.f;
                }
            }
        }
    };

    if (ntaps == 0) {       // a phase no tap reaches: zeros (+ epilogue)
        for (int tile = tile_begin; tile < tile_end; tile += tstep) finish_tile(tile);
        return;
    }
    const int nchunks = (kz1 - kz0 + KCB - 1) / KCB;
    const int items = (tile_end - tile_begin + tstep - 1) / tstep * nchunks;
    int tile_n = tile_begin, k0_n = kz0;      // cursor of the staging side (prefetch / commit)
    int tile_c = tile_begin, k0_c = kz0;      // cursor of the compute side (MFMA / stores)
    prefetch(tile_n, k0_n);
    for (int k = tid; k < ((p.K + KCB - 1) / KCB) * KCB; k += 256) s_si[k] = k < p.K ? (sib ? sib[k] : 1.f) : 0.f;
    if (tid < OCT) {
        const int oc = min(n0 + tid, p.N - 1);
        s_so[tid] = sob ? sob[oc] : 1.f;
        s_bias[tid] = p.bias ? p.bias[oc] : 0.f;
    }
    __syncthreads();
    commit(tile_n, k0_n);
    k0_n += KCB; if (k0_n >= kz1) { k0_n = kz0; tile_n += tstep; }
    __syncthreads();
    // Steady state.  Every step is unconditional, so no control-flow path reaches the loop header with staged loads
    // in flight and the compiler plants no wait inside the next prefetch; the last item is peeled below.
    for (int it = 1; it < items; ++it) {
        prefetch(tile_n, k0_n);
        __builtin_amdgcn_s_setprio(GC_MFMA_PRIO);
        mfma_phase();
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        commit(tile_n, k0_n);                               // retires the loads first: no store is outstanding yet
        k0_n += KCB; if (k0_n >= kz1) { k0_n = kz0; tile_n += tstep; }
        if (k0_c + KCB >= kz1) finish_tile(tile_c);         // stores drain while the next MFMA phase runs
        k0_c += KCB; if (k0_c >= kz1) { k0_c = kz0; tile_c += tstep; }
        __syncthreads();
    }
    mfma_phase();
    finish_tile(tile_c);
}

// ---------------------------------------------------------------------------------------------------------
// Wave-specialised variant of conv_bf16x3_kernel for the wide layers (up = down = 1, K a multiple of 16, N a multiple of 64):
// ONE workgroup of 12 waves per CU -- eight MULTIPLYING waves (two per SIMD; each owns 64 oc x 2 rows x 32 px of a
// 64 oc x 16 rows x 32 px tile and issues nothing but LDS fragment reads and MFMAs) and four STAGING waves (one per SIMD:
// global loads, the per-sample scale, the hi / lo split and the LDS writes of the NEXT 16-channel chunk) -- over two LDS stages
// with one barrier per chunk.  In conv_bf16x3_kernel every wave alternates between the two jobs and the matrix pipes only stay
// busy while the co-resident workgroup happens to be in the other phase (matrix pipes busy 51 %, 21 % of the wave time at the two
// barriers per chunk, profiles/pmc_r01.md); here the multiplying waves never convert and never wait for a load.
//  * The pre-split weight slab goes HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: a [tap][kg] row of 64 oc units is 1 KiB,
//    contiguous on both sides = one wave-level instruction): no registers, no ds_write, no vector ALU work for 63 % of the staged bytes.
//  * A 16-row tile halves the weight staging and the halo rows (18 / 16 instead of 10 / 8) per MFMA.
//  * Ordering of the LDS-DMA data: the staging wave waits vmcnt(0) before the barrier, the multiplying waves read the stage after it;
//    a stage is rewritten one full item after its last read (the barrier in between retires the reads).
// EPK: 0 = the full fused epilogue; 1 = out_scale and / or residual only (the input-gradient launches: G's modulated layers, D's ResBlock
// convolutions); 2 = nothing to apply.  Compile-time: the epilogue runs on the MULTIPLYING waves (6 vector instructions + 2 LDS reads per output
// element in its full form, 384 per lane and tile, both waves of a SIMD at the same moment) next to only 108 MFMAs per tile at 32 input channels.
// RES: the launch adds a residual (gc_conv_epilogue.residual).  Compile-time as well: the residual form keeps 64 loads in flight next to the
// accumulators, and as a run-time branch of the SAME kernel its register pressure spilled values that live across the whole kernel (34 VGPRs,
// a -9 .. -20 % on the 64-channel layers WITHOUT a residual, same-box A/B profiles/kernel_ab_r05_b.log).
#ifndef GC_WS_TRACE
#define GC_WS_TRACE 0        // dev instrumentation (tools/ws_trace.py): workgroup 0 records s_memtime at the phase boundaries of its first items -- one multiplying and one staging wave
#endif
#if GC_WS_TRACE
__device__ unsigned long long gc_ws_trace[2][512];      // [role][event]: (tag << 56) | time
#define GC_TR(role, tag) do { if (tr_on && tr_n[role] < 512) { gc_ws_trace[role][tr_n[role]++] = ((unsigned long long)(tag) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull); } } while (0)
#else
#define GC_TR(role, tag) do { } while (0)
#endif
template <int KS, int WOC, int CB, int EPK = 0, bool RES = false>
__global__ __launch_bounds__(768) void conv_bf16x3_ws_kernel(Bf16Args a) {
    using C = BCfg<1, 8, WOC, 2, 1, 1, KS, CB>;      // CB = 1: 16 rows x 32 px tiles; CB = 2: 8 rows x 64 px (longer contiguous runs per row: the HBM-bound layers)
    constexpr int OCT = 32 * WOC, TPH = C::TPH, PLANE = C::PLANE, WPX = 2, NTAP = KS * KS;
    constexpr int STAGE = C::SMEM_UNITS;                 // one stage: [weights hi | weights lo | patch hi | patch lo]
    static_assert(C::OCT == OCT && C::TPH * CB == 16, "(32 | 64) oc x 512 px tiles");
    static_assert(2 * STAGE * 16 + (MAX_K_BF16X3 + KCB + 2 * OCT) * 4 <= 160 * 1024, "two stages fit the 160 KiB of LDS");
    const ConvArgs& p = a.c;
    __shared__ uint4 smem[2 * STAGE];
    __shared__ __attribute__((aligned(16))) float s_si[MAX_K_BF16X3 + KCB];
    __shared__ __attribute__((aligned(16))) float s_so[OCT], s_bias[OCT];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

#if GC_WS_TRACE
    const bool tr_on = blockIdx.x == GC_WS_TRACE - 1 && blockIdx.y == 0 && (threadIdx.x == 0 || threadIdx.x == 512);
    int tr_n[2] = {0, 0};
#endif
    int bid = blockIdx.x, boc = blockIdx.y;
#if GC_WS_XCD
    // XCD-aware order: workgroups go to the eight XCDs round-robin in linear block order, so the output-channel blocks of ONE pixel group (ids gridDim.x
    // apart) land on different L2s -- or on the same one a whole round later -- and the patch crosses the fabric once per block.  Re-deal the ids so that
    // the blocks of a pixel group are 8 apart: same XCD, dispatched back to back.
    if (gridDim.y > 1 && gridDim.x % 8 == 0) {
        const unsigned l = blockIdx.x + gridDim.x * blockIdx.y, slot = l >> 3;
        boc = (int)(slot % gridDim.y);
        bid = (int)((slot / gridDim.y) * 8 + (l & 7));
    }
#endif
    const int grp = bid % a.groups;
    const int b = bid / a.groups;
    const int n0 = boc * OCT;
    // The `tpb` tiles of a workgroup are `groups` apart (GC_WS_STRIDED): at any moment the 256 resident workgroups then work on ~256
    // NEIGHBOURING tiles -- a band of rows of one sample, contiguous per channel in DRAM -- instead of 256 bands spread over the batch.
    const int tiles_all = p.tiles_x * p.tiles_y;
    const int tstep = GC_WS_STRIDED ? a.groups : 1;
    const int tile_begin = GC_WS_STRIDED ? grp : grp * a.tpb;
    const int ntiles = GC_WS_STRIDED ? (tiles_all - grp + a.groups - 1) / a.groups : min(tiles_all, tile_begin + a.tpb) - tile_begin;
    const int nchunks = p.K / KCB;
    const int items = ntiles * nchunks;
    const int chan = p.in_h * p.in_w;

    for (int k = tid; k < p.K; k += 768) s_si[k] = p.si ? p.si[(size_t)b * p.K + k] : 1.f;
    if (tid < OCT) {
        const int oc = n0 + tid;
        s_so[tid] = p.so ? p.so[(size_t)b * p.N + oc] : 1.f;
        s_bias[tid] = p.bias ? p.bias[oc] : 0.f;
    }
    __syncthreads();

    // The weight slab of the NEXT item: rows (half, tap, kg) of 64 units, one LDS-DMA instruction each, dealt round-robin to the eight
    // multiplying waves (4 or 5 each) at the start of their MFMA phase.  The instruction is issued from an asm statement: the
    // compiler-tracked builtin makes every later LDS read of the wave (the fragment reads of THIS item) wait for the DMA first, which
    // puts its latency at the head of each MFMA phase.  Untracked, its completion is counted by hand: vmcnt(0) before the barrier.
#ifdef GC_SINGLE
    constexpr int ROWS = NTAP * KG;
#else
    constexpr int ROWS = 2 * NTAP * KG;
#endif
    // one instruction moves 64 units = 64 / OCT consecutive rows (rows are adjacent in LDS; the halves hold an even number of rows)
    constexpr int RPI = 64 / OCT, INSTR = ROWS / RPI;
    static_assert(ROWS % RPI == 0 && (NTAP * KG) % RPI == 0, "row groups do not straddle the hi / lo halves");
    // GC_WS_DMA_STAGER = 1: the four staging waves issue it instead (right after their patch loads, vmcnt(0) before their barrier)
    // GC_WS_DMA_HALF (round 6): only multiplying waves 0..3 -- one per SIMD -- issue the slab.  A wave-level LDS-DMA instruction costs 100-185 cycles to issue next to
    // fragment reads, and with all eight waves issuing their share right after the barrier BOTH waves of every SIMD were busy with it for ~750 of an item's ~8 900
    // cycles while the matrix pipe idled (tools/ws_trace.py, profiles/ws_trace_r06_m.log); now the partner wave starts its MFMAs at once and has the pipe to itself meanwhile.
    constexpr int DMA_WAVES = (GC_WS_DMA_STAGER || GC_WS_DMA_HALF) ? 4 : 8;
    const int dma_wave = GC_WS_DMA_STAGER ? (wave - 8) & 3 : wave;
    const bool dma_mine = GC_WS_DMA_STAGER || wave < DMA_WAVES;          // (wave-uniform)
    auto weights = [&](int k0, int buf) {
        if (!dma_mine) return;
        uint4* const base = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < (INSTR + DMA_WAVES - 1) / DMA_WAVES; ++j) {
            const int q = dma_wave + DMA_WAVES * j;
            if (DMA_WAVES * j + DMA_WAVES - 1 < INSTR || q < INSTR) {
                const int r0 = q * RPI;                                   // first row of the group (wave-uniform)
                const int half = r0 / (NTAP * KG), rr0 = r0 % (NTAP * KG);
                const int rr = rr0 + lane / OCT;                          // this lane's row
                const int t = rr / KG, kg = rr % KG;
                const uint4* src = (half ? a.wl : a.wh) + ((size_t)(t * a.kgroups + k0 / 8 + kg) * p.N + n0 + lane % OCT);
                glds16(src, base + half * C::WUNITS + rr0 * OCT);
            }
        }
    };
    if (wave >= 8) {
        // ---------------- staging waves ----------------
        if (GC_WS_STAGER_PRIO) __builtin_amdgcn_s_setprio(GC_WS_STAGER_PRIO);
        const int st = tid - 512;
        const int kgl = __builtin_amdgcn_readfirstlane(st >> 7), tb = st & 127;     // waves 8, 9: channel group 0; waves 10, 11: group 1
        const float* xb = p.x + (size_t)b * p.K * chan;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)p.K * chan * 4u);
        const int lead = gc::pos_mod(p.pad_x, 32);           // patch rows start `lead` floats before a 128-byte boundary
        // Two register sets: the loads of item i + 2 are in flight while item i + 1 is converted and written, so the load latency
        // is never on this wave's critical path (which is then ~300 vector instructions + 16 ds_write_b128 per item).
        uint4 pa[C::NT][8], pb[C::NT][8];
        auto loads = [&](uint4 (&preg)[C::NT][8], int tile, int k0) {
            const int iy0 = (tile / p.tiles_x) * TPH - p.pad_y, ix0 = (tile % p.tiles_x) * (32 * CB) - p.pad_x;
#pragma unroll
            for (int j = 0; j < C::NT; ++j) {
                const typename C::Task tk = C::task_of(tb + 128 * j, lead);
                const int iy = iy0 + tk.row, ix = ix0 + tk.col;
                const bool ok = tk.used > 0 && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;      // rows past the last tile: zeros
                const unsigned boff = ok ? (unsigned)(iy * p.in_w + ix) * 4u : OOB;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    preg[j][q] = GC_WS_NT_LOAD ? __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)boff, (int)__builtin_amdgcn_readfirstlane((unsigned)(k0 + kgl * 8 + q) * chan * 4u), 2))
                                               : buf_load_u128(rx, boff, (unsigned)(k0 + kgl * 8 + q) * chan * 4u);     // channels past K: beyond the descriptor, zeros
            }
        };
        auto convert = [&](uint4 (&preg)[C::NT][8], int tile, int k0, int buf) {
            uint4* const p_h = smem + buf * STAGE + 2 * C::WUNITS;
            uint4* const p_l = p_h + C::PUNITS;
            const int kk = min(k0, p.K - KCB);          // the item past the last one is converted into a stage nobody reads
            const float4 sa = *reinterpret_cast<const float4*>(&s_si[kk + kgl * 8]), sb = *reinterpret_cast<const float4*>(&s_si[kk + kgl * 8 + 4]);
            const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
            const int ix0 = (tile % p.tiles_x) * (32 * CB) - p.pad_x;
            const bool ragged_rows = (p.in_w & 3) != 0 && ix0 + lead + 32 * C::SEG_M + 4 > p.in_w;
            auto body = [&](auto masked, auto scaled) {
#pragma unroll
                for (int j = 0; j < C::NT; ++j) {
                    const typename C::Task tk = C::task_of(tb + 128 * j, lead);
                    const int inrow = p.in_w - (ix0 + tk.col);
                    const int rbase = kgl * PLANE + tk.row * C::RP;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const unsigned raw = i == 0 ? preg[j][q].x : (i == 1 ? preg[j][q].y : (i == 2 ? preg[j][q].z : preg[j][q].w));
                            v[q] = (!decltype(masked)::value || i < inrow) ? __uint_as_float(raw) : 0.f;
                        }
                        uint4 h, l;
                        split8s<decltype(scaled)::value>(v, sc, &h, &l);
                        if (i < tk.used) {
                            const int u = rbase + C::ucol(tk.col + i);
                            p_h[u] = h;
                            GC_LO(p_l[u] = l;)
                        }
                    }
                }
            };
            // without modulation (every layer of D) the multiply by one is not issued
            if (p.si) { if (ragged_rows) body(std::true_type{}, std::true_type{}); else body(std::false_type{}, std::true_type{}); }
            else      { if (ragged_rows) body(std::true_type{}, std::false_type{}); else body(std::false_type{}, std::false_type{}); }
        };
        auto advance = [&](int& tile, int& k0) { k0 += KCB; if (k0 >= p.K) { k0 = 0; tile += tstep; } };
        int t0 = tile_begin, k0 = 0;                    // item 0 -> set A
        loads(pa, t0, k0);
        int t1 = t0, k1 = k0; advance(t1, k1);          // item 1 -> set B
        loads(pb, t1, k1);
        if (GC_WS_DMA_STAGER) weights(0, 0);
        convert(pa, t0, k0, 0);
        if (GC_WS_DMA_STAGER) wait_staged_loads();
        __syncthreads();
        // interval `it`: the multiplying waves work on item it; item it + 1 is converted here, item it + 2 is fetched
        for (int it = 0; it < items; it += 2) {
            int t2 = t1, k2 = k1; advance(t2, k2);
            GC_TR(1, 1);
            if (!(GC_WS_ABL & 1)) { loads(pa, t2, k2); GC_TR(1, 2); if (GC_WS_TRACE) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C::NT * 8) : "memory"); GC_TR(1, 3); } if (GC_WS_DMA_STAGER) weights(k1, 1); convert(pb, t1, k1, 1); }
            if (GC_WS_DMA_STAGER) wait_staged_loads();
            if (GC_WS_TRACE) { __builtin_amdgcn_s_waitcnt(0xC07F); GC_TR(1, 4); }
            __syncthreads();
            GC_TR(1, 5);
            if (it + 1 >= items) break;
            t1 = t2; k1 = k2; advance(t1, k1);
            GC_TR(1, 1);
            if (!(GC_WS_ABL & 1)) { loads(pb, t1, k1); GC_TR(1, 2); if (GC_WS_TRACE) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C::NT * 8) : "memory"); GC_TR(1, 3); } if (GC_WS_DMA_STAGER) weights(k2, 0); convert(pa, t2, k2, 0); }
            if (GC_WS_DMA_STAGER) wait_staged_loads();
            if (GC_WS_TRACE) { __builtin_amdgcn_s_waitcnt(0xC07F); GC_TR(1, 4); }
            __syncthreads();
            GC_TR(1, 5);
        }
#if GC_WS_TRACE
        if (tr_on) gc_ws_trace[1][511] = tr_n[1];
#endif
        return;
    }

    // ---------------- multiplying waves ----------------
    const int wave_row = (wave / CB) * 2, wave_col = (wave % CB) * 32;        // this wave: rows wave_row, wave_row + 1 of column block wave % CB
    f32x16 acc[WOC][WPX];
#pragma unroll
    for (int i = 0; i < WOC; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int boff[WPX];
#pragma unroll
    for (int j = 0; j < WPX; ++j) boff[j] = hi * PLANE + (wave_row + j) * C::RP + wave_col;
    const int aoff = hi * OCT + l31;

    const unsigned oplane = (unsigned)(p.out_h * p.out_w) * 4u;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + (size_t)b * p.N * p.out_h * p.out_w, (unsigned)p.N * oplane);
    const EpilogueConsts ec = epilogue_consts(p);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.residual ? p.residual + (size_t)b * p.N * p.out_h * p.out_w : p.y, p.residual ? (unsigned)p.N * oplane : 0u);
    auto finish_tile = [&](int tile) {
        const int qy0 = (tile / p.tiles_x) * TPH, qx0 = (tile % p.tiles_x) * (32 * CB);
        const int nb = opaque_s(n0);
        // every load of the epilogue (noise, residual) is issued before the first store: a load between two stores waits for the stores
        unsigned voff[WPX];
        float nz[WPX];
#pragma unroll
        for (int j = 0; j < WPX; ++j) {
            const int qy = qy0 + wave_row + j, qx = qx0 + wave_col + l31;
            const bool inside = qy < p.out_h && qx < p.out_w;
            voff[j] = inside ? (unsigned)(qy * p.out_w + qx) * 4u + (unsigned)(4 * hi) * oplane : OOB;
            nz[j] = (p.noise && inside) ? p.noise[((size_t)b * p.out_h + qy) * p.out_w + qx] : 0.f;
        }
        // Two phases, no control flow inside either.  (Round 5: the first version evaluated the epilogue AT each store and chose between the
        // residual / plain forms per element; the compiler turned that into one basic block per element -- two ds_read_b32 of out_scale / bias,
        // each waited for on the spot, ~12 vector instructions, the store, a branch: 128 exposed LDS round trips per lane and tile on the
        // MULTIPLYING waves, all eight of them at the same moment, next to 108 MFMAs per chunk.  tools/kernel_regs.py / the disassembly show it.)
        // Phase 1: every accumulator becomes its final value in place.  The 16 out_scale / bias values of a 32-channel block that this lane's
        // registers belong to are four runs of four consecutive channels: four 16-byte LDS reads each, issued together.
        // (with a residual its loads belong to the same block as the arithmetic: issued from a separate block after phase 1 their latency was
        // exposed once per tile -- 512 -> 512 @64^2 with scale + residual 199 -> 217 us in the first version of this change)
        auto phase1 = [&](auto with_res) {
#pragma unroll
            for (int i = 0; i < WOC; ++i) {
                float so16[16], bi16[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 s4 = *reinterpret_cast<const float4*>(&s_so[i * 32 + 8 * q + 4 * hi]);
                    so16[4 * q] = s4.x; so16[4 * q + 1] = s4.y; so16[4 * q + 2] = s4.z; so16[4 * q + 3] = s4.w;
                    if (EPK == 0) {
                        const float4 b4 = *reinterpret_cast<const float4*>(&s_bias[i * 32 + 8 * q + 4 * hi]);
                        bi16[4 * q] = b4.x; bi16[4 * q + 1] = b4.y; bi16[4 * q + 2] = b4.z; bi16[4 * q + 3] = b4.w;
                    }
                }
#pragma unroll
                for (int j = 0; j < WPX; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // EPK 1 (out_scale and / or residual only): the product is rounded on its own, as conv_epilogue rounds it; values differ from
                        // the full epilogue's only in the sign of an exact zero (it adds +0 for the absent bias)
                        float v = EPK == 0 ? conv_epilogue(ec, acc[i][j][r], so16[r], bi16[r], nz[j]) : plain_mul(acc[i][j][r], so16[r]);
                        if (decltype(with_res)::value) v = plain_sum(v, buf_load_f32(rres, voff[j], (unsigned)(nb + i * 32 + (r & 3) + 8 * (r >> 2)) * oplane));
                        acc[i][j][r] = v;
                    }
            }
        };
        if (EPK < 2) phase1(std::integral_constant<bool, RES>{});
        // Phase 2: nothing but stores
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int i = 0; i < WOC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ocs = nb + i * 32 + (r & 3) + 8 * (r >> 2);
                    const float v = acc[i][j][r];
                    if (!(GC_WS_ABL & 8) || v == 12345.678f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, (int)voff[j], (int)((unsigned)ocs * oplane), GC_CONV_ST_AUX);
                    acc[i][j][r] = 0.f;
                }
    };
    int tile_c = tile_begin, k0_c = 0;
    if (!GC_WS_DMA_STAGER) { weights(0, 0); wait_staged_loads(); }
    __syncthreads();                 // stage 0 is staged
    // GC_WS_EARLY_DMA (round 5): the weight slab of item it + 2 is requested right AFTER the barrier that ends item it -- its stage is free from
    // that moment -- and BEFORE the stores of a finished tile, instead of at the top of item it + 1 after them.  vmcnt counts loads and stores in
    // issue order, so with the request after the stores the `vmcnt(0)` in front of the next barrier also waited for the whole tile's stores to
    // reach memory -- once per tile, with only one or two items per tile at 32 / 64 input channels to hide it behind.  With the request older than
    // the stores the wait is counted: `vmcnt(S)`, S = the stores a lane issues per tile (at most 63), lets them stay in flight across the barrier.
    // The barrier is then the raw instruction (the `__syncthreads()` fence would drain the stores again).
    // GC_WS_DMA_MID (round 6 experiment): the slab of item it + 1 is requested INSIDE the MFMA phase of item it -- waves 0..3 in front of tap 1, waves 4..7 in front of
    // tap 5 -- so that the two waves of a SIMD are never both busy issuing LDS-DMA instructions (the ~750 idle cycles per item of profiles/ws_trace_r06.md)
    constexpr bool MID = GC_WS_DMA_MID && !GC_WS_DMA_STAGER && !(GC_WS_ABL & 2);
    constexpr bool EARLY = GC_WS_EARLY_DMA && !MID && !GC_WS_DMA_STAGER && !(GC_WS_ABL & (2 | 8));      // (the no-store ablation makes the stores conditional: no counted wait)
    constexpr int NSTORES = WOC * WPX * 16 > 63 ? 63 : WOC * WPX * 16;
    // (the counted wait below is only right while finish_tile issues exactly WOC * WPX * 16 unconditional stores per lane AFTER the newest request,
    //  which is why the no-store ablation GC_WS_ABL & 8 is excluded from EARLY)
    bool stored = false;             // the previous item ended a tile: its stores were issued after the newest weight request
    if (EARLY) weights(KCB < p.K ? KCB : 0, 1);                       // item 1
    for (int it = 0; it < items; ++it) {
        if (!EARLY && !MID && !(GC_WS_ABL & 2) && !GC_WS_DMA_STAGER) weights(k0_c + KCB < p.K ? k0_c + KCB : 0, (it + 1) & 1);          // after the last item: a valid slab into a stage nobody reads
        const uint4* const wl_h = smem + (it & 1) * STAGE;
        const uint4* const wl_l = wl_h + C::WUNITS;
        const uint4* const p_h = wl_l + C::WUNITS;
        const uint4* const p_l = p_h + C::PUNITS;
        GC_TR(0, 1);
        __builtin_amdgcn_s_setprio(GC_MFMA_PRIO);
        // Fragment double buffer: the eight ds_read_b128 of tap t + 1 are issued BEFORE the twelve MFMAs of tap t (the scheduling
        // barriers pin that order; left alone the compiler sinks every read to 1-3 MFMAs before its use, far less than the LDS latency).
        bf16x8 fa[2][2 * WOC], fb[2][2 * WPX];          // [set][hi 0..1, lo 0..1]
        auto load_tap = [&](int t, int set) {
            const int jy = t / KS, jx = t % KS;
            const int wbase = t * KG * OCT + aoff;
            const int pbase = jy * C::RP + l31 + jx;
#pragma unroll
            for (int i = 0; i < WOC; ++i) {
                const uint4 uh = wl_h[wbase + i * 32];
                fa[set][i] = *reinterpret_cast<const bf16x8*>(&uh);
                GC_LO(const uint4 ul = wl_l[wbase + i * 32]; fa[set][WOC + i] = *reinterpret_cast<const bf16x8*>(&ul);)
            }
            if (GC_WS_SHIFT && KS == 3 && jx > 0) {
                // The patch fragment of tap (jy, jx) is that of (jy, jx - 1) one pixel on: lane l wants what lane l + 1 holds.  A whole-wave DPP
                // shift (gfx9: wave_shl) moves it between registers -- four v_mov_b32_dpp per fragment -- instead of a second and third LDS
                // read of the same units; only the last pixel of the 32 (lanes 31 and 63: the shift brings the other channel group's / nothing)
                // is fetched from LDS, by an exec-masked read.
#pragma unroll
                for (int f = 0; f < (2 * WPX); ++f) {
#ifdef GC_SINGLE
                    if (f >= WPX) continue;
#endif
                    const uint4 prev = *reinterpret_cast<const uint4*>(&fb[set ^ 1][f]);
                    uint4 cur;
                    cur.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)prev.x, 0x130, 0xf, 0xf, false);
                    cur.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)prev.y, 0x130, 0xf, 0xf, false);
                    cur.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)prev.z, 0x130, 0xf, 0xf, false);
                    cur.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)prev.w, 0x130, 0xf, 0xf, false);
                    if (l31 == 31) cur = (f < WPX ? p_h : p_l)[pbase + boff[f % WPX]];
                    fb[set][f] = *reinterpret_cast<const bf16x8*>(&cur);
                }
            } else {
#pragma unroll
                for (int j = 0; j < WPX; ++j) {
                    const uint4 uh = p_h[pbase + boff[j]];
                    fb[set][j] = *reinterpret_cast<const bf16x8*>(&uh);
                    GC_LO(const uint4 ul = p_l[pbase + boff[j]]; fb[set][WPX + j] = *reinterpret_cast<const bf16x8*>(&ul);)
                }
            }
        };
        load_tap(0, 0);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            if (t + 1 < NTAP && !(GC_WS_ABL & 4)) load_tap(t + 1, (t + 1) & 1);
            if (MID && (NTAP == 1 || t == 1 || t == 5)) {
                if (NTAP == 1 || (t == 1) == (wave < 4)) weights(k0_c + KCB < p.K ? k0_c + KCB : 0, (it + 1) & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < WOC; ++i)
#pragma unroll
                for (int j = 0; j < WPX; ++j) { constexpr int fs = (GC_WS_ABL & 4) ? 0 : 1; GC_MFMA3(acc[i][j], fa[t & fs][i], fa[t & fs][WOC + i], fb[t & fs][j], fb[t & fs][WPX + j]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        GC_TR(0, 2);
        if (EARLY) {
            // the rows of item it + 1 were requested one item ago, before any store still in flight
            // (a wave that issued no slab rows has nothing to wait for: its stores stay in flight, the issuing waves' waits + the barrier order the slab)
            if (dma_mine) {
                if (stored) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSTORES) : "memory");
                else        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this wave's LDS reads have returned
            GC_TR(0, 3);
            __builtin_amdgcn_s_barrier();
            GC_TR(0, 4);
            // item it + 2 goes into the stage item it has just left: (it + 2) chunks on from the start, modulo the chunks of a tile
            const int k2 = k0_c + 2 * KCB;
            weights(k2 < p.K ? k2 : (k2 - p.K < p.K ? k2 - p.K : 0), it & 1);
            stored = false;
        } else {
        if (!GC_WS_DMA_STAGER) wait_staged_loads();         // the LDS-DMA rows of this wave have landed (they were issued a whole MFMA phase ago)
        __syncthreads();             // this stage may be rewritten from the next item on; the other one is staged
        }
        k0_c += KCB;
        if (k0_c >= p.K) { GC_TR(0, 5); finish_tile(tile_c); GC_TR(0, 6); k0_c = 0; tile_c += tstep; stored = true; }
    }
#if GC_WS_TRACE
    if (tr_on) gc_ws_trace[0][511] = tr_n[0];
#endif
    // the slabs requested for the two items past the last one (valid rows into stages nobody reads) must have landed before the wave ends and the
    // LDS is handed to the next workgroup: costs nothing, the wave is ending (round-5 advisor finding)
    if (EARLY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient in split-bf16:  dW[tap][k][n] = sum_px X[k][px + tap] * dY[n][px]   (down = 1)
// The MFMA reduction index is the PIXEL, so a lane's fragment is 8 consecutive pixels of one channel.  Both
// tiles sit in LDS pixel-contiguous as 16-byte units of 8 pixels (hi and lo parts).  A horizontal tap shift of
// tx pixels is a funnel shift over two neighbouring units (4 v_perm for tx = 1, register moves for tx = 2) --
// every ds_read_b128 stays 16-byte aligned.  Each wave owns a 32k x 32n block for all taps (144 accumulators).
#if defined(GC_ABL) && GC_ABL == 3      // dev ablation: no global loads in the weight-gradient staging
#define WG_LOAD(r, off, imm) make_uint4((off), (off) + 1u, (off) + 2u, (off) + 3u)
#else
#define WG_LOAD(r, off, imm) buf_load_u128(r, off, imm)
#endif
// non-temporal variant (dev knob GC_WG_NT_LOAD: 1 = every weight-gradient load, 2 = only where the operands are read once: K = N = 32).
// Measured much SLOWER in both forms (round 4, B = 4: 32 -> 32 @1024^2 344 -> 555 us; with 1 also 64 -> 64 @512^2 267 -> 427, stride 2 233 -> 426):
// the halo rows of a tile and the neighbouring pixel splits re-read the lines the hint evicts.
#ifndef GC_WG_NT_LOAD
#define GC_WG_NT_LOAD 0
#endif
__device__ __forceinline__ uint4 buf_load_u128_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)__builtin_amdgcn_readfirstlane(soff), 2));
}
#define WG_LOAD1(once, r, off, imm) ((GC_WG_NT_LOAD == 1 || (GC_WG_NT_LOAD == 2 && (once))) ? buf_load_u128_nt(r, off, imm) : WG_LOAD(r, off, imm))
// XCD-aware block order for the weight-gradient grids (k blocks x n blocks x pixel splits).  Workgroups go to the eight XCDs round-robin
// in linear block order, so the 8 x 8 (k, n) blocks of ONE pixel split -- which all stream the same X and dY tiles -- land on eight
// different L2s and every tile crosses the fabric eight times.  Re-deal the linear ids so that each XCD gets a contiguous range of
// (x fastest, then y, then z): the blocks that share operands then share one L2.
struct WgBlock { int x, y, z; };
template <bool XCD>
__device__ __forceinline__ WgBlock wg_block() {
  if (XCD) {
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    unsigned l = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if (total % 8 == 0) l = (l % 8) * (total / 8) + l / 8;
    return {(int)(l % gx), (int)((l / gx) % gy), (int)(l / (gx * gy))};
  }
    return {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
}

struct WgArgs {
    const float* x; const float* dy; const float* si; const float* so; float* ws;
    int B, K, N, in_h, in_w, out_h, out_w, pad_y, pad_x;
    int tiles_x, tiles_y, tiles_per_split;
    int x_pitch;          // floats between the rows of x (wgrad_bf16x3_s2_kernel; in_w when dense)
    int spb;              // > 0: per-sample mode -- split z works on sample z / spb only (its tiles spb apart), so ws[z] is a partial sum of ONE sample
};

__device__ __forceinline__ uint4 shift_px(const uint4 a, const uint4 b, int tx) {
    if (tx == 0) return a;
    if (tx == 2) return make_uint4(a.y, a.z, a.w, b.x);
    return make_uint4(__builtin_amdgcn_alignbit(a.y, a.x, 16), __builtin_amdgcn_alignbit(a.z, a.y, 16),
                      __builtin_amdgcn_alignbit(a.w, a.z, 16), __builtin_amdgcn_alignbit(b.x, a.w, 16));
}

template <bool SCALED = true>
__device__ __forceinline__ void split8(const float (&v)[8], float scale, uint4* h, uint4* l) {
#if defined(GC_ABL) && GC_ABL == 2      // dev ablation: staging without the conversions
    *h = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    *l = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
    return;
#endif
#if GC_PAIR_SPLIT
    unsigned hh[4], ll[4];
#pragma unroll
    for (int q = 0; q < 8; q += 2) {       // pairs: see cvt_pk_bf16 (the same bits as the element-by-element form below)
        const float f0 = SCALED ? v[q] * scale : v[q], f1 = SCALED ? v[q + 1] * scale : v[q + 1];
        const unsigned pk = cvt_pk_bf16(f0, f1);
        const float t0 = __uint_as_float(pk << 16), t1 = __uint_as_float(pk & 0xffff0000u);
        float d0, d1;
        asm("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(f0), "v"(t0));      // plain, not packed: v_pk_add_f32 stalls the matrix pipe (profiles/pmc_r01.md); +3..4 % at >= 64 channels
        asm("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(f1), "v"(t1));
        hh[q / 2] = pk;
        ll[q / 2] = cvt_pk_bf16(d0, d1);
    }
    *h = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    *l = make_uint4(ll[0], ll[1], ll[2], ll[3]);
#else
    bf16x8 hh, ll;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float f = SCALED ? v[q] * scale : v[q];
        const __bf16 t = (__bf16)f;
        hh[q] = t;
        const float tf = (float)t;
        float dlo;
        asm("v_sub_f32 %0, %1, %2" : "=v"(dlo) : "v"(f), "v"(tf));      // plain, not packed: v_pk_add_f32 stalls the matrix pipe (profiles/pmc_r01.md); +3..4 % at >= 64 channels
        ll[q] = (__bf16)dlo;
    }
    *h = *reinterpret_cast<uint4*>(&hh);
    *l = *reinterpret_cast<uint4*>(&ll);
#endif
}

template <int WK, int WN, int WP, int TR, int KS>
struct WgCfg {
    static constexpr int KT = WK * 32, NTL = WN * 32;
    static constexpr int PH = TR + KS - 1;
    static constexpr int XU = KS == 3 ? 5 : 4, YU = 4;            // 8-pixel units per patch row / dY row
    static constexpr int CSX = (PH * XU) | 1, CSY = (TR * YU) | 1;   // odd unit strides between channels: conflict-free b128 reads
    static constexpr int NXU = KT * PH * XU, NYU = NTL * TR * YU;
    static constexpr int NPX = (NXU + 255) / 256, NPY = (NYU + 255) / 256;
    static constexpr int RED_UNITS = (WP - 1) * WK * WN * 16 * 64 / 4;      // cross-wave reduction scratch (floats / 4)
    static constexpr int SMEM_UNITS = cmax(2 * (KT * CSX + NTL * CSY), RED_UNITS);
    static constexpr int NT = KS * KS;
};

template <int WK, int WN, int WP, int TR, int KS>
__global__ __launch_bounds__(256, 2) void wgrad_bf16x3_kernel(WgArgs p) {
    using C = WgCfg<WK, WN, WP, TR, KS>;
    static_assert(WK * WN * WP == 4, "4 waves per workgroup");
    static_assert((2 * TR) % WP == 0, "pixel steps split evenly over the pixel waves");
    constexpr int KT = C::KT, NTL = C::NTL, PH = C::PH, XU = C::XU, YU = C::YU, NT = C::NT;
    __shared__ uint4 smem[C::SMEM_UNITS];
    uint4* xh = smem;
    uint4* xl = xh + KT * C::CSX;
    uint4* yh = xl + KT * C::CSX;
    uint4* yl = yh + NTL * C::CSY;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wp = wave % WP, wn = (wave / WP) % WN, wk = wave / (WP * WN);
    const WgBlock blk = wg_block<false>();      // measured: no gain at stride 1 (same-box A/B within +-3 %)
    const int k0 = blk.x * KT, n0 = blk.y * NTL, split = blk.z;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int tiles_per_sample = p.tiles_x * p.tiles_y;
    const int total_tiles = tiles_per_sample * p.B;
    // GC_WG_STRIDED: tiles run ACROSS the rows and the tiles of one split are gridDim.z apart, so the resident workgroups read a band
    // of neighbouring rows (contiguous in DRAM, halo rows shared through L2) instead of one 32-column strip each, all over the batch.
    const int sb = p.spb ? split / p.spb : 0;        // per-sample mode (gc_conv2d_wgrad_samples_*): the splits of one sample walk that sample's tiles only
    const int tstep = p.spb ? p.spb : (GC_WG_STRIDED ? (int)gridDim.z : 1);
    const int t_begin = p.spb ? sb * tiles_per_sample + (split - sb * p.spb) : (GC_WG_STRIDED ? split : split * p.tiles_per_split);
    const int t_end = p.spb ? (sb + 1) * tiles_per_sample : (GC_WG_STRIDED ? total_tiles : min(total_tiles, t_begin + p.tiles_per_split));
    const int xchan = p.in_h * p.in_w, ychan = p.out_h * p.out_w;

    // Staging is kept LEAN: with two workgroups per CU the vector ALUs (index arithmetic, masks, conversions), not the
    // matrix pipes, bound this kernel.  Everything that depends only on the lane is computed once -- the byte offset of each
    // staged unit inside a sample and a packed descriptor (LDS unit offset, patch row, unit column, channel) -- so a tile costs
    // ~6 vector instructions per unit to address and ~40 to convert.  The per-sample scales sit in an LDS table (refilled
    // when a split crosses into the next sample).
    __shared__ float s_scale[KT + NTL];
    int b_tab = -1;
    float4 xreg[C::NPX][2], yreg[C::NPY][2];
    unsigned xdesc[C::NPX], ydesc[C::NPY];       // LDS unit offset | unit column << 16 | patch row << 20 | channel << 24 | idle lane << 31
    constexpr unsigned OUTSIDE = 0x80000000u;    // beyond every buffer
#pragma unroll
    for (int j = 0; j < C::NPX; ++j) {
        const int u = tid + 256 * j;
        const int xu = u % XU, row = u / XU;
        const int r = row % PH, kk = min(row / PH, KT - 1);
        const bool live = u < C::NXU && k0 + kk < p.K;
        xdesc[j] = (unsigned)(kk * C::CSX + r * XU + xu) | (unsigned)xu << 16 | (unsigned)r << 20 | (unsigned)kk << 24 | (live ? 0u : OUTSIDE);
    }
#pragma unroll
    for (int j = 0; j < C::NPY; ++j) {
        const int u = tid + 256 * j;
        const int yu = u % YU, row = u / YU;
        const int r = row % TR, nn = min(row / TR, NTL - 1);
        const bool live = u < C::NYU && n0 + nn < p.N;
        ydesc[j] = (unsigned)(nn * C::CSY + r * YU + yu) | (unsigned)yu << 16 | (unsigned)r << 20 | (unsigned)nn << 24 | (live ? 0u : OUTSIDE);
    }
    const unsigned xbytes = (unsigned)p.K * xchan * 4u, ybytes = (unsigned)p.N * ychan * 4u;
    auto prefetch = [&](int tile) {
        const int b = tile / tiles_per_sample;
        const int rem = tile - b * tiles_per_sample;
        const int oy0 = GC_WG_STRIDED ? (rem / p.tiles_x) * TR : (rem % p.tiles_y) * TR, ox0 = GC_WG_STRIDED ? (rem % p.tiles_x) * 32 : (rem / p.tiles_y) * 32;      // see the tile loop
#if defined(GC_ABL) && (GC_ABL == 4 || GC_ABL == 5)      // dev ablation (wrong results): the patch rows start on the tile's own 128-byte line instead of one pixel left of it
        const int iy0 = oy0 - p.pad_y, ix0 = ox0;
#else
        const int iy0 = oy0 - p.pad_y, ix0 = ox0 - p.pad_x;
#endif
        const int xoff = (k0 * xchan + iy0 * p.in_w + ix0) * 4, yoff = (n0 * ychan + oy0 * p.out_w + ox0) * 4;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)b * p.K * xchan, xbytes);
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.dy + (size_t)b * p.N * ychan, ybytes);
#pragma unroll
        for (int j = 0; j < C::NPX; ++j) {
            // rows start at odd offsets (pad - 1, 1025-wide planes): the 16-byte loads are only 4-byte aligned, which
            // buffer_load_dwordx4 accepts; each dword is range-checked separately.  The very first unit of a sample
            // (channel 0, row 0, left halo) would start at a NEGATIVE offset, which the range check rejects as a whole:
            // it is loaded from offset 0 and patched after the commit (fix_first_unit).
            const unsigned d = (unsigned)opaque((int)xdesc[j]);       // opaque: nothing derived from the descriptor may be hoisted out of the tile loop (registers)
            const int r = (int)((d >> 20) & 15u);
            const int lin = (int)((d >> 24) & 63u) * (xchan * 4) + r * (p.in_w * 4) + (int)((d >> 16) & 15u) * 32 + xoff;
#if defined(GC_ABL) && GC_ABL == 5      // ... and the fifth (halo) unit of every row is not fetched: exactly one 128-byte line per row
            const unsigned off = ((int)d >= 0 && (unsigned)(iy0 + r) < (unsigned)p.in_h && ((d >> 16) & 15u) < 4u) ? (unsigned)max(lin, 0) : OUTSIDE;
#else
            const unsigned off = ((int)d >= 0 && (unsigned)(iy0 + r) < (unsigned)p.in_h) ? (unsigned)max(lin, 0) : OUTSIDE;
#endif
            xreg[j][0] = __builtin_bit_cast(float4, WG_LOAD1(WK * WN == 1, rx, off, 0));
            xreg[j][1] = __builtin_bit_cast(float4, WG_LOAD1(WK * WN == 1, rx, off, 16));
        }
#pragma unroll
        for (int j = 0; j < C::NPY; ++j) {
            const unsigned d = (unsigned)opaque((int)ydesc[j]);
            const int r = (int)((d >> 20) & 15u);
            const int lin = (int)((d >> 24) & 63u) * (ychan * 4) + r * (p.out_w * 4) + (int)((d >> 16) & 15u) * 32 + yoff;
            const unsigned off = ((int)d >= 0 && oy0 + r < p.out_h) ? (unsigned)lin : OUTSIDE;
            yreg[j][0] = __builtin_bit_cast(float4, WG_LOAD1(WK * WN == 1, ry, off, 0));
            yreg[j][1] = __builtin_bit_cast(float4, WG_LOAD1(WK * WN == 1, ry, off, 16));
        }
    };
    // `edge_t`: the column masks exist only in the variant that border tiles take.  (Round 5: written as a per-lane `if (unit straddles a border)`
    // the compiler predicated the masks for EVERY lane and tile -- two compares, a scalar and, a select per value: 448 of the 1 021 vector
    // instructions of the conversion phase; the tile-uniform switch in commit() makes it a scalar branch that 30 of 32 tile columns skip.)
    auto unit8 = [&](auto scaled_t, auto edge_t, const float4 (&r)[2], int col0, int width, float scale, uint4* h, uint4* l) {
        float v[8] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w};
        if (decltype(edge_t)::value) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (col0 + q >= 0 && col0 + q < width) ? v[q] : 0.f;
        }
        split8<decltype(scaled_t)::value>(v, scale, h, l);
    };
    auto commit = [&](int tile) {
        const int b = tile / tiles_per_sample;
        const int rem = tile - b * tiles_per_sample;
        const int oy0 = GC_WG_STRIDED ? (rem / p.tiles_x) * TR : (rem % p.tiles_y) * TR, ox0 = GC_WG_STRIDED ? (rem % p.tiles_x) * 32 : (rem / p.tiles_y) * 32;
        const bool scaled = p.si != nullptr || p.so != nullptr;
        if (scaled && b != b_tab) {          // uniform: every lane of the workgroup sees the same tile
            __syncthreads();
            if (tid < KT) s_scale[tid] = p.si ? p.si[(size_t)b * p.K + min(k0 + tid, p.K - 1)] : 1.f;
            else if (tid < KT + NTL) s_scale[tid] = p.so ? p.so[(size_t)b * p.N + min(n0 + tid - KT, p.N - 1)] : 1.f;
            __syncthreads();
            b_tab = b;
        }
        wait_staged_loads();
        auto items = [&](auto scaled_t, auto edge_t) {          // without modulation (every layer of D) the multiply by one is not issued: it is packed fp32, which stalls the matrix pipe
            constexpr bool SC = decltype(scaled_t)::value;
#pragma unroll
            for (int j = 0; j < C::NPX; ++j) {
                const unsigned d = (unsigned)opaque((int)xdesc[j]);       // opaque: nothing derived from the descriptor may be hoisted out of the tile loop (registers)
                const float sc = SC ? s_scale[(d >> 24) & 63u] : 1.f;
                uint4 h, l;
                unit8(scaled_t, edge_t, xreg[j], ox0 - p.pad_x + 8 * (int)((d >> 16) & 15u), p.in_w, sc, &h, &l);      // rows / channels outside the image were loaded as zeros
                if (256 * (j + 1) <= C::NXU || tid + 256 * j < C::NXU) { xh[d & 0xffffu] = h; GC_LO(xl[d & 0xffffu] = l;) }
            }
#pragma unroll
            for (int j = 0; j < C::NPY; ++j) {
                const unsigned d = (unsigned)opaque((int)ydesc[j]);
                const float sc = SC ? s_scale[KT + ((d >> 24) & 63u)] : 1.f;
                uint4 h, l;
                unit8(scaled_t, edge_t, yreg[j], ox0 + 8 * (int)((d >> 16) & 15u), p.out_w, sc, &h, &l);
                if (256 * (j + 1) <= C::NYU || tid + 256 * j < C::NYU) { yh[d & 0xffffu] = h; GC_LO(yl[d & 0xffffu] = l;) }
            }
        };
        // tile-uniform: does any staged unit of this tile reach over the left / right image border?
        const bool edge = ox0 - p.pad_x < 0 || ox0 - p.pad_x + 8 * XU > p.in_w || ox0 + 8 * YU > p.out_w;
        if (scaled) { if (edge) items(std::true_type{}, std::true_type{}); else items(std::true_type{}, std::false_type{}); }
        else        { if (edge) items(std::false_type{}, std::true_type{}); else items(std::false_type{}, std::false_type{}); }
        // the one unit per sample that was fetched from offset 0 instead of -pad (see prefetch): channel 0, image row 0, left halo
        if (k0 == 0 && ox0 == 0 && p.pad_x > 0 && oy0 < PH && oy0 - p.pad_y <= 0) {          // uniform and rare
            __syncthreads();
            if (tid == 0) {
                const int r = p.pad_y - oy0;                 // patch row that holds image row 0
                const float* row0 = p.x + (size_t)b * p.K * xchan;
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int c = q - p.pad_x; v[q] = (c >= 0 && c < p.in_w) ? row0[c] : 0.f; }
                uint4 h, l;
                split8(v, scaled ? s_scale[0] : 1.f, &h, &l);
                xh[r * XU] = h; GC_LO(xl[r * XU] = l;)
            }
        }
    };

    if (t_begin < t_end) {
        prefetch(t_begin);
        commit(t_begin);
        __syncthreads();
        const int xa = (wk * 32 + l31) * C::CSX + hi, yb_ = (wn * 32 + l31) * C::CSY + hi;
        for (int tile = t_begin; tile < t_end; tile += tstep) {
            wait_staged_loads();    // no-op in hardware (commit retired them); clears the compiler's pending-load model at the loop header
            const bool more = tile + tstep < t_end;
            prefetch(more ? tile + tstep : tile);       // unconditional: a conditional prefetch merges through register copies, which wait for the loads
            __builtin_amdgcn_s_setprio(GC_MFMA_PRIO);
#pragma unroll 1
            for (int step = 0; step < 2 * TR / WP; ++step) {
                {
                    const int sidx = step * WP + wp;            // this wave's pixel step: row r, half-row st
                    const int r = sidx >> 1, st = sidx & 1;
                    const uint4 ubh = yh[yb_ + r * YU + 2 * st], ubl = yl[yb_ + r * YU + 2 * st];
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&ubh), bl = *reinterpret_cast<const bf16x8*>(&ubl);
#pragma unroll
                    for (int ty = 0; ty < KS; ++ty) {
                        const int o = xa + (r + ty) * XU + 2 * st;
                        const uint4 a0h = xh[o], a0l = xl[o];
                        uint4 a1h = a0h, a1l = a0l;
                        if (KS == 3) { a1h = xh[o + 1]; a1l = xl[o + 1]; }
#pragma unroll
                        for (int tx = 0; tx < KS; ++tx) {
                            const uint4 uh = shift_px(a0h, a1h, tx), ul = shift_px(a0l, a1l, tx);
                            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&uh), al = *reinterpret_cast<const bf16x8*>(&ul);
                            f32x16 c = acc[ty * KS + tx];
                            GC_MFMA3(c, ah, al, bh, bl);
                            acc[ty * KS + tx] = c;
                        }
                        if (KS == 3) __builtin_amdgcn_sched_barrier(0x100);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            if (!more) break;       // leave here: no path may reach the loop header with staged loads in flight
            {
                commit(tile + tstep);
                __syncthreads();
            }
        }
    }

    if (WP > 1) {
        // the WP pixel-waves of a (wk, wn) group hold partial sums of the same (k, n) block: add them through LDS
        float* red = reinterpret_cast<float*>(smem) + (wk * WN + wn) * (WP - 1) * 16 * 64;
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

// ---------------------------------------------------------------------------------------------------------
// Wave-specialised weight gradient (round 5; 3 x 3, stride 1, K and N multiples of 64): the role split and the two-deep staging of the
// stride-1 forward kernel applied to dW.  wgrad_bf16x3_kernel alternates stage -> barrier -> multiply -> barrier with one LDS stage; two
// co-resident workgroups overlap by luck (1.45 x of one).  Here ONE workgroup of 16 waves owns a CU:
//  * 12 MULTIPLYING waves = 3 tap rows x (2 x 2) blocks of 32 k x 32 n: wave (ty, wk, wn) holds the three accumulators of taps (ty, 0..2)
//    -- 48 registers instead of 144 -- and issues nothing but LDS fragment reads, the funnel shifts of the tap columns and MFMAs;
//  * 4 STAGING waves load, split and write what the NEXT item needs while the loads of the item after next are in flight.
// The pixel space is walked in STRIPS: a strip is 32 columns x RB consecutive output rows of one sample, an ITEM is one output row of a strip.
// Item r needs X rows r - 1, r, r + 1 (tap row ty reads row r - 1 + ty) and dY row r: the X rows live in a ring of 6 row slots, so an item
// inside a strip stages ONE new X row and one dY row (the first item of a strip: three X rows) -- every input row is converted once per strip
// instead of (TR + 2) / TR times per tile, and the staging waves are idle most of an item.  One barrier per item.
// Sums are accumulated in a fixed order (strips of a split in order, rows in order, half-rows in order): bit-identical run to run; the order
// differs from wgrad_bf16x3_kernel's, so the two agree to rounding, not bit for bit.
#ifndef GC_WG_WS
#define GC_WG_WS 2          // 2: wgrad_bf16x3_ws2_kernel (two-row items, stream staging: +5..8 % over the one-role kernel, below); 0: one-role kernel only;
                            // 1: the first, one-row form -- MEASURED AT PARITY with wgrad_bf16x3_kernel, not enabled (round 5, tools/kbench.py, same box, profiles/kernel_ab_r05_{d,e}.log): B = 4
                            // 277-298 vs 281-299 TF/s, B = 8 315-323 vs 317-333.  Correct (all weight-gradient tests, race screen) and bit-reproducible.  The
                            // staging waves bound it: the first version (every lane converted four X slots per item, three of them dead inside a strip) ran at
                            // 216-220 TF/s, lean staging 261-278, staging waves at priority 3 277-298; an item (18 MFMAs per wave between two 16-wave barriers,
                            // its fragment reads issued by all twelve multiplying waves at the same moment) is too short.  Two-row items would halve the
                            // barrier / read-burst share but need 56 registers per staging set at a strip start (two sets do not fit 128).
                            // Ablations (GC_WGWS_ABL, profiles/kernel_ab_r05_h.log, B = 4, 128 -> 128 @256^2): complete 264 us; the multiplying waves ALONE
                            // (staging waves keep only the barriers) 145 us = 534 TF/s; the staging waves alone (no MFMAs, no fragment reads) 60 us; staging
                            // + fragment reads without MFMAs 233 us.  The two sides do not overlap -- together they cost more than their sum -- which is the
                            // thing to understand (PMC: SQ wait / issue counters per role) before this kernel is worth enabling: its matrix side is the
                            // fastest in the library.  It is NOT the staging waves' instruction count: a version with the three always-used slots static per
                            // lane (six registers, no integer divisions) and the strip-start rows fetched on the spot ran no faster at >= 128 channels and
                            // 16 % slower at 64 (the on-the-spot fetch stalls once per strip): profiles/kernel_ab_r05_i.log.
#endif
#ifndef GC_WGWS_ABL
#define GC_WGWS_ABL 0       // dev ablations of wgrad_bf16x3_ws_kernel (wrong results): 1 the staging waves only keep the barriers, 2 the multiplying waves issue no MFMAs,
                            // 4 ... and no fragment reads either
#endif
#ifndef GC_WGWS_STAGER_PRIO
#define GC_WGWS_STAGER_PRIO 3      // the staging waves bound this kernel (kbench, B = 4: 261-278 TF/s at priority 0, 277-298 at 3): they issue first
#endif
#if GC_WG_WS == 1
#include "experiments/wgrad_ws1.inc.h"      // the one-row form: measured at parity, not shipped
#endif

// ---------------------------------------------------------------------------------------------------------
// Second form of the wave-specialised weight gradient (GC_WG_WS = 2): TWO output rows per item and the input rows staged as a stream.
// The one-row form's ablations (profiles/kernel_ab_r05_{h,j}.log) say its matrix side alone runs at ~530 TF/s and that the staging waves' path --
// load latency, conversion, LDS write -- is what an item waits for: an item was 18 MFMAs per wave, ~0.7 us, and its loads were requested two
// items = ~1.5 us ahead.  Here an item is 36 MFMAs per wave (half the barriers), the loads of an item are requested two items = ~3 us ahead, and
// every staging step is exactly five unit slots per lane:
//   * X rows 2i + 2, 2i + 3 of the strip (the two new rows of item i: 640 units = 2.5 slots), dY rows 2i, 2i + 1 (512 units = 2 slots);
//   * the idle half of the third X slot carries 128 units of the NEXT strip's first two input rows (its top halo: 640 units over the steps of
//     items 2..6), so a strip boundary costs no extra step: those two rows live in two dedicated row slots (6, 7), the other rows of all strips
//     form one running sequence through a ring of six.
// Strips are 16 rows (out_h a multiple of 16).  Same partial-sum layout and reduce pass as the other weight-gradient kernels.
struct WgWs2Cfg {
    static constexpr int XRING = 6, XR = 8, YR = 4, XU = 5, YU = 4, RB = 16;
    static constexpr int CSX = (XR * XU) | 1, CSY = (YR * YU) | 1;
    static constexpr int XUNITS = 64 * CSX, YUNITS = 64 * CSY;
    static constexpr int SMEM_UNITS = 2 * (XUNITS + YUNITS);
    static constexpr int ROW_X = 64 * XU, ROW_Y = 64 * YU;
};

__global__ __launch_bounds__(1024) void wgrad_bf16x3_ws2_kernel(WgArgs p, int bands) {
    using C = WgWs2Cfg;
    constexpr int XRING = C::XRING, YR = C::YR, XU = C::XU, YU = C::YU, CSX = C::CSX, CSY = C::CSY, RB = C::RB, IPS = RB / 2;
    __shared__ uint4 smem[C::SMEM_UNITS];
    uint4* xh = smem;
    uint4* xl = xh + C::XUNITS;
    uint4* yh = xl + C::XUNITS;
    uint4* yl = yh + C::YUNITS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 64, split = blockIdx.z;

    const int strips_per_sample = p.tiles_x * bands;
    const int sb = p.spb ? split / p.spb : 0;
    const int sstep = p.spb ? p.spb : (int)gridDim.z;
    const int s_begin = p.spb ? sb * strips_per_sample + (split - sb * p.spb) : split;
    const int s_end = p.spb ? (sb + 1) * strips_per_sample : strips_per_sample * p.B;
    const int nstrips = s_begin < s_end ? (s_end - s_begin + sstep - 1) / sstep : 0;
    const int items = nstrips * IPS;
    const int xchan = p.in_h * p.in_w, ychan = p.out_h * p.out_w;

    // row slot of input row xr (0 .. RB + 1) of the strip with ordinal `ord`: the two top rows in the dedicated slots, the rest in the running ring
    auto xslot_of = [&](int ord, int xr) { return xr < 2 ? XRING + xr : (ord * RB + xr - 2) % XRING; };

    if (wave >= 12) {
        // ---------------- staging waves ----------------
        if (GC_WGWS_STAGER_PRIO) __builtin_amdgcn_s_setprio(GC_WGWS_STAGER_PRIO);
        const int st = tid - 768;
        constexpr unsigned OUTSIDE = 0x80000000u;
        const unsigned xbytes = (unsigned)p.K * xchan * 4u, ybytes = (unsigned)p.N * ychan * 4u;
        struct Strip { int sidx, b, oy0, ox0, ord; };
        auto place = [&](Strip& c) {
            c.b = c.sidx / strips_per_sample;
            const int rem = c.sidx - c.b * strips_per_sample;
            c.oy0 = (rem / p.tiles_x) * RB;
            c.ox0 = (rem % p.tiles_x) * 32;
        };
        // One X unit of (strip c, input row xr): u in [0, 320) = (channel, unit column)
        auto x_load = [&](float4 (&v)[2], float& sc, const Strip& c, int xr, int u, bool live) {
            const int ch = min(u / XU, 63), xu = u - (u / XU) * XU;
            const int b = min(c.b, p.B - 1);
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)b * p.K * xchan, xbytes);
            const int iy = c.oy0 + xr - p.pad_y;
            // (the unit at channel 0, row 0, column -pad of a sample would start at a negative offset, which the range check rejects as a whole:
            // it is loaded from offset 0 and shifted by one pixel when it is converted)
            const int lin = ((k0 + ch) * xchan + iy * p.in_w + c.ox0 - p.pad_x) * 4 + xu * 32;
            const unsigned off = (live && (unsigned)iy < (unsigned)p.in_h && c.b < p.B) ? (unsigned)max(lin, 0) : OUTSIDE;
            v[0] = __builtin_bit_cast(float4, buf_load_u128(rx, off, 0));
            v[1] = __builtin_bit_cast(float4, buf_load_u128(rx, off, 16));
            sc = p.si ? p.si[(size_t)b * p.K + k0 + ch] : 1.f;
        };
        auto x_store = [&](auto scaled_t, auto edge_t, const float4 (&r2)[2], float sc, const Strip& c, int xr, int u, bool live) {
            const int ch = min(u / XU, 63), xu = u - (u / XU) * XU;
            float v[8] = {r2[0].x, r2[0].y, r2[0].z, r2[0].w, r2[1].x, r2[1].y, r2[1].z, r2[1].w};
            if (decltype(edge_t)::value) {
                const int col0 = c.ox0 - p.pad_x + 8 * xu;
                if (col0 < 0 && k0 + ch == 0 && c.oy0 + xr - p.pad_y == 0) {
#pragma unroll
                    for (int e = 7; e > 0; --e) v[e] = v[e - 1];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (col0 + e >= 0 && col0 + e < p.in_w) ? v[e] : 0.f;
            }
            uint4 h, l;
            split8<decltype(scaled_t)::value>(v, sc, &h, &l);
            if (live) { const int o = ch * CSX + xslot_of(c.ord, xr) * XU + xu; xh[o] = h; GC_LO(xl[o] = l;) }
        };
        auto y_load = [&](float4 (&v)[2], float& sc, const Strip& c, int r, bool live) {
            const int ych = st >> 2, yu = st & 3;
            const int b = min(c.b, p.B - 1);
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.dy + (size_t)b * p.N * ychan, ybytes);
            const unsigned off = (live && c.b < p.B) ? (unsigned)(((n0 + ych) * ychan + (c.oy0 + r) * p.out_w + c.ox0) * 4 + yu * 32) : OUTSIDE;
            v[0] = __builtin_bit_cast(float4, buf_load_u128(ry, off, 0));
            v[1] = __builtin_bit_cast(float4, buf_load_u128(ry, off, 16));
            sc = p.so ? p.so[(size_t)b * p.N + n0 + ych] : 1.f;
        };
        auto y_store = [&](auto scaled_t, auto edge_t, const float4 (&r2)[2], float sc, const Strip& c, int r) {
            const int ych = st >> 2, yu = st & 3;
            float v[8] = {r2[0].x, r2[0].y, r2[0].z, r2[0].w, r2[1].x, r2[1].y, r2[1].z, r2[1].w};
            if (decltype(edge_t)::value) {
                const int col0 = c.ox0 + 8 * yu;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = col0 + e < p.out_w ? v[e] : 0.f;
            }
            uint4 h, l;
            split8<decltype(scaled_t)::value>(v, sc, &h, &l);
            const int o = ych * CSY + ((c.ord * RB + r) % YR) * YU + yu;
            yh[o] = h; GC_LO(yl[o] = l;)
        };
        // A step = what item i of strip c needs that is not staged yet + (steps 2..6) a fifth of the next strip's top rows.
        // Slots: 0, 1 = X units st, st + 256 of the 640 (rows 2i + 2, 2i + 3); 2 = X unit st + 512 for st < 128, else unit (i - 2) * 128 + st - 128 of
        // the next strip's rows 0, 1; 3, 4 = dY rows 2i, 2i + 1.
        struct Step { Strip c, n; int i; bool live; };       // strip, the strip after it, item
        auto step_loads = [&](float4 (&v)[5][2], float (&sc)[5], const Step& s) {
            const int u0 = opaque(st), u1 = opaque(st) + 256, u2 = opaque(st) + 512;
            x_load(v[0], sc[0], s.c, 2 * s.i + 2, u0, s.live);                       // u0 < 320: row 2i + 2
            x_load(v[1], sc[1], s.c, 2 * s.i + 2 + (u1 >= C::ROW_X ? 1 : 0), u1 >= C::ROW_X ? u1 - C::ROW_X : u1, s.live);
            if (st < 128) {
                x_load(v[2], sc[2], s.c, 2 * s.i + 3, u2 - C::ROW_X, s.live);
            } else {
                const int hu = (s.i - 2) * 128 + st - 128;                           // unit of the next strip's top rows, [0, 640)
                x_load(v[2], sc[2], s.n, hu >= C::ROW_X ? 1 : 0, hu >= C::ROW_X ? hu - C::ROW_X : hu, s.live && s.i >= 2 && s.i <= 6);
            }
            y_load(v[3], sc[3], s.c, 2 * s.i, s.live);
            y_load(v[4], sc[4], s.c, 2 * s.i + 1, s.live);
        };
        auto step_stores = [&](const float4 (&v)[5][2], const float (&sc)[5], const Step& s) {
            if (!s.live) return;
            const bool scaled = p.si != nullptr || p.so != nullptr;
            auto is_edge = [&](const Strip& c) { return c.ox0 - p.pad_x < 0 || c.ox0 - p.pad_x + 8 * XU > p.in_w || c.ox0 + 8 * YU > p.out_w; };
            const bool edge = is_edge(s.c) || is_edge(s.n);
            auto body = [&](auto scaled_t, auto edge_t) {
                const int u0 = opaque(st), u1 = opaque(st) + 256, u2 = opaque(st) + 512;
                x_store(scaled_t, edge_t, v[0], sc[0], s.c, 2 * s.i + 2, u0, true);
                x_store(scaled_t, edge_t, v[1], sc[1], s.c, 2 * s.i + 2 + (u1 >= C::ROW_X ? 1 : 0), u1 >= C::ROW_X ? u1 - C::ROW_X : u1, true);
                if (st < 128) {
                    x_store(scaled_t, edge_t, v[2], sc[2], s.c, 2 * s.i + 3, u2 - C::ROW_X, true);
                } else {
                    const int hu = (s.i - 2) * 128 + st - 128;
                    x_store(scaled_t, edge_t, v[2], sc[2], s.n, hu >= C::ROW_X ? 1 : 0, hu >= C::ROW_X ? hu - C::ROW_X : hu, s.i >= 2 && s.i <= 6 && s.n.b < p.B);
                }
                y_store(scaled_t, edge_t, v[3], sc[3], s.c, 2 * s.i);
                y_store(scaled_t, edge_t, v[4], sc[4], s.c, 2 * s.i + 1);
            };
            if (scaled) { if (edge) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
            else        { if (edge) body(std::false_type{}, std::true_type{}); else body(std::false_type{}, std::false_type{}); }
        };
        auto next_step = [&](Step& s, int t) {           // the step after s, which is item t overall
            if (++s.i == IPS) {
                s.i = 0;
                s.c = s.n;
                s.n.sidx += sstep; ++s.n.ord; place(s.n);
                if (s.n.sidx >= s_end) s.n.b = p.B;      // no strip after the last one: its loads read as zeros, nothing of it is stored
            }
            s.live = t < items;
        };
        Step sl;                                        // cursor of the loads
        sl.c = Strip{s_begin, 0, 0, 0, 0}; place(sl.c);
        sl.n = Strip{s_begin + sstep, 0, 0, 0, 1}; place(sl.n);
        if (sl.n.sidx >= s_end) sl.n.b = p.B;
        sl.i = 0; sl.live = items > 0;
        Step sc_ = sl;                                  // cursor of the conversions
        float4 va[5][2], vb[5][2];
        float sa[5], sb5[5];
        // prologue: the first strip's top rows (nobody staged them ahead): three slots, on the spot
        if (items > 0) {
            const bool scaled = p.si != nullptr || p.so != nullptr;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int u = opaque(st) + 256 * j;
                x_load(va[j], sa[j], sl.c, u >= C::ROW_X ? 1 : 0, u >= C::ROW_X ? u - C::ROW_X : u, u < 2 * C::ROW_X);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int u = opaque(st) + 256 * j;
                if (scaled) x_store(std::true_type{}, std::true_type{}, va[j], sa[j], sl.c, u >= C::ROW_X ? 1 : 0, u >= C::ROW_X ? u - C::ROW_X : u, u < 2 * C::ROW_X);
                else        x_store(std::false_type{}, std::true_type{}, va[j], sa[j], sl.c, u >= C::ROW_X ? 1 : 0, u >= C::ROW_X ? u - C::ROW_X : u, u < 2 * C::ROW_X);
            }
        }
        // interval t: the multiplying waves work on item t; item t + 1 is converted here (its loads were issued one interval ago), item t + 2 is fetched
        step_loads(va, sa, sl); next_step(sl, 1);
        step_loads(vb, sb5, sl); next_step(sl, 2);
        step_stores(va, sa, sc_); next_step(sc_, 1);
        __syncthreads();
        for (int t = 0; t < items; t += 2) {
            step_loads(va, sa, sl); next_step(sl, t + 3);
            step_stores(vb, sb5, sc_); next_step(sc_, t + 2);
            __syncthreads();
            if (t + 1 >= items) break;
            step_loads(vb, sb5, sl); next_step(sl, t + 4);
            step_stores(va, sa, sc_); next_step(sc_, t + 3);
            __syncthreads();
        }
        return;
    }

    // ---------------- multiplying waves ----------------
    const int ty = wave >> 2, wk = (wave >> 1) & 1, wn = wave & 1;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int xa = (wk * 32 + l31) * CSX + hi, yb_ = (wn * 32 + l31) * CSY + hi;
    int i = 0, ord = 0;
    __syncthreads();                 // item 0 is staged
    for (int it = 0; it < items; ++it) {
        __builtin_amdgcn_s_setprio(GC_MFMA_PRIO);
        // four quarter-steps (row, half-row): the fragments of the next one are read before the MFMAs of the current one
        uint4 fbh[2], fbl[2], a0h[2], a1h[2], a0l[2], a1l[2];
        auto read_q = [&](int q, int set) {
            const int row = q >> 1, half = q & 1;
            const int yo = yb_ + ((ord * RB + 2 * i + row) % YR) * YU + 2 * half;
            const int o = xa + xslot_of(ord, 2 * i + row + ty) * XU + 2 * half;
            fbh[set] = yh[yo]; a0h[set] = xh[o]; a1h[set] = xh[o + 1];
            GC_LO(fbl[set] = yl[yo]; a0l[set] = xl[o]; a1l[set] = xl[o + 1];)
        };
        read_q(0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q + 1 < 4) read_q(q + 1, (q + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&fbh[q & 1]);
#ifndef GC_SINGLE
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&fbl[q & 1]);
#endif
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const uint4 uh = shift_px(a0h[q & 1], a1h[q & 1], tx);
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&uh);
#ifndef GC_SINGLE
                const uint4 ul = shift_px(a0l[q & 1], a1l[q & 1], tx);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(&ul);
#endif
                GC_MFMA3(acc[tx], ah, al, bh, bl);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (++i == IPS) { i = 0; ++ord; }
        __syncthreads();             // the rows of this item may be rewritten from the next interval on; the next item is staged
    }
    float* out = p.ws + (size_t)split * 9 * p.K * p.N;
    const int n = n0 + wn * 32 + l31;
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int k = k0 + wk * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
            out[((size_t)(ty * 3 + tx) * p.K + k) * p.N + n] = acc[tx][rr];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stride-2 variant (down = 2, pad = 0): dW[tap][k][n] = sum_px X[k][2 px + tap] * dY[n][px] -- the weight gradient
// of D's 3x3 / 1x1 stride-2 convolutions and (operands swapped) of G's transposed convolutions.  Each input row is
// staged DE-INTERLEAVED: units of 8 even columns and units of 8 odd columns, so tap tx = 0 reads an even unit,
// tx = 1 an odd unit and tx = 2 the even units funnel-shifted by one pixel -- every ds_read_b128 stays aligned.

template <int TR, int KS, int WK, int WN = 2>
struct WgS2Cfg {
    // WK = 2: 64k x 64n, one 32 x 32 block per wave; WK = 1: 32k x 64n, two pixel-waves per block; WK = 1, WN = 4 (round 6): 32k x 128n, one block per wave --
    // the X tile (at stride 2 four times the pixels of the dY tile, and de-interleaved while staged) is then shared by four output-channel blocks instead of two:
    // 29 % fewer operand bytes and conversions per MFMA than the 64k x 64n tile
    static constexpr int KT = 32 * WK, NTL = 32 * WN;
    static constexpr int PH = (TR - 1) * 2 + KS;
    static constexpr int XE = KS == 3 ? 5 : 4, XO = KS == 3 ? 4 : 0, RU = XE + XO, YU = 4;
    static constexpr int NI = XE;                                    // 16-column staging items per row
    static constexpr int CSX = (PH * RU) | 1, CSY = (TR * YU) | 1;
    static constexpr int NXI = KT * PH * NI, NYU = NTL * TR * YU;
    static constexpr int NPX = (NXI + 255) / 256, NPY = (NYU + 255) / 256;
    static constexpr int SMEM_UNITS = 2 * (KT * CSX + NTL * CSY);
    static constexpr int NT = KS * KS;
};

// Dispatched with two workgroups per CU (64 KB of LDS each): 64k x 64n tiles of ONE output row (the second workgroup hides the
// staging phases of the first; every input row is fetched 3 instead of 2.5 times, from L2 since tiles run down a column strip),
// or 32k x 64n tiles of two rows for 32..63 input channels.  A two-row 64k x 64n tile needs 110 KB -- one workgroup per CU --
// and measured 130 against 171 TFLOP/s on 64 -> 128 channels at 513^2.
template <int TR, int KS, int WK, int WN = 2>
__global__ __launch_bounds__(256, (TR == 1 || WK == 1) ? 2 : 1) void wgrad_bf16x3_s2_kernel(WgArgs p) {
    using C = WgS2Cfg<TR, KS, WK, WN>;
    static_assert(WK * WN == 4 || WK * WN == 2, "four waves: WK x WN blocks, two pixel-waves per block when there are only two blocks");
    constexpr int WP = 4 / (WK * WN);       // waves sharing a (k, n) block: they split the half-rows and are summed at the end
    constexpr int KT = C::KT, NTL = C::NTL, PH = C::PH, XE = C::XE, RU = C::RU, YU = C::YU, NI = C::NI, NT = C::NT;
    __shared__ uint4 smem[C::SMEM_UNITS];
    uint4* xh = smem;
    uint4* xl = xh + KT * C::CSX;
    uint4* yh = xl + KT * C::CSX;
    uint4* yl = yh + NTL * C::CSY;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wn = wave % WN, wk = (wave / WN) % WK, wp = wave / (WN * WK);
    const WgBlock blk = wg_block<GC_WG_XCD != 0>();      // stride 2: +3..10 % (same-box A/B)
    const int k0 = blk.x * KT, n0 = blk.y * NTL, split = blk.z;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int tiles_per_sample = p.tiles_x * p.tiles_y;
    const int total_tiles = tiles_per_sample * p.B;
    const int sb = p.spb ? split / p.spb : 0;
    const int tstep = p.spb ? p.spb : (GC_WG2_STRIDED ? (int)gridDim.z : 1);
    const int t_begin = p.spb ? sb * tiles_per_sample + (split - sb * p.spb) : (GC_WG2_STRIDED ? split : split * p.tiles_per_split);
    const int t_end = p.spb ? (sb + 1) * tiles_per_sample : (GC_WG2_STRIDED ? total_tiles : min(total_tiles, t_begin + p.tiles_per_split));
    const int xchan = p.in_h * p.x_pitch, ychan = p.out_h * p.out_w;
    const unsigned xbytes = (unsigned)p.K * xchan * 4u, ybytes = (unsigned)p.N * ychan * 4u;

    // Only the loaded data lives in registers between prefetch and commit: the per-sample scales sit in an LDS table
    // (refilled when a split crosses into the next sample), and every staged item has ONE packed per-lane descriptor
    // (LDS unit offset | item column << 16 | patch row << 20 | channel << 24 | idle lane << 31), made opaque per use so that
    // nothing derived from it is hoisted into registers -- with 144 accumulators the kernel otherwise spills inside the tile loop.
    __shared__ float s_scale[KT + NTL];
    int b_tab = -1;
    float4 xreg[C::NPX][4], yreg[C::NPY][2];
    constexpr unsigned OUTSIDE = 0x80000000u;    // beyond every buffer
    auto xdesc_of = [&](int u) -> unsigned {
        const int it = u % NI, row = u / NI;
        const int r = row % PH, kk = min(row / PH, KT - 1);
        const bool live = u < C::NXI && k0 + kk < p.K;
        return (unsigned)(kk * C::CSX + r * RU + it) | (unsigned)it << 16 | (unsigned)r << 20 | (unsigned)kk << 24 | (live ? 0u : OUTSIDE);
    };
    auto ydesc_of = [&](int u) -> unsigned {
        const int yu = u % YU, row = u / YU;
        const int r = row % TR, nn = min(row / TR, NTL - 1);
        const bool live = u < C::NYU && n0 + nn < p.N;
        return (unsigned)(nn * C::CSY + r * YU + yu) | (unsigned)yu << 16 | (unsigned)r << 20 | (unsigned)nn << 24 | (live ? 0u : OUTSIDE);
    };
    // 64k x 64n: the descriptors stay in registers (6 of them); 32k x 64n has two more staged items per lane and no register to
    // spare -- it rebuilds them from the lane index per use (measured: keeping them there costs scratch reloads in front of the loads)
    constexpr bool KEEP = WK * WN == 4;
    unsigned xdesc[KEEP ? C::NPX : 1], ydesc[KEEP ? C::NPY : 1];
    if (KEEP) {
#pragma unroll
        for (int j = 0; j < C::NPX; ++j) xdesc[j] = xdesc_of(tid + 256 * j);
#pragma unroll
        for (int j = 0; j < C::NPY; ++j) ydesc[j] = ydesc_of(tid + 256 * j);
    }
    auto xd = [&](int j) -> unsigned { return KEEP ? (unsigned)opaque((int)xdesc[KEEP ? j : 0]) : xdesc_of(opaque(tid) + 256 * j); };
    auto yd = [&](int j) -> unsigned { return KEEP ? (unsigned)opaque((int)ydesc[KEEP ? j : 0]) : ydesc_of(opaque(tid) + 256 * j); };
    auto prefetch = [&](int tile) {
        const int b = tile / tiles_per_sample;
        const int rem = tile - b * tiles_per_sample;
        const int oy0 = GC_WG2_STRIDED ? (rem / p.tiles_x) * TR : (rem % p.tiles_y) * TR, ox0 = GC_WG2_STRIDED ? (rem % p.tiles_x) * 32 : (rem / p.tiles_y) * 32;      // as in wgrad_bf16x3_kernel
        const int iy0 = oy0 * 2, ix0 = ox0 * 2;                      // pad = 0 (checked on the host)
        const int xoff = (k0 * xchan + iy0 * p.x_pitch + ix0) * 4, yoff = (n0 * ychan + oy0 * p.out_w + ox0) * 4;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)b * p.K * xchan, xbytes);
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.dy + (size_t)b * p.N * ychan, ybytes);
#pragma unroll
        for (int j = 0; j < C::NPX; ++j) {
            const unsigned d = xd(j);
            const int r = (int)((d >> 20) & 15u);
            const int lin = (int)((d >> 24) & 63u) * (xchan * 4) + r * (p.x_pitch * 4) + (int)((d >> 16) & 15u) * 64 + xoff;
            const unsigned off = ((int)d >= 0 && iy0 + r < p.in_h) ? (unsigned)lin : OUTSIDE;
#pragma unroll
            for (int v = 0; v < 4; ++v) xreg[j][v] = __builtin_bit_cast(float4, WG_LOAD1(false, rx, off, 16 * v));
        }
#pragma unroll
        for (int j = 0; j < C::NPY; ++j) {
            const unsigned d = yd(j);
            const int r = (int)((d >> 20) & 15u);
            const int lin = (int)((d >> 24) & 127u) * (ychan * 4) + r * (p.out_w * 4) + (int)((d >> 16) & 15u) * 32 + yoff;      // (dY channel: bits 24..30, up to 128 per tile)
            const unsigned off = ((int)d >= 0 && oy0 + r < p.out_h) ? (unsigned)lin : OUTSIDE;
            yreg[j][0] = __builtin_bit_cast(float4, WG_LOAD1(false, ry, off, 0));
            yreg[j][1] = __builtin_bit_cast(float4, WG_LOAD1(false, ry, off, 16));
        }
    };
    auto commit = [&](int tile) {
        const int b = tile / tiles_per_sample;
        const int rem = tile - b * tiles_per_sample;
        const int ox0 = GC_WG2_STRIDED ? (rem % p.tiles_x) * 32 : (rem / p.tiles_y) * 32;
        const bool scaled = p.si != nullptr || p.so != nullptr;
        if (scaled && b != b_tab) {          // uniform: every lane of the workgroup sees the same tile
            __syncthreads();
            if (tid < KT) s_scale[tid] = p.si ? p.si[(size_t)b * p.K + min(k0 + tid, p.K - 1)] : 1.f;
            else if (tid < KT + NTL) s_scale[tid] = p.so ? p.so[(size_t)b * p.N + min(n0 + tid - KT, p.N - 1)] : 1.f;
            __syncthreads();
            b_tab = b;
        }
        wait_staged_loads();
        auto items = [&](auto scaled_t, auto edge_t) {          // edge_t: the right-border masks, compiled only into the variant the last tile column takes (see wgrad_bf16x3_kernel)
            constexpr bool SC = decltype(scaled_t)::value, EDGE = decltype(edge_t)::value;
#pragma unroll
            for (int j = 0; j < C::NPX; ++j) {
                const unsigned d = xd(j);
                const int it = (int)((d >> 16) & 15u), o = (int)(d & 0xffffu);
                const int col0 = 2 * ox0 + 16 * it;          // rows / channels outside the image were loaded as zeros already
                const float sc = SC ? s_scale[(d >> 24) & 63u] : 1.f;
                const float4* q4 = xreg[j];
                const float v[16] = {q4[0].x, q4[0].y, q4[0].z, q4[0].w, q4[1].x, q4[1].y, q4[1].z, q4[1].w,
                                     q4[2].x, q4[2].y, q4[2].z, q4[2].w, q4[3].x, q4[3].y, q4[3].z, q4[3].w};
                const int room = p.in_w - col0;             // columns of this item inside the image (pad = 0: only the right border cuts)
                float ev[8], od[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { ev[q] = (!EDGE || 2 * q < room) ? v[2 * q] : 0.f; od[q] = (!EDGE || 2 * q + 1 < room) ? v[2 * q + 1] : 0.f; }
                uint4 eh, el, oh, ol;
                split8<SC>(ev, sc, &eh, &el);
                if (256 * (j + 1) <= C::NXI || tid + 256 * j < C::NXI) {
                    xh[o] = eh; GC_LO(xl[o] = el;)
                    if (KS == 3 && it < C::XO) {
                        split8<SC>(od, sc, &oh, &ol);
                        xh[o + XE] = oh; GC_LO(xl[o + XE] = ol;)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // one item at a time: interleaving the conversions of several items costs more registers than there are
            }
#pragma unroll
            for (int j = 0; j < C::NPY; ++j) {
                const unsigned d = yd(j);
                const int col0 = ox0 + 8 * (int)((d >> 16) & 15u);
                const float sc = SC ? s_scale[KT + ((d >> 24) & 127u)] : 1.f;
                float v[8] = {yreg[j][0].x, yreg[j][0].y, yreg[j][0].z, yreg[j][0].w, yreg[j][1].x, yreg[j][1].y, yreg[j][1].z, yreg[j][1].w};
                const int room = p.out_w - col0;
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (!EDGE || q < room) ? v[q] : 0.f;
                uint4 h, l;
                split8<SC>(v, sc, &h, &l);
                if (256 * (j + 1) <= C::NYU || tid + 256 * j < C::NYU) { yh[d & 0xffffu] = h; GC_LO(yl[d & 0xffffu] = l;) }
            }
        };
        const bool edge = 2 * ox0 + 16 * NI > p.in_w || ox0 + 8 * YU > p.out_w;      // tile-uniform (pad = 0: only the right border cuts)
        if (scaled) { if (edge) items(std::true_type{}, std::true_type{}); else items(std::true_type{}, std::false_type{}); }
        else        { if (edge) items(std::false_type{}, std::true_type{}); else items(std::false_type{}, std::false_type{}); }
    };

    if (t_begin < t_end) {
        prefetch(t_begin);
        commit(t_begin);
        __syncthreads();
        const int xa = (wk * 32 + l31) * C::CSX + hi, yb_ = (wn * 32 + l31) * C::CSY + hi;
        for (int tile = t_begin; tile < t_end; tile += tstep) {
            wait_staged_loads();    // no-op in hardware (commit retired them); clears the compiler's pending-load model at the loop header
            const bool more = tile + tstep < t_end;
            prefetch(more ? tile + tstep : tile);       // unconditional: a conditional prefetch merges through register copies, which wait for the loads
            __builtin_amdgcn_s_setprio(GC_MFMA_PRIO);
#pragma unroll ((TR == 1 || WK == 1) ? 1 : 2)
            for (int r = 0; r < TR; ++r) {
#pragma unroll ((TR == 1 || WK == 1) ? 1 : 2)
                for (int st_ = 0; st_ < 2 / WP; ++st_) {
                    const int st = WP == 2 ? wp : st_;          // two pixel-waves: each takes one half-row
                    const uint4 ubh = yh[yb_ + r * YU + 2 * st], ubl = yl[yb_ + r * YU + 2 * st];
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&ubh), bl = *reinterpret_cast<const bf16x8*>(&ubl);
#pragma unroll
                    for (int ty = 0; ty < KS; ++ty) {
                        const int o = xa + (2 * r + ty) * RU + 2 * st;
                        auto tap = [&](int tx, const uint4 uh, const uint4 ul) {
                            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&uh), al = *reinterpret_cast<const bf16x8*>(&ul);
                            f32x16 c = acc[ty * KS + tx];
                            GC_MFMA3(c, ah, al, bh, bl);
                            acc[ty * KS + tx] = c;
                        };
                        // tap order 0, 2, 1: the even units (and their one-pixel shift) retire before the odd unit is live --
                        // with 144 accumulators and the staged tile in registers there is no room for all three fragments at once
                        const uint4 e0h = xh[o], e0l = xl[o];
                        if (KS == 3) {
                            const uint4 e1h = xh[o + 1], e1l = xl[o + 1];
                            const uint4 sh = shift_px(e0h, e1h, 1), sl = shift_px(e0l, e1l, 1);
                            tap(0, e0h, e0l);
                            __builtin_amdgcn_sched_barrier(0x100);
                            const uint4 o0h = xh[o + XE], o0l = xl[o + XE];
                            tap(2, sh, sl);
                            __builtin_amdgcn_sched_barrier(0x100);
                            tap(1, o0h, o0l);
                            __builtin_amdgcn_sched_barrier(0x100);
                        } else {
                            tap(0, e0h, e0l);
                        }
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            if (!more) break;       // leave here: no path may reach the loop header with staged loads in flight
            {
                commit(tile + tstep);
                __syncthreads();
            }
        }
    }

    if (WP == 2) {
        // the two pixel-waves of a (k, n) block hold partial sums: add them through LDS (the staging buffers are free now)
        float* red = reinterpret_cast<float*>(smem) + wn * 16 * 64;
        for (int t = 0; t < NT; ++t) {
            __syncthreads();
            if (wp == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[t][r];
            }
            __syncthreads();
            if (wp == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += red[r * 64 + lane];
            }
        }
        if (wp != 0) return;
    }
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

struct WgPlan { int small, ct, kt, tr, splits, tiles_per_split, tiles_x, tiles_y; };
#ifndef GC_WG_S2_N128
#define GC_WG_S2_N128 1      // stride-2 weight gradients with N % 128 == 0 and K % 32 == 0 (K >= 64) on 32k x 128n tiles (wgrad_bf16x3_s2_kernel<1, KS, 1, 4>); 0: 64k x 64n
#endif
#ifndef GC_WG_SPLIT_TARGET
#define GC_WG_SPLIT_TARGET 512      // workgroups a weight-gradient launch aims for (pixel splits x channel tiles)
#endif

// 64k x 64n tiles (2 rows per pixel tile) when both channel counts reach 64, else 32k x 32n tiles with the four
// waves splitting the pixel steps of a 4-row tile
WgPlan plan_wg(const gc_conv_desc* d) {
    WgPlan pl;
    pl.small = d->down == 1 && !(d->in_ch >= 64 && d->out_ch >= 64);
    pl.ct = pl.small ? 32 : 64;
    pl.kt = (d->down == 2 && d->in_ch < 64) ? 32 : pl.ct;      // stride 2 with 32..63 input channels: 32k x 64n tiles
    // (round 3: THREE rows for the 64 x 64 tiles -- 162 MFMAs per wave between barriers, 78 KB of LDS, still two workgroups per CU -- measured
    // 15-26 % SLOWER: 64 -> 64 @512^2, B = 8: 557 -> 748 us; 512 -> 512 @64^2: 470 -> 543 us: the two extra staging register sets spill 108 bytes per lane)
    pl.tr = pl.small ? 6 : 2;          // 32 x 32 channel tiles: six rows (81 MFMAs per wave between barriers, 65 KB of LDS; four rows: 923 vs 880 us at 32 -> 32 @1024^2)
    if (d->down == 2 && pl.kt == 64) pl.tr = 1;
    // round 6: 32k x 128n tiles at stride 2 where both channel counts allow it (GC_WG_S2_N128): the X tile is shared by four output-channel blocks
    if (GC_WG_S2_N128 && d->down == 2 && d->in_ch >= 64 && d->in_ch % 32 == 0 && d->out_ch % 128 == 0) { pl.kt = 32; pl.ct = 128; pl.tr = 1; }     // stride 2, 64k x 64n: one output row per tile keeps two workgroups per CU (two-row tiles need 110 KB of LDS: 130 vs 171 TFLOP/s)     // stride 2, small planes: one output row per tile, two workgroups per CU
    pl.tiles_x = gc::ceil_div(d->out_w, 32);
    pl.tiles_y = gc::ceil_div(d->out_h, pl.tr);
    const int total = pl.tiles_x * pl.tiles_y * d->batch;
    const int ctiles = gc::ceil_div(d->in_ch, pl.kt) * gc::ceil_div(d->out_ch, pl.ct);
    // one workgroup per CU is resident (512 registers per lane): two rounds -- except on the shapes wgrad_bf16x3_ws2_kernel takes (see wgrad_launch): its
    // 16-wave workgroups run longer per strip and ONE full round of 256 measured 3..12 % faster at every channel count (same box, B = 2 / 4 / 8,
    // profiles/wg_ab_r05.log: 512 ch @64^2 232 -> 220 us, 256 @128^2 222 -> 209, 128 @256^2 235 -> 221, 64 @512^2 251 -> 231 at B = 4); everything else
    // is 20..50 % slower with 256
    const bool ws2_shape = GC_WG_WS == 2 && d->down == 1 && d->kh == 3 && !pl.small && d->in_ch % 64 == 0 && d->out_ch % 64 == 0 && d->pad_x == 1 && d->pad_y == 1 &&
                           d->out_w >= 32 && d->out_h == d->in_h && d->out_w == d->in_w && d->out_h % 16 == 0;
    int want = gc::ceil_div(GC_WG_SPLIT_TARGET, ctiles);
    if (ws2_shape) {
        // ... provided that kernel really takes the launch with the halved split count (wgrad_launch: at least two 16-row strips per split);
        // 512 -> 512 @32^2 at B = 2 does not, and the one-role kernel with half the splits is 9 % slower
        const int half = std::max(1, std::min(gc::ceil_div(GC_WG_SPLIT_TARGET / 2, ctiles), total));
        const int splits = gc::ceil_div(total, gc::ceil_div(total, half));
        if ((long long)pl.tiles_x * d->batch * (d->out_h / 16) >= 2LL * splits) want = half;
    }
    if (want > total) want = total;
    if (want < 1) want = 1;
    pl.tiles_per_split = gc::ceil_div(total, want);
    pl.splits = gc::ceil_div(total, pl.tiles_per_split);
    return pl;
}

bool wg_eligible(const gc_conv_desc* d) {
    // Narrow planes included: a 4 .. 16-pixel row fills an eighth .. half of the 32-pixel tile (the rest is masked zeros), and the
    // split-bf16 kernels are still 2-3x the fp32 MFMA path there (512 -> 512 @16^2, B = 8: 96 vs 298 us; @4^2: 48 vs 90 us).
    if (d->up != 1 || d->out_w < 4 || pointwise_thin_wgrad(d)) return false;
    if (d->down == 1) return d->in_ch >= 32 && d->out_ch >= 32 && d->pad_x >= 0 && d->pad_x <= 1;
    return d->in_ch >= 32 && d->out_ch >= 64 && d->pad_x == 0 && d->pad_y == 0;      // stride-2 kernel: 64 (or 32) k x 64 n tiles, no padding
}

// ---------------------------------------------------------------------------------------------------------
// Transposed 3x3 stride-2 convolution (up = 2, pad' = 2: ModulatedConv2d's up-sampling branch gan_model.py:295-306
// and the input gradient of every 3x3 stride-2 conv) with the four output phases FUSED in one workgroup.
// Output pixel (2q + py, 2q' + px) of phase (py, px) reads input pixels q + {-1, 0}: all phases share the same
// 2x2 input neighbourhood, so a tile of q positions is staged once and each lane keeps one accumulator per phase.
// Taps per axis: phase 0 -> (t = 0, d = -1), (t = 2, d = 0); phase 1 -> (t = 1, d = 0): 9 (phase, tap) pairs = the
// MFMA count of a plain 3x3 tile, every workgroup does the same work, and B fragments are shared across phases.
template <int WG_OC, int WG_PX, int WPX, int TPW>
struct TCfg {
    static constexpr int OCT = WG_OC * 32, RPB = 32 / TPW;
    static constexpr int TQH = WG_PX * WPX * RPB;
    static constexpr int PH = TQH + 1, PWD = TPW + 1, PLANE = PH * PWD;
    static constexpr int WUNITS = 9 * KG * OCT, PUNITS = KG * PLANE;
    static constexpr int SMEM_UNITS = 2 * (WUNITS + PUNITS);
    static constexpr int NWU = (WUNITS + 255) / 256;
};

// EPI: 0 = store the accumulators as they are (input-gradient launches), 1 = out_scale only (modulated up-sampling
// convolution), 2 = the full fused epilogue.  The epilogue is ~6 VALU instructions per output element on 256 elements per
// lane; compiled out where the launch does not need it (bare stores are 10 % faster at <= 128 input channels).
#ifndef GC_CT_OCC32
#define GC_CT_OCC32 2        // workgroups per CU the 32-output-channel instance (WG_OC = 1: the store-bound 64 -> 32 @512^2 layer) is compiled for
#endif
template <int WG_OC, int WG_PX, int WPX, int TPW, int EPI, bool WDMA = false>
__global__ __launch_bounds__(256, WG_OC == 1 ? GC_CT_OCC32 : 2) void convt_fused_bf16x3_kernel(Bf16Args a) {
    using C = TCfg<WG_OC, WG_PX, WPX, TPW>;
    static_assert(WG_OC * WG_PX == 4, "4 waves per workgroup");
    constexpr int OCT = C::OCT, RPB = C::RPB, TQH = C::TQH, PWD = C::PWD, PLANE = C::PLANE;
    const ConvArgs& p = a.c;
    __shared__ uint4 smem[C::SMEM_UNITS];
    uint4* wl_h = smem;                         // [tap][kg][OCT]
    uint4* wl_l = wl_h + C::WUNITS;
    uint4* p_h = wl_l + C::WUNITS;              // [kg][PH][PWD]
    uint4* p_l = p_h + C::PUNITS;
    __shared__ __attribute__((aligned(16))) float s_so[OCT], s_bias[OCT];    // out_scale / bias of this workgroup's channels (see conv_epilogue)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wave_px = wave % WG_PX, wave_oc = wave / WG_PX;

    int bid = blockIdx.x;
    const int tile_x = bid % p.tiles_x; bid /= p.tiles_x;
    const int tile_y = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int n0 = blockIdx.y * OCT;
    if (EPI > 0 && tid < OCT) {                 // read in the epilogue, many barriers later
        const int oc = min(n0 + tid, p.N - 1);
        s_so[tid] = p.so ? p.so[(size_t)b * p.N + oc] : 1.f;
        s_bias[tid] = p.bias ? p.bias[oc] : 0.f;
    }
    const int qy0 = tile_y * TQH, qx0 = tile_x * TPW;

    f32x16 acc[4][WPX];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ph][j][r] = 0.f;

    int boff[WPX];
#pragma unroll
    for (int j = 0; j < WPX; ++j) boff[j] = hi * PLANE + ((wave_px * WPX + j) * RPB + l31 / TPW) * PWD + l31 % TPW;
    const int aoff = hi * OCT + wave_oc * 32 + l31;

    const float* xb = p.x + (size_t)b * p.K * p.in_h * p.in_w;
    const float* sib = p.si ? p.si + (size_t)b * p.K : nullptr;
    const int chan = p.in_h * p.in_w;
    const int iy0 = qy0 - 1, ix0 = qx0 - 1;

    // WDMA: the weight slab never touches a register.  Its rows ([tap][kg] x OCT units, contiguous in HBM and in LDS) are copied by
    // LDS-DMA into the SINGLE weight stage right after the barrier that ends the MFMA phase -- every wave has read its fragments by then --
    // and land while the patch of the next chunk is converted and written; vmcnt(0) before the second barrier.  40 registers and
    // 10 ds_write_b128 per lane and chunk less than the register path (launch_t takes this path when N and K need no masking).
    uint4 wreg_h[WDMA ? 1 : C::NWU], wreg_l[WDMA ? 1 : C::NWU];
#ifdef GC_SINGLE
    constexpr int DROWS = 9 * KG;
#else
    constexpr int DROWS = 2 * 9 * KG;
#endif
    constexpr int RPI = 64 / OCT, DINSTR = DROWS / RPI;
    static_assert(!WDMA || (DROWS % RPI == 0 && (9 * KG) % RPI == 0), "row groups do not straddle the hi / lo halves");
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto dma_weights = [&](int k0) {
#pragma unroll
        for (int j = 0; j < (DINSTR + 3) / 4; ++j) {
            const int q = wave_u + 4 * j;
            if (4 * j + 3 < DINSTR || q < DINSTR) {
                const int r0 = q * RPI;
                const int half = r0 / (9 * KG), rr0 = r0 % (9 * KG);
                const int rr = rr0 + lane / OCT;
                const int t = rr / KG, kg = rr % KG;
                const uint4* src = (half ? a.wl : a.wh) + ((size_t)(t * a.kgroups + k0 / 8 + kg) * p.N + n0 + lane % OCT);
                glds16(src, (half ? wl_l : wl_h) + rr0 * OCT);
            }
        }
    };
    // Patch staging as in conv_bf16x3_kernel: a lane fetches FOUR consecutive pixels of a channel with one 16-byte load (eight
    // channels = eight loads) and transposes them in registers into four channel-last units.  The texture-address unit spends
    // ~16 cycles per wave-level load whatever its width; with 24 dword loads per lane per chunk that was more than the MFMAs of a
    // chunk at <= 64 output channels.  A patch row is the halo column (one pixel, "edge" task) + TPW / 4 aligned groups.
    constexpr int GR = TPW / 4, TASKS = C::PH * (GR + 1);
    static_assert(TASKS <= 128, "one staging task per lane and channel group");
    __shared__ __attribute__((aligned(16))) float s_si[MAX_K_BF16X3 + KCB];     // in_scale of this sample, zero past K (a ragged last chunk contributes nothing)
    for (int k = tid; k < ((p.K + KCB - 1) / KCB) * KCB; k += 256) s_si[k] = k < p.K ? (sib ? sib[k] : 1.f) : 0.f;
    const int kgl_p = __builtin_amdgcn_readfirstlane(tid >> 7), tb = tid & 127;     // waves 0,1: channel group 0; waves 2,3: group 1
    const int t_row = tb / (GR + 1), t_g = tb % (GR + 1);
    const int t_col = t_g == 0 ? 0 : 4 * t_g - 3, t_used = tb < TASKS ? (t_g == 0 ? 1 : 4) : 0;
    uint4 preg[8];
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)p.K * chan * 4u);
    const unsigned wbytes = 9u * a.kgroups * p.N * 16u;
    const __amdgpu_buffer_rsrc_t rwh = make_rsrc(a.wh, wbytes), rwl = make_rsrc(a.wl, wbytes);
    auto prefetch = [&](int k0) {
        const int t_ = tid;
        if (!WDMA) {
#pragma unroll
            for (int j = 0; j < C::NWU; ++j) {
                const int u = t_ + 256 * j;
                const int oc = u % OCT, rest = u / OCT;
                const int kgl = rest % KG, tap = rest / KG;
                const int kg = k0 / 8 + kgl, n = n0 + oc;
                const bool ok = u < C::WUNITS && kg < a.kgroups && n < p.N;
                const unsigned gb = ok ? (unsigned)((tap * a.kgroups + kg) * p.N + n) * 16u : OOB;
                wreg_h[j] = (GC_CT_ABL & 4) ? make_uint4(gb, gb, gb, gb) : buf_load_u128(rwh, gb, 0);
                wreg_l[j] = (GC_CT_ABL & 4) ? make_uint4(gb, gb, gb, gb) : buf_load_u128(rwl, gb, 0);
            }
        }
        const int iy = iy0 + t_row, ix = ix0 + t_col;
        const bool ok = t_used > 0 && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
        const unsigned boff_ = ok ? (unsigned)(iy * p.in_w + ix) * 4u : OOB;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = min(k0 + kgl_p * 8 + q, p.K - 1);          // wave-uniform -> scalar offset
            preg[q] = (GC_CT_ABL & 4) ? make_uint4(boff_, k, boff_ + 1, k + 1) : buf_load_u128(rx, boff_, (unsigned)k * chan * 4u);
        }
    };
    auto commit = [&](int k0) {
        wait_staged_loads();
        const int t_ = tid;
        if (WDMA) {
            if (!(GC_CT_ABL & 4)) dma_weights(k0);          // in flight during the conversion below
        } else {
#pragma unroll
            for (int j = 0; j < C::NWU; ++j) {
                const int u = t_ + 256 * j;
                if (u < C::WUNITS) { wl_h[u] = wreg_h[j]; GC_LO(wl_l[u] = wreg_l[j];) }
            }
        }
        const float4 sa = *reinterpret_cast<const float4*>(&s_si[k0 + kgl_p * 8]), sb = *reinterpret_cast<const float4*>(&s_si[k0 + kgl_p * 8 + 4]);
        const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
        const int inrow = p.in_w - (ix0 + t_col);                    // pixels of this group that are still inside the image row
        const int ubase = kgl_p * PLANE + t_row * PWD + t_col;
        if (GC_CT_ABL & 8) {        // ablation: what a pre-split input would leave of the staging -- the loaded registers go to LDS as they are
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < t_used) { p_h[ubase + i] = preg[i]; p_l[ubase + i] = preg[4 + i]; }
        } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned raw = i == 0 ? preg[q].x : (i == 1 ? preg[q].y : (i == 2 ? preg[q].z : preg[q].w));
                v[q] = i < inrow ? __uint_as_float(raw) : 0.f;
            }
            uint4 h, l;
            if (p.si) split8s<true>(v, sc, &h, &l);        // plain (un-packed) multiplies and subtractions: see split8s
            else      split8s<false>(v, sc, &h, &l);       // D's input-gradient launches: no per-sample scale, no multiply by one
            if (i < t_used) {
                p_h[ubase + i] = h;
                GC_LO(p_l[ubase + i] = l;)
            }
        }
        }
        if (WDMA) wait_staged_loads();           // the LDS-DMA rows of this wave have landed (untracked by the compiler: counted by hand)
    };

    // split over the input channels (small planes, see plan_splitk_bf16): slice blockIdx.z covers [kz0, kz1) and stores raw partial sums
    const int kz0 = a.k_per_split ? (int)blockIdx.z * a.k_per_split : 0;
    const int kz1 = a.k_per_split ? min(p.K, kz0 + a.k_per_split) : p.K;
    prefetch(kz0);
    __syncthreads();        // s_si
    commit(kz0);
    __syncthreads();
    for (int k0 = kz0; k0 < kz1; k0 += KCB) {
        wait_staged_loads();    // no-op in hardware (commit retired them); clears the compiler's pending-load model at the loop header
        const bool more = k0 + KCB < kz1;
        prefetch(more ? k0 + KCB : k0);       // unconditional: a conditional prefetch merges through register copies, which wait for the loads
        __builtin_amdgcn_s_setprio(GC_MFMA_PRIO);
#if GC_FRAG_PIPE
        if (!(GC_CT_ABL & 2)) {
            // The nine (phase, tap) steps of a chunk as one software pipeline: the weight fragment of step s + 1 -- and the patch fragments of the
            // next neighbour group when the group changes -- are read BEFORE the MFMAs of step s (scheduling barriers pin the order); the compiler's
            // own order waited `lgkmcnt(0)` a dozen times per chunk with one to five MFMAs in between.
            // step s -> neighbour group g = (dyi, dxi): s = 0: (0,0); 1, 2: (0,1); 3, 4: (1,0); 5..8: (1,1)
            bf16x8 fbh[WDMA ? 2 : 1][WPX], fbl[WDMA ? 2 : 1][WPX], fah[2], fal[2];
            auto grp = [](int s_) { return s_ == 0 ? 0 : (s_ < 3 ? 1 : (s_ < 5 ? 2 : 3)); };
            auto load_b = [&](int g, int set) {
                const int dyi = g >> 1, dxi = g & 1;
#pragma unroll
                for (int j = 0; j < WPX; ++j) {
                    const uint4 uh = p_h[boff[j] + dyi * PWD + dxi];
                    fbh[set][j] = *reinterpret_cast<const bf16x8*>(&uh);
                    GC_LO(const uint4 ul = p_l[boff[j] + dyi * PWD + dxi]; fbl[set][j] = *reinterpret_cast<const bf16x8*>(&ul);)
                }
            };
            auto step_of = [&](int s_, int& py, int& px, int& ty, int& tx) {
                const int g = grp(s_), dyi = g >> 1, dxi = g & 1;
                const int iy = g == 2 ? s_ - 3 : (g == 3 ? (s_ - 5) >> 1 : 0), ix = g == 1 ? s_ - 1 : (g == 3 ? (s_ - 5) & 1 : 0);
                py = (dyi == 1 && iy == 1) ? 1 : 0; ty = dyi == 0 ? 0 : (iy == 0 ? 2 : 1);
                px = (dxi == 1 && ix == 1) ? 1 : 0; tx = dxi == 0 ? 0 : (ix == 0 ? 2 : 1);
            };
            auto load_a = [&](int s_, int set) {
                int py, px, ty, tx;
                step_of(s_, py, px, ty, tx);
                const int wbase = (ty * 3 + tx) * KG * OCT + aoff;
                const uint4 uh = wl_h[wbase];
                fah[set] = *reinterpret_cast<const bf16x8*>(&uh);
                GC_LO(const uint4 ul = wl_l[wbase]; fal[set] = *reinterpret_cast<const bf16x8*>(&ul);)
            };
            // (two sets of patch fragments only where the registers allow it: with the weight slab staged through registers -- WDMA = false,
            // 40 more live registers -- the second set spilled INSIDE the chunk loop, a scratch reload in front of every prefetch pair)
            constexpr int BSETS = WDMA ? 2 : 1;
            load_b(0, 0);
            load_a(0, 0);
#pragma unroll
            for (int s_ = 0; s_ < 9; ++s_) {
                if (BSETS == 1 && s_ > 0 && grp(s_) != grp(s_ - 1)) load_b(grp(s_), 0);
                if (s_ + 1 < 9) {
                    load_a(s_ + 1, (s_ + 1) & 1);
                    if (BSETS == 2 && grp(s_ + 1) != grp(s_)) load_b(grp(s_ + 1), grp(s_ + 1) & 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                int py, px, ty, tx;
                step_of(s_, py, px, ty, tx);
                const int bs = BSETS == 2 ? grp(s_) & 1 : 0;
#pragma unroll
                for (int j = 0; j < WPX; ++j) { GC_MFMA3(acc[py * 2 + px][j], fah[s_ & 1], fal[s_ & 1], fbh[bs][j], fbl[bs][j]); }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#endif
#pragma unroll
        for (int dyi = 0; dyi < 2; ++dyi) {
#pragma unroll
            for (int dxi = 0; dxi < 2; ++dxi) {
                bf16x8 bh[WPX], bl[WPX];
#pragma unroll
                for (int j = 0; j < WPX; ++j) {
                    const uint4 uh = p_h[boff[j] + dyi * PWD + dxi], ul = p_l[boff[j] + dyi * PWD + dxi];
                    bh[j] = *reinterpret_cast<const bf16x8*>(&uh);
                    bl[j] = *reinterpret_cast<const bf16x8*>(&ul);
                }
                // (phase, tap) pairs reading the neighbour at offset d = dyi - 1: d = -1 -> (0, t=0); d = 0 -> (0, t=2), (1, t=1)
#pragma unroll
                for (int iy = 0; iy < 1 + dyi; ++iy) {
                    const int py = (dyi == 1 && iy == 1) ? 1 : 0, ty = dyi == 0 ? 0 : (iy == 0 ? 2 : 1);
#pragma unroll
                    for (int ix = 0; ix < 1 + dxi; ++ix) {
                        const int px = (dxi == 1 && ix == 1) ? 1 : 0, tx = dxi == 0 ? 0 : (ix == 0 ? 2 : 1);
                        const int wbase = (ty * 3 + tx) * KG * OCT + aoff;
                        const uint4 uh = wl_h[wbase], ul = wl_l[wbase];
                        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&uh), al = *reinterpret_cast<const bf16x8*>(&ul);
#pragma unroll
                        for (int j = 0; j < WPX; ++j) {
                            f32x16 c = acc[py * 2 + px][j];
                            if (GC_CT_ABL & 2) { c[0] += __builtin_bit_cast(float, ((const uint4&)ah).x ^ ((const uint4&)bh[j]).x ^ ((const uint4&)al).x ^ ((const uint4&)bl[j]).x); }
                            else { GC_MFMA3(c, ah, al, bh[j], bl[j]); }
                            acc[py * 2 + px][j] = c;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        if (!more) break;       // leave here: no path may reach the loop header with staged loads in flight
        {
            commit(k0 + KCB);
            __syncthreads();
        }
    }

    // The phases px = 0 / 1 of one input column are NEIGHBOURS in the output row: they leave as one 8-byte store (4-byte aligned:
    // rows of a 1025-wide plane start anywhere), half the store instructions and whole 128-byte segments per 16 lanes.
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    const int opitch = a.out_pitch;          // rows of a (2H + 1)-wide output are never 16-byte aligned: a pitch that is a multiple of 32 floats gives every 128-byte store run whole cache lines
    float* yb = (a.k_per_split ? a.part + (size_t)blockIdx.z * a.per_slice : p.y) + (size_t)b * p.N * p.out_h * opitch;
    const EpilogueConsts ec = epilogue_consts(p);
    float nz[WPX][2][2];         // fetched before the first store: a load between stores waits for every store before it
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        const int qy = qy0 + (wave_px * WPX + j) * RPB + l31 / TPW, qx = qx0 + l31 % TPW;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int oy = min(2 * qy + (ph >> 1), p.out_h - 1), ox = min(2 * qx + (ph & 1), p.out_w - 1);
            nz[j][ph >> 1][ph & 1] = (EPI == 2 && p.noise) ? p.noise[((size_t)b * p.out_h + oy) * p.out_w + ox] : 0.f;
        }
    }
    // out_scale / bias of this lane's 16 channels (four runs of four consecutive ones), fetched ONCE before the store loops: read at each
    // store they cost one exposed LDS round trip per output pair (round 5, found in the disassembly: 124 of 128 stores behind an lgkmcnt wait)
    float so16[16], bi16[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 s4 = EPI > 0 ? *reinterpret_cast<const float4*>(&s_so[wave_oc * 32 + 8 * q + 4 * hi]) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 b4 = EPI == 2 ? *reinterpret_cast<const float4*>(&s_bias[wave_oc * 32 + 8 * q + 4 * hi]) : make_float4(0.f, 0.f, 0.f, 0.f);
        so16[4 * q] = s4.x; so16[4 * q + 1] = s4.y; so16[4 * q + 2] = s4.z; so16[4 * q + 3] = s4.w;
        bi16[4 * q] = b4.x; bi16[4 * q + 1] = b4.y; bi16[4 * q + 2] = b4.z; bi16[4 * q + 3] = b4.w;
    }
#pragma unroll
    for (int j = 0; j < WPX; ++j) {
        const int qy = qy0 + (wave_px * WPX + j) * RPB + l31 / TPW, qx = qx0 + l31 % TPW;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            const int oy = 2 * qy + py, ox = 2 * qx;
            if (oy >= p.out_h || ox >= p.out_w) continue;
            const bool pair = ox + 1 < p.out_w;
            float res[2][16];
            if (EPI == 2 && p.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int oc = min(n0 + wave_oc * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, p.N - 1);
                    const float* rp = p.residual + (((size_t)b * p.N + oc) * p.out_h + oy) * p.out_w + ox;
                    res[0][r] = rp[0];
                    res[1][r] = pair ? rp[1] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ocl = wave_oc * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (n0 + ocl < p.N) {
                    float v0 = acc[py * 2][j][r], v1 = acc[py * 2 + 1][j][r];
                    if (EPI == 1) { v0 *= so16[r]; v1 *= so16[r]; }
                    if (EPI == 2) { v0 = conv_epilogue(ec, v0, so16[r], bi16[r], nz[j][py][0]); v1 = conv_epilogue(ec, v1, so16[r], bi16[r], nz[j][py][1]); }
                    if (EPI == 2 && p.residual) { v0 += res[0][r]; v1 += res[1][r]; }
                    float* yp = yb + ((size_t)(n0 + ocl) * p.out_h + oy) * opitch + ox;
                    if ((GC_CT_ABL & 1) && v0 != 12345.678f) continue;
#if GC_CONV_NT
                    if (pair) { f2u v = {v0, v1}; __builtin_nontemporal_store(v, reinterpret_cast<f2u*>(yp)); }
                    else __builtin_nontemporal_store(v0, yp);
#else
                    if (pair) { f2u v = {v0, v1}; *reinterpret_cast<f2u*>(yp) = v; }
                    else yp[0] = v0;
#endif
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The last output row and column of a (2H + 1) x (2W + 1) transposed convolution (round 5).  q-space is (H + 1) x (W + 1): the kernel above
// tiles it in 4 x 32 or 8 x 16 blocks, and at H = W = 32 / 64 / 128 the one extra q-row and q-column cost 47 / 29 / 16 % more tiles than the
// H x W region, which tiles exactly.  With GC_CT_EDGE the fused kernel is launched over the H x W region only (output rows 0 .. 2H - 1, columns
// 0 .. 2W - 1) and this kernel computes the rest: output row 2H (2W + 1 values, from input row H - 1 under the taps ty = 0) and output column 2W
// (2H values, from input column W - 1 under the taps tx = 0) -- 1-D problems, (H + W + 1) q-positions of three taps each instead of H + W + 1
// positions padded to whole 2-D tiles.  One workgroup = 32 q-positions x 64 output channels; its four waves take a quarter of the input
// channels each, straight from global memory into registers (no LDS staging: 16 scalar loads of x and 12 16-byte loads of the packed weights
// per lane and chunk, two chunks in flight), and wave 0 adds the quarters in a fixed order and applies the epilogue.
// Same arithmetic as the fused kernel (split operands, three MFMAs per product); the sums run over the quarters one after the other instead
// of chunk by chunk, so the edge values differ from the one-kernel form in the last bits.
#ifndef GC_CT_EDGE
#define GC_CT_EDGE 1
#endif
#ifndef GC_CT_EDGE_MIN_WGS
#define GC_CT_EDGE_MIN_WGS 512      // workgroups of the main region from which the two-launch form is used (see ct_edge_eligible)
#endif
__global__ __launch_bounds__(256) void convt_edge_bf16x3_kernel(Bf16Args a) {
    const ConvArgs& p = a.c;
    __shared__ float red[3][64][64];                       // [wave - 1][accumulator register][lane]
    const int H = p.in_h, W = p.in_w, chan = H * W;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rblocks = (W + 1 + 31) / 32;                 // blocks of the bottom row first, then those of the right column
    const bool col = (int)blockIdx.x >= rblocks;
    const int e = ((int)blockIdx.x - (col ? rblocks : 0)) * 32 + l31;         // q-position along the edge
    const int n0 = blockIdx.y * 64, b = blockIdx.z;
    // the two input pixels of this position: `cur` (offset d = 0) and `prev` (d = -1) along the edge
    const bool okc = col ? e < H : e < W, okp = col ? (e >= 1 && e < H) : (e >= 1 && e <= W);
    const int cur = okc ? (col ? e * W + W - 1 : (H - 1) * W + e) : 0;
    const int prev = okp ? (col ? (e - 1) * W + W - 1 : (H - 1) * W + e - 1) : 0;
    // taps: prev -> phase 0 under (0, 0); cur -> phase 0 under (0, 2) | (2, 0) and -> phase 1 under (0, 1) | (1, 0)
    const int tB = col ? 6 : 2, tC = col ? 3 : 1;
    const float* xb = p.x + (size_t)b * p.K * chan;
    const float* sib = p.si ? p.si + (size_t)b * p.K : nullptr;
    const int chunks = p.K / KCB, c0 = wave * chunks / 4, c1 = (wave + 1) * chunks / 4;
    f32x16 acc[2][2];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ph][i][r] = 0.f;
#pragma unroll 2
    for (int c = c0; c < c1; ++c) {
        const int kb = c * KCB + hi * 8;
        float vp[8], vc[8], sc[8];
        // `prev` of a lane is `cur` of the lane before it: only the first lane of each 32-lane half loads it (a column block's loads touch one
        // cache line per lane -- 64 line requests per instruction -- so loading both pixels everywhere doubled what the texture unit had to do)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            vc[q] = xb[(size_t)(kb + q) * chan + cur];
            vp[q] = l31 == 0 ? xb[(size_t)(kb + q) * chan + prev] : 0.f;
            sc[q] = sib ? sib[kb + q] : 1.f;
        }
        uint4 wa_h[2], wa_l[2], wb_h[2], wb_l[2], wc_h[2], wc_l[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t col_ = (size_t)n0 + i * 32 + l31, kg = c * KG + hi;
            wa_h[i] = a.wh[(0 * (size_t)a.kgroups + kg) * p.N + col_];  GC_LO(wa_l[i] = a.wl[(0 * (size_t)a.kgroups + kg) * p.N + col_];)
            wb_h[i] = a.wh[(tB * (size_t)a.kgroups + kg) * p.N + col_]; GC_LO(wb_l[i] = a.wl[(tB * (size_t)a.kgroups + kg) * p.N + col_];)
            wc_h[i] = a.wh[(tC * (size_t)a.kgroups + kg) * p.N + col_]; GC_LO(wc_l[i] = a.wl[(tC * (size_t)a.kgroups + kg) * p.N + col_];)
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            vc[q] = okc ? vc[q] : 0.f;
            const float up = __shfl_up(vc[q], 1, 32);             // (zero where the lane before is past the plane, like its own `cur`)
            vp[q] = okp ? (l31 == 0 ? vp[q] : up) : 0.f;
        }
        uint4 ph_, pl_, ch_, cl_;
        if (sib) { split8s<true>(vp, sc, &ph_, &pl_); split8s<true>(vc, sc, &ch_, &cl_); }
        else     { split8s<false>(vp, sc, &ph_, &pl_); split8s<false>(vc, sc, &ch_, &cl_); }
        const bf16x8 bph = *reinterpret_cast<const bf16x8*>(&ph_), bch = *reinterpret_cast<const bf16x8*>(&ch_);
        GC_LO(const bf16x8 bpl = *reinterpret_cast<const bf16x8*>(&pl_); const bf16x8 bcl = *reinterpret_cast<const bf16x8*>(&cl_);)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&wa_h[i]), bh = *reinterpret_cast<const bf16x8*>(&wb_h[i]), chh = *reinterpret_cast<const bf16x8*>(&wc_h[i]);
            GC_LO(const bf16x8 al = *reinterpret_cast<const bf16x8*>(&wa_l[i]); const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&wb_l[i]); const bf16x8 cll = *reinterpret_cast<const bf16x8*>(&wc_l[i]);)
            GC_MFMA3(acc[0][i], ah, al, bph, bpl);
            GC_MFMA3(acc[0][i], bh, bl, bch, bcl);
            GC_MFMA3(acc[1][i], chh, cll, bch, bcl);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave - 1][(ph * 2 + i) * 16 + r][lane] = acc[ph][i][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][i][r] += red[w][(ph * 2 + i) * 16 + r][lane];
    // Epilogue in two phases like the other kernels: every value this lane needs (out_scale / bias of its 32 channels, noise, residual) is loaded BEFORE
    // the first store -- a load between two stores waits for every store issued so far, and as first written (loads inside the store loop) this
    // kernel took 45 us, most of it in 64 such round trips.
    const EpilogueConsts ec = epilogue_consts(p);
    const int opitch = a.out_pitch;
    float so_[2][16], bi_[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            so_[i][r] = p.so ? p.so[(size_t)b * p.N + n] : 1.f;
            bi_[i][r] = p.bias ? p.bias[n] : 0.f;
        }
    int oy_[2], ox_[2];
    bool ok_[2];
    float nz_[2];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        oy_[ph] = col ? 2 * e + ph : 2 * H;
        ox_[ph] = col ? 2 * W : 2 * e + ph;
        ok_[ph] = oy_[ph] < p.out_h && ox_[ph] < p.out_w && !(col && e >= H);
        nz_[ph] = (p.noise && ok_[ph]) ? p.noise[((size_t)b * p.out_h + oy_[ph]) * p.out_w + ox_[ph]] : 0.f;
    }
    if (p.residual) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float rv = ok_[ph] ? p.residual[(((size_t)b * p.N + n) * p.out_h + oy_[ph]) * p.out_w + ox_[ph]] : 0.f;
                    acc[ph][i][r] = conv_epilogue(ec, acc[ph][i][r], so_[i][r], bi_[i][r], nz_[ph]) + rv;
                }
    } else {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][i][r] = conv_epilogue(ec, acc[ph][i][r], so_[i][r], bi_[i][r], nz_[ph]);      // absent parts are exact no-ops (conv_common.h)
    }
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        if (!ok_[ph]) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                p.y[(((size_t)b * p.N + n) * p.out_h + oy_[ph]) * opitch + ox_[ph]] = acc[ph][i][r];
            }
    }
}

// the launches that take the H x W main region + edge form: the (2H + 1) x (2W + 1) geometry, whole chunks and 64-channel output blocks, >= 256 input
// channels (below that the layer is bound by its stores, not by its tiles), an H x W region that the 4 x 32 tile covers exactly, and >= 10 % fewer tiles
inline bool ct_edge_eligible(const Bf16Args& a) {
    const ConvArgs& c = a.c;
    if (!GC_CT_EDGE || a.k_per_split || c.out_h != 2 * c.in_h + 1 || c.out_w != 2 * c.in_w + 1) return false;
    if (c.K % KCB != 0 || c.K < 256 || c.K / KCB < 4 || c.N % 64 != 0 || c.in_w % 32 != 0 || c.in_h % 4 != 0) return false;
    const int qh = c.in_h + 1, qw = c.in_w + 1;
    const bool narrow = GC_CONVT_NARROW && gc::ceil_div(qw, 16) * 16 < gc::ceil_div(qw, 32) * 32;
    const long long full = (long long)gc::ceil_div(qw, narrow ? 16 : 32) * gc::ceil_div(qh, narrow ? 8 : 4), main_ = (long long)(c.in_w / 32) * (c.in_h / 4);
    // ... and enough workgroups for two per CU: with one per CU nothing overlaps its staging (512 -> 512 @32^2, B = 4: 256 workgroups, 127 -> 137 us;
    // 512 -> 256 @64^2, B = 2: 99 -> 133 us -- against B = 8 / B = 4 of the same layers: 226 -> 175, 197 -> 162 us; profiles/convt_ab_r05.log)
    return 10 * main_ <= 9 * full && main_ * c.B * (c.N / 64) >= GC_CT_EDGE_MIN_WGS;
}

// ---------------------------------------------------------------------------------------------------------
// Wave-specialised form of the transposed 3x3 convolution above (round 4): same geometry, same packed weights, same pitched output and
// the same order of accumulation per output element (bit-identical results), on the structure of conv_bf16x3_ws_kernel -- ONE workgroup
// of 12 waves per CU, eight MULTIPLYING waves that issue nothing but LDS fragment reads and MFMAs, four STAGING waves (loads of the item
// after next in flight while the next item is converted), one barrier per item.  convt_fused_bf16x3_kernel spends 22-31 % of its time
// waiting for the patch of the next chunk (GC_CT_ABL = 4) because four accumulator sets (128 registers) leave room for ONE chunk of
// prefetch at two workgroups per CU.  Here the four output phases are produced in TWO PASSES over the input channels:
//   pass 0: output rows 2 qy     = phases (0,0), (0,1): taps ty in {0, 2} -> 6 taps, patch rows qy - 1 and qy
//   pass 1: output rows 2 qy + 1 = phases (1,0), (1,1): tap  ty = 1       -> 3 taps, patch row qy
// so a multiplying wave carries 2 phases x 2 pixel blocks x 32 oc = 64 accumulator registers, as in the stride-1 kernel.  The patch of a
// chunk is staged once per pass (twice per chunk: 1.9 x the conversions per MFMA of the stride-1 kernel, well inside what four staging
// waves do), the weight slab of a (pass, chunk) item is its 6 or 3 tap rows.  An item is short (36 / 18 MFMAs per wave), shorter than an
// LDS-DMA round trip: the weight slabs therefore live in a THREE-slot ring filled TWO items ahead (the DMA of item i + 2 is issued at the
// start of item i), and completion is counted by hand -- at the end of item i a wave waits `vmcnt(n)` with n = the DMA instructions it has
// just issued for item i + 2; loads complete in order, so everything older (the rows of item i + 1, and any store of a finished tile) is
// done, whatever the stores' own completion order.
// MEASURED AND NOT ENABLED (round 4, tools/kbench.py, B = 4, same box; profiles/convt_ws_r04.md): correct on every test shape and bit-identical
// run to run, but no faster than the one-role kernel -- 512 -> 256 @64^2 199 vs 213 us, 256 -> 128 @128^2 190 vs 180, 128 -> 64 @256^2 205 vs 191,
// 64 -> 32 @512^2 302 vs 243.  Ablation builds (GC_CTWS_ABL) say why: with neither patch staging nor weight DMA the multiplying side alone
// runs 159 / 131 / 125 / 173 us -- (i) the (H + 1)^2 q-space of a (2H + 1)-wide output tiles badly (65 = 4 x 16 + 1: 66 % of the MFMA work of a
// 512 -> 256 @64^2 launch is useful) and one long-lived workgroup per CU quantises what is left (208 of 256 CUs busy); (ii) at <= 128 input
// channels the second pass re-reads the patch the layer is HBM-bound on (64 -> 32: 170 us without patch staging).  An edge-row / edge-column
// path that would make the main region H x H (perfect tiling: ~115 us projected for 512 -> 256) is the open continuation.
#ifndef GC_CTWS
#define GC_CTWS 0             // 1: transposed 3x3 convolutions with K % 16 == 0, N % 32 == 0 on convt_bf16x3_ws_kernel
#endif
#ifndef GC_CTWS_ABL
#define GC_CTWS_ABL 0         // dev ablations (wrong results): 1 no patch staging, 2 no weight DMA, 8 no stores
#endif
#if GC_CTWS
#include "experiments/convt_ws.inc.h"
#endif      // GC_CTWS

template <int WG_OC, int WG_PX, int WPX, int TPW>
int launch_t(Bf16Args a, hipStream_t s, bool main_only = false) {
    using C = TCfg<WG_OC, WG_PX, WPX, TPW>;
    const int qh = main_only ? a.c.in_h : gc::ceil_div(a.c.out_h, 2), qw = main_only ? a.c.in_w : gc::ceil_div(a.c.out_w, 2);     // main_only: the H x W region (convt_edge_bf16x3_kernel does the rest)
    a.c.tiles_y = gc::ceil_div(qh, C::TQH);
    a.c.tiles_x = gc::ceil_div(qw, TPW);
    const long long gx = (long long)a.c.tiles_x * a.c.tiles_y * a.c.B;
    if (gx > 2147483647LL) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_bf16x3_f32: grid too large");
    if (gc::probing()) return gc::probe_name("convt_fused_bf16x3_kernel<%d,%d,%d,%d>|up2,down1,k3", WG_OC, WG_PX, WPX, TPW);
    dim3 grid((unsigned)gx, gc::ceil_div(a.c.N, C::OCT), a.k_per_split ? gc::ceil_div(a.c.K, a.k_per_split) : 1);
    const int epi = (a.c.bias || a.c.noise || a.c.act || a.c.residual) ? 2 : (a.c.so ? 1 : 0);
    // LDS-DMA copies whole rows unmasked: every output-channel block and every 16-channel chunk must be complete
    const bool dma = GC_CT_DMA && a.c.N % C::OCT == 0 && a.c.K % KCB == 0;
    if (dma) {
        if (epi == 2)      hipLaunchKernelGGL((convt_fused_bf16x3_kernel<WG_OC, WG_PX, WPX, TPW, 2, true>), grid, dim3(256), 0, s, a);
        else if (epi == 1) hipLaunchKernelGGL((convt_fused_bf16x3_kernel<WG_OC, WG_PX, WPX, TPW, 1, true>), grid, dim3(256), 0, s, a);
        else               hipLaunchKernelGGL((convt_fused_bf16x3_kernel<WG_OC, WG_PX, WPX, TPW, 0, true>), grid, dim3(256), 0, s, a);
        return gc::check_launch("gc_conv2d_bf16x3_f32(fused transposed, weight DMA)");
    }
    if (epi == 2)      hipLaunchKernelGGL((convt_fused_bf16x3_kernel<WG_OC, WG_PX, WPX, TPW, 2>), grid, dim3(256), 0, s, a);
    else if (epi == 1) hipLaunchKernelGGL((convt_fused_bf16x3_kernel<WG_OC, WG_PX, WPX, TPW, 1>), grid, dim3(256), 0, s, a);
    else               hipLaunchKernelGGL((convt_fused_bf16x3_kernel<WG_OC, WG_PX, WPX, TPW, 0>), grid, dim3(256), 0, s, a);
    return gc::check_launch("gc_conv2d_bf16x3_f32(fused transposed)");
}

// q-space is (H + 1) wide for a (2H + 1)-wide output: take the tile width that wastes fewer lanes
int dispatch_t(const Bf16Args& a, hipStream_t s) {
    const int qw = gc::ceil_div(a.c.out_w, 2);
    const bool narrow = GC_CONVT_NARROW && gc::ceil_div(qw, 16) * 16 < gc::ceil_div(qw, 32) * 32;
#if GC_CTWS == 1
    if (tws_eligible(a)) {
        if (a.c.N % 64 != 0) return launch_tws<1, 32>(a, s);                 // 32 oc x 16 rows x 32 q-columns
        return narrow ? launch_tws<2, 16>(a, s) : launch_tws<2, 32>(a, s);    // 64 oc x (16 x 16 | 8 x 32) q-pixels
    }
#endif
    // <= 32 output channels: the layer is bound by its stores, and 32-column q-tiles write 256-byte runs per row instead of 128-byte
    // ones (64 -> 32 @512^2: 254 -> 232 us) -- worth more than the 16 columns of lanes a 513-wide q-row wastes
    if (a.c.N <= 32) return launch_t<1, 4, 2, 32>(a, s);
    if (ct_edge_eligible(a)) {
#if GC_CTWS == 2
        // round 6 experiment: the H x W main region on the wave-specialised kernel (its 8 x 32 q-tiles then cover the region exactly)
        if (a.c.in_h % 8 == 0 && !(a.c.bias || a.c.noise || a.c.act || a.c.residual)) {
            if (gc::probing()) return gc::probe_name("convt_bf16x3_ws_kernel<2,32>+edge|up2,down1,k3");
            if (int rc = launch_tws<2, 32>(a, s, true)) return rc;
            const dim3 grid((unsigned)(gc::ceil_div(a.c.in_w + 1, 32) + gc::ceil_div(a.c.in_h, 32)), (unsigned)(a.c.N / 64), (unsigned)a.c.B);
            hipLaunchKernelGGL(convt_edge_bf16x3_kernel, grid, dim3(256), 0, s, a);
            return gc::check_launch("gc_conv2d_bf16x3_f32(transposed ws, edge)");
        }
#endif
        if (gc::probing()) return gc::probe_name("convt_fused_bf16x3_kernel<2,2,2,32>+edge|up2,down1,k3");
        if (int rc = launch_t<2, 2, 2, 32>(a, s, true)) return rc;
        const dim3 grid((unsigned)(gc::ceil_div(a.c.in_w + 1, 32) + gc::ceil_div(a.c.in_h, 32)), (unsigned)(a.c.N / 64), (unsigned)a.c.B);
        hipLaunchKernelGGL(convt_edge_bf16x3_kernel, grid, dim3(256), 0, s, a);
        return gc::check_launch("gc_conv2d_bf16x3_f32(transposed, edge)");
    }
    return narrow ? launch_t<2, 2, 2, 16>(a, s) : launch_t<2, 2, 2, 32>(a, s);
}

template <int WG_OC, int WG_PX, int WOC, int WPX, int UP, int DOWN, int KS>
int launch(Bf16Args a, hipStream_t s) {
    using C = BCfg<WG_OC, WG_PX, WOC, WPX, UP, DOWN, KS>;
    const int qh = gc::ceil_div(a.c.out_h, UP), qw = gc::ceil_div(a.c.out_w, UP);
    a.c.tiles_y = gc::ceil_div(qh, C::TPH);
    a.c.tiles_x = gc::ceil_div(qw, 32);
    const int tiles = a.c.tiles_x * a.c.tiles_y, ocb = gc::ceil_div(a.c.N, C::OCT);
    // tiles per workgroup: as many as keep >= 8 workgroups per CU-slot pair (2048 on 256 CUs x 2) in the launch, at most 8;
    // few channel chunks per tile is where the per-tile latencies dominate, many chunks need no help
    a.tpb = 1;
    if (UP == 1) {
        const long long wgs = (long long)tiles * a.c.B * ocb;
        const int nchunks = gc::ceil_div(a.c.K, KCB);
        int want = nchunks <= 2 ? 8 : (nchunks <= 4 ? 4 : (nchunks <= 8 ? 2 : 1));
        while (want > 1 && wgs / want < 2048) want >>= 1;
        a.tpb = want;
    }
    if (a.k_per_split) a.tpb = 1;
    a.groups = gc::ceil_div(tiles, a.tpb);
    const long long gx = (long long)a.groups * UP * UP * a.c.B;
    if (gx > 2147483647LL) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_bf16x3_f32: grid too large");
    if (gc::probing()) return gc::probe_name("conv_bf16x3_kernel<%d,%d,%d,%d>|up%d,down%d,k%d", WG_OC, WG_PX, WOC, WPX, UP, DOWN, KS);
    dim3 grid((unsigned)gx, ocb, a.k_per_split ? gc::ceil_div(a.c.K, a.k_per_split) : 1);
    hipLaunchKernelGGL((conv_bf16x3_kernel<WG_OC, WG_PX, WOC, WPX, UP, DOWN, KS>), grid, dim3(256), 0, s, a);
    return gc::check_launch("gc_conv2d_bf16x3_f32");
}

// wave-specialised kernel (conv_bf16x3_ws_kernel): 64 oc x 16 rows x 32 px tiles, one 12-wave workgroup per CU
template <int KS, int WOC, int CB>
int launch_ws(Bf16Args a, hipStream_t s) {
    a.c.tiles_y = gc::ceil_div(a.c.out_h, 16 / CB);
    a.c.tiles_x = gc::ceil_div(a.c.out_w, 32 * CB);
    const int tiles = a.c.tiles_x * a.c.tiles_y, ocb = a.c.N / (32 * WOC);
    const long long wgs = (long long)tiles * a.c.B * ocb;
    // One workgroup per CU is resident, so nothing overlaps a workgroup's start-up (two exposed load latencies) and its store drain:
    // give every workgroup ALL the consecutive tiles its CU would get over the rounds of the launch (the staging waves then run ahead
    // into the next tile while the multiplying waves store the current one).
    a.tpb = (int)std::min<long long>(std::max<long long>((wgs + GC_WS_SLOTS - 1) / GC_WS_SLOTS, 1), tiles);
    a.groups = gc::ceil_div(tiles, a.tpb);
    const long long gx = (long long)a.groups * a.c.B;
    if (gx > 2147483647LL) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_bf16x3_f32: grid too large");
    if (gc::probing()) return gc::probe_name("conv_bf16x3_ws_kernel<%d,%d,%d>|up1,down1,k%d", KS, WOC, CB, KS);
    const bool plain = GC_WS_BARE && !a.c.bias && !a.c.noise && !a.c.act;
    const int epk = !plain ? 0 : ((a.c.so || a.c.residual) ? 1 : 2);
    const bool res = a.c.residual != nullptr;
    if (epk == 2)             hipLaunchKernelGGL((conv_bf16x3_ws_kernel<KS, WOC, CB, 2, false>), dim3((unsigned)gx, ocb), dim3(768), 0, s, a);
    else if (epk == 1 && res) hipLaunchKernelGGL((conv_bf16x3_ws_kernel<KS, WOC, CB, 1, true>), dim3((unsigned)gx, ocb), dim3(768), 0, s, a);
    else if (epk == 1)        hipLaunchKernelGGL((conv_bf16x3_ws_kernel<KS, WOC, CB, 1, false>), dim3((unsigned)gx, ocb), dim3(768), 0, s, a);
    else if (res)             hipLaunchKernelGGL((conv_bf16x3_ws_kernel<KS, WOC, CB, 0, true>), dim3((unsigned)gx, ocb), dim3(768), 0, s, a);
    else                      hipLaunchKernelGGL((conv_bf16x3_ws_kernel<KS, WOC, CB, 0, false>), dim3((unsigned)gx, ocb), dim3(768), 0, s, a);
    return gc::check_launch("gc_conv2d_bf16x3_f32(ws)");
}

// the layers the wave-specialised kernel takes: whole 16-channel chunks and 64-channel output blocks, enough chunks per tile to
// amortise its two-stage start-up, and enough tiles to give every CU a workgroup
inline bool ws_eligible(const Bf16Args& a) {
#ifdef GC_NO_WS
    return false;
#endif
    const ConvArgs& c = a.c;
    const int oct = c.N % 64 == 0 ? 64 : 32;          // 32-channel output blocks for the layers whose N is not a multiple of 64 (the 1024^2 layers: N = 32)
    if (a.k_per_split || c.K % KCB != 0 || c.N % 32 != 0 || c.K < (oct == 64 ? GC_WS_MIN_K : GC_WS_MIN_K32) || c.out_w < 32 || c.out_h < 16) return false;
    const long long wgs = (long long)gc::ceil_div(c.out_w, 32) * gc::ceil_div(c.out_h, 16) * c.B * (c.N / oct);
    return wgs >= 192;
}

template <int UP, int DOWN, int KS>
int dispatch(const Bf16Args& a, hipStream_t s) {
    if constexpr (UP == 1 && DOWN == 1) {
        if (ws_eligible(a)) {
            // wide tiles (8 rows x 64 px) where HBM, not the matrix pipe, bounds the layer: few input channels per output byte
            const bool wide = a.c.K <= GC_WS_WIDE_MAX_K && a.c.out_w >= 64;
            if (a.c.N % 64 == 0) return wide ? launch_ws<KS, 2, 2>(a, s) : launch_ws<KS, 2, 1>(a, s);      // (4 x 128 px tiles of 64 channels do not fit two LDS stages)
            return wide ? (a.c.out_w >= 128 ? launch_ws<KS, 1, GC_WS_WIDE_CB32>(a, s) : launch_ws<KS, 1, 2>(a, s)) : launch_ws<KS, 1, 1>(a, s);
        }
    }
    if constexpr (DOWN == 2) {
#ifndef GC_SINGLE
        // round 6: the large 3x3 layers on the wave-specialised stride-2 kernel (conv_s2ws.hip: E / O half stages, 128 oc x 8 rows x 32 px tiles)
        if constexpr (KS == 3) { if (s2ws_eligible(a)) return launch_s2ws(a, s); }
#endif
        // patch extents double: 4-row tiles only
        if (a.c.N <= 32) return launch<1, 4, 1, 1, UP, DOWN, KS>(a, s);
        return launch<1, 4, 2, 1, UP, DOWN, KS>(a, s);
    } else {
        if (a.c.N <= 32) return launch<1, 4, 1, 2, UP, DOWN, KS>(a, s);     // 32oc x (8 rows x 32 px)
        // 64oc x (8 rows x 32 px) unless that leaves most CUs idle (32x32 / 64x64 planes at batch 4): 4-row tiles
        const int qw = gc::ceil_div(a.c.out_w, UP), qh = gc::ceil_div(a.c.out_h, UP);
        const long long big = (long long)gc::ceil_div(qw, 32) * gc::ceil_div(qh, 8) * UP * UP * a.c.B * gc::ceil_div(a.c.N, 64);
        if (big < 512) return launch<1, 4, 2, 1, UP, DOWN, KS>(a, s);
        return launch<1, 4, 2, 2, UP, DOWN, KS>(a, s);
    }
}

// shapes the split-bf16 kernel is built for; everything else runs on the exact fp32 kernel
bool eligible(const gc_conv_desc* d) {
    const int qw = gc::ceil_div(d->out_w, d->up);
    // 9 .. 16-pixel rows fill part of the 32-pixel tile only, yet beat the fp32 MFMA path (512 -> 512 @16^2, B = 8: 97 vs 201 us;
    // stride 2 @33^2: 128 vs 281 us); at <= 8 pixels the fp32 kernel with its split over K is faster (36 vs 92 us @8^2)
    if (d->in_ch < 16 || d->in_ch > MAX_K_BF16X3 || qw <= 8 || pointwise_thin(d)) return false;
    // the staging scheme wants the patch rows to start at most EDGE floats before a 32-float boundary (see BCfg)
    if (d->up == 1) {
        const int pwd = 31 * d->down + d->kw, edge = pwd - 32 * (pwd % 32 == 0 ? pwd / 32 : (pwd - 1) / 32);
        return d->pad_x >= 0 && d->pad_x <= edge;
    }
    return d->pad_x >= 1 && d->pad_x <= 2;      // up = 2: phase rows start 0 or 1 floats before the boundary only for these
}

// Planes of 9 .. 32 pixels (the 16^2 / 32^2 layers, 512 channels) give the launch only B * tiles * N / 64 = 64 .. 256 workgroups, each
// walking all 32 channel chunks one after the other with nothing to hide the load latency behind (512 -> 512 @16^2, B = 4: 88 us for
// 4.8 GFLOP).  Splitting K over blockIdx.z fills the chip and shortens the dependent chain; slices of >= 4 chunks.
#ifndef GC_CT_SPLITK
#define GC_CT_SPLITK 1        // transposed 3x3 convolutions on small planes split over the input channels
#endif
#ifndef GC_SPLIT_TARGET
#define GC_SPLIT_TARGET 512   // workgroups a split launch aims for
#endif
struct SplitPlan { int slices, k_per_split; };
SplitPlan plan_splitk_bf16(const gc_conv_desc* d) {
    SplitPlan sp{1, 0};
    if (d->in_ch < 128) return sp;
    long long wgs;
    if (d->up == 2) {
        // the fused transposed kernel (round 5: 512 -> 512 @8^2 / @16^2 were 64 / 160 workgroups walking all 32 chunks: 90 us each whatever the plane);
        // dense rows only (the finish pass writes dense rows), > 32 output channels (dispatch_t's 64-channel tiles)
        if (!GC_CT_SPLITK || d->kh != 3 || d->pad_y != 2 || d->pad_x != 2 || d->out_ch <= 32 || (d->out_pitch != 0 && d->out_pitch != d->out_w)) return sp;
        const int qw = gc::ceil_div(d->out_w, 2), qh = gc::ceil_div(d->out_h, 2);
        const bool narrow = GC_CONVT_NARROW && gc::ceil_div(qw, 16) * 16 < gc::ceil_div(qw, 32) * 32;
        wgs = (long long)gc::ceil_div(qw, narrow ? 16 : 32) * gc::ceil_div(qh, narrow ? 8 : 4) * d->batch * gc::ceil_div(d->out_ch, 64);
    } else {
        // tile rows as dispatch() picks them: 4 at stride 2; at stride 1 eight unless that gives < 512 workgroups, then four (round 5: this plan
        // assumed eight rows throughout and cut the stride-1 layers into twice the slices they needed -- 512 -> 512 @32^2: B = 4 79 -> 71 us with
        // two slices instead of four, B = 8 133 -> 114 us unsplit; the partial sums are the cost of a slice)
        int rows = d->down == 2 ? 4 : 8;
        wgs = (long long)gc::ceil_div(d->out_w, 32) * gc::ceil_div(d->out_h, rows) * d->batch * gc::ceil_div(d->out_ch, 64);
        if (d->down == 1 && d->out_ch > 32 && wgs < 512) wgs = (long long)gc::ceil_div(d->out_w, 32) * gc::ceil_div(d->out_h, 4) * d->batch * gc::ceil_div(d->out_ch, 64);
    }
    const int want = (int)std::min<long long>(GC_SPLIT_TARGET / std::max<long long>(wgs, 1), d->in_ch / 64);
    if (want <= 1) return sp;
    sp.k_per_split = gc::ceil_div(gc::ceil_div(d->in_ch, want), KCB) * KCB;
    sp.slices = gc::ceil_div(d->in_ch, sp.k_per_split);
    if (sp.slices <= 1) { sp.slices = 1; sp.k_per_split = 0; }
    return sp;
}

size_t splitk_bytes(const gc_conv_desc* d) {
    const SplitPlan sp = plan_splitk_bf16(d);
    return sp.slices > 1 ? (size_t)sp.slices * d->batch * d->out_ch * d->out_h * d->out_w * sizeof(float) : 0;
}

}  // namespace

#ifdef GC_SINGLE
// the plain-bf16 build shares the queries, the weight pack (its lo half is simply not read) and the workspace layout of the split build
#define gc_conv2d_fused_bf16x3_packed_f32 gc_conv2d_fused_bf16_packed_f32
#define gc_conv2d_wgrad_bf16x3_f32 gc_conv2d_wgrad_bf16_f32
#define gc_conv2d_wgrad_samples_bf16x3_f32 gc_conv2d_wgrad_samples_bf16_f32
#else
extern "C" size_t gc_conv2d_bf16x3_workspace(const gc_conv_desc* d) {
    if (!d || d->in_ch <= 0 || d->out_ch <= 0 || d->kh <= 0 || d->kw <= 0) return 0;
    if (!eligible(d)) return conv2d_f32_workspace(d);        // runs on the fp32 kernel: split-K partial sums (small planes) or nothing
    const size_t units = (size_t)d->kh * d->kw * ((d->in_ch + 7) / 8) * d->out_ch;
    return 2 * units * sizeof(uint4) + splitk_bytes(d);      // the split weights (gc_conv2d_fused_bf16x3_f32 packs them here), then the K slices
}

extern "C" size_t gc_conv2d_bf16x3_splitk_bytes(const gc_conv_desc* d) {
    if (!d || d->batch <= 0 || d->in_ch <= 0 || d->out_ch <= 0 || d->kh <= 0 || d->kw <= 0 || !eligible(d)) return 0;
    return splitk_bytes(d);
}
#endif

#ifndef GC_SINGLE
extern "C" int gc_conv2d_out_pitch(const gc_conv_desc* d, int mode) {
    if (!d || mode == 0 || d->in_ch <= 0 || d->out_ch <= 0 || d->out_w <= 0) return 0;          // the fp32 kernels write dense rows
    if (!(eligible(d) && d->kh == 3 && d->kw == 3 && d->up == 2 && d->pad_y == 2 && d->pad_x == 2)) return 0;
    if (d->out_w % 32 == 0 || d->out_w < 129) return 0;       // already aligned, or too small to matter
    return (d->out_w + 31) / 32 * 32;
}

extern "C" int gc_conv2d_in_pitch_ok(const gc_conv_desc* d, int mode, int wgrad) {
    if (!d || mode == 0 || d->in_ch <= 0 || d->out_ch <= 0 || d->up != 1 || d->down != 2) return 0;
    return wgrad ? (wg_eligible(d) ? 1 : 0) : (eligible(d) ? 1 : 0);
}

extern "C" size_t gc_conv2d_bf16x3_packed_bytes(const gc_conv_desc* d) {
    if (!d || d->in_ch <= 0 || d->out_ch <= 0 || d->kh <= 0 || d->kw <= 0 || !eligible(d)) return 0;
    return 2 * (size_t)d->kh * d->kw * ((d->in_ch + 7) / 8) * d->out_ch * sizeof(uint4);
}

extern "C" int gc_conv2d_pack_weights_bf16x3(const gc_conv_desc* d, const float* w, void* packed, size_t packed_bytes, gc_stream_t stream) {
    int rc = validate(d, "gc_conv2d_pack_weights_bf16x3", false);
    if (rc) return rc;
    const size_t need = gc_conv2d_bf16x3_packed_bytes(d);
    if (need == 0) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_pack_weights_bf16x3: this shape runs on the fp32 kernel and takes no packed weights");
    if (!w || !packed || packed_bytes < need || (reinterpret_cast<uintptr_t>(packed) & 15))
        return gc::fail(GC_ERR_WORKSPACE, "gc_conv2d_pack_weights_bf16x3: buffer %zu < %zu bytes (or null / not 16-byte aligned)", packed_bytes, need);
    const int kgroups = (d->in_ch + 7) / 8, taps = d->kh * d->kw;
    const size_t units = (size_t)taps * kgroups * d->out_ch;
    uint4* wh = static_cast<uint4*>(packed);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)std::min<size_t>((units + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       w, wh, wh + units, taps, d->in_ch, d->out_ch, kgroups);
    return gc::check_launch("gc_conv2d_pack_weights_bf16x3");
}

extern "C" int gc_conv2d_pack_weights_bf16x3_grouped(const gc_wpack_group* groups, int n_groups, gc_stream_t stream) {
    if (n_groups < 0 || (n_groups > 0 && !groups)) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_pack_weights_bf16x3_grouped: bad group table");
    for (int first = 0; first < n_groups; first += MAXPG) {
        PackGroupArgs a;
        a.n_groups = std::min(MAXPG, n_groups - first);
        long long blocks = 0;
        for (int i = 0; i < a.n_groups; ++i) {
            const gc_wpack_group& g = groups[first + i];
            int rc = validate(&g.desc, "gc_conv2d_pack_weights_bf16x3_grouped", false);
            if (rc) return rc;
            const size_t need = gc_conv2d_bf16x3_packed_bytes(&g.desc);
            if (need == 0) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_pack_weights_bf16x3_grouped: group %d runs on the fp32 kernel and takes no packed weights", first + i);
            if (!g.w || !g.packed || g.packed_bytes < need || (reinterpret_cast<uintptr_t>(g.packed) & 15))
                return gc::fail(GC_ERR_WORKSPACE, "gc_conv2d_pack_weights_bf16x3_grouped: group %d: buffer %zu < %zu bytes (or null / not 16-byte aligned)", first + i, (size_t)g.packed_bytes, need);
            const int kgroups = (g.desc.in_ch + 7) / 8, taps = g.desc.kh * g.desc.kw;
            const size_t units = (size_t)taps * kgroups * g.desc.out_ch;
            uint4* wh = static_cast<uint4*>(g.packed);
            a.g[i] = PackGroup{g.w, wh, wh + units, taps, g.desc.in_ch, g.desc.out_ch, kgroups, (int)blocks};
            blocks += (long long)((units + 255) / 256);
            if (blocks > 2147483647LL) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_pack_weights_bf16x3_grouped: too many blocks");
        }
        if (blocks) hipLaunchKernelGGL(pack_weights_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    }
    return gc::check_launch("gc_conv2d_pack_weights_bf16x3_grouped");
}
#endif

extern "C" int gc_conv2d_fused_bf16x3_packed_f32(const gc_conv_desc* d, const float* x, const float* w, const void* packed, size_t packed_bytes,
                                                 const float* in_scale, const float* out_scale, const gc_conv_epilogue* ep, float* y,
                                                 void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    int rc = validate(d, "gc_conv2d_bf16x3_f32", false);
    if (rc) return rc;
    if (!x || !w || !y) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_bf16x3_f32: null pointer");
    if (d->batch == 0) return GC_OK;
    if ((rc = validate_epilogue(ep, "gc_conv2d_bf16x3_f32"))) return rc;
    if (d->in_pitch != 0 && d->in_pitch != d->in_w && !(eligible(d) && d->up == 1 && d->down == 2))
        return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_bf16x3_f32: in_pitch %d: only the split-bf16 stride-2 kernel reads pitched rows (gc_conv2d_in_pitch_ok)", d->in_pitch);
    const bool pitched_ok = eligible(d) && d->kh == 3 && d->up == 2 && d->pad_y == 2 && d->pad_x == 2;      // the fused transposed kernel
    if (!dense_output(d) && !pitched_ok)
        return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_bf16x3_f32: out_pitch %d: only the fused transposed 3x3 convolution writes pitched rows (gc_conv2d_out_pitch)", d->out_pitch);
    if (!eligible(d)) return conv2d_f32_ws(d, x, w, in_scale, out_scale, ep, y, workspace, workspace_bytes, stream);
    const size_t need = gc_conv2d_bf16x3_packed_bytes(d);
    if (!packed || packed_bytes < need || (reinterpret_cast<uintptr_t>(packed) & 15))
        return gc::fail(GC_ERR_WORKSPACE, "gc_conv2d_bf16x3_f32: packed weights %zu < %zu bytes (or null / not 16-byte aligned)", packed_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    const int kgroups = (d->in_ch + 7) / 8, taps = d->kh * d->kw;
    const size_t units = (size_t)taps * kgroups * d->out_ch;
    const uint4* wh = static_cast<const uint4*>(packed);
    const uint4* wl = wh + units;
    Bf16Args a{{x, w, in_scale, out_scale, y, d->batch, d->in_ch, d->out_ch, d->in_h, d->in_w, d->out_h, d->out_w,
                d->pad_y, d->pad_x, 0, 0}, wh, wl, kgroups, 1, 1, 0, nullptr, 0, d->out_pitch ? d->out_pitch : d->out_w, d->in_pitch ? d->in_pitch : d->in_w};
    set_epilogue(a.c, ep);
    // small planes: split over K when the caller brought room for the slices (gc_conv2d_bf16x3_splitk_bytes)
    const SplitPlan sp = plan_splitk_bf16(d);
    const size_t slice_bytes = splitk_bytes(d);
    const bool split = sp.slices > 1 && workspace && workspace_bytes >= slice_bytes && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0;
    ConvArgs fin = a.c;
    if (split) {
        a.c.so = nullptr;
        set_epilogue(a.c, nullptr);
        a.k_per_split = sp.k_per_split;
        a.part = static_cast<float*>(workspace);
        a.per_slice = (long long)d->batch * d->out_ch * d->out_h * d->out_w;
    }
    if (d->kh == 3) {
        if (d->up == 2 && d->pad_y == 2 && d->pad_x == 2) rc = dispatch_t(a, s);
        else if (d->up == 2) return dispatch<2, 1, 3>(a, s);
        else rc = d->down == 2 ? dispatch<1, 2, 3>(a, s) : dispatch<1, 1, 3>(a, s);
    } else {
        if (d->up == 2) return dispatch<2, 1, 1>(a, s);
        rc = d->down == 2 ? dispatch<1, 2, 1>(a, s) : dispatch<1, 1, 1>(a, s);
    }
    if (rc || !split) return rc;
    fin.part = a.part;
    return launch_splitk_finish(fin, sp.slices, a.per_slice, s);
}

#ifndef GC_SINGLE
extern "C" int gc_conv2d_fused_bf16x3_f32(const gc_conv_desc* d, const float* x, const float* w,
                                          const float* in_scale, const float* out_scale, const gc_conv_epilogue* ep, float* y,
                                          void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    int rc = validate(d, "gc_conv2d_bf16x3_f32", false);
    if (rc) return rc;
    if (d->batch == 0) return GC_OK;
    const size_t need = gc_conv2d_bf16x3_packed_bytes(d);
    if (need == 0) return gc_conv2d_fused_bf16x3_packed_f32(d, x, w, nullptr, 0, in_scale, out_scale, ep, y, workspace, workspace_bytes, stream);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
        return gc::fail(GC_ERR_WORKSPACE, "gc_conv2d_bf16x3_f32: workspace %zu < %zu bytes (or not 16-byte aligned)", workspace_bytes, need);
    if (!w) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_bf16x3_f32: null pointer");
    if ((rc = gc_conv2d_pack_weights_bf16x3(d, w, workspace, workspace_bytes, stream))) return rc;
    char* tail = static_cast<char*>(workspace) + need;
    return gc_conv2d_fused_bf16x3_packed_f32(d, x, w, workspace, need, in_scale, out_scale, ep, y, workspace_bytes > need ? tail : nullptr,
                                             workspace_bytes > need ? workspace_bytes - need : 0, stream);
}

extern "C" int gc_conv2d_bf16x3_f32(const gc_conv_desc* d, const float* x, const float* w,
                                    const float* in_scale, const float* out_scale, float* y,
                                    void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    return gc_conv2d_fused_bf16x3_f32(d, x, w, in_scale, out_scale, nullptr, y, workspace, workspace_bytes, stream);
}

extern "C" size_t gc_conv2d_wgrad_bf16x3_workspace(const gc_conv_desc* d) {
    if (!d || d->batch <= 0 || d->in_ch <= 0 || d->out_ch <= 0 || d->out_h <= 0 || d->out_w <= 0) return 0;
    size_t need = gc_conv2d_wgrad_workspace(d);
    if (wg_eligible(d)) {
        const WgPlan pl = plan_wg(d);
        need = std::max(need, (size_t)pl.splits * d->kh * d->kw * d->in_ch * d->out_ch * sizeof(float));
    }
    return need;
}
#endif

namespace {

// per-sample mode: the pixel splits of plan_wg regrouped as B x spb, every split inside one sample
WgPlan plan_wg_samples(const gc_conv_desc* d) {
    WgPlan pl = plan_wg(d);
    const int per_sample = pl.tiles_x * pl.tiles_y;
    const int ctiles = gc::ceil_div(d->in_ch, pl.kt) * gc::ceil_div(d->out_ch, pl.ct);
    int spb = gc::ceil_div(gc::ceil_div(512, ctiles), d->batch);
    spb = std::max(1, std::min(spb, per_sample));
    pl.tiles_per_split = gc::ceil_div(per_sample, spb);
    pl.splits = spb * d->batch;
    return pl;
}

// dw_samples == nullptr: dw = the sum over the batch (gc_conv2d_wgrad_bf16x3_f32); else also dw_samples[b] = sample b's share of it
int wgrad_launch(const gc_conv_desc* d, const float* x, const float* dy, const float* in_scale, const float* out_scale, float* dw,
                 float* dw_samples, void* workspace, size_t workspace_bytes, hipStream_t s, const char* who) {
    const WgPlan pl = dw_samples ? plan_wg_samples(d) : plan_wg(d);
    const size_t count = (size_t)d->kh * d->kw * d->in_ch * d->out_ch;
    const size_t need = (size_t)pl.splits * count * sizeof(float);
    const bool direct = pl.splits == 1 && !dw_samples;
    if (!direct && (!workspace || workspace_bytes < need)) return gc::fail(GC_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", who, workspace_bytes, need);
    WgArgs a{x, dy, in_scale, out_scale, direct ? dw : static_cast<float*>(workspace), d->batch, d->in_ch, d->out_ch,
             d->in_h, d->in_w, d->out_h, d->out_w, d->pad_y, d->pad_x, pl.tiles_x, pl.tiles_y, pl.tiles_per_split, d->in_pitch ? d->in_pitch : d->in_w,
             dw_samples ? pl.splits / d->batch : 0};
    dim3 grid(gc::ceil_div(d->in_ch, pl.kt), gc::ceil_div(d->out_ch, pl.ct), pl.splits);
#if GC_WG_WS
    // the wave-specialised kernel: 3 x 3 "same" convolutions with whole 64-channel blocks on both sides; strips of rb rows, at least two per split
    if (d->down == 1 && d->kh == 3 && !pl.small && d->in_ch % 64 == 0 && d->out_ch % 64 == 0 && d->pad_x == 1 && d->pad_y == 1 && d->out_w >= 32 &&
        d->out_h == d->in_h && d->out_w == d->in_w) {
        const int per = dw_samples ? pl.splits / d->batch : pl.splits;                     // splits that share one pool of strips
        const long long pool = (long long)pl.tiles_x * (dw_samples ? 1 : d->batch);          // ... strips per row band in that pool
        int rb = 0;
        for (int cand = 16; cand >= 2; cand >>= 1)
            if (d->out_h % cand == 0 && pool * (d->out_h / cand) >= 2LL * per) { rb = cand; break; }
        if (GC_WG_WS == 2) rb = (d->out_h % 16 == 0 && pool * (d->out_h / 16) >= 2LL * per) ? 16 : 0;
        if (rb) {
            if (gc::probing()) return gc::probe_name("wgrad_bf16x3_ws_kernel|rb%d", rb);
#if GC_WG_WS == 2
            hipLaunchKernelGGL(wgrad_bf16x3_ws2_kernel, grid, dim3(1024), 0, s, a, d->out_h / rb);
#else
            hipLaunchKernelGGL(wgrad_bf16x3_ws_kernel, grid, dim3(1024), 0, s, a, rb, d->out_h / rb);      // experiments/wgrad_ws1.inc.h
#endif
            int rc = gc::check_launch(who);
            if (rc || direct) return rc;
            if (dw_samples) return launch_wgrad_reduce_samples(static_cast<const float*>(workspace), dw, dw_samples, count, d->batch, pl.splits / d->batch, s);
            return launch_wgrad_reduce(static_cast<const float*>(workspace), dw, count, pl.splits, s);
        }
    }
#endif
    if (d->down == 2) {
        if (pl.ct == 128) {
            if (gc::probing()) return gc::probe_name("wgrad_bf16x3_s2_kernel<1,%d,1,4>", d->kh);
            if (d->kh == 3) hipLaunchKernelGGL((wgrad_bf16x3_s2_kernel<1, 3, 1, 4>), grid, dim3(256), 0, s, a);
            else            hipLaunchKernelGGL((wgrad_bf16x3_s2_kernel<1, 1, 1, 4>), grid, dim3(256), 0, s, a);
        } else if (pl.kt == 32) {
            if (d->kh == 3) hipLaunchKernelGGL((wgrad_bf16x3_s2_kernel<2, 3, 1>), grid, dim3(256), 0, s, a);
            else            hipLaunchKernelGGL((wgrad_bf16x3_s2_kernel<2, 1, 1>), grid, dim3(256), 0, s, a);
        } else if (pl.tr == 1) {
            if (d->kh == 3) hipLaunchKernelGGL((wgrad_bf16x3_s2_kernel<1, 3, 2>), grid, dim3(256), 0, s, a);
            else            hipLaunchKernelGGL((wgrad_bf16x3_s2_kernel<1, 1, 2>), grid, dim3(256), 0, s, a);
        } else {
            return gc::fail(GC_ERR_UNSUPPORTED, "%s: no stride-2 kernel for this tile plan", who);
        }
    } else if (pl.small) {
        if (d->kh == 3) hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 1, 4, 6, 3>), grid, dim3(256), 0, s, a);
        else            hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 1, 4, 6, 1>), grid, dim3(256), 0, s, a);
    } else {
#if defined(GC_ABL)      // dev ablation: GC_ABL_DYNLDS=<bytes> of dynamic LDS forces one workgroup per CU
        static const int dyn = getenv("GC_ABL_DYNLDS") ? atoi(getenv("GC_ABL_DYNLDS")) : 0;
        if (dyn > 0 && d->kh == 3) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16x3_kernel<2, 2, 1, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
            hipLaunchKernelGGL((wgrad_bf16x3_kernel<2, 2, 1, 2, 3>), grid, dim3(256), dyn, s, a);
        } else
#endif
        if (d->kh == 3) hipLaunchKernelGGL((wgrad_bf16x3_kernel<2, 2, 1, 2, 3>), grid, dim3(256), 0, s, a);
        else            hipLaunchKernelGGL((wgrad_bf16x3_kernel<2, 2, 1, 2, 1>), grid, dim3(256), 0, s, a);
    }
    int rc = gc::check_launch(who);
    if (rc || direct) return rc;
    if (dw_samples) return launch_wgrad_reduce_samples(static_cast<const float*>(workspace), dw, dw_samples, count, d->batch, pl.splits / d->batch, s);
    return launch_wgrad_reduce(static_cast<const float*>(workspace), dw, count, pl.splits, s);
}

}  // namespace

extern "C" int gc_conv2d_wgrad_bf16x3_f32(const gc_conv_desc* d, const float* x, const float* dy,
                                          const float* in_scale, const float* out_scale, float* dw,
                                          void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    int rc = validate(d, "gc_conv2d_wgrad_bf16x3_f32", true);
    if (rc) return rc;
    if (!x || !dy || !dw) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_wgrad_bf16x3_f32: null pointer");
    if (d->in_pitch != 0 && d->in_pitch != d->in_w && !(wg_eligible(d) && d->down == 2))
        return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_wgrad_bf16x3_f32: in_pitch %d: only the split-bf16 stride-2 kernel reads pitched rows (gc_conv2d_in_pitch_ok)", d->in_pitch);
    if (d->batch == 0 || !wg_eligible(d) || wgrad_small_eligible(d)) return gc_conv2d_wgrad_f32(d, x, dy, in_scale, out_scale, dw, workspace, workspace_bytes, stream);
    return wgrad_launch(d, x, dy, in_scale, out_scale, dw, nullptr, workspace, workspace_bytes, (hipStream_t)stream, "gc_conv2d_wgrad_bf16x3_f32");
}

#ifndef GC_SINGLE
extern "C" size_t gc_conv2d_wgrad_samples_workspace(const gc_conv_desc* d, int mode) {
    if (!d || d->batch <= 0 || d->in_ch <= 0 || d->out_ch <= 0 || d->out_h <= 0 || d->out_w <= 0 || d->kh <= 0 || d->kw <= 0) return 0;
    if (pointwise_thin_wgrad(d)) return pointwise_wgrad_workspace(d);
    if (mode == 0 || !wg_eligible(d) || d->batch > 16384) return 0;      // the splits are a grid dimension: B x spb <= 65535 with room to spare
    return (size_t)plan_wg_samples(d).splits * d->kh * d->kw * d->in_ch * d->out_ch * sizeof(float);
}
#endif

// gc_conv2d_wgrad_samples_bf16x3_f32 / _bf16_f32: the weight gradient AND each sample's share of it (header: what the shares are for)
extern "C" int gc_conv2d_wgrad_samples_bf16x3_f32(const gc_conv_desc* d, const float* x, const float* dy, const float* in_scale, const float* out_scale,
                                                  float* dw, float* dw_samples, void* workspace, size_t workspace_bytes, gc_stream_t stream) {
    int rc = validate(d, "gc_conv2d_wgrad_samples_bf16x3_f32", true);
    if (rc) return rc;
    if (!x || !dy || !dw || !dw_samples) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_wgrad_samples_bf16x3_f32: null pointer");
    if (d->batch == 0) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_wgrad_samples_bf16x3_f32: empty batch");
    if (pointwise_thin_wgrad(d)) return gc_conv2d_wgrad_samples_f32(d, x, dy, in_scale, out_scale, dw, dw_samples, workspace, workspace_bytes, stream);
    if (!wg_eligible(d) || d->batch > 16384)
        return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_wgrad_samples_bf16x3_f32: shape not taken by the split-bf16 weight-gradient kernels (gc_conv2d_wgrad_samples_workspace() == 0)");
    if (d->in_pitch != 0 && d->in_pitch != d->in_w && d->down != 2)
        return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_wgrad_samples_bf16x3_f32: in_pitch %d: only the stride-2 kernel reads pitched rows", d->in_pitch);
    return wgrad_launch(d, x, dy, in_scale, out_scale, dw, dw_samples, workspace, workspace_bytes, (hipStream_t)stream, "gc_conv2d_wgrad_samples_bf16x3_f32");
}

#if GC_WS_TRACE && !defined(GC_SINGLE)
// dev: copy the trace of the last conv_bf16x3_ws_kernel launches to the host (tools/ws_trace.py)
extern "C" int gc_debug_ws_trace(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gc_ws_trace), sizeof(unsigned long long) * 2 * 512, 0, hipMemcpyDeviceToHost);
}
#endif
