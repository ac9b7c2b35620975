// Stride-2 3x3 convolution in split-bf16 on the wave-specialised structure (round 6).
//
// Replaces: the `F.conv2d(..., stride=2)` behind Blur + EqualConv2d in D's ResBlocks (reference src/gan_control/models/gan_model.py:857-870,
// :152-162) and the input gradient of G's transposed up-sampling convolution (:295-307) -- pad 0, (2H + 1)^2 -> H^2.
//
// Why a kernel of its own.  At stride 2 a tile's input patch is four times the size of a stride-1 tile's: two LDS stages of
// [weights | patch] -- what conv_bf16x3_ws_kernel keeps -- only fit for 4-row tiles of 64 channels, and those leave a wave one 64 oc x 1 row
// block (one LDS fragment read per MFMA, 39 % of the wave time waiting: profiles/pmc_r03_stride2.md, presplit_kill_r05.md).  Here an item
// (tile x 16-channel chunk) is split by the PARITY OF THE INPUT ROW instead:
//   E sub-item: tap rows ky = 0 and 2 (six taps) read the tile's EVEN input rows (9 of its 17);
//   O sub-item: tap row  ky = 1     (three taps) reads the ODD rows (8).
// Both accumulate into the same registers.  The two LDS stages are then the E buffer and the O buffer of ONE 8-row tile: while the multiplying
// waves work on E(c) the staging waves fill O(c); while they work on O(c), E(c + 1) is filled.  87 + 58 KB hold a tile of 128 oc x 8 rows x
// 32 px, so that
//   * a multiplying wave owns 64 oc x 2 rows x 32 px: the 2 x 2 register blocking of the stride-1 kernel (0.67 fragment reads per MFMA);
//   * a patch is staged once per 128 output channels (half the staging work per MFMA of a 64-channel tile);
//   * the multiplying waves (8; 4 at 64 output channels) issue only fragment reads and MFMAs; the staging waves (4; 8) load, scale, split and
//     write the next sub-item; the pre-split weight slab goes HBM -> LDS by LDS-DMA -- exactly the roles of conv_bf16x3_ws_kernel.
// Patch columns are de-interleaved in LDS (33 even columns, then 32 odd ones), so lane l's fragment of column 2 l + kx is a run of consecutive
// 16-byte units (conflict-free ds_read_b128), as in conv_bf16x3_kernel.
// pad = 0 only: every input element outside the image then feeds only output pixels outside the output plane (output pixel (y, x) reads rows
// 2y .. 2y + 2 <= in_h - 1), and an MFMA keeps output pixels apart (B column = pixel), so nothing is masked: out-of-buffer reads return zero by
// the buffer descriptor, in-buffer garbage (the next row, the pad of a pitched row) lands in accumulators that are never stored.
#include "conv_bf16x3_shared.h"

#ifndef GC_S2WS_ABL
#define GC_S2WS_ABL 0       // dev ablations (wrong results): 1 no patch staging (loads, conversion, LDS writes), 2 no weight DMA, 4 no conversion (raw registers to LDS), 8 no stores
#endif
#ifndef GC_S2WS_STAGER_PRIO
#define GC_S2WS_STAGER_PRIO 0
#endif
#ifndef GC_S2WS_NT_LOAD
#define GC_S2WS_NT_LOAD 0   // 1: non-temporal patch loads where the patch is read by ONE output-channel block (N == the block width); 2: always
#endif
#ifndef GC_S2WS_STRIDED
#define GC_S2WS_STRIDED 1   // a workgroup's tiles are `groups` apart (0: consecutive)
#endif
#ifndef GC_S2WS_DEEP
#define GC_S2WS_DEEP 0      // 1: the 64-channel variant (eight staging waves) keeps two register sets per row parity, loads two intervals ahead.  MEASURED NEUTRAL (round 6,
                            // profiles/s2ws_deep_r06_i.log: 32 -> 64 @1025^2, B = 8, 400 vs 403 us): that layer is not bound by what the staging waves have in flight -- without any
                            // patch staging it still takes 255 us of the ~300 us its HBM traffic needs: with two chunks per tile the four multiplying waves spend a
                            // third of a tile in the epilogue and the 64 stores per lane, during which no MFMA issues
#endif
#ifndef GC_S2WS_WOCB1_MAX_K
#define GC_S2WS_WOCB1_MAX_K 0    // input channels up to which layers with N % 128 == 0 take the 64-channel variant too (eight staging waves, the patch staged per 64 channels): 64 -> 128 @513^2 measured 8 % SLOWER on it
#endif
#ifndef GC_S2WS_MIN_WGS
#define GC_S2WS_MIN_WGS 192  // tiles x samples x output-channel blocks from which the kernel is used (one workgroup per CU is resident)
#endif

namespace {

using namespace gcconv;

template <int WOCB>
struct S2Cfg {
    static constexpr int OCT = 64 * WOCB;                  // output channels per workgroup
    static constexpr int MW = 4 * WOCB;                    // multiplying waves: (oc half) x (row pair)
    static constexpr int SW = 12 - MW;                     // staging waves
    static constexpr int TR = 8;                           // output rows per tile (x 32 output columns)
    static constexpr int RP = 65;                          // units per patch row: columns 0, 2, .., 64, then 1, 3, .., 63
    static constexpr int ROWS_E = TR + 1, ROWS_O = TR;
    static constexpr int PLANE_E = ROWS_E * RP, PLANE_O = ROWS_O * RP;             // units per 8-channel group
    static constexpr int WU_E = 6 * KG * OCT, WU_O = 3 * KG * OCT;                 // weight units per half (hi or lo)
    static constexpr int STAGE_E = 2 * (WU_E + KG * PLANE_E), STAGE_O = 2 * (WU_O + KG * PLANE_O);      // [W hi | W lo | P hi | P lo]
    static constexpr int LANES = 32 * SW;                  // staging lanes per 8-channel group
    static constexpr int TPR = 17;                         // staging tasks per patch row: 16 groups of four columns + column 64
    static constexpr int NT_E = (ROWS_E * TPR + LANES - 1) / LANES, NT_O = (ROWS_O * TPR + LANES - 1) / LANES;
};

// EPK / RES: as conv_bf16x3_ws_kernel (0 full fused epilogue, 1 out_scale and / or residual only, 2 nothing; RES: a residual is added)
template <int WOCB, int EPK, bool RES>
__global__ __launch_bounds__(768) void conv_s2ws_bf16x3_kernel(Bf16Args a) {
    using C = S2Cfg<WOCB>;
    constexpr int OCT = C::OCT, MW = C::MW, RP = C::RP, TR = C::TR;
    constexpr bool DEEP = GC_S2WS_DEEP && C::NT_E == 1 && C::NT_O == 1;
    static_assert((C::STAGE_E + C::STAGE_O) * 16 + (MAX_K_BF16X3 + KCB + 2 * OCT) * 4 <= 160 * 1024, "the E and the O stage fit the 160 KiB of LDS");
    const ConvArgs& p = a.c;
    __shared__ uint4 smem[C::STAGE_E + C::STAGE_O];
    __shared__ __attribute__((aligned(16))) float s_si[MAX_K_BF16X3 + KCB];
    __shared__ __attribute__((aligned(16))) float s_so[OCT], s_bias[OCT];
    uint4* const stE = smem;
    uint4* const stO = smem + C::STAGE_E;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    const int bid = blockIdx.x;
    const int grp = bid % a.groups, b = bid / a.groups;
    const int n0 = blockIdx.y * OCT;
    // a workgroup's tiles are `groups` apart (the resident workgroups work on neighbouring tiles of one sample: DRAM locality)
    const int tiles_all = p.tiles_x * p.tiles_y;
    const int tstep = GC_S2WS_STRIDED ? a.groups : 1, tile_begin = GC_S2WS_STRIDED ? grp : grp * a.tpb;
    const int ntiles = GC_S2WS_STRIDED ? (tiles_all - grp + a.groups - 1) / a.groups : min(tiles_all, tile_begin + a.tpb) - tile_begin;
    const int nchunks = p.K / KCB;
    const int items = ntiles * nchunks;          // an item = E sub-item + O sub-item
    const int chan = p.in_h * a.in_pitch;        // floats per input channel (pitched rows: the Blur's output)

    for (int k = tid; k < p.K; k += 768) s_si[k] = p.si ? p.si[(size_t)b * p.K + k] : 1.f;
    if (tid < OCT) {
        const int oc = n0 + tid;
        s_so[tid] = p.so ? p.so[(size_t)b * p.N + oc] : 1.f;
        s_bias[tid] = p.bias ? p.bias[oc] : 0.f;
    }
    __syncthreads();

    if (wave >= MW) {
        // ---------------- staging waves ----------------
        if (GC_S2WS_STAGER_PRIO) __builtin_amdgcn_s_setprio(GC_S2WS_STAGER_PRIO);
        const int st = tid - 64 * MW;
        const int kgl = __builtin_amdgcn_readfirstlane(st / C::LANES), tb = st % C::LANES;
        const float* xb = p.x + (size_t)b * p.K * chan;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)p.K * chan * 4u);
        // one register set per row parity: the loads of the sub-item after the next are in flight while the next one is converted and written
        uint4 pe[C::NT_E][8], po[C::NT_O][8];
        const bool nt_loads = p.N == OCT;
        auto loads = [&](auto ph, auto& preg, int tile, int k0, bool valid) {
            constexpr int PHASE = decltype(ph)::value, ROWS = PHASE == 0 ? C::ROWS_E : C::ROWS_O, NT = PHASE == 0 ? C::NT_E : C::NT_O;
            const int iy0 = (tile / p.tiles_x) * (2 * TR) + PHASE, ix0 = (tile % p.tiles_x) * 64;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int t = tb + C::LANES * j;
                const int row = t / C::TPR, g = t % C::TPR;
                const bool ok = valid && t < ROWS * C::TPR;
                const unsigned boff = ok ? (unsigned)((iy0 + 2 * row) * a.in_pitch + ix0 + 4 * g) * 4u : OOB;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const unsigned soff = (unsigned)(k0 + kgl * 8 + q) * chan * 4u;
                    if (GC_S2WS_NT_LOAD == 2 || (GC_S2WS_NT_LOAD == 1 && nt_loads))
                        preg[j][q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)boff, (int)__builtin_amdgcn_readfirstlane(soff), 2));
                    else
                        preg[j][q] = buf_load_u128(rx, boff, soff);
                }
            }
        };
        auto convert = [&](auto ph, auto& preg, int k0) {
            constexpr int PHASE = decltype(ph)::value, ROWS = PHASE == 0 ? C::ROWS_E : C::ROWS_O, NT = PHASE == 0 ? C::NT_E : C::NT_O;
            constexpr int PL = PHASE == 0 ? C::PLANE_E : C::PLANE_O, WU = PHASE == 0 ? C::WU_E : C::WU_O;
            uint4* const p_h = (PHASE == 0 ? stE : stO) + 2 * WU;
            uint4* const p_l = p_h + KG * PL;
            const int kk = min(k0, p.K - KCB);
            const float4 sa = *reinterpret_cast<const float4*>(&s_si[kk + kgl * 8]), sb = *reinterpret_cast<const float4*>(&s_si[kk + kgl * 8 + 4]);
            const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
            auto body = [&](auto scaled) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int t = tb + C::LANES * j;
                    const int row = t / C::TPR, g = t % C::TPR;
                    if (NT * C::LANES > ROWS * C::TPR && t >= ROWS * C::TPR) continue;
                    const int rbase = kgl * PL + row * RP;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i > 0 && g == 16) continue;            // column 64: the first pixel of the group only
                        // column 4 g + i: even columns at unit c / 2, odd ones at 33 + c / 2
                        const int u = rbase + ((i & 1) ? 33 : 0) + 2 * g + (i >> 1);
                        if (GC_S2WS_ABL & 4) { p_h[u] = preg[j][i]; p_l[u] = preg[j][4 + i]; continue; }
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const unsigned raw = i == 0 ? preg[j][q].x : (i == 1 ? preg[j][q].y : (i == 2 ? preg[j][q].z : preg[j][q].w));
                            v[q] = __uint_as_float(raw);
                        }
                        uint4 h, l;
                        split8s<decltype(scaled)::value>(v, sc, &h, &l);
                        p_h[u] = h;
                        p_l[u] = l;
                    }
                }
            };
            if (p.si) body(std::true_type{}); else body(std::false_type{});       // without modulation (every layer of D) the multiply by one is not issued
        };
        using PE = std::integral_constant<int, 0>;
        using PO = std::integral_constant<int, 1>;
        auto advance = [&](int& tile, int& k0) { k0 += KCB; if (k0 >= p.K) { k0 = 0; tile += tstep; } };
        if constexpr (DEEP) {
            // Eight staging waves, one task per lane and sub-item: a register set is 32 registers, so every parity keeps TWO sets and the loads of a
            // sub-item are in flight for two intervals instead of one (the 64-channel variant serves the layers with 32 / 64 input channels: they are
            // bound by what the staging waves have in flight, not by the matrix pipes).  Fetch order E(0) O(0) E(1) | O(1) E(2) O(2) E(3) ...
            uint4 pe2[C::NT_E][8], po2[C::NT_O][8];
            int tF = tile_begin, kF = 0, nFE = 0, nFO = 0;      // the E and the O fetch of an item share its (tile, chunk): one cursor, advanced after the O fetch ...
            int tG = tile_begin, kG = 0;                        // ... but the E fetches run one item ahead of the O fetches: a cursor of their own
            int kcE = 0, kcO = 0;                               // chunk of the next E / O sub-item to convert (its scales)
            auto fetchE = [&](auto& set) { if (!(GC_S2WS_ABL & 1)) loads(PE{}, set, tG, kG, nFE < items); advance(tG, kG); ++nFE; };
            auto fetchO = [&](auto& set) { if (!(GC_S2WS_ABL & 1)) loads(PO{}, set, tF, kF, nFO < items); advance(tF, kF); ++nFO; };
            auto convE = [&](auto& set) { if (!(GC_S2WS_ABL & 1)) convert(PE{}, set, kcE); kcE += KCB; if (kcE >= p.K) kcE = 0; };
            auto convO = [&](auto& set) { if (!(GC_S2WS_ABL & 1)) convert(PO{}, set, kcO); kcO += KCB; if (kcO >= p.K) kcO = 0; };
            fetchE(pe); fetchO(po); fetchE(pe2);
            convE(pe);
            __syncthreads();
            for (int it = 0; it < items; it += 2) {
                fetchO(po2); convO(po);  __syncthreads();       // the multiplying waves are on E(it)
                fetchE(pe);  convE(pe2); __syncthreads();       // ... on O(it)
                if (it + 1 >= items) break;
                fetchO(po);  convO(po2); __syncthreads();       // ... on E(it + 1)
                fetchE(pe2); convE(pe);  __syncthreads();       // ... on O(it + 1)
            }
            return;
        }
        int tE = tile_begin, kE = 0, tO = tile_begin, kO = 0, nE = 0, nO = 0;         // cursors of the next E / O sub-item to fetch, and their item numbers
        if (!(GC_S2WS_ABL & 1)) { loads(PE{}, pe, tE, kE, true); loads(PO{}, po, tO, kO, true); convert(PE{}, pe, kE); }
        __syncthreads();
        // interval 2 it: the multiplying waves are on E(it); O(it) is converted here and E(it + 1) is fetched.  Interval 2 it + 1: they are on
        // O(it); E(it + 1) is converted and O(it + 1) is fetched.  Items past the last one are fetched as zeros into a stage nobody reads.
        for (int it = 0; it < items; ++it) {
            const int kO_cur = kO;
            advance(tE, kE); ++nE;
            if (!(GC_S2WS_ABL & 1)) { loads(PE{}, pe, tE, kE, nE < items); convert(PO{}, po, kO_cur); }
            __syncthreads();
            advance(tO, kO); ++nO;
            if (!(GC_S2WS_ABL & 1)) { loads(PO{}, po, tO, kO, nO < items); convert(PE{}, pe, kE); }
            __syncthreads();
        }
        return;
    }

    // ---------------- multiplying waves ----------------
    const int och = wave % WOCB, rp = wave / WOCB;           // this wave: output channels och * 64 .. + 63, tile rows 2 rp and 2 rp + 1
    constexpr int WOC = 2, WPX = 2;
    f32x16 acc[WOC][WPX];
#pragma unroll
    for (int i = 0; i < WOC; ++i)
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int aoff = hi * OCT + och * 64 + l31;

    // the weight slab of a sub-item: rows (half, tap, kg) of OCT units, 64 units = 1 KiB per LDS-DMA instruction, dealt round-robin to the
    // multiplying waves (6 each for E, 3 for O).  Issued from an asm statement (untracked: see conv_bf16x3_ws_kernel), completion counted by hand.
    auto weights = [&](auto ph, int k0) {
        constexpr int PHASE = decltype(ph)::value, NTAPS = PHASE == 0 ? 6 : 3, WU = NTAPS * KG * OCT;
        constexpr int INSTR = 2 * NTAPS * KG * WOCB;
        static_assert(INSTR % MW == 0, "whole rounds of DMA instructions");
        uint4* const base = PHASE == 0 ? stE : stO;
#pragma unroll
        for (int j = 0; j < INSTR / MW; ++j) {
            const int q = wave + MW * j;                     // wave-uniform
            const int r = q / WOCB, part = q % WOCB;
            const int half = r / (NTAPS * KG), rr = r % (NTAPS * KG);
            const int t = rr / KG, kg = rr % KG;
            const int gt = PHASE == 0 ? (t / 3) * 6 + t % 3 : 3 + t;             // tap index in the [ky][kx] order of the packed slab
            const uint4* src = (half ? a.wl : a.wh) + ((size_t)(gt * a.kgroups + k0 / 8 + kg) * p.N + n0 + part * 64 + lane);
            glds16(src, base + half * WU + rr * OCT + part * 64);
        }
    };
    auto multiply = [&](auto ph) {
        constexpr int PHASE = decltype(ph)::value, NTAPS = PHASE == 0 ? 6 : 3, WU = NTAPS * KG * OCT;
        constexpr int PL = PHASE == 0 ? C::PLANE_E : C::PLANE_O;
        const uint4* const wl_h = PHASE == 0 ? stE : stO;
        const uint4* const wl_l = wl_h + WU;
        const uint4* const p_h = wl_l + WU;
        const uint4* const p_l = p_h + KG * PL;
        int boff[WPX];
#pragma unroll
        for (int j = 0; j < WPX; ++j) boff[j] = hi * PL + (2 * rp + j) * RP + l31;
        // fragment double buffer: the eight ds_read_b128 of tap t + 1 are issued before the twelve MFMAs of tap t
        bf16x8 fa[2][2 * WOC], fb[2][2 * WPX];
        auto load_tap = [&](int t, int set) {
            const int ky2 = PHASE == 0 ? t / 3 : 0, kx = t % 3;
            const int wbase = t * KG * OCT + aoff;
            const int pbase = ky2 * RP + (kx == 1 ? 33 : kx / 2);
#pragma unroll
            for (int i = 0; i < WOC; ++i) {
                const uint4 uh = wl_h[wbase + i * 32], ul = wl_l[wbase + i * 32];
                fa[set][i] = *reinterpret_cast<const bf16x8*>(&uh);
                fa[set][WOC + i] = *reinterpret_cast<const bf16x8*>(&ul);
            }
#pragma unroll
            for (int j = 0; j < WPX; ++j) {
                const uint4 uh = p_h[pbase + boff[j]], ul = p_l[pbase + boff[j]];
                fb[set][j] = *reinterpret_cast<const bf16x8*>(&uh);
                fb[set][WPX + j] = *reinterpret_cast<const bf16x8*>(&ul);
            }
        };
        load_tap(0, 0);
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            if (t + 1 < NTAPS) load_tap(t + 1, (t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < WOC; ++i)
#pragma unroll
                for (int j = 0; j < WPX; ++j) { GC_MFMA3(acc[i][j], fa[t & 1][i], fa[t & 1][WOC + i], fb[t & 1][j], fb[t & 1][WPX + j]); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const unsigned oplane = (unsigned)(p.out_h * p.out_w) * 4u;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + (size_t)b * p.N * p.out_h * p.out_w, (unsigned)p.N * oplane);
    const EpilogueConsts ec = epilogue_consts(p);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.residual ? p.residual + (size_t)b * p.N * p.out_h * p.out_w : p.y, p.residual ? (unsigned)p.N * oplane : 0u);
    // one finished tile: two phases without control flow (values in place, then bare stores), every epilogue load before the first store
    auto finish_tile = [&](int tile) {
        const int qy0 = (tile / p.tiles_x) * TR + 2 * rp, qx = (tile % p.tiles_x) * 32 + l31;
        const int nb = opaque_s(n0 + och * 64);
        unsigned voff[WPX];
        float nz[WPX];
#pragma unroll
        for (int j = 0; j < WPX; ++j) {
            const int qy = qy0 + j;
            const bool inside = qy < p.out_h && qx < p.out_w;
            voff[j] = inside ? (unsigned)(qy * p.out_w + qx) * 4u + (unsigned)(4 * hi) * oplane : OOB;
            nz[j] = (p.noise && inside) ? p.noise[((size_t)b * p.out_h + qy) * p.out_w + qx] : 0.f;
        }
        auto phase1 = [&](auto with_res) {
#pragma unroll
            for (int i = 0; i < WOC; ++i) {
                float so16[16], bi16[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 s4 = *reinterpret_cast<const float4*>(&s_so[och * 64 + i * 32 + 8 * q + 4 * hi]);
                    so16[4 * q] = s4.x; so16[4 * q + 1] = s4.y; so16[4 * q + 2] = s4.z; so16[4 * q + 3] = s4.w;
                    if (EPK == 0) {
                        const float4 b4 = *reinterpret_cast<const float4*>(&s_bias[och * 64 + i * 32 + 8 * q + 4 * hi]);
                        bi16[4 * q] = b4.x; bi16[4 * q + 1] = b4.y; bi16[4 * q + 2] = b4.z; bi16[4 * q + 3] = b4.w;
                    }
                }
#pragma unroll
                for (int j = 0; j < WPX; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = EPK == 0 ? conv_epilogue(ec, acc[i][j][r], so16[r], bi16[r], nz[j]) : plain_mul(acc[i][j][r], so16[r]);
                        if (decltype(with_res)::value) v = plain_sum(v, buf_load_f32(rres, voff[j], (unsigned)(nb + i * 32 + (r & 3) + 8 * (r >> 2)) * oplane));
                        acc[i][j][r] = v;
                    }
            }
        };
        if (EPK < 2) phase1(std::integral_constant<bool, RES>{});
#pragma unroll
        for (int j = 0; j < WPX; ++j)
#pragma unroll
            for (int i = 0; i < WOC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ocs = nb + i * 32 + (r & 3) + 8 * (r >> 2);
                    const float v = acc[i][j][r];
                    if (!(GC_S2WS_ABL & 8) || v == 12345.678f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, (int)voff[j], (int)((unsigned)ocs * oplane), GC_CONV_ST_AUX);
                    acc[i][j][r] = 0.f;
                }
    };

    using PE = std::integral_constant<int, 0>;
    using PO = std::integral_constant<int, 1>;
    constexpr bool DMA = !(GC_S2WS_ABL & 2);
    // The slab of sub-item s + 2 is requested right after the barrier that ends sub-item s (its stage is free from then on) and BEFORE the
    // stores of a finished tile: vmcnt counts loads and stores in issue order, so the wait in front of the next barrier is `vmcnt(63)` -- the 64
    // stores of a lane's tile may stay in flight, everything older has landed -- and the barrier is the raw instruction.
    constexpr int NSTORES = 63;
    static_assert(WOC * WPX * 16 >= NSTORES + 1, "finish_tile issues at least NSTORES + 1 stores per lane after the newest weight request");
    if (DMA) { weights(PE{}, 0); wait_staged_loads(); }
    __syncthreads();                                   // E(0) is staged (patch by the staging waves, weights here)
    if (DMA) weights(PO{}, 0);
    int tile_c = tile_begin, k0_c = 0;
    bool stored = false;
    for (int it = 0; it < items; ++it) {
        const int k_next = k0_c + KCB < p.K ? k0_c + KCB : 0;                // the chunk of item it + 1 (the next tile starts at channel 0 again)
        // ---- E sub-item ----
        __builtin_amdgcn_s_setprio(GC_MFMA_PRIO);
        multiply(PE{});
        __builtin_amdgcn_s_setprio(0);
        if (stored && !(GC_S2WS_ABL & 8)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSTORES) : "memory");
        else        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): this wave's fragment reads have returned
        __builtin_amdgcn_s_barrier();
        if (DMA && it + 1 < items) weights(PE{}, k_next);                    // E(it + 1) into the stage E(it) has just left
        stored = false;
        // ---- O sub-item ----
        __builtin_amdgcn_s_setprio(GC_MFMA_PRIO);
        multiply(PO{});
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        if (DMA && it + 1 < items) weights(PO{}, k_next);
        k0_c += KCB;
        if (k0_c >= p.K) { finish_tile(tile_c); k0_c = 0; tile_c += tstep; stored = true; }
    }
    // every LDS-DMA request has been waited for (none is issued for sub-items past the last one)
}

template <int WOCB>
int launch_s2ws_t(Bf16Args a, hipStream_t s) {
    using C = S2Cfg<WOCB>;
    a.c.tiles_y = gc::ceil_div(a.c.out_h, C::TR);
    a.c.tiles_x = gc::ceil_div(a.c.out_w, 32);
    const int tiles = a.c.tiles_x * a.c.tiles_y, ocb = a.c.N / C::OCT;
    const long long wgs = (long long)tiles * a.c.B * ocb;
    // one workgroup per CU is resident: a workgroup takes all the tiles its CU would get over the rounds of the launch
    a.tpb = (int)std::min<long long>(std::max<long long>((wgs + GC_WS_SLOTS - 1) / GC_WS_SLOTS, 1), tiles);
    a.groups = gc::ceil_div(tiles, a.tpb);
    const long long gx = (long long)a.groups * a.c.B;
    if (gx > 2147483647LL) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_bf16x3_f32: grid too large");
    if (gc::probing()) return gc::probe_name("conv_s2ws_bf16x3_kernel<%d>|up1,down2,k3", WOCB);
    const bool plain = !a.c.bias && !a.c.noise && !a.c.act;
    const int epk = !plain ? 0 : ((a.c.so || a.c.residual) ? 1 : 2);
    const bool res = a.c.residual != nullptr;
    const dim3 grid((unsigned)gx, ocb);
    if (epk == 2)             hipLaunchKernelGGL((conv_s2ws_bf16x3_kernel<WOCB, 2, false>), grid, dim3(768), 0, s, a);
    else if (epk == 1 && res) hipLaunchKernelGGL((conv_s2ws_bf16x3_kernel<WOCB, 1, true>), grid, dim3(768), 0, s, a);
    else if (epk == 1)        hipLaunchKernelGGL((conv_s2ws_bf16x3_kernel<WOCB, 1, false>), grid, dim3(768), 0, s, a);
    else if (res)             hipLaunchKernelGGL((conv_s2ws_bf16x3_kernel<WOCB, 0, true>), grid, dim3(768), 0, s, a);
    else                      hipLaunchKernelGGL((conv_s2ws_bf16x3_kernel<WOCB, 0, false>), grid, dim3(768), 0, s, a);
    return gc::check_launch("gc_conv2d_bf16x3_f32(stride 2, ws)");
}

}  // namespace

namespace gcconv {

#ifndef GC_S2WS
#define GC_S2WS 1             // 0: stride-2 3x3 convolutions stay on conv_bf16x3_kernel
#endif

// whole 16-channel chunks, whole 64-channel output blocks, no padding, enough tiles to give most CUs a workgroup
bool s2ws_eligible(const Bf16Args& a) {
    if (!GC_S2WS) return false;
    const ConvArgs& c = a.c;
    if (a.k_per_split || c.K % KCB != 0 || c.K < 32 || c.K > MAX_K_BF16X3 || c.N % 64 != 0 || c.pad_x != 0 || c.pad_y != 0) return false;
    if (c.out_w < 32 || c.out_h < 8) return false;
    const int oct = (c.N % 128 == 0 && c.K > GC_S2WS_WOCB1_MAX_K) ? 128 : 64;
    const long long wgs = (long long)gc::ceil_div(c.out_w, 32) * gc::ceil_div(c.out_h, 8) * c.B * (c.N / oct);
    return wgs >= GC_S2WS_MIN_WGS;
}

int launch_s2ws(Bf16Args a, hipStream_t s) {
    return (a.c.N % 128 == 0 && a.c.K > GC_S2WS_WOCB1_MAX_K) ? launch_s2ws_t<2>(a, s) : launch_s2ws_t<1>(a, s);
}

}  // namespace gcconv
