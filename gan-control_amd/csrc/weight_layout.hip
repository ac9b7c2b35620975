// Weight re-layout for K3/K4: one pass that scales, permutes and (optionally) mirrors the taps of a convolution weight.
// See gc_weight_layout_f32 in include/gancontrol_hip.h.
//
// The reference prepares weights with separate ATen passes on every call (gan_model.py:154 `weight * scale`,
// :284-306 scale, modulate, view, transpose; convolution_backward permutes again).  The kernels here want
// [taps, K, N] (N contiguous), the parameters are stored [N, K, taps] (or [K, N, taps] / [1, N, K, taps]), the input
// gradient needs [mirrored taps, N, K].  All of these are the same operation:
//
//     dst[t' * dt + k * dk + n * dn] = scale * src[t * st + k * sk + n * sn],   t' = flip ? T - 1 - t : t
//
// A workgroup moves a 32(k) x 32(n) x (<= 9 taps) tile through LDS: it is read in the memory order of `src` and written
// in the memory order of `dst`, so both sides are coalesced.  The layout pairs the model uses are compiled with constant
// index arithmetic (weight_layout_kernel<read order, write order, taps>); anything else takes a plain gather kernel.
#include <algorithm>
#include <vector>

#include "common.h"

namespace {

#ifndef GC_LAYOUT_TT3
#define GC_LAYOUT_TT3 3
#endif
#ifndef GC_LAYOUT_PER_TAP
#define GC_LAYOUT_PER_TAP 1
#endif

constexpr int TILE = 32;

struct LayoutArgs {
    const float* src; float* dst;
    int T, K, N;
    long long st, sk, sn, dt, dk, dn;
    int flip; float scale;
};

// Memory order of a layout, slowest -> fastest axis (t = tap, k, n).  The four orders the model needs:
//   NKT  parameter [N,K,taps]           KNT  conv_transpose2d parameter [K,N,taps]
//   TKN  kernel layout [taps,K,N]       TNK  input-gradient layout [taps,N,K]
enum Order { NKT = 0, KNT = 1, TKN = 2, TNK = 3 };

// Coordinates (t, k, n) of the e-th element of a full 32 x 32 x TT tile walked in memory order ORD (constant divisors only).
template <int ORD, int TT>
__device__ __forceinline__ void decode(int e, int& t, int& k, int& n) {
    if (ORD == NKT) { t = e % TT; k = (e / TT) % TILE; n = e / (TT * TILE); }
    if (ORD == KNT) { t = e % TT; n = (e / TT) % TILE; k = e / (TT * TILE); }
    if (ORD == TKN) { n = e % TILE; k = (e / TILE) % TILE; t = e / (TILE * TILE); }
    if (ORD == TNK) { k = e % TILE; n = (e / TILE) % TILE; t = e / (TILE * TILE); }
}

// The tile sits in LDS in the READ order, with odd strides for the middle and slow axes so that the write phase -- whose
// lanes run along a different axis -- is bank-conflict free as well.
template <int ORD, int TT>
struct TileLds {
    static constexpr int FAST = (ORD == NKT || ORD == KNT) ? TT : TILE;
    static constexpr int MID = TILE;
    static constexpr int SLOW = (ORD == NKT || ORD == KNT) ? TILE : TT;
    static constexpr int MIDS = FAST | 1;
    static constexpr int SLOWS = (MID * MIDS) | 1;
    static constexpr int SIZE = SLOW * SLOWS;
    __device__ static __forceinline__ int at(int t, int k, int n) {
        if (ORD == NKT) return n * SLOWS + k * MIDS + t;
        if (ORD == KNT) return k * SLOWS + n * MIDS + t;
        if (ORD == TKN) return t * SLOWS + k * MIDS + n;
        return t * SLOWS + n * MIDS + k;
    }
};

template <int RORD, int WORD, int TT>
__device__ __forceinline__ void layout_tile(const LayoutArgs& a, int bx, int by, int bz, float* tile) {
    using L = TileLds<RORD, TT>;
    const int k0 = bx * TILE, n0 = by * TILE, t0 = bz * TT;
    const int ke = min(TILE, a.K - k0), ne = min(TILE, a.N - n0), te = min(TT, a.T - t0);
    constexpr int TOTAL = TT * TILE * TILE, PER = TOTAL / 256;
    // all loads of the lane first (36 in flight for 3x3 taps): a weight tensor is at most a few hundred tiles, so the
    // launch has no other parallelism to hide the memory latency with
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        int t, k, n;
        decode<RORD, TT>(threadIdx.x + 256 * i, t, k, n);
        v[i] = (t < te && k < ke && n < ne) ? a.src[(t0 + t) * a.st + (k0 + k) * a.sk + (n0 + n) * a.sn] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        int t, k, n;
        decode<RORD, TT>(threadIdx.x + 256 * i, t, k, n);
        tile[L::at(t, k, n)] = v[i];
    }
    __syncthreads();
#pragma unroll 4
    for (int e = threadIdx.x; e < TOTAL; e += 256) {
        int t, k, n;
        decode<WORD, TT>(e, t, k, n);
        if (t < te && k < ke && n < ne) {
            const int td = a.flip ? a.T - 1 - (t0 + t) : t0 + t;
            a.dst[td * a.dt + (k0 + k) * a.dk + (n0 + n) * a.dn] = a.scale * tile[L::at(t, k, n)];
        }
    }
}

template <int RORD, int WORD, int TT>
__global__ __launch_bounds__(256) void weight_layout_kernel(LayoutArgs a) {
    __shared__ float tile[TileLds<RORD, TT>::SIZE];
    layout_tile<RORD, WORD, TT>(a, blockIdx.x, blockIdx.y, blockIdx.z, tile);
}

// The same pass over MANY tensors in one launch (gc_weight_layout_grouped_f32): after every optimiser step all ~25 convolution weights of a
// network change together and each is needed as kernel layout and as input-gradient layout -- ~50 launches of 4 - 6 us whose work is
// a few hundred tiles each.  One tap per block (the variants the two forward-side re-layouts use); a block finds its tensor in a table.
constexpr int MAXLG = 32;
struct LayoutGroupArgs {
    LayoutArgs g[MAXLG];
    int first[MAXLG + 1];        // prefix sum of blocks
    int n_groups;
};

template <int RORD, int WORD>
__global__ __launch_bounds__(256) void weight_layout_grouped_kernel(LayoutGroupArgs a) {
    __shared__ float tile[TileLds<RORD, 1>::SIZE];
    int gi = 0;
#pragma unroll 1
    for (int i = 1; i < a.n_groups; ++i) gi = ((int)blockIdx.x >= a.first[i]) ? i : gi;
    const LayoutArgs& L = a.g[gi];
    int r = blockIdx.x - a.first[gi];
    const int kb = (L.K + TILE - 1) / TILE, nb = (L.N + TILE - 1) / TILE;
    const int bx = r % kb; r /= kb;
    const int by = r % nb;
    const int bz = r / nb;
    layout_tile<RORD, WORD, 1>(L, bx, by, bz, tile);
}

// any other pair of layouts: one element per lane, gathered reads (correct for every stride triple, not fast)
__global__ __launch_bounds__(256) void weight_layout_generic_kernel(LayoutArgs a) {
    const long long total = (long long)a.T * a.K * a.N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int n = (int)(e % a.N);
        const long long r = e / a.N;
        const int k = (int)(r % a.K), t = (int)(r / a.K);
        const int td = a.flip ? a.T - 1 - t : t;
        a.dst[td * a.dt + k * a.dk + n * a.dn] = a.scale * a.src[t * a.st + k * a.sk + n * a.sn];
    }
}

// which of the four orders a stride triple is (for extents > 1); -1 if none
int order_of(const int64_t s[3], int T, int K, int N) {
    const int64_t t = s[0], k = s[1], n = s[2];
    if (t == 1 && k == T && n == (int64_t)K * T) return NKT;
    if (t == 1 && n == T && k == (int64_t)N * T) return KNT;
    if (n == 1 && k == N && t == (int64_t)K * N) return TKN;
    if (k == 1 && n == K && t == (int64_t)N * K) return TNK;
    return -1;
}

template <int RORD, int WORD>
void launch_tt(const LayoutArgs& a, hipStream_t s) {
    // both sides tap-major (kernel layout -> input-gradient weights): every tap is a [K, N] transpose of its own, so one tap per block
    // gives nine times the blocks (a 512 x 512 x 9 tensor is only 256 tiles: the pass is latency-bound) at unchanged coalescing
    constexpr bool per_tap = (RORD == TKN || RORD == TNK) && (WORD == TKN || WORD == TNK);
    if (a.T == 1 || (per_tap && GC_LAYOUT_PER_TAP)) {
        dim3 grid(gc::ceil_div(a.K, TILE), gc::ceil_div(a.N, TILE), a.T);
        hipLaunchKernelGGL((weight_layout_kernel<RORD, WORD, 1>), grid, dim3(256), 0, s, a);
    } else if (GC_LAYOUT_TT3 && (RORD == NKT || RORD == KNT) && (WORD == TKN || WORD == TNK)) {
        // tap-minor on the READ side only (parameter -> kernel layout): one tap per block as well -- the 4-byte reads of the nine blocks
        // of a tile share their lines in L2 (3.9 us against 5.4 with three taps and 9.5 with nine, rocprofv3, any size up to 512 x 512 x 9)
        dim3 grid(gc::ceil_div(a.K, TILE), gc::ceil_div(a.N, TILE), a.T);
        hipLaunchKernelGGL((weight_layout_kernel<RORD, WORD, 1>), grid, dim3(256), 0, s, a);
    } else if (GC_LAYOUT_TT3 && a.T % 3 == 0) {      // tap-minor on the WRITE side (weight gradient -> parameter layout): three taps per block (12-byte runs: 5.7 against 9.8 us)
        dim3 grid(gc::ceil_div(a.K, TILE), gc::ceil_div(a.N, TILE), a.T / 3);
        hipLaunchKernelGGL((weight_layout_kernel<RORD, WORD, 3>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid(gc::ceil_div(a.K, TILE), gc::ceil_div(a.N, TILE), gc::ceil_div(a.T, 9));
        hipLaunchKernelGGL((weight_layout_kernel<RORD, WORD, 9>), grid, dim3(256), 0, s, a);
    }
}

template <int RORD, int WORD>
void launch_grouped(const LayoutArgs* list, int count, hipStream_t s) {
    for (int first = 0; first < count; first += MAXLG) {
        LayoutGroupArgs a;
        a.n_groups = std::min(MAXLG, count - first);
        int blocks = 0;
        for (int i = 0; i < a.n_groups; ++i) {
            a.g[i] = list[first + i];
            a.first[i] = blocks;
            blocks += gc::ceil_div(a.g[i].K, TILE) * gc::ceil_div(a.g[i].N, TILE) * a.g[i].T;
        }
        a.first[a.n_groups] = blocks;
        hipLaunchKernelGGL((weight_layout_grouped_kernel<RORD, WORD>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    }
}

int fill_args(LayoutArgs& a, const float* src, float* dst, int taps, int k, int n, const int64_t src_stride[3], const int64_t dst_stride[3],
              int flip_taps, float scale, const char* what) {
    if (!src || !dst || !src_stride || !dst_stride) return gc::fail(GC_ERR_BAD_ARG, "%s: null pointer", what);
    if (taps <= 0 || k <= 0 || n <= 0) return gc::fail(GC_ERR_BAD_ARG, "%s: non-positive extent", what);
    for (int i = 0; i < 3; ++i)
        if (src_stride[i] < 0 || dst_stride[i] < 0) return gc::fail(GC_ERR_BAD_ARG, "%s: negative stride", what);
    if (gc::ceil_div(n, TILE) > 65535 || gc::ceil_div(taps, 9) > 65535) return gc::fail(GC_ERR_UNSUPPORTED, "%s: extent too large", what);
    a.src = src; a.dst = dst; a.T = taps; a.K = k; a.N = n;
    a.st = src_stride[0]; a.sk = src_stride[1]; a.sn = src_stride[2];
    a.dt = dst_stride[0]; a.dk = dst_stride[1]; a.dn = dst_stride[2];
    a.flip = flip_taps ? 1 : 0; a.scale = scale;
    return GC_OK;
}

}  // namespace

extern "C" int gc_weight_layout_grouped_f32(const gc_wlayout_group* groups, int n_groups, gc_stream_t stream) {
    if (n_groups < 0 || (n_groups > 0 && !groups)) return gc::fail(GC_ERR_BAD_ARG, "gc_weight_layout_grouped_f32: bad group table");
    hipStream_t s = (hipStream_t)stream;
    // the two forward-side re-layouts (parameter -> kernel layout, kernel layout -> input-gradient weights) are batched; anything else
    // goes through the single-tensor entry point, group by group (same results either way: the tile routine is shared)
    std::vector<LayoutArgs> fwd, fwd_t, adj;
    for (int i = 0; i < n_groups; ++i) {
        const gc_wlayout_group& g = groups[i];
        LayoutArgs a;
        int rc = fill_args(a, g.src, g.dst, g.taps, g.k, g.n, g.src_stride, g.dst_stride, g.flip_taps, g.scale, "gc_weight_layout_grouped_f32");
        if (rc) return rc;
        const int ro = order_of(g.src_stride, g.taps, g.k, g.n), wo = order_of(g.dst_stride, g.taps, g.k, g.n);
        if (ro == NKT && wo == TKN) fwd.push_back(a);
        else if (ro == KNT && wo == TKN) fwd_t.push_back(a);
        else if (ro == TKN && wo == TNK) adj.push_back(a);
        else if ((rc = gc_weight_layout_f32(g.src, g.dst, g.taps, g.k, g.n, g.src_stride, g.dst_stride, g.flip_taps, g.scale, stream))) return rc;
    }
    if (!fwd.empty()) launch_grouped<NKT, TKN>(fwd.data(), (int)fwd.size(), s);
    if (!fwd_t.empty()) launch_grouped<KNT, TKN>(fwd_t.data(), (int)fwd_t.size(), s);
    if (!adj.empty()) launch_grouped<TKN, TNK>(adj.data(), (int)adj.size(), s);
    return gc::check_launch("gc_weight_layout_grouped_f32");
}

extern "C" int gc_weight_layout_f32(const float* src, float* dst, int taps, int k, int n,
                                    const int64_t src_stride[3], const int64_t dst_stride[3],
                                    int flip_taps, float scale, gc_stream_t stream) {
    LayoutArgs a;
    int rc = fill_args(a, src, dst, taps, k, n, src_stride, dst_stride, flip_taps, scale, "gc_weight_layout_f32");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int ro = order_of(src_stride, taps, k, n), wo = order_of(dst_stride, taps, k, n);
    if (ro == NKT && wo == TKN)      launch_tt<NKT, TKN>(a, s);     // parameter -> kernel layout
    else if (ro == KNT && wo == TKN) launch_tt<KNT, TKN>(a, s);     // conv_transpose2d parameter -> kernel layout
    else if (ro == TKN && wo == TNK) launch_tt<TKN, TNK>(a, s);     // kernel layout -> input-gradient weights
    else if (ro == TKN && wo == NKT) launch_tt<TKN, NKT>(a, s);     // weight gradient -> parameter layout
    else if (ro == TKN && wo == KNT) launch_tt<TKN, KNT>(a, s);
    else {
        const long long total = (long long)taps * k * n;
        hipLaunchKernelGGL(weight_layout_generic_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, s, a);
    }
    return gc::check_launch("gc_weight_layout_f32");
}
