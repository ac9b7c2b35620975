// NOT SHIPPED (moved out of conv_bf16x3.hip in round 6 so that the shipped path can be read): the first, one-row form of the wave-specialised weight
// gradient (GC_WG_WS = 1).  Measured at parity with the one-role kernel in round 5; what it taught is in the GC_WG_WS comment of conv_bf16x3.hip and in
// DESIGN.md section 8.  Included from conv_bf16x3.hip inside its anonymous namespace, only when GC_WG_WS == 1.

struct WgWsCfg {
    static constexpr int XR = 6, YR = 2, XU = 5, YU = 4;
    static constexpr int CSX = (XR * XU) | 1, CSY = (YR * YU) | 1;          // odd unit strides between channels: conflict-free b128 reads
    static constexpr int XUNITS = 64 * CSX, YUNITS = 64 * CSY;
    static constexpr int SMEM_UNITS = 2 * (XUNITS + YUNITS);
    static constexpr int ROW_X = 64 * XU, ROW_Y = 64 * YU;                   // units of one staged X / dY row
};

__global__ __launch_bounds__(1024) void wgrad_bf16x3_ws_kernel(WgArgs p, int rb, int bands) {
    using C = WgWsCfg;
    constexpr int XR = C::XR, YR = C::YR, XU = C::XU, YU = C::YU, CSX = C::CSX, CSY = C::CSY;
    constexpr int NXJ = (3 * C::ROW_X + 255) / 256;          // X units per staging lane and item (three rows at the top of a strip): 4, the last partly idle
    constexpr int NXJ1 = (C::ROW_X + 255) / 256;              // ... inside a strip (one row): 2
    static_assert(C::ROW_Y == 256, "one dY unit per staging lane and item");
    __shared__ uint4 smem[C::SMEM_UNITS];
    uint4* xh = smem;
    uint4* xl = xh + C::XUNITS;
    uint4* yh = xl + C::XUNITS;
    uint4* yl = yh + C::YUNITS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 64, split = blockIdx.z;

    // the strips of this split: `sstep` apart (neighbouring splits work on neighbouring column strips of one row band: contiguous rows in DRAM)
    const int strips_per_sample = p.tiles_x * bands;
    const int sb = p.spb ? split / p.spb : 0;        // per-sample mode (gc_conv2d_wgrad_samples_*): the splits of one sample walk that sample's strips only
    const int sstep = p.spb ? p.spb : (int)gridDim.z;
    const int s_begin = p.spb ? sb * strips_per_sample + (split - sb * p.spb) : split;
    const int s_end = p.spb ? (sb + 1) * strips_per_sample : strips_per_sample * p.B;
    const int nstrips = s_begin < s_end ? (s_end - s_begin + sstep - 1) / sstep : 0;
    const int items = nstrips * rb;
    const int xchan = p.in_h * p.in_w, ychan = p.out_h * p.out_w;

    if (wave >= 12) {
        // ---------------- staging waves ----------------
        if (GC_WGWS_STAGER_PRIO) __builtin_amdgcn_s_setprio(GC_WGWS_STAGER_PRIO);
        const int st = tid - 768;
        constexpr unsigned OUTSIDE = 0x80000000u;
        const unsigned xbytes = (unsigned)p.K * xchan * 4u, ybytes = (unsigned)p.N * ychan * 4u;
        // Which units a lane stages is fixed: X unit u = st + 256 j (j < NXJ) = (row q of the up-to-three new rows, channel, unit column), and ONE dY
        // unit (channel, unit column).  Everything per unit is recomputed from the lane index where it is used (a few integer instructions): nothing
        // but the loaded data lives in registers between an item's loads and its conversion.
        const int ych = st >> 2, yu = st & 3;
        // (keeping each slot's channel / unit column / offsets in registers instead of recomputing them from the lane index -- two integer divisions
        // per unit -- was tried: 128 registers, 4-10 spilled, a scratch reload in front of the staging waves' LDS writes)
        auto statics = [&](int u, int& meta, int& lo, int& go) {
            const int q = u / C::ROW_X, rem = u - q * C::ROW_X, ch = rem / XU, xu = rem - ch * XU;
            meta = q | xu << 4 | ch << 8;
            lo = ch * CSX + xu;
            go = ((k0 + ch) * xchan) * 4 + xu * 32;
        };
#define GC_WGWS_SLOT(j, meta, lo, go) int meta, lo, go; statics(opaque(st) + 256 * (j), meta, lo, go)
        const int ygo = ((n0 + ych) * ychan) * 4 + yu * 32, ylo = ych * CSY + yu;
        // an item's position: strip (sample b, first column ox0, first row oy0) and row r of the strip; advanced one item at a time
        struct Cur { int sidx, b, oy0, ox0, r, ord; };
        auto place = [&](Cur& c) {
            c.b = c.sidx / strips_per_sample;
            const int rem = c.sidx - c.b * strips_per_sample;
            c.oy0 = (rem / p.tiles_x) * rb;
            c.ox0 = (rem % p.tiles_x) * 32;
        };
        auto advance = [&](Cur& c) {
            if (++c.r == rb) { c.r = 0; c.sidx += sstep; ++c.ord; place(c); }
        };
        auto loads = [&](float4 (&xv)[NXJ][2], float (&xs)[NXJ], float4 (&yv)[2], float& ys, const Cur& c, bool live_item) {
            const int b = min(c.b, p.B - 1);
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)b * p.K * xchan, xbytes);
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.dy + (size_t)b * p.N * ychan, ybytes);
            const int oy = c.oy0 + c.r;
            const int nx = c.r == 0 ? 3 : 1;                     // new X rows of this item: all three at the top of a strip, else the bottom one
            const int nj = nx == 3 ? NXJ : NXJ1;                 // unit slots in use (item-uniform): inside a strip one row = 320 units = 1.25 per lane
#pragma unroll
            for (int j = 0; j < NXJ; ++j) {
                if (j >= nj) { xv[j][0] = xv[j][1] = make_float4(0.f, 0.f, 0.f, 0.f); xs[j] = 1.f; continue; }
                GC_WGWS_SLOT(j, meta, lo_, go);
                const int q = meta & 15;
                const int iy = oy - p.pad_y + (nx == 3 ? q : 2);
                const int lin = go + (iy * p.in_w + c.ox0 - p.pad_x) * 4;
                // (the unit at channel 0, row 0, column -pad of a sample would start at a negative offset, which the range check rejects as a whole:
                // it is loaded from offset 0 and shifted by one pixel in convert())
                const bool ok = live_item && q < nx && (unsigned)iy < (unsigned)p.in_h;
                const unsigned off = ok ? (unsigned)max(lin, 0) : OUTSIDE;
                xv[j][0] = __builtin_bit_cast(float4, buf_load_u128(rx, off, 0));
                xv[j][1] = __builtin_bit_cast(float4, buf_load_u128(rx, off, 16));
                xs[j] = p.si ? p.si[(size_t)b * p.K + k0 + min(meta >> 8, 63)] : 1.f;
            }
            const unsigned yoff = live_item ? (unsigned)(ygo + (oy * p.out_w + c.ox0) * 4) : OUTSIDE;
            yv[0] = __builtin_bit_cast(float4, buf_load_u128(ry, yoff, 0));
            yv[1] = __builtin_bit_cast(float4, buf_load_u128(ry, yoff, 16));
            ys = p.so ? p.so[(size_t)b * p.N + n0 + ych] : 1.f;
        };
        auto convert = [&](const float4 (&xv)[NXJ][2], const float (&xs)[NXJ], const float4 (&yv)[2], float ys, const Cur& c, int yslot) {
            const int oy = c.oy0 + c.r;
            const int nx = c.r == 0 ? 3 : 1;
            const int xseq = c.ord * (rb + 2) + c.r;                     // sequence number of this item's X row of tap row 0; rows ty = 1, 2 follow
            const bool scaled = p.si != nullptr || p.so != nullptr;
            const bool edge = c.ox0 - p.pad_x < 0 || c.ox0 - p.pad_x + 8 * XU > p.in_w || c.ox0 + 8 * YU > p.out_w;      // strip-uniform
            const int nj = nx == 3 ? NXJ : NXJ1;
            auto body = [&](auto scaled_t, auto edge_t) {
                constexpr bool SC = decltype(scaled_t)::value, EDGE = decltype(edge_t)::value;
#pragma unroll
                for (int j = 0; j < NXJ; ++j) {
                    if (j >= nj) continue;                   // item-uniform: a scalar branch
                    GC_WGWS_SLOT(j, meta, lo_, go);
                    const int q = meta & 15;
                    float v[8] = {xv[j][0].x, xv[j][0].y, xv[j][0].z, xv[j][0].w, xv[j][1].x, xv[j][1].y, xv[j][1].z, xv[j][1].w};
                    if (EDGE) {
                        const int col0 = c.ox0 - p.pad_x + 8 * ((meta >> 4) & 15);
                        if (col0 < 0 && k0 + (meta >> 8) == 0 && oy - p.pad_y + (nx == 3 ? q : 2) == 0) {
                            // the unit fetched from offset 0 instead of -pad (see loads): what was loaded is columns 0..7, wanted is -1..6
#pragma unroll
                            for (int e = 7; e > 0; --e) v[e] = v[e - 1];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (col0 + e >= 0 && col0 + e < p.in_w) ? v[e] : 0.f;
                    }
                    uint4 h, l;
                    split8<SC>(v, xs[j], &h, &l);
                    if (q < nx) {
                        const int slot = (xseq + (nx == 3 ? q : 2)) % XR;
                        const int o = lo_ + slot * XU;
                        xh[o] = h; GC_LO(xl[o] = l;)
                    }
                }
                {
                    float v[8] = {yv[0].x, yv[0].y, yv[0].z, yv[0].w, yv[1].x, yv[1].y, yv[1].z, yv[1].w};
                    if (EDGE) {
                        const int col0 = c.ox0 + 8 * yu;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = col0 + e < p.out_w ? v[e] : 0.f;
                    }
                    uint4 h, l;
                    split8<SC>(v, ys, &h, &l);
                    const int o = ylo + yslot * YU;
                    yh[o] = h; GC_LO(yl[o] = l;)
                }
            };
            if (scaled) { if (edge) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
            else        { if (edge) body(std::false_type{}, std::true_type{}); else body(std::false_type{}, std::false_type{}); }
        };
        // interval t: the multiplying waves work on item t; item t + 1 is converted here (its loads were issued one interval ago), item t + 2 is fetched
        float4 xva[NXJ][2], xvb[NXJ][2], yva[2], yvb[2];
        float xsa[NXJ], xsb[NXJ], ysa, ysb;
        Cur cl{s_begin, 0, 0, 0, 0, 0};                 // cursor of the loads
        place(cl);
        Cur cc = cl;                                    // cursor of the conversions
        if (GC_WGWS_ABL & 1) {
            __syncthreads();
            for (int t = 0; t < items; ++t) __syncthreads();
            return;
        }
        loads(xva, xsa, yva, ysa, cl, 0 < items); advance(cl);
        loads(xvb, xsb, yvb, ysb, cl, 1 < items); advance(cl);
        if (items > 0) convert(xva, xsa, yva, ysa, cc, 0);
        advance(cc);
        __syncthreads();
        for (int t = 0; t < items; t += 2) {
            loads(xva, xsa, yva, ysa, cl, t + 2 < items); advance(cl);
            if (t + 1 < items) convert(xvb, xsb, yvb, ysb, cc, 1);
            advance(cc);
            __syncthreads();
            if (t + 1 >= items) break;
            loads(xvb, xsb, yvb, ysb, cl, t + 3 < items); advance(cl);
            if (t + 2 < items) convert(xva, xsa, yva, ysa, cc, 0);
            advance(cc);
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
    int r = 0, xslot = ty % XR, yslot = 0;           // row of the strip, ring slot of this wave's X row (tap row ty), slot of the dY row
    __syncthreads();                 // item 0 is staged
    for (int it = 0; it < items; ++it) {
        __builtin_amdgcn_s_setprio(GC_MFMA_PRIO);
        // both half-rows' fragments are read before the first MFMA (twelve ds_read_b128, 18 MFMAs)
        uint4 fbh0, fbh1, fbl0, fbl1, a0h0, a0h1, a1h0, a1h1, a0l0, a0l1, a1l0, a1l1;
        if (!(GC_WGWS_ABL & 4)) {
            const int yo = yb_ + yslot * YU, o = xa + xslot * XU;
            fbh0 = yh[yo]; a0h0 = xh[o]; a1h0 = xh[o + 1];
            GC_LO(fbl0 = yl[yo]; a0l0 = xl[o]; a1l0 = xl[o + 1];)
            __builtin_amdgcn_sched_barrier(0);
            fbh1 = yh[yo + 2]; a0h1 = xh[o + 2]; a1h1 = xh[o + 3];
            GC_LO(fbl1 = yl[yo + 2]; a0l1 = xl[o + 2]; a1l1 = xl[o + 3];)
            __builtin_amdgcn_sched_barrier(0);
        }
        auto half = [&](const uint4& fbh, const uint4& fbl, const uint4& a0h, const uint4& a1h, const uint4& a0l, const uint4& a1l) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&fbh);
#ifndef GC_SINGLE
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&fbl);
#endif
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const uint4 uh = shift_px(a0h, a1h, tx);
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&uh);
#ifndef GC_SINGLE
                const uint4 ul = shift_px(a0l, a1l, tx);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(&ul);
#endif
                GC_MFMA3(acc[tx], ah, al, bh, bl);
            }
        };
        if (!(GC_WGWS_ABL & 2)) {
        half(fbh0, fbl0, a0h0, a1h0, a0l0, a1l0);
        half(fbh1, fbl1, a0h1, a1h1, a0l1, a1l1);
        } else if (!(GC_WGWS_ABL & 4)) {
            acc[0][0] += __builtin_bit_cast(float, fbh0.x ^ fbh1.x ^ a0h0.x ^ a1h0.x ^ a0h1.x ^ a1h1.x GC_LO(^ fbl0.x ^ fbl1.x ^ a0l0.x ^ a1l0.x ^ a0l1.x ^ a1l1.x));
        }
        __builtin_amdgcn_s_setprio(0);
        // next item: one ring slot on inside a strip, three at a strip boundary (the new strip brings three new rows)
        if (++r == rb) { r = 0; xslot += 3; } else { xslot += 1; }
        if (xslot >= XR) xslot -= XR;
        yslot ^= 1;
        __syncthreads();             // the slots of this item may be rewritten from the next interval on; the next item is staged
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

