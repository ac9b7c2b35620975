// NOT SHIPPED (moved out of conv_bf16x3.hip in round 6): the wave-specialised form of the transposed 3x3 convolution, built with -DGC_CTWS=1 (whole
// (H + 1) x (W + 1) q-space) or -DGC_CTWS=2 (the H x W main region next to convt_edge_bf16x3_kernel: +3 % over the one-role kernel, round 6,
// profiles/ctws_main_edge_r06_c.log).  The design notes are the comment in front of its include in conv_bf16x3.hip.  Included inside the anonymous namespace.

template <int WOCB, int TPW>
struct TWCfg {
    static constexpr int OCT = 32 * WOCB, RPB = 32 / TPW, RG = 8 / WOCB;      // RG row groups; a multiplying wave owns 32 oc x 2 blocks of RPB rows x TPW q-columns
    static constexpr int TQH = RG * 2 * RPB;
    static constexpr int PH = TQH + 1, PWD = TPW + 1, PLANE = PH * PWD;
    static constexpr int WHALF = 6 * KG * OCT;          // units of one half (hi / lo) of a weight slot: [tap slot 0..5][kg][OCT]
    static constexpr int WSLOT = 2 * WHALF;
    static constexpr int PSTAGE = 2 * KG * PLANE;       // one patch stage: [hi | lo][kg][PH][PWD]
    static constexpr int GR = TPW / 4, TASKS = PH * (GR + 1), NT = (TASKS + 127) / 128;
    static constexpr int SMEM_UNITS = 3 * WSLOT + 2 * PSTAGE;
};

__device__ __forceinline__ void wait_vmcnt_le(int n) {      // n is wave-uniform, 0..3
    if (n >= 3) __builtin_amdgcn_s_waitcnt(0x0F73);
    else if (n == 2) __builtin_amdgcn_s_waitcnt(0x0F72);
    else if (n == 1) __builtin_amdgcn_s_waitcnt(0x0F71);
    else __builtin_amdgcn_s_waitcnt(0x0F70);
}

template <int WOCB, int TPW, int EPI>
__global__ __launch_bounds__(768) void convt_bf16x3_ws_kernel(Bf16Args a) {
    using C = TWCfg<WOCB, TPW>;
    constexpr int OCT = C::OCT, RPB = C::RPB, TQH = C::TQH, PWD = C::PWD, PLANE = C::PLANE, WHALF = C::WHALF, WSLOT = C::WSLOT, PSTAGE = C::PSTAGE;
    static_assert(C::SMEM_UNITS * 16 + (MAX_K_BF16X3 + KCB + 2 * OCT) * 4 <= 160 * 1024, "three weight slots and two patch stages fit the 160 KiB of LDS");
    const ConvArgs& p = a.c;
    __shared__ uint4 smem[C::SMEM_UNITS];
    __shared__ __attribute__((aligned(16))) float s_si[MAX_K_BF16X3 + KCB];
    __shared__ __attribute__((aligned(16))) float s_so[OCT], s_bias[OCT];
    uint4* const patches = smem + 3 * WSLOT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    int bid = blockIdx.x;
    const int grp = bid % a.groups;
    const int b = bid / a.groups;
    const int n0 = blockIdx.y * OCT;
    const int tiles_all = p.tiles_x * p.tiles_y;
    const int tstep = a.groups;
    const int ntiles = (tiles_all - grp + a.groups - 1) / a.groups;
    const int nchunks = p.K / KCB;
    const int items = ntiles * 2 * nchunks;
    const int chan = p.in_h * p.in_w;

    for (int k = tid; k < p.K; k += 768) s_si[k] = p.si ? p.si[(size_t)b * p.K + k] : 1.f;
    if (tid < OCT) {
        const int oc = n0 + tid;
        s_so[tid] = p.so ? p.so[(size_t)b * p.N + oc] : 1.f;
        s_bias[tid] = p.bias ? p.bias[oc] : 0.f;
    }
    __syncthreads();

    // items in order: for tile (grp, grp + groups, ...): for pass 0, 1: for chunk
    struct Item { int tile, pass, k0; };
    auto advance = [&](Item& it) {
        it.k0 += KCB;
        if (it.k0 >= p.K) { it.k0 = 0; if (++it.pass == 2) { it.pass = 0; it.tile += tstep; } }
    };

    if (wave >= 8) {
        // ---------------- staging waves ----------------
        const int st = tid - 512;
        const int kgl = __builtin_amdgcn_readfirstlane(st >> 7), tb = st & 127;     // waves 8, 9: channel group 0; waves 10, 11: group 1
        const float* xb = p.x + (size_t)b * p.K * chan;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)p.K * chan * 4u);
        uint4 pa[C::NT][8], pb[C::NT][8];
        // a patch row = the halo column (one pixel, the "edge" task) + TPW / 4 aligned groups of four pixels
        auto task = [&](int j, int& row, int& col, int& used) {
            const int t = tb + 128 * j;
            const int g = t % (C::GR + 1);
            row = t / (C::GR + 1);
            col = g == 0 ? 0 : 4 * g - 3;
            used = t < C::TASKS ? (g == 0 ? 1 : 4) : 0;
        };
        auto loads = [&](uint4 (&preg)[C::NT][8], const Item& it) {
            const int iy0 = (it.tile / p.tiles_x) * TQH - 1, ix0 = (it.tile % p.tiles_x) * TPW - 1;
#pragma unroll
            for (int j = 0; j < C::NT; ++j) {
                int row, col, used;
                task(j, row, col, used);
                const int iy = iy0 + row, ix = ix0 + col;
                const bool ok = used > 0 && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;      // tiles past the last one: zeros
                const unsigned boff = ok ? (unsigned)(iy * p.in_w + ix) * 4u : OOB;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    preg[j][q] = buf_load_u128(rx, boff, (unsigned)(it.k0 + kgl * 8 + q) * chan * 4u);
            }
        };
        auto convert = [&](uint4 (&preg)[C::NT][8], const Item& it, int buf) {
            uint4* const p_h = patches + buf * PSTAGE;
            uint4* const p_l = p_h + KG * PLANE;
            const float4 sa = *reinterpret_cast<const float4*>(&s_si[it.k0 + kgl * 8]), sb = *reinterpret_cast<const float4*>(&s_si[it.k0 + kgl * 8 + 4]);
            const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
            const int ix0 = (it.tile % p.tiles_x) * TPW - 1;
            auto body = [&](auto scaled) {
#pragma unroll
                for (int j = 0; j < C::NT; ++j) {
                    int row, col, used;
                    task(j, row, col, used);
                    const int inrow = p.in_w - (ix0 + col);              // pixels of this group that are still inside the image row
                    const int ubase = kgl * PLANE + row * PWD + col;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const unsigned raw = i == 0 ? preg[j][q].x : (i == 1 ? preg[j][q].y : (i == 2 ? preg[j][q].z : preg[j][q].w));
                            v[q] = i < inrow ? __uint_as_float(raw) : 0.f;
                        }
                        uint4 h, l;
                        split8s<decltype(scaled)::value>(v, sc, &h, &l);
                        if (i < used) {
                            p_h[ubase + i] = h;
                            GC_LO(p_l[ubase + i] = l;)
                        }
                    }
                }
            };
            if (p.si) body(std::true_type{}); else body(std::false_type{});       // without modulation (D's input gradients) the multiply by one is not issued
        };
        Item i0{grp, 0, 0};                            // item 0 -> set A
        loads(pa, i0);
        Item i1 = i0; advance(i1);                     // item 1 -> set B
        loads(pb, i1);
        convert(pa, i0, 0);
        __syncthreads();
        // interval `it`: the multiplying waves work on item it; item it + 1 is converted here, item it + 2 is fetched
        for (int it = 0; it < items; it += 2) {
            Item i2 = i1; advance(i2);
            if (!(GC_CTWS_ABL & 1)) { loads(pa, i2); convert(pb, i1, 1); }
            __syncthreads();
            if (it + 1 >= items) break;
            i1 = i2; advance(i1);
            if (!(GC_CTWS_ABL & 1)) { loads(pb, i1); convert(pa, i2, 0); }
            __syncthreads();
        }
        return;
    }

    // ---------------- multiplying waves ----------------
    const int ocb = wave % WOCB, rg = wave / WOCB;
    f32x16 acc[2][2];                                  // [px][block j]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int boff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) boff[j] = hi * PLANE + ((rg * 2 + j) * RPB + l31 / TPW) * PWD + l31 % TPW;      // patch row 0 = input row qy0 - 1
    const int aoff = hi * OCT + ocb * 32 + l31;

    // The slab of one item: rows (half, tap slot, kg) of OCT units; one LDS-DMA instruction moves 64 units = 64 / OCT rows (adjacent in
    // LDS), dealt round-robin to the eight multiplying waves.  Pass 0 holds taps (0,tx) in slots 0..2 and (2,tx) in slots 3..5; pass 1 taps
    // (1,tx) in slots 0..2.  Returns how many instructions THIS wave issued.
#ifdef GC_SINGLE
    constexpr int HALVES = 1;
#else
    constexpr int HALVES = 2;
#endif
    constexpr int RPI = 64 / OCT;
    auto weights = [&](auto pass_c, int k0, int slot) -> int {
        constexpr int PASS = decltype(pass_c)::value;
        constexpr int PER_HALF = (PASS == 0 ? 6 : 3) * KG, INSTR = HALVES * PER_HALF / RPI;
        uint4* const base = smem + slot * WSLOT;
        const int ln = opaque(tid) & 63;                                  // recomputed per call: nothing lane-dependent of this lambda stays live across the MFMAs
        const unsigned wlane = (unsigned)(n0 + ln % OCT) * 16u;           // this lane's unit inside a row of N units (bytes)
        int issued = 0;
#pragma unroll
        for (int j = 0; j < (INSTR + 7) / 8; ++j) {
            const int q = wave + 8 * j;
            if (8 * j + 7 < INSTR || q < INSTR) {
                const int r0 = q * RPI;                                   // first row of the group (wave-uniform)
                const int half = r0 / PER_HALF, rr0 = r0 % PER_HALF;
                const int rr = rr0 + ln / OCT;                            // this lane's row
                const int sl = rr / KG, kg = rr % KG;
                const int tap = PASS == 0 ? (sl < 3 ? sl : sl + 3) : sl + 3;
                const unsigned voff = (unsigned)((tap * a.kgroups + k0 / 8 + kg) * p.N) * 16u + wlane;
                if (!(GC_CTWS_ABL & 2)) glds16_s(half ? a.wl : a.wh, voff, base + half * WHALF + rr0 * OCT);
                ++issued;
            }
        }
        return (GC_CTWS_ABL & 2) ? 0 : issued;
    };
    auto weights_of = [&](const Item& it, int slot) -> int {
        return it.pass == 0 ? weights(std::integral_constant<int, 0>{}, it.k0, slot) : weights(std::integral_constant<int, 1>{}, it.k0, slot);
    };

    // Output through a buffer descriptor: lane offset = pixel (+ the hi half's 4 channels), scalar offset = channel plane -- no 64-bit
    // address per register row (hoisted out of the item loop, sixteen of them were spilled to scratch and reloaded in front of the stores).
    const int opitch = a.out_pitch;
    const unsigned oplane = (unsigned)(p.out_h * opitch) * 4u;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + (size_t)b * p.N * p.out_h * opitch, (unsigned)p.N * oplane);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc((EPI == 2 && p.residual) ? p.residual + (size_t)b * p.N * p.out_h * p.out_w : p.y,
                                                  (EPI == 2 && p.residual) ? (unsigned)(p.N * p.out_h * p.out_w) * 4u : 0u);
    const EpilogueConsts ec = epilogue_consts(p);
    // store the two phases of one finished (tile, pass) -- px = 0 / 1 of one input column leave as ONE 8-byte store -- and clear the accumulators
    auto finish = [&](int tile, int py) {
        const int qy0 = (tile / p.tiles_x) * TQH, qx0 = (tile % p.tiles_x) * TPW;
        const int nb = opaque_s(n0) + ocb * 32;         // recomputed here rather than carried across the item loop
        unsigned voff[2], roff[2];
        bool pair[2];
        float nz[2][2];          // every load of the epilogue is issued before the first store: a load between stores waits for the stores
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int qy = qy0 + (rg * 2 + j) * RPB + l31 / TPW, qx = qx0 + l31 % TPW;
            const int oy = 2 * qy + py, ox = 2 * qx;
            const bool inside = oy < p.out_h && ox < p.out_w;
            pair[j] = ox + 1 < p.out_w;
            voff[j] = inside ? (unsigned)(oy * opitch + ox) * 4u + (unsigned)(4 * hi) * oplane : OOB;
            roff[j] = inside ? (unsigned)(oy * p.out_w + ox) * 4u + (unsigned)(4 * hi * p.out_h * p.out_w) * 4u : OOB;
#pragma unroll
            for (int px = 0; px < 2; ++px)
                nz[j][px] = (EPI == 2 && p.noise && inside && (px == 0 || pair[j])) ? p.noise[((size_t)b * p.out_h + oy) * p.out_w + ox + px] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (EPI == 2 && p.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned so = (unsigned)((nb + (r & 3) + 8 * (r >> 2)) * p.out_h * p.out_w) * 4u;
                    const float r0 = buf_load_f32(rres, roff[j], so), r1 = buf_load_f32(rres, pair[j] ? roff[j] + 4u : OOB, so);
                    const int ocl = ocb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    acc[0][j][r] = conv_epilogue(ec, acc[0][j][r], s_so[ocl], s_bias[ocl], nz[j][0]) + r0;
                    acc[1][j][r] = conv_epilogue(ec, acc[1][j][r], s_so[ocl], s_bias[ocl], nz[j][1]) + r1;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ocl = ocb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                float v0 = acc[0][j][r], v1 = acc[1][j][r];
                if (EPI == 1) { v0 *= s_so[ocl]; v1 *= s_so[ocl]; }
                if (EPI == 2 && !p.residual) { v0 = conv_epilogue(ec, v0, s_so[ocl], s_bias[ocl], nz[j][0]); v1 = conv_epilogue(ec, v1, s_so[ocl], s_bias[ocl], nz[j][1]); }
                const int soff = (int)((unsigned)(nb + (r & 3) + 8 * (r >> 2)) * oplane);
                if (!(GC_CTWS_ABL & 8) || v0 == 12345.678f) {
                    typedef int i32x2 __attribute__((ext_vector_type(2)));
                    if (pair[j]) __builtin_amdgcn_raw_buffer_store_b64(i32x2{__builtin_bit_cast(int, v0), __builtin_bit_cast(int, v1)}, ry, (int)voff[j], soff, GC_CONV_ST_AUX);
                    else         __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v0), ry, (int)voff[j], soff, GC_CONV_ST_AUX);
                }
                acc[0][j][r] = 0.f; acc[1][j][r] = 0.f;
            }
        }
    };

    // one item's MFMAs.  Stages (patch offset (dyi, dxi); tap slots feeding px = 0 [, px = 1]):
    //   pass 0: (0,0): s0 | (0,1): s2, s1 | (1,0): s3 | (1,1): s5, s4        pass 1: (1,0): s0 | (1,1): s2, s1
    // -- the order in which convt_fused_bf16x3_kernel adds the taps of a phase.  The fragment reads of stage s + 1 are issued before the
    // MFMAs of stage s (two fragment sets; the scheduling barriers pin the order).
    auto multiply = [&](auto pass_c, int slot, int buf) {
        constexpr int PASS = decltype(pass_c)::value;
        constexpr int NS = PASS == 0 ? 4 : 2;
        const uint4* const wl_h = smem + slot * WSLOT;
        const uint4* const wl_l = wl_h + WHALF;
        const uint4* const p_h = patches + buf * PSTAGE;
        const uint4* const p_l = p_h + KG * PLANE;
        bf16x8 fa[2][2][2], fb[2][2][2];            // [set][tap of the stage | block j][hi, lo]
        auto load_stage = [&](int s, int set) {
            const int dyi = PASS == 0 ? s >> 1 : 1, dxi = s & 1;
            const int sbase = PASS == 0 ? 3 * (s >> 1) : 0;
#pragma unroll
            for (int i = 0; i < 1 + dxi; ++i) {
                const int sl = sbase + (dxi == 0 ? 0 : (i == 0 ? 2 : 1));
                const uint4 uh = wl_h[sl * KG * OCT + aoff];
                fa[set][i][0] = *reinterpret_cast<const bf16x8*>(&uh);
                GC_LO(const uint4 ul = wl_l[sl * KG * OCT + aoff]; fa[set][i][1] = *reinterpret_cast<const bf16x8*>(&ul);)
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint4 uh = p_h[boff[j] + dyi * PWD + dxi];
                fb[set][j][0] = *reinterpret_cast<const bf16x8*>(&uh);
                GC_LO(const uint4 ul = p_l[boff[j] + dyi * PWD + dxi]; fb[set][j][1] = *reinterpret_cast<const bf16x8*>(&ul);)
            }
        };
        load_stage(0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) load_stage(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 1 + (s & 1); ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) { GC_MFMA3(acc[i][j], fa[s & 1][i][0], fa[s & 1][i][1], fb[s & 1][j][0], fb[s & 1][j][1]); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    Item cur{grp, 0, 0};
    Item nx1 = cur; advance(nx1);
    weights_of(cur, 0);
    weights_of(nx1, 1);
    wait_staged_loads();
    __syncthreads();                 // patch stage 0 and weight slots 0, 1 are staged
    Item nx2 = nx1;
    int slot = 0, it = 0;
    // One item.  The two passes are two LOOPS, not a branch inside one loop: with both MFMA bodies under one loop the compiler keeps a
    // separate accumulator set per body and copies 64 registers at the join of every item.
    auto step = [&](auto pass_c) {
        advance(nx2);                                    // item it + 2 (past the last item: a valid slab into a slot nobody reads)
        const int s2 = slot >= 1 ? slot - 1 : 2;         // (slot + 2) % 3
        const int issued = weights_of(nx2, s2);
        __builtin_amdgcn_s_setprio(GC_MFMA_PRIO);
        multiply(pass_c, slot, it & 1);
        __builtin_amdgcn_s_setprio(0);
        wait_vmcnt_le(issued);       // the rows of item it + 1 (issued one item ago) have landed; those of item it + 2 may still be in flight
        __syncthreads();             // this patch stage and this weight slot may be rewritten from the next item on
        slot = slot == 2 ? 0 : slot + 1;
        ++it;
    };
    for (int t = 0; t < ntiles; ++t) {
        const int tile = grp + t * tstep;
        for (int c = 0; c < nchunks; ++c) step(std::integral_constant<int, 0>{});
        finish(tile, 0);
        for (int c = 0; c < nchunks; ++c) step(std::integral_constant<int, 1>{});
        finish(tile, 1);
    }
}

template <int WOCB, int TPW>
int launch_tws(Bf16Args a, hipStream_t s, bool main_only = false) {
    using C = TWCfg<WOCB, TPW>;
    const int qh = main_only ? a.c.in_h : gc::ceil_div(a.c.out_h, 2), qw = main_only ? a.c.in_w : gc::ceil_div(a.c.out_w, 2);     // main_only: the H x W region (convt_edge_bf16x3_kernel does the rest)
    a.c.tiles_y = gc::ceil_div(qh, C::TQH);
    a.c.tiles_x = gc::ceil_div(qw, TPW);
    const int tiles = a.c.tiles_x * a.c.tiles_y, ocb = a.c.N / C::OCT;
    const long long wgs = (long long)tiles * a.c.B * ocb;
    a.tpb = (int)std::min<long long>(std::max<long long>((wgs + GC_WS_SLOTS - 1) / GC_WS_SLOTS, 1), tiles);
    a.groups = gc::ceil_div(tiles, a.tpb);
    const long long gx = (long long)a.groups * a.c.B;
    if (gx > 2147483647LL) return gc::fail(GC_ERR_UNSUPPORTED, "gc_conv2d_bf16x3_f32: grid too large");
    if (gc::probing()) return gc::probe_name("convt_bf16x3_ws_kernel<%d,%d>|up2,down1,k3", WOCB, TPW);
    const dim3 grid((unsigned)gx, ocb);
    if (a.c.so) hipLaunchKernelGGL((convt_bf16x3_ws_kernel<WOCB, TPW, 1>), grid, dim3(768), 0, s, a);
    else        hipLaunchKernelGGL((convt_bf16x3_ws_kernel<WOCB, TPW, 0>), grid, dim3(768), 0, s, a);
    return gc::check_launch("gc_conv2d_bf16x3_f32(transposed, ws)");
}

// the transposed layers the wave-specialised kernel takes: whole 16-channel chunks, whole 32- / 64-channel output blocks, enough chunks
// per item sequence to amortise its start-up, enough tiles to give most CUs a workgroup
inline bool tws_eligible(const Bf16Args& a) {
    const ConvArgs& c = a.c;
    if (a.k_per_split || c.K % KCB != 0 || c.K < 32 || c.N % 32 != 0) return false;
    if (c.bias || c.noise || c.act || c.residual) return false;       // the full fused epilogue (no caller in the training step) stays on the one-role kernel
    const int qh = gc::ceil_div(c.out_h, 2), qw = gc::ceil_div(c.out_w, 2);
    if (qw < 32 || qh < 16) return false;
    const int oct = c.N % 64 == 0 ? 64 : 32;
    const long long wgs = (long long)gc::ceil_div(qw, 32) * gc::ceil_div(qh, oct == 64 ? 8 : 16) * c.B * (c.N / oct);
    return wgs >= 192;
}
