// Shared by the split-bf16 convolution translation units (conv_bf16x3.hip, conv_s2ws.hip): build knobs, the argument block, the hi / lo split of
// eight staged values, LDS-DMA helpers.  Everything sits in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "conv_common.h"

// The same source builds a second time with -DGC_SINGLE (object conv_bf16.o): plain bf16 arithmetic -- ONE MFMA per product on the
// hi parts only (bf16 operands, fp32 accumulate, fp32 in HBM; ~3e-3 relative error per layer, the precision of BASELINE.json's
// config[1] "bf16").  The lo parts are then neither converted, stored to LDS nor read; entry points gc_conv2d_fused_bf16_packed_f32 /
// gc_conv2d_wgrad_bf16_f32.  It is never the default nor the parity mode.
#ifdef GC_SINGLE
#define GC_LO(...)
#define GC_MFMA3(c, ah, al, bh, bl) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0)
#else
#define GC_LO(...) __VA_ARGS__
#define GC_MFMA3(c, ah, al, bh, bl)                                        \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);       \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);       \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0)
#endif

// Wave priority during the MFMA phases: the co-resident workgroup is staging (vector ALU, LDS, loads) meanwhile; letting the
// multiplying wave issue first keeps the matrix pipe fed (same-box A/B: -2 % forward, -1..3 % weight gradients).
#ifndef GC_MFMA_PRIO
#define GC_MFMA_PRIO 2
#endif
#ifndef GC_PLAIN_SPLIT
#define GC_PLAIN_SPLIT 1      // dev knob: 0 = leave the scale-and-split arithmetic to the compiler (it forms packed fp32 instructions)
#endif
#ifndef GC_WS_MIN_K
#define GC_WS_MIN_K 64      // input channels from which the wave-specialised forward kernel takes over
#endif
#ifndef GC_WG_XCD
#define GC_WG_XCD 1          // XCD-aware block order of the weight-gradient kernels (0: hardware order)
#endif
#ifndef GC_WS_MIN_K32
#define GC_WS_MIN_K32 32    // ... and its 32-output-channel variant
#endif
#ifndef GC_WS_WIDE_MAX_K
#define GC_WS_WIDE_MAX_K 64   // input channels up to which the wave-specialised kernel uses 8 x 64 tiles
#endif
#ifndef GC_WS_WIDE_CB32
#define GC_WS_WIDE_CB32 4      // column blocks of the wide tiles of the 32-output-channel variant (4: 4 rows x 128 px)
#endif
#ifndef GC_CT_ABL
#define GC_CT_ABL 0           // dev ablations of convt_fused_bf16x3_kernel (wrong results): 1 no stores, 2 no MFMAs, 4 no global loads, 8 no conversion (a pre-split input)
#endif
#ifndef GC_CT_DMA
#define GC_CT_DMA 0           // 1: convt_fused_bf16x3_kernel copies the pre-split weight slab of a chunk HBM -> LDS by LDS-DMA instead of through registers.
                              // Measured SLOWER at >= 256 input channels (512 -> 256 @64^2, B = 4: 211 -> 239 us; 256 -> 128 @128^2: 183 -> 191; neutral at <= 128): with
                              // one weight stage the DMA is issued after the barrier that ends the MFMA phase and its latency is exposed before the next one.
#endif
#ifndef GC_WS_DMA_STAGER
#define GC_WS_DMA_STAGER 0     // 1: the staging waves issue the weight LDS-DMA -- measured SLOWER (512 -> 512 @64^2: 177 -> 203 us): the DMA wait lands on the staging waves' critical path
#endif
#ifndef GC_CONVT_NARROW
#define GC_CONVT_NARROW 1       // 16-column q-tiles where they waste fewer lanes than 32-column ones (0: always 32)
#endif
#ifndef GC_WS_SLOTS
#define GC_WS_SLOTS 256     // workgroups the wave-specialised kernel keeps resident: one per CU
#endif
#ifndef GC_WS_STAGER_PRIO
#define GC_WS_STAGER_PRIO 0
#endif
#ifndef GC_CONV_STRIDED
#define GC_CONV_STRIDED 0   // ... and of conv_bf16x3_kernel's tiles per workgroup: measured neutral (its >= 2048 short-lived workgroups are dispatched in tile order anyway)
#endif
#ifndef GC_WG2_STRIDED
#define GC_WG2_STRIDED 1    // ... and of the stride-2 weight gradient
#endif
#ifndef GC_WG_STRIDED
#define GC_WG_STRIDED 1     // the same for the pixel splits of the stride-1 weight gradient (tiles across the rows, a split's tiles gridDim.z apart)
#endif
#ifndef GC_WS_STRIDED
#define GC_WS_STRIDED 1     // a workgroup's tiles are `groups` apart instead of consecutive (DRAM locality of the resident workgroups)
#endif
#ifndef GC_S2_ABL
#define GC_S2_ABL 0         // dev ablations of conv_bf16x3_kernel at stride 2 (wrong results): 1 the loaded registers are written to LDS as they are -- no scale, no hi / lo
                            // split, no masks: what a PRE-SPLIT input (the producer emitting channel-last bf16 pairs, same bytes) would leave of the staging;
                            // 2 no patch loads and no patch commit at all (weights, fragment reads, MFMAs, stores only)
#endif
#ifndef GC_FRAG_PIPE
#define GC_FRAG_PIPE 1      // conv_bf16x3_kernel (up = 1): fragment reads of the next tap issued before the MFMAs of the current one (0: the compiler's order)
#endif
#ifndef GC_WS_XCD
#define GC_WS_XCD 0           // 1: XCD-aware block order of conv_bf16x3_ws_kernel (see the kernel)
#endif
#ifndef GC_WS_EARLY_DMA
#define GC_WS_EARLY_DMA 1   // conv_bf16x3_ws_kernel: weight slabs requested before a finished tile's stores, counted vmcnt, raw barrier (see the multiplying waves' loop)
#endif
#ifndef GC_WS_ABL
#define GC_WS_ABL 0         // dev ablations of conv_bf16x3_ws_kernel (wrong results): 1 no patch staging, 2 no weight DMA, 4 fragments read once, 8 no stores
#endif
#include <type_traits>

// Output stores of the convolution kernels: non-temporal (aux bit 1 = the `nt` bit of gfx94x / gfx950 buffer stores) when GC_CONV_NT is set.
// Round 4, same-box A/B on the whole step (two alternations): 75.15 -> 75.44 images/s on top of the streaming kernels' own non-temporal stores
// (GC_NT_STORE, common.h: 74.17 -> 75.15).
#ifndef GC_CONV_NT
#define GC_CONV_NT 1
#endif
#define GC_CONV_ST_AUX (GC_CONV_NT ? 2 : 0)
#ifndef GC_WS_SHIFT
#define GC_WS_SHIFT 0        // 1: patch fragments of the jx > 0 taps by a whole-wave DPP shift of the previous tap's instead of an LDS read: correct (all
                             // convolution tests) and SLOWER -- 512 -> 512 @64^2 181 -> 216 us, 64 -> 64 @512^2 197 -> 280 (round 4, same box): 16 v_mov_b32_dpp per tap
                             // on the MULTIPLYING waves (and 20 spilled registers) cost more than the 4 ds_read_b128 they replace
#endif
#ifndef GC_WS_DMA_HALF
#define GC_WS_DMA_HALF 0      // 1: the weight slab is issued by multiplying waves 0..3 only (one per SIMD), so that the partner wave has the matrix pipe meanwhile.  MEASURED NEUTRAL
                              // (round 6, profiles/ws_dma_half_r06_n.log: every shape within +-2 %): the trace then shows the issuing wave spending 4 260 cycles on its nine LDS-DMA
                              // instructions (470 each; 166 each when all eight waves issue 4-5) -- the instructions queue behind the staging waves' loads in the CU's one
                              // vector-memory path, whoever issues them
#endif
#ifndef GC_WS_DMA_MID
#define GC_WS_DMA_MID 0       // 1: weight slab of the next item requested inside the MFMA phase, staggered by wave group (see the multiplying waves' loop).  MEASURED neutral at >= 128
                              // channels and 3-5 % SLOWER at 32 / 64 (round 6, profiles/ws_dma_mid_r06_n2.log): where the slab is requested does not matter
#endif
#ifndef GC_WS_BARE
#define GC_WS_BARE 1         // reduced epilogues (EPK 1 / 2) of the wave-specialised kernel for launches without bias / noise / activation (0: always the full epilogue)
#endif
#ifndef GC_WS_NT_LOAD
#define GC_WS_NT_LOAD 0      // non-temporal patch loads in the wave-specialised forward kernel: measured SLOWER (dominant kernel 409 -> 386 TF/s, step -2 %):
                             // every patch is re-read by the other output-channel blocks and by the neighbouring tiles' halos
#endif


namespace gcconv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KCB = 16;   // input channels per chunk = K of one bf16 MFMA
constexpr int KG = 2;     // 8-channel groups per chunk
constexpr int MAX_K_BF16X3 = 1024;   // in_scale of one sample is kept in LDS

struct Bf16Args {
    ConvArgs c;
    const uint4* wh; const uint4* wl;   // packed weights [tap][ceil(K/8)][N] units of 8 bf16 (hi / lo parts)
    int kgroups;                        // ceil(K / 8)
    int tpb, groups;                    // conv_bf16x3_kernel: pixel tiles per workgroup, workgroups per (sample, phase)
    // grid-level split over the input channels (small planes): slice blockIdx.z covers channels [z * k_per_split, (z + 1) * k_per_split)
    // and writes RAW partial sums to part + z * per_slice; splitk_finish_kernel (conv.hip) adds the slices and applies the epilogue
    int k_per_split; float* part; long long per_slice;
    int out_pitch;                      // floats between output rows (convt_fused_bf16x3_kernel only; out_w elsewhere)
    int in_pitch;                       // floats between input rows (conv_bf16x3_kernel; in_w when dense)
};

// defined in conv_s2ws.hip (split build only): the wave-specialised stride-2 3x3 kernel and the shapes it takes
bool s2ws_eligible(const Bf16Args& a);
int launch_s2ws(Bf16Args a, hipStream_t s);

}  // namespace gcconv

namespace {

using namespace gcconv;

// Eight values with one scale EACH (a channel-last patch unit: eight channels of one pixel) -> hi / lo bf16 units.  Plain v_mul_f32 /
// v_sub_f32 by asm: left to itself the compiler pairs neighbouring channels into v_pk_mul_f32 / v_pk_add_f32, and packed fp32 does not
// run under another wave's MFMA (profiles/pmc_r01.md, co-issue table) -- in the wave-specialised kernel the staging wave shares its SIMD
// with two multiplying waves, so every packed instruction is time taken from the matrix pipe.
// Two fp32 values -> one dword of two bf16 (a in the low half), round to nearest even: ONE v_cvt_pk_bf16_f32.  Written `(__bf16)f` element by
// element the compiler spends one conversion per VALUE on the hi parts (it needs each hi back as a float for the lo part) and v_perm to pair
// them up: 5.2 vector instructions per value in the staging loops (round 5, opcode histogram of the commit phase); from the PAIR the two hi
// parts come back as floats by one shift and one mask -- 3 per value unscaled, 4 scaled, the same bits.
#ifndef GC_PAIR_SPLIT
#define GC_PAIR_SPLIT 1
#endif
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <bool SCALED>
__device__ __forceinline__ void split8s(const float (&v)[8], const float (&sc)[8], uint4* h, uint4* l) {
#if GC_PAIR_SPLIT && GC_PLAIN_SPLIT
    unsigned hh[4], ll[4];
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        float f0 = v[q], f1 = v[q + 1];
        if (SCALED) { asm("v_mul_f32 %0, %1, %2" : "=v"(f0) : "v"(v[q]), "v"(sc[q])); asm("v_mul_f32 %0, %1, %2" : "=v"(f1) : "v"(v[q + 1]), "v"(sc[q + 1])); }
        const unsigned pk = cvt_pk_bf16(f0, f1);
        const float t0 = __uint_as_float(pk << 16), t1 = __uint_as_float(pk & 0xffff0000u);
        float d0, d1;
        // the low part of a scaled value is taken from the EXACT product (one fused multiply-subtract), so the rounding of v * sc to fp32
        // is captured as well -- what the compiler's own contraction of `v * sc - hi` does
        if (SCALED) { asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(d0) : "v"(v[q]), "v"(sc[q]), "v"(t0)); asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(d1) : "v"(v[q + 1]), "v"(sc[q + 1]), "v"(t1)); }
        else        { asm("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(f0), "v"(t0)); asm("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(f1), "v"(t1)); }
        hh[q / 2] = pk;
        ll[q / 2] = cvt_pk_bf16(d0, d1);
    }
    *h = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    *l = make_uint4(ll[0], ll[1], ll[2], ll[3]);
#else
    bf16x8 hh, ll;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float f = v[q];
#if GC_PLAIN_SPLIT
        if (SCALED) asm("v_mul_f32 %0, %1, %2" : "=v"(f) : "v"(v[q]), "v"(sc[q]));
#else
        if (SCALED) f = v[q] * sc[q];
#endif
        const __bf16 t = (__bf16)f;
        hh[q] = t;
        const float tf = (float)t;
        float dlo;
#if GC_PLAIN_SPLIT
        if (SCALED) asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(dlo) : "v"(v[q]), "v"(sc[q]), "v"(tf));
        else        asm("v_sub_f32 %0, %1, %2" : "=v"(dlo) : "v"(f), "v"(tf));
#else
        dlo = f - tf;
#endif
        ll[q] = (__bf16)dlo;
    }
    *h = *reinterpret_cast<uint4*>(&hh);
    *l = *reinterpret_cast<uint4*>(&ll);
#endif
}


// LDS-DMA of one 1 KiB row: lane l copies 16 bytes from its own global address to (wave-uniform LDS address) + 16 l.  M0 carries the LDS
// address and is compiler-reserved: saved and restored inside the statement (cdna guide, "M0 ... write it in the same statement").
__device__ __forceinline__ void glds16(const void* gsrc, const void* lds_row) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)lds_row);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// The same with a wave-uniform 64-bit base (SGPR pair) and a 32-bit lane byte offset: no 64-bit vector address arithmetic, one VGPR per lane.
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, const void* lds_row) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)lds_row);
    const unsigned long long sb = (unsigned long long)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
    const unsigned long long base = ((unsigned long long)hi << 32) | lo;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

// a + b / a * s, never contracted with a neighbouring operation
__device__ __forceinline__ float plain_sum(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float plain_mul(float a, float s) {
#pragma clang fp contract(off)
    return a * s;
}

}  // namespace
