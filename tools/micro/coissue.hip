// Do matrix (MFMA) and vector (VALU) instructions of DIFFERENT waves on one SIMD overlap?  A 512-lane workgroup puts two waves on each
// SIMD (waves w and w + 4).  ROLE0 / ROLE1 = what waves 0-3 / 4-7 run: 0 nothing, 1 MFMA, 2 VALU (packed FMA), 3 both interleaved in one
// instruction stream, 4 integer VALU, 5 MFMA + integer VALU.  No memory traffic.  Build: hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ROLE>
__device__ __forceinline__ float work(int iters, const float* seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(seed[q] + threadIdx.x); b[q] = (__bf16)(seed[q + 8]); }
    float v[16];
    for (int q = 0; q < 16; ++q) v[q] = seed[q] + threadIdx.x;
    const float m = seed[16], c = seed[17];
    unsigned iv[8];
    for (int q = 0; q < 8; ++q) iv[q] = __float_as_uint(seed[q]) + threadIdx.x;
    const unsigned im = __float_as_uint(seed[16]), ic = __float_as_uint(seed[17]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ROLE & 1) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u], 0, 0, 0);
            if (ROLE & 2) {
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = __builtin_fmaf(v[q], m, c);      // 8 v_pk_fma_f32 = 16 FMAs per MFMA slot
            }
            if (ROLE & 8) {          // un-packed fp32: 16 v_fma_f32
#pragma unroll
                for (int q = 0; q < 16; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(m), "v"(c));
            }
            if (ROLE & 16) {         // the bf16 conversion instruction: 8 v_cvt_pk_bf16_f32
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(iv[q]) : "v"(v[2 * q]), "v"(v[2 * q + 1]));
            }
            if (ROLE & 4) {          // integer / bit ALU work (what index arithmetic, masks and bf16 packing are made of): 8 v_xad_u32-like ops
#pragma unroll
                for (int q = 0; q < 8; ++q) iv[q] = (iv[q] ^ im) + ic;
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int q = 0; q < 16; ++q) s += v[q];
    for (int q = 0; q < 8; ++q) s += __uint_as_float(iv[q]);
    return s;
}

template <int ROLE0, int ROLE1>
__global__ __launch_bounds__(512) void mix(float* out, int iters, const float* seed) {
    float s = 0.f;
    if ((threadIdx.x >> 8) == 0) { if (ROLE0) s = work<ROLE0>(iters, seed); }
    else                         { if (ROLE1) s = work<ROLE1>(iters, seed); }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int ROLE0, int ROLE1>
void run(const char* name, int iters, const float* seed, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix<ROLE0, ROLE1><<<256, 512>>>(out, iters, seed); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mix<ROLE0, ROLE1><<<256, 512>>>(out, iters, seed);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-70s %8.3f ms\n", name, best);
}

int main() {
    float h[18]; for (int i = 0; i < 18; ++i) h[i] = 0.5f + 0.01f * i;
    h[16] = 0.999f; h[17] = 0.001f;
    float* seed; hipMalloc(&seed, sizeof(h)); hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;       // per wave: 80 000 MFMAs (32 cycles each = 2.56 M cycles) and / or 640 000 packed FMAs
    run<1, 0>("one wave per SIMD: MFMA", iters, seed, out);
    run<2, 0>("one wave per SIMD: VALU (8 v_pk_fma per MFMA slot)", iters, seed, out);
    run<3, 0>("one wave per SIMD: MFMA + VALU interleaved in the instruction stream", iters, seed, out);
    run<1, 1>("two waves per SIMD: MFMA | MFMA", iters, seed, out);
    run<2, 2>("two waves per SIMD: VALU | VALU", iters, seed, out);
    run<1, 2>("two waves per SIMD: MFMA | VALU", iters, seed, out);
    run<3, 3>("two waves per SIMD: interleaved | interleaved", iters, seed, out);
    run<4, 0>("one wave per SIMD: integer VALU (8 ops per MFMA slot)", iters, seed, out);
    run<5, 0>("one wave per SIMD: MFMA + integer VALU interleaved", iters, seed, out);
    run<1, 4>("two waves per SIMD: MFMA | integer VALU", iters, seed, out);
    run<5, 5>("two waves per SIMD: MFMA + integer interleaved | same", iters, seed, out);
    run<8, 0>("one wave per SIMD: 16 v_fma_f32 per MFMA slot", iters, seed, out);
    run<9, 0>("one wave per SIMD: MFMA + v_fma_f32 interleaved", iters, seed, out);
    run<1, 8>("two waves per SIMD: MFMA | v_fma_f32", iters, seed, out);
    run<16, 0>("one wave per SIMD: 8 v_cvt_pk_bf16_f32 per MFMA slot", iters, seed, out);
    run<17, 0>("one wave per SIMD: MFMA + v_cvt_pk_bf16_f32 interleaved", iters, seed, out);
    run<1, 16>("two waves per SIMD: MFMA | v_cvt_pk_bf16_f32", iters, seed, out);
    return 0;
}
