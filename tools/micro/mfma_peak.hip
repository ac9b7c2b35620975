// Measures the sustained v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32 issue rate of the whole chip (no memory traffic).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool F32>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(float)(threadIdx.x + q); b[q] = (__bf16)(float)(q + 1); }
    float fa = threadIdx.x, fb = 2.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (F32) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// random operands, 8 different fragment pairs cycled -> realistic data toggling in the multipliers
__global__ __launch_bounds__(256) void spin_rand(const uint4* __restrict__ src, float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[8], b[8];
    for (int q = 0; q < 8; ++q) {
        uint4 u = src[(q * 2) * 256 + threadIdx.x], v = src[(q * 2 + 1) * 256 + threadIdx.x];
        a[q] = *reinterpret_cast<bf16x8*>(&u); b[q] = *reinterpret_cast<bf16x8*>(&v);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + i) & 7], b[(u + 3 * i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

void run_rand(int blocks, int iters) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    uint4* src; hipMalloc(&src, 16 * 256 * 16);
    unsigned short* h = new unsigned short[16 * 256 * 8];
    unsigned seed = 12345;
    for (int i = 0; i < 16 * 256 * 8; ++i) { seed = seed * 1664525u + 1013904223u; h[i] = (unsigned short)(((seed >> 16) & 0x807f) | 0x3f00 | ((seed >> 8) & 0x80)); }  // random sign/mantissa, exponent ~1
    hipMemcpy(src, h, 16 * 256 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    spin_rand<<<blocks, 256>>>(src, out, iters); hipDeviceSynchronize();
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        spin_rand<<<blocks, 256>>>(src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = (double)blocks * 4 * iters * 32;
        printf("bf16 random operands         blocks=%5d  %8.3f ms  %8.1f TFLOP/s   implied clock at 100%% pipe: %.2f GHz\n", blocks, ms, mfmas * 32768.0 / ms / 1e9,
               mfmas * 32 / 1024.0 / (ms * 1e-3) / 1e9);
    }
}

template <int NACC, bool F32>
void run(const char* name, int blocks, int iters) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    spin<NACC, F32><<<blocks, 256>>>(out, iters);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        spin<NACC, F32><<<blocks, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = (double)blocks * 4 * iters * 8 * NACC;
        const double flops = mfmas * (F32 ? 32.0 * 32 * 2 * 2 : 32.0 * 32 * 16 * 2);
        const double cyc = F32 ? 64 : 32;   // nominal pipe cycles per instruction
        printf("%-28s blocks=%5d  %8.3f ms  %8.1f TFLOP/s   implied clock at 100%% pipe: %.2f GHz\n", name, blocks, ms, flops / ms / 1e9,
               mfmas * cyc / 1024.0 / (ms * 1e-3) / 1e9 * (blocks >= 256 ? 1.0 : 256.0 / blocks));
    }
    hipFree(out);
}

int main() {
    run<4, false>("bf16 4acc 1 wave/SIMD", 256, 20000);
    run<4, false>("bf16 4acc 2 waves/SIMD", 512, 10000);
    run<2, false>("bf16 2acc 2 waves/SIMD", 512, 20000);
    run<4, false>("bf16 4acc 4 waves/SIMD", 1024, 5000);
    run<4, true>("f32 4acc 2 waves/SIMD", 512, 5000);
    run<4, false>("bf16 4acc 1 wave/SIMD, 32 CUs", 32, 20000);
    run_rand(512, 10000);
    run_rand(512, 100000);
    return 0;
}
