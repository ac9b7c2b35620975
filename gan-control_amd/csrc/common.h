// Shared helpers for the gfx950 kernels: error plumbing and launch checks.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "gancontrol_hip.h"

namespace gc {

// thread-local error text behind gc_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

// Dispatch probe (gc_conv2d_variant_name): while probe_buf() is non-null the convolution launchers write the name of the kernel
// variant they WOULD launch into it and return without launching -- the name comes from the dispatch code itself, so a host-side
// profiler cannot drift from it.
char*& probe_buf();
inline bool probing() { return probe_buf() != nullptr; }
int probe_name(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GC_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return GC_OK;
}

// Per-block LDS limit of the CURRENT device in bytes (hipDeviceAttributeMaxSharedMemoryPerBlock, cached per device); 160 KiB -- the gfx950
// figure -- where no device can be asked (the build container's no-launch probes and workspace queries).
size_t device_lds_limit();
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device); GC_OK or GC_ERR_HIP.
int allow_dynamic_lds(const void* kernel, size_t bytes, bool (&done)[16], const char* what);

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// floor division / modulo for possibly negative numerators
__host__ __device__ inline int floor_div(int a, int b) {
    int q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}
__host__ __device__ inline int pos_mod(int a, int b) {
    int m = a % b;
    return m < 0 ? m + b : m;
}

// Streaming stores.  The big activation / gradient tensors of the training step (134 MB .. 1 GB) are written once and read again only by a
// later kernel, long after the 4 MB L2s and most of the 256 MB Infinity Cache have turned over; a NON-TEMPORAL store keeps them from
// displacing the lines the running kernel still wants (halo rows, noise planes) and lets the memory side stream the writes.  Round 4,
// same-box A/B on the 4 x 4 FIR tile kernel: 4.81 -> 5.77 TB/s at [4, 32, 1025^2] (profiles/pmc_r04_fir.md).  GC_NT_STORE=0 restores plain stores.
#ifndef GC_NT_STORE
#define GC_NT_STORE 1
#endif
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u_t __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ void stream_store4(float* dst, float a, float b, float c, float d) {           // 16-byte aligned
#if GC_NT_STORE
    __builtin_nontemporal_store(f32x4_t{a, b, c, d}, reinterpret_cast<f32x4_t*>(dst));
#else
    *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{a, b, c, d};
#endif
}
__device__ __forceinline__ void stream_store4u(float* dst, float a, float b, float c, float d) {          // 4-byte aligned (gfx9 takes 16-byte accesses there)
#if GC_NT_STORE
    __builtin_nontemporal_store(f32x4u_t{a, b, c, d}, reinterpret_cast<f32x4u_t*>(dst));
#else
    *reinterpret_cast<f32x4u_t*>(dst) = f32x4u_t{a, b, c, d};
#endif
}
// ... and the matching loads for operands a kernel reads exactly once (never for anything with halo / cross-workgroup re-use)
#ifndef GC_NT_LOAD
#define GC_NT_LOAD 1      // round 4, same-box A/B on the whole step: 76.03 -> 76.24 images/s with the activation kernels' operands loaded non-temporally
#endif
__device__ __forceinline__ f32x4_t stream_load4(const float* src) {
#if GC_NT_LOAD
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src));
#else
    return *reinterpret_cast<const f32x4_t*>(src);
#endif
}
__device__ __forceinline__ f32x4u_t stream_load4u(const float* src) {
#if GC_NT_LOAD
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4u_t*>(src));
#else
    return *reinterpret_cast<const f32x4u_t*>(src);
#endif
}
__device__ __forceinline__ void stream_store1(float* dst, float a) {
#if GC_NT_STORE
    __builtin_nontemporal_store(a, dst);
#else
    *dst = a;
#endif
}

}  // namespace gc
