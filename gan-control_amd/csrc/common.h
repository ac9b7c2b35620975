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

}  // namespace gc
