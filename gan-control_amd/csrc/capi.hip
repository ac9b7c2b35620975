// Error plumbing and version entry points of the C ABI (include/gancontrol_hip.h).
#include "common.h"

namespace gc {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

char*& probe_buf() {
    static thread_local char* p = nullptr;
    return p;
}

int probe_name(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(probe_buf(), 128, fmt, ap);
    va_end(ap);
    return GC_OK;
}

size_t device_lds_limit() {
    static size_t cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 160 * 1024; }
    const int slot = dev & 15;
    if (cached[slot] == 0) {
        // On AMD parts one workgroup may take the whole LDS of its CU once the kernel asks for it (hipFuncAttributeMaxDynamicSharedMemorySize);
        // "per block" reports the default static limit (64 KiB), "per multiprocessor" the LDS of a CU (160 KiB on gfx950, 64 KiB before).
        int per_block = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); per_block = 0; }
        if (hipDeviceGetAttribute(&per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        const int v = per_block > per_cu ? per_block : per_cu;
        cached[slot] = v > 0 ? (size_t)v : (size_t)160 * 1024;
    }
    return cached[slot];
}

int allow_dynamic_lds(const void* kernel, size_t bytes, bool (&done)[16], const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(GC_ERR_HIP, "%s: no current device", what);
    if (done[dev & 15]) return GC_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(GC_ERR_HIP, "%s: cannot reserve %zu bytes of LDS: %s", what, bytes, hipGetErrorString(e)); }
    done[dev & 15] = true;
    return GC_OK;
}

}  // namespace gc

extern "C" int gc_conv2d_variant_name(const gc_conv_desc* d, int mode, char* name, int name_bytes) {
    if (!d || !name || name_bytes < 128) return gc::fail(GC_ERR_BAD_ARG, "gc_conv2d_variant_name: null pointer or fewer than 128 bytes");
    name[0] = 0;
    gc::probe_buf() = name;
    static float dummy[4];          // never dereferenced: the launchers return before any launch while probing
    int rc;
    if (mode == 0) rc = gc_conv2d_fused_f32(d, dummy, dummy, nullptr, nullptr, nullptr, dummy, nullptr);
    else {
        const size_t pb = gc_conv2d_bf16x3_packed_bytes(d);
        void* packed = pb ? reinterpret_cast<void*>(uintptr_t(16)) : nullptr;      // aligned, non-null, never dereferenced
        rc = mode == 2 ? gc_conv2d_fused_bf16_packed_f32(d, dummy, dummy, packed, pb, nullptr, nullptr, nullptr, dummy, nullptr, 0, nullptr)
                       : gc_conv2d_fused_bf16x3_packed_f32(d, dummy, dummy, packed, pb, nullptr, nullptr, nullptr, dummy, nullptr, 0, nullptr);
    }
    gc::probe_buf() = nullptr;
    return rc;
}

extern "C" int gc_abi_version(void) { return GC_ABI_VERSION; }
extern "C" int gc_struct_sizes(size_t* sizes, int n) {
    const size_t all[GC_STRUCT_COUNT] = {sizeof(gc_conv_desc), sizeof(gc_conv_epilogue), sizeof(gc_wlayout_group),
                                         sizeof(gc_wpack_group), sizeof(gc_glin_group), sizeof(gc_wsq_group)};
    for (int i = 0; sizes && i < n && i < GC_STRUCT_COUNT; ++i) sizes[i] = all[i];
    return GC_STRUCT_COUNT;
}
extern "C" const char* gc_last_error(void) { return gc::err_buf(); }
#ifndef GC_SOURCE_HASH
#define GC_SOURCE_HASH "unstamped"
#endif
extern "C" const char* gc_source_hash(void) { return GC_SOURCE_HASH; }
