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
extern "C" const char* gc_last_error(void) { return gc::err_buf(); }
