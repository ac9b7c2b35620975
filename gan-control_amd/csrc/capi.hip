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

}  // namespace gc

extern "C" int gc_abi_version(void) { return GC_ABI_VERSION; }
extern "C" const char* gc_last_error(void) { return gc::err_buf(); }
