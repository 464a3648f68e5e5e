// Host-side plumbing shared by every entry point of libtoist_hip.so: version + per-thread error text.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "common.h"

namespace toist {

static thread_local char g_err[512] = {0};

void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error("%s: %s", what, hipGetErrorString(e));
        return TOIST_EHIP;
    }
    return TOIST_OK;
}

}  // namespace toist

extern "C" int toist_version(void) { return TOIST_ABI_VERSION; }

extern "C" int toist_last_error(char* buf, size_t cap) {
    const size_t n = strlen(toist::g_err);
    if (buf && cap) {
        const size_t m = n < cap - 1 ? n : cap - 1;
        memcpy(buf, toist::g_err, m);
        buf[m] = 0;
    }
    return (int)n;
}
