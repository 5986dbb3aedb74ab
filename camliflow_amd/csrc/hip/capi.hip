// Library-level entry points of libcamli_hip.so: version, last-error string, launch checking.
#include "camli_common.h"

#include <stdarg.h>
#include <stdio.h>

namespace {
thread_local char g_err[512] = "";
}

void camli_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int camli_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        camli_set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
        return CAMLI_ELAUNCH;
    }
    return CAMLI_OK;
}

extern "C" int camli_version(void) { return 100; }  // 0.1.0

extern "C" const char* camli_last_error_string(void) { return g_err; }
