// Error reporting and library identification for the C-ABI (include/stx_hip.h).
#include "stx_common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

int stx_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int stx_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return stx_set_error(STX_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return STX_OK;
}

extern "C" const char* stx_last_error(void) { return g_err; }

extern "C" const char* stx_build_info(void) {
#ifdef STX_HIPEMU
    return "stx-hipemu (host SIMT emulator build: tests only)";
#else
    return "stx-gfx950 (hipcc --offload-arch=gfx950)";
#endif
}
