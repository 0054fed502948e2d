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

// ---------------------------------------------------------------------------------------------- tuning switches
#include <stdlib.h>
#include <string.h>
namespace {
struct TuneEntry { const char* name; int value; };
TuneEntry g_tune[STX_TUNE_COUNT] = {
    {"STX_MARCH_BS", 1}, {"STX_MARCH_EPI", 1}, {"STX_MARCH_ABLATE", 0}, {"STX_WGRAD_ABLATE", 0}, {"STX_WGRAD_MARCH", 3}, {"STX_WGRAD_GRID", 0}, {"STX_CONV_L1_MARCH", 0}, {"STX_CONV_S2_DENSE", 1}, {"STX_CONV_WN", 2},
    {"STX_CV_OLD", 0}, {"STX_CV_GRID", 0}, {"STX_CV_PF", 0}, {"STX_CV_UNITS", 1}, {"STX_CV_WIN", 0}, {"STX_CVB_OLD", 0}, {"STX_CVB_TEAM", 0}, {"STX_CVB_GRID", 0}, {"STX_CVB_NSET", 3},
    {"STX_SV_BWD_V1", 0}, {"STX_DWCONV_ROLL", 1},
};
struct TuneInit {                       // environment read once, when the library is loaded
    TuneInit() {
        for (int i = 0; i < STX_TUNE_COUNT; ++i)
            if (const char* e = getenv(g_tune[i].name)) g_tune[i].value = atoi(e);
    }
} g_tune_init;
}  // namespace

int stx_tune(StxTune id) { return g_tune[id].value; }

extern "C" int stx_get_tuning(const char* name) {
    for (int i = 0; i < STX_TUNE_COUNT; ++i)
        if (name && !strcmp(name, g_tune[i].name)) return g_tune[i].value;
    return -1;
}

extern "C" int stx_set_tuning(const char* name, int value) {
    for (int i = 0; i < STX_TUNE_COUNT; ++i)
        if (name && !strcmp(name, g_tune[i].name)) { g_tune[i].value = value; return STX_OK; }
    return stx_set_error(STX_ERR_ARG, "stx_set_tuning: unknown switch %s", name ? name : "(null)");
}
