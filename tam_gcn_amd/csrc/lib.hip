// Library-level entry points: version and per-thread error text.
#include "common.h"
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void tamgcn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local char g_kernel[96] = "";

void tamgcn_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

extern "C" const char* tamgcn_last_kernel(void) { return g_kernel; }
extern "C" int tamgcn_version(void) { return TAMGCN_VERSION; }
extern "C" const char* tamgcn_last_error(void) { return g_err; }

// Split-fp32 (3 x bf16 MFMA) policy for the GEMM kernels, read once from the environment.
static int g_split_mode = -1;

int tamgcn_split_mode(void) {
    int mode = __atomic_load_n(&g_split_mode, __ATOMIC_RELAXED);
    if (mode < 0) {
        const char* e = getenv("TAMGCN_SPLIT_BF16");
        mode = e ? atoi(e) : 0;          // the reference is fp32 throughout: exact unless asked otherwise
        if (mode < 0 || mode > 1) mode = 0;
        __atomic_store_n(&g_split_mode, mode, __ATOMIC_RELAXED);
    }
    return mode;
}

int tamgcn_wgrad_taps(void) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TAMGCN_WGRAD_TAPS"); v = e ? atoi(e) : 1; }
    return v;
}

extern "C" int tamgcn_get_split_mode(void) { return tamgcn_split_mode(); }
extern "C" int tamgcn_set_split_mode(int mode) {
    TG_CHECK(mode >= 0 && mode <= 1, "tamgcn_set_split_mode: mode %d outside 0..1", mode);
    __atomic_store_n(&g_split_mode, mode, __ATOMIC_RELAXED);
    return 0;
}
