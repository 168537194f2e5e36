// Library-level entry points: version and per-thread error text.
#include "common.h"
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
