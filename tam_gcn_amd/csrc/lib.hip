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

extern "C" int tamgcn_version(void) { return TAMGCN_VERSION; }
extern "C" const char* tamgcn_last_error(void) { return g_err; }
