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
        mode = e ? atoi(e) : 1;
        if (mode < 0 || mode > 2) mode = 1;
        __atomic_store_n(&g_split_mode, mode, __ATOMIC_RELAXED);
    }
    return mode;
}

// Three-term split (fp32-exact to ~1e-6 of the output scale) in the FORWARD 1x1 GEMMs into >= 128 channels: opt-in
// (TAMGCN_SPLIT3_FWD=1, with split mode >= 1).  Measured r02: 11 % on those GEMMs (config 4: 151 -> 149 ms / step, N-UCLA
// and NTU steps unchanged) -- the 128-row split kernels are bound by the operand splitting and their two-stage DMA ring, not
// by the matrix pipe -- while its 2-8x fp32 rounding noise moves more ReLU masks in the 4-clip SGD fixtures: not the default.
// two-term split data-gradient GEMMs also for the 64-channel layers (64-row tiles): opt-in (TAMGCN_SPLIT64=1).  Measured r02: the
// 2.18 ms of exact kernels it replaces become 2.31 ms -- those launches are bound by their prologue / epilogue latency, not by the matrix pipe.
static int g_rows128 = -1;
int tamgcn_rows128(void) {
    if (g_rows128 < 0) { const char* e = getenv("TAMGCN_ROWS128"); g_rows128 = e ? (atoi(e) != 0) : 0; }
    return g_rows128;
}
extern "C" int tamgcn_set_rows128(int on) { g_rows128 = on ? 1 : 0; return 0; }

int tamgcn_wgrad_taps(void) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TAMGCN_WGRAD_TAPS"); v = e ? atoi(e) : 1; }
    return v;
}

int tamgcn_split64(void) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TAMGCN_SPLIT64"); v = e ? (atoi(e) != 0) : 0; }
    return v;
}

static int g_split3 = -1;
int tamgcn_split3_fwd(void) {
    if (g_split3 < 0) { const char* e = getenv("TAMGCN_SPLIT3_FWD"); g_split3 = e ? (atoi(e) != 0) : 0; }
    return g_split3 && tamgcn_split_mode() >= 1;
}
extern "C" int tamgcn_set_split3_fwd(int on) { g_split3 = on ? 1 : 0; return 0; }

extern "C" int tamgcn_get_split_mode(void) { return tamgcn_split_mode(); }
extern "C" int tamgcn_set_split_mode(int mode) {
    TG_CHECK(mode >= 0 && mode <= 2, "tamgcn_set_split_mode: mode %d outside 0..2", mode);
    __atomic_store_n(&g_split_mode, mode, __ATOMIC_RELAXED);
    return 0;
}
