// Error channel and version of libyolo3hip.so.
#include "common.h"

static thread_local char g_err[512] = "";

void y3_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* y3_last_error(void) { return g_err; }

// host twin of y3_div (common.h): same magic numbers, the multiply-high written out in 64 bits
extern "C" int y3_debug_div(int x, int d) {
    const Y3Div m = y3_make_div(d);
    return m.mul ? (int)((unsigned)(((unsigned long long)(unsigned)x * m.mul) >> 32) >> m.shift) : x;
}
extern "C" int y3_version(void) { return 1; }
