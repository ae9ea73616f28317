#include <stdarg.h>
#include <stdio.h>
#include "rrnet_hip.h"

static thread_local char g_err[512] = "";

void rr_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *rr_last_error(void) { return g_err; }
extern "C" int rr_abi_version(void) { return RR_ABI_VERSION; }
