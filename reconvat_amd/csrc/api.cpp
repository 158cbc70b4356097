// Error text for the C ABI (include/reconvat_hip.h): thread-local last-error string; ABI version; digest of the sources this
// library was built from (reconvat_amd/build.py passes it in: a stale prebuilt .so cannot pass for the sources next to it).
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" void rv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* rv_last_error(void) { return g_err; }

extern "C" int rv_abi_version(void) { return 1; }

#ifndef RV_SOURCE_DIGEST
#define RV_SOURCE_DIGEST "unknown"
#endif
extern "C" const char* rv_source_digest(void) { return RV_SOURCE_DIGEST; }
