// Error text and version of libgcn_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/gcn_hip.h"

namespace gcn {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int fail(gcn_status code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return (int)code;
}

}  // namespace gcn

extern "C" {

int gcn_version(void) { return 100; }

// sha256 of the sources this library was built from (written by the Makefile)
const char* gcn_build_hash(void) {
    return
#include "build_hash.inc"
        ;
}

const char* gcn_last_error_string(void) { return gcn::g_error; }

}  // extern "C"
