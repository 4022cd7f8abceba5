// error.cpp -- thread-local error string + ABI version of libadvgrpo_hip.so
#include <stdarg.h>

#include "common.hpp"

namespace advgrpo {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace advgrpo

extern "C" int advgrpo_abi_version(void) { return ADVGRPO_ABI_VERSION; }
extern "C" const char* advgrpo_last_error(void) { return advgrpo::g_err; }
