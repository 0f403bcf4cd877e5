#include "common.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

void da_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {
const char* da_last_error(void) { return g_err; }
int da_abi_version(void) { return 1; }
}
