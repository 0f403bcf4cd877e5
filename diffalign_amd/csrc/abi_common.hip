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
// 3 (round 3): da_init added, the two-pass output stage (da_edge_out_g / da_edge_out_finish) removed; 2 (round 2, not bumped at the time):
// da_y_film / da_y_update* take the y-chain weights k-major ((in, out)), da_edge_layer_stats resets its ticket word itself
int da_abi_version(void) { return 3; }
}
