// Error reporting and small queries of the C-ABI (include/tracevae_hip.h).
#include <stdarg.h>
#include <string.h>

#include "tv_common.h"

static thread_local char g_error[512] = "";

void tv_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

extern "C" const char *tv_last_error(void) { return g_error; }

extern "C" int tv_version(void) { return 1; }

extern "C" int64_t tv_stat_blocks(int64_t rows) { return tv_stat_blocks_host(rows); }
