// Library-level entry points of libskg.so: ABI version and last-error text.
#include "common.h"
#include <stdio.h>

static thread_local char g_err[256] = "";

void skg_set_error(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

extern "C" int skg_abi_version(void) { return SKG_ABI_VERSION; }
extern "C" const char* skg_last_error(void) { return g_err; }
