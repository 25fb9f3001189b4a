// Error state + version of the C-ABI library (see include/gm_amd.h).
#include <stdio.h>
#include <string.h>
static thread_local char g_err[512] = "";
extern "C" void gm_set_error(const char* where, int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s: error %d: %s", where ? where : "?", code, msg ? msg : "");
}
extern "C" const char* gm_last_error(void) { return g_err; }
extern "C" int gm_abi_version(void) { return 1; }
