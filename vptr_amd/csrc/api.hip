// Error reporting and ABI version for libvptr_hip.so.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void vptr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vptr_last_error(void) { return g_err; }
extern "C" int vptr_abi_version(void) { return 10; }

int g_vptr_deterministic = 0;
// process-wide switch (host-side launch decisions only; set it before the launches it should affect are enqueued): returns the previous value
extern "C" int vptr_set_deterministic(int on) {
  const int prev = g_vptr_deterministic;
  g_vptr_deterministic = on ? 1 : 0;
  return prev;
}
extern "C" int vptr_get_deterministic(void) { return g_vptr_deterministic; }
