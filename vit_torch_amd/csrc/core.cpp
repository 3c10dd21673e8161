// Error plumbing + version for libvitmi (host only).
#include <cstdarg>
#include <cstdio>
#include "common.h"

static thread_local std::string g_last_error;

void vitmi_set_error(const std::string& s) { g_last_error = s; }

int vitmi_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

int vitmi_check_launch(const char* what) {
  hipError_t err = hipGetLastError();
  if (err == hipSuccess) return 0;
  g_last_error = std::string(what) + ": " + hipGetErrorString(err);
  return (int)err;
}

extern "C" int vitmi_version(void) { return VITMI_VERSION; }
extern "C" const char* vitmi_last_error_string(void) { return g_last_error.c_str(); }
