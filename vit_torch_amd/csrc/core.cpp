// Error plumbing + version for libvitmi (host only).
#include <cstdarg>
#include <cstdio>
#include <map>
#include <atomic>
#include <mutex>
#include <set>
#include <utility>
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

namespace {
std::mutex g_dev_mu;
std::map<int, int> g_cus;                              // device -> CU count
std::set<std::pair<const void*, int>> g_lds_raised;    // (kernel, device)
int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  return dev;
}
}  // namespace

int vitmi_cu_count() {
  const int dev = current_device();
  std::lock_guard<std::mutex> lk(g_dev_mu);
  auto it = g_cus.find(dev);
  if (it != g_cus.end()) return it->second;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  g_cus[dev] = cus;
  return cus;
}

// test / profiling hook (not part of the ABI): -1 = follow each call's launch_flags (default), 0 = force
// one tile / pair per workgroup, 1 = force the persistent grids
static std::atomic<int> g_persist_override{-1};
void vitmi_debug_reset_attention();
void vitmi_debug_reset_attention_f32();
void vitmi_debug_reset_cait();
void vitmi_debug_reset_cait_fused();
void vitmi_debug_reset_gemm();
void vitmi_debug_reset_gemm_fast();
void vitmi_debug_reset_gemm_fast2();
void vitmi_debug_reset_gemm_small();
void vitmi_debug_reset_layernorm();
void vitmi_debug_reset_swin();
// all vitmi_debug_* switches back to their defaults.  They are process-wide std::atomic<int>s read at launch time
// (diagnostics and A/B hooks, not part of include/vitmi.h); tests/conftest.py calls this after EVERY test, so a test that
// fails between a set and its reset cannot leave the next one on another kernel
extern "C" void vitmi_debug_reset(void) {
  g_persist_override.store(-1, std::memory_order_relaxed);
  vitmi_debug_reset_attention();
  vitmi_debug_reset_attention_f32();
  vitmi_debug_reset_cait();
  vitmi_debug_reset_cait_fused();
  vitmi_debug_reset_gemm();
  vitmi_debug_reset_gemm_fast();
  vitmi_debug_reset_gemm_fast2();
  vitmi_debug_reset_gemm_small();
  vitmi_debug_reset_layernorm();
  vitmi_debug_reset_swin();
}
extern "C" void vitmi_debug_gemm_persist(int on) { g_persist_override.store(on < 0 ? -1 : (on != 0), std::memory_order_relaxed); }
int vitmi_persist_on(int launch_flags) {
  const int o = g_persist_override.load(std::memory_order_relaxed);
  if (o >= 0) return o;
  return (launch_flags & VITMI_LAUNCH_SHARED_DEVICE) ? 0 : 1;
}

int vitmi_raise_dynamic_lds(const void* kern, int bytes, const char* who) {
  const int dev = current_device();
  std::lock_guard<std::mutex> lk(g_dev_mu);
  const auto key = std::make_pair(kern, dev);
  if (g_lds_raised.count(key)) return 0;
  hipError_t err = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (err != hipSuccess) return vitmi_fail((int)err, "%s: cannot raise dynamic LDS to %d: %s", who, bytes, hipGetErrorString(err));
  g_lds_raised.insert(key);
  return 0;
}

extern "C" int vitmi_version(void) { return VITMI_VERSION; }
extern "C" const char* vitmi_last_error_string(void) { return g_last_error.c_str(); }
