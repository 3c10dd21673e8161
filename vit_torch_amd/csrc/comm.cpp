// libvitmi_comm.so: bucketed gradient all-reduce on an OWN RCCL communicator (include/vitmi_comm.h).
// Host only.  RCCL is bound at run time so that the process keeps ONE librccl (PyTorch ships its own copy).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/vitmi_comm.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;

struct Comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;     // the exchange runs here, beside the caller's compute stream
  hipEvent_t fork = nullptr;        // compute stream -> comm stream
  hipEvent_t join = nullptr;        // comm stream -> compute stream
  int world = 0, rank = 0, device = 0;
  bool pending = false;             // an async exchange has been enqueued since the last join
};

int nccl_fail(ncclResult_t r, const char* what) {
  return fail(10000 + (int)r, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}
int hip_fail(hipError_t e, const char* what) { return fail((int)e, "%s: %s", what, hipGetErrorString(e)); }

#define NEED_RCCL() \
  if (!g_rccl.handle) return fail(-2, "vitmi_comm: RCCL is not bound (call vitmi_comm_load first)")
#define NEED_COMM(c) \
  NEED_RCCL();       \
  if (!(c)) return fail(-1, "vitmi_comm: null communicator")
#define HIPC(call, what)                                   \
  do {                                                     \
    hipError_t e_ = (call);                                \
    if (e_ != hipSuccess) return hip_fail(e_, what);       \
  } while (0)
#define NCCLC(call, what)                                  \
  do {                                                     \
    ncclResult_t r_ = (call);                              \
    if (r_ != ncclSuccess) return nccl_fail(r_, what);     \
  } while (0)

template <typename F>
bool bind(void* h, const char* name, F& fn) {
  fn = reinterpret_cast<F>(dlsym(h, name));
  return fn != nullptr;
}

}  // namespace

extern "C" int vitmi_comm_version(void) { return VITMI_COMM_VERSION; }
extern "C" const char* vitmi_comm_last_error(void) { return g_err.c_str(); }

extern "C" int vitmi_comm_load(const char* rccl_path) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rccl.handle) return 0;
  const char* path = (rccl_path && rccl_path[0]) ? rccl_path : "librccl.so.1";
  void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(-3, "vitmi_comm_load: dlopen(%s) failed: %s", path, dlerror());
  Rccl r;
  const bool ok = bind(h, "ncclGetVersion", r.GetVersion) && bind(h, "ncclGetUniqueId", r.GetUniqueId) &&
                  bind(h, "ncclCommInitRank", r.CommInitRank) && bind(h, "ncclCommDestroy", r.CommDestroy) &&
                  bind(h, "ncclCommCount", r.CommCount) && bind(h, "ncclCommUserRank", r.CommUserRank) &&
                  bind(h, "ncclCommCuDevice", r.CommCuDevice) && bind(h, "ncclAllReduce", r.AllReduce) &&
                  bind(h, "ncclBroadcast", r.Broadcast) && bind(h, "ncclGetErrorString", r.GetErrorString);
  if (!ok) {
    dlclose(h);
    return fail(-3, "vitmi_comm_load: %s does not export the RCCL entry points", path);
  }
  r.handle = h;
  g_rccl = r;
  return 0;
}

extern "C" int vitmi_comm_rccl_version(int* version) {
  NEED_RCCL();
  if (!version) return fail(-1, "vitmi_comm_rccl_version: null argument");
  NCCLC(g_rccl.GetVersion(version), "ncclGetVersion");
  return 0;
}

extern "C" int vitmi_comm_unique_id(void* out128) {
  NEED_RCCL();
  if (!out128) return fail(-1, "vitmi_comm_unique_id: null argument");
  static_assert(sizeof(ncclUniqueId) == VITMI_COMM_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  NCCLC(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(out128, &id, sizeof(id));
  return 0;
}

extern "C" int vitmi_comm_init(const void* unique_id128, int world, int rank, int device, void** comm_out) {
  NEED_RCCL();
  if (!unique_id128 || !comm_out || world < 1 || rank < 0 || rank >= world || device < 0)
    return fail(-1, "vitmi_comm_init: bad argument (world %d, rank %d, device %d)", world, rank, device);
  HIPC(hipSetDevice(device), "hipSetDevice");
  ncclUniqueId id;
  memcpy(&id, unique_id128, sizeof(id));
  Comm* c = new Comm;
  c->world = world; c->rank = rank; c->device = device;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { delete c; return nccl_fail(r, "ncclCommInitRank"); }
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join, hipEventDisableTiming);
  if (e != hipSuccess) {
    if (c->join) (void)hipEventDestroy(c->join);
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    g_rccl.CommDestroy(c->comm);
    delete c;
    return hip_fail(e, "vitmi_comm_init: stream / event creation");
  }
  *comm_out = c;
  return 0;
}

extern "C" int vitmi_comm_info(void* comm, int* world, int* rank, int* device) {
  NEED_COMM(comm);
  Comm* c = static_cast<Comm*>(comm);
  int w = 0, r = 0, d = 0;
  NCCLC(g_rccl.CommCount(c->comm, &w), "ncclCommCount");
  NCCLC(g_rccl.CommUserRank(c->comm, &r), "ncclCommUserRank");
  NCCLC(g_rccl.CommCuDevice(c->comm, &d), "ncclCommCuDevice");
  if (world) *world = w;
  if (rank) *rank = r;
  if (device) *device = d;
  return 0;
}

extern "C" int vitmi_comm_allreduce_sum_f32_async(void* comm, float* buf, int64_t count, void* compute_stream) {
  NEED_COMM(comm);
  if (!buf || count <= 0) return fail(-1, "vitmi_comm_allreduce_sum_f32_async: bad buffer");
  Comm* c = static_cast<Comm*>(comm);
  hipStream_t cs = reinterpret_cast<hipStream_t>(compute_stream);
  // the bucket's gradients are final once everything queued on the compute stream so far has run
  HIPC(hipEventRecord(c->fork, cs), "hipEventRecord(fork)");
  HIPC(hipStreamWaitEvent(c->stream, c->fork, 0), "hipStreamWaitEvent(comm stream)");
  NCCLC(g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, c->comm, c->stream), "ncclAllReduce");
  c->pending = true;
  return 0;
}

extern "C" int vitmi_comm_join(void* comm, void* compute_stream) {
  NEED_COMM(comm);
  Comm* c = static_cast<Comm*>(comm);
  if (!c->pending) return 0;
  hipStream_t cs = reinterpret_cast<hipStream_t>(compute_stream);
  HIPC(hipEventRecord(c->join, c->stream), "hipEventRecord(join)");
  HIPC(hipStreamWaitEvent(cs, c->join, 0), "hipStreamWaitEvent(compute stream)");
  c->pending = false;
  return 0;
}

extern "C" int vitmi_comm_broadcast_f32(void* comm, float* buf, int64_t count, int root, void* stream) {
  NEED_COMM(comm);
  Comm* c = static_cast<Comm*>(comm);
  if (!buf || count <= 0 || root < 0 || root >= c->world) return fail(-1, "vitmi_comm_broadcast_f32: bad argument");
  NCCLC(g_rccl.Broadcast(buf, buf, (size_t)count, ncclFloat, root, c->comm, reinterpret_cast<hipStream_t>(stream)), "ncclBroadcast");
  return 0;
}

extern "C" int vitmi_comm_allreduce_sum_f32(void* comm, float* buf, int64_t count, void* stream) {
  NEED_COMM(comm);
  if (!buf || count <= 0) return fail(-1, "vitmi_comm_allreduce_sum_f32: bad buffer");
  Comm* c = static_cast<Comm*>(comm);
  NCCLC(g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, c->comm, reinterpret_cast<hipStream_t>(stream)), "ncclAllReduce");
  return 0;
}

extern "C" int vitmi_comm_destroy(void* comm) {
  NEED_COMM(comm);
  Comm* c = static_cast<Comm*>(comm);
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  ncclResult_t r = g_rccl.CommDestroy(c->comm);
  (void)hipEventDestroy(c->join);
  (void)hipEventDestroy(c->fork);
  (void)hipStreamDestroy(c->stream);
  delete c;
  if (r != ncclSuccess) return nccl_fail(r, "ncclCommDestroy");
  return 0;
}
