// Shared device/host helpers for libvitmi (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/vitmi.h"

typedef __bf16 bf16;
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// ---- host-side error plumbing ------------------------------------------
void vitmi_set_error(const std::string& s);
int vitmi_fail(int code, const char* fmt, ...);
int vitmi_check_launch(const char* what);
// Per-device facts, filled once per device under a lock (the library's only process-wide
// state besides the vitmi_debug_* switches): CU count of the current device, and the
// dynamic-LDS limit of a kernel raised once per (kernel, device).
int vitmi_cu_count();
// persistent grids (one workgroup per CU walking a fixed list of tiles / (image, head) pairs) unless the call
// carries VITMI_LAUNCH_SHARED_DEVICE (other kernels, e.g. RCCL's, hold CUs); vitmi_debug_gemm_persist overrides
int vitmi_persist_on(int launch_flags);
int vitmi_raise_dynamic_lds(const void* kern, int bytes, const char* who);
// out[c] = sum_{r<S} part[r*ld + c], c < N (elementwise.hip)
int vitmi_reduce_rows(const float* part, int S, int64_t N, int64_t ld, float* out, hipStream_t stream);
int vitmi_reduce_rows_segs(const float* part, int S, int64_t ld, float* const out[4], const int width[4],
                           hipStream_t stream);   // four consecutive column segments, four destinations
int vitmi_reduce_rows3(const float* part, int S, int64_t N, int64_t ld, float* out0, float* out1, float* out2,
                       hipStream_t stream);   // out2 may be null

#define VITMI_REQUIRE(cond, code, ...)                  \
  do {                                                  \
    if (!(cond)) return vitmi_fail((code), __VA_ARGS__); \
  } while (0)

static inline bool is_aligned(const void* p, size_t a) {
  return (reinterpret_cast<uintptr_t>(p) % a) == 0;
}
static inline size_t dtype_size(int dt) { return dt == VITMI_BF16 ? 2 : 4; }

// ---- device helpers ----------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// The same reductions on the DPP network (no LDS crossbar): quad swaps, the two row mirrors,
// then row_bcast:15 / row_bcast:31 carry the row totals up to lane 63, which v_readlane
// hands to every lane as a scalar.  Six dependent VALU instructions instead of six
// ds_bpermute round trips (~100 cycles each): for kernels that reduce many short rows.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_take<0xB1, 0xf>(0.f, v);        // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xf>(0.f, v);        // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xf>(0.f, v);       // row_half_mirror
  v += dpp_take<0x140, 0xf>(0.f, v);       // row_mirror: every lane holds its row's total
  v += dpp_take<0x142, 0xa>(0.f, v);       // row_bcast:15 into rows 1 and 3
  v += dpp_take<0x143, 0xc>(0.f, v);       // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_take<0xB1, 0xf>(-INFINITY, v));
  v = fmaxf(v, dpp_take<0x4E, 0xf>(-INFINITY, v));
  v = fmaxf(v, dpp_take<0x141, 0xf>(-INFINITY, v));
  v = fmaxf(v, dpp_take<0x140, 0xf>(-INFINITY, v));
  v = fmaxf(v, dpp_take<0x142, 0xa>(-INFINITY, v));
  v = fmaxf(v, dpp_take<0x143, 0xc>(-INFINITY, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// exact (erf) GELU and its derivative, as nn.GELU() default
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float dgelu_erf(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// 4-element vector load/store of T as fp32 (T=float: 16 B, T=bf16: 8 B)
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}
template <> __device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) {
  *reinterpret_cast<f32x4*>(p) = v;
}
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, f32x4 v) {
  bf16x4 r = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *reinterpret_cast<bf16x4*>(p) = r;
}

// MFMA C/D layout of the 32x32 shapes: register r of lane l holds
// row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31   (guide §3)
__device__ __forceinline__ int mfma32_row(int reg, int lane_hi) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * lane_hi;
}
