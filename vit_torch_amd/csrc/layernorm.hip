// LayerNorm forward / backward: one wavefront per row, row held in registers,
// fp32 statistics, 16-B (fp32) / 8-B (bf16) vector accesses.  HBM-bound:
// fwd moves (sizeof(X)+sizeof(Y))*D bytes per row, bwd
// (sizeof(dY)+sizeof(X)+2*sizeof(G)+sizeof(Gb))*D.
#include <atomic>
#include <initializer_list>
#include "common.h"

namespace {

constexpr int LN_MAXV = 8;   // float4 chunks per lane -> D <= 8*64*4 = 2048

// NV = float4 chunks per lane actually instantiated (D <= NV*256): the row lives in
// registers, so a D = 768 row must not pay for the 2048-wide case's registers (occupancy
// is what hides the HBM latency here).
template <int NV, typename TX, typename TY>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, int64_t xs,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     TY* __restrict__ y, int64_t ys,
                                                     float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out,
                                                     int64_t M, int D, float eps) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < M; row += (int64_t)gridDim.x * 4) {
  const TX* xr = x + row * xs;
  f32x4 v[NV];
  float s = 0.f;
  // loads are unconditional from a clamped column (a branch around a load makes hipcc
  // drain vmcnt after each: one HBM round trip per chunk instead of one per row)
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = load4<TX>(xr + min((i * 64 + lane) * 4, D - 4));
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
    }
  }
  const float var = wave_sum(q) / (float)D;
  const float rstd = rsqrtf(var + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
  TY* yr = y + row * ys;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const int cc = min(c, D - 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + cc);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + cc);
    if (c < D) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      store4<TY>(yr + c, o);
    }
  }
  }
}

// Backward.  Each wave walks rows row0, row0+stride, ...; dgamma/dbeta are kept
// per lane in registers, reduced over the block's 4 waves through LDS and
// written as one partial row per block: part[blockIdx][0..D) = dgamma, [D..2D) = dbeta,
// [2D..3D) = column sum of g_out (the bias gradient of the Linear that produced this
// LayerNorm's input: it costs three adds per element here instead of a separate pass over
// g_out).  reduce_rows sums the partial rows.
template <int NV, typename TDY, typename TX, typename TG, typename TGB>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dy, int64_t dys,
                                                     const TX* __restrict__ x, int64_t xs,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma,
                                                     const TG* g_in, TG* g_out, int64_t gstride,
                                                     TGB* __restrict__ gb_out, int64_t gbs,
                                                     float* __restrict__ part, int want_gsum,
                                                     const float* __restrict__ gb_scale,
                                                     const float* __restrict__ gb_rowscale, int64_t rpg,
                                                     int64_t M, int D) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4][D]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4 gam[NV], dg[NV], db[NV], gs[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    dg[i] = z; db[i] = z; gam[i] = z; gs[i] = z;
    if (c < D) gam[i] = *reinterpret_cast<const f32x4*>(gamma + c);
  }
  const float invD = 1.f / (float)D;
  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < M; row += (int64_t)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    const TDY* dyr = dy + row * dys;
    const TX* xr = x + row * xs;
    f32x4 xh[NV], dv[NV];
    float s1 = 0.f, s2 = 0.f;
    // unconditional loads from clamped columns first (see ln_fwd_kernel)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int cc = min((i * 64 + lane) * 4, D - 4);
      xh[i] = load4<TX>(xr + cc);
      dv[i] = load4<TDY>(dyr + cc);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        const f32x4 xv = xh[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[i][j] = (xv[j] - mu) * rs;
          dg[i][j] += dv[i][j] * xh[i][j];
          db[i][j] += dv[i][j];
          dv[i][j] *= gam[i][j];          // dy * gamma
          s1 += dv[i][j];
          s2 += dv[i][j] * xh[i][j];
        }
      }
    }
    const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
    float rsc = 1.f;
    if (gb_rowscale) rsc = gb_rowscale[(uint32_t)row / (uint32_t)rpg];   // wave-uniform branch; M < 2^32
    f32x4 gin[NV];
    if (g_in) {                                    // wave-uniform; loads again unconditional
#pragma unroll
      for (int i = 0; i < NV; ++i) gin[i] = load4<TG>(g_in + row * gstride + min((i * 64 + lane) * 4, D - 4));
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rs * (dv[i][j] - c1 - xh[i][j] * c2);
        if (g_in) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += gin[i][j];
        }
        store4<TG>(g_out + row * gstride + c, o);
        // the GEMM-operand copy (and its column sum) carry the LayerScale of the branch
        // that will consume them: d(branch out) = g_out * gamma_branch
        // (gb_scale is re-read per row, from L1: holding it would cost NV*4 registers
        // and a wave of occupancy on a kernel that lives on memory-level parallelism)
        if (gb_scale) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(gb_scale + c);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] *= sc[j];
        }
        if (gb_rowscale) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] *= rsc;
        }
        if (gb_out) store4<TGB>(gb_out + row * gbs + c, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) gs[i][j] += o[j];
      }
    }
  }
  // block reduction of dgamma then dbeta
  float* prow = part + (int64_t)blockIdx.x * 3 * D;
#pragma unroll 1
  for (int pass = 0; pass < (want_gsum ? 3 : 2); ++pass) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) *reinterpret_cast<f32x4*>(red + w * D + c) = pass == 0 ? dg[i] : (pass == 1 ? db[i] : gs[i]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256)
      prow[pass * D + c] = (red[c] + red[D + c]) + (red[2 * D + c] + red[3 * D + c]);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// 8-element form (round 3): each lane owns chunks of 8 consecutive elements (16 B of bf16, 2 x 16 B of
// fp32) and a row is spread over LPR = 16 / 32 / 64 lanes, so a wave normalises 64 / LPR rows at once.
// Why: with 4 elements per lane the bf16 rows of the bf16 residual stream moved as 8-B accesses (ln_bwd
// 4.3 TB/s against 5.7 with the fp32 stream's 16-B accesses), and narrow rows left lanes idle (Swin's
// C = 96: 24 of 64 lanes; now 12 of every 16).  Same arithmetic and reduction order within a row group
// (two-pass mean / variance, fp32), reductions by xor-shuffles that stay inside the LPR lanes of a row.
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&o)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float (&o)[8]) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
  *reinterpret_cast<bf16x8*>(p) = o;
}
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int LPR, int NV, typename TX, typename TY>
__global__ __launch_bounds__(256) void ln_fwd8_kernel(const TX* __restrict__ x, int64_t xs, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, TY* __restrict__ y, int64_t ys,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                      int64_t M, int D, float eps) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int64_t groups = (M + RPW - 1) / RPW;
  const float invD = 1.f / (float)D;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + w; grp < groups; grp += (int64_t)gridDim.x * 4) {
    const int64_t row = grp * RPW + sub;
    const bool live = row < M;
    const TX* xr = x + (live ? row : M - 1) * xs;
    float v[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) load8<TX>(xr + min((i * LPR + l) * 8, D - 8), v[i]);     // unconditional, clamped
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if ((i * LPR + l) * 8 < D) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[i][j];
      }
    const float mean = group_sum<LPR>(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if ((i * LPR + l) * 8 < D) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
      }
    const float rstd = rsqrtf(group_sum<LPR>(q) * invD + eps);
    if (l == 0 && live) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
    TY* yr = y + row * ys;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * LPR + l) * 8;
      float g[8], b[8];
      load8<float>(gamma + min(c, D - 8), g);
      load8<float>(beta + min(c, D - 8), b);
      if (c < D && live) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
        store8<TY>(yr + c, o);
      }
    }
  }
}

// backward, same layout; part[blockIdx][0..D) = dgamma, [D..2D) = dbeta, [2D..3D) = column sum of gb (see ln_bwd_kernel)
template <int LPR, int NV, typename TDY, typename TX, typename TG, typename TGB>
__global__ __launch_bounds__(256) void ln_bwd8_kernel(const TDY* __restrict__ dy, int64_t dys, const TX* __restrict__ x, int64_t xs,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const TG* g_in, TG* g_out, int64_t gstride,
                                                      TGB* __restrict__ gb_out, int64_t gbs, float* __restrict__ part, int want_gsum,
                                                      const float* __restrict__ gb_scale, const float* __restrict__ gb_rowscale,
                                                      int64_t rpg, int64_t M, int D) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4 * RPW][D]
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  float gam[NV][8], dg[NV][8], db[NV][8], gs[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    load8<float>(gamma + min((i * LPR + l) * 8, D - 8), gam[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; gs[i][j] = 0.f; }
  }
  const float invD = 1.f / (float)D;
  const int64_t groups = (M + RPW - 1) / RPW;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + w; grp < groups; grp += (int64_t)gridDim.x * 4) {
    const int64_t row = grp * RPW + sub;
    const bool live = row < M;
    const int64_t r = live ? row : M - 1;
    const float mu = mean[r], rs = rstd[r];
    float xh[NV][8], dv[NV][8], gin[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {                    // every load of the row group in one batch (unconditional, clamped)
      const int cc = min((i * LPR + l) * 8, D - 8);
      load8<TX>(x + r * xs + cc, xh[i]);
      load8<TDY>(dy + r * dys + cc, dv[i]);
    }
    if (g_in) {                                       // wave-uniform
#pragma unroll
      for (int i = 0; i < NV; ++i) load8<TG>(g_in + r * gstride + min((i * LPR + l) * 8, D - 8), gin[i]);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if ((i * LPR + l) * 8 < D && live) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[i][j] = (xh[i][j] - mu) * rs;
          dg[i][j] += dv[i][j] * xh[i][j];
          db[i][j] += dv[i][j];
          dv[i][j] *= gam[i][j];
          s1 += dv[i][j];
          s2 += dv[i][j] * xh[i][j];
        }
      }
    const float c1 = group_sum<LPR>(s1) * invD, c2 = group_sum<LPR>(s2) * invD;
    float rsc = 1.f;
    if (gb_rowscale) rsc = gb_rowscale[(uint32_t)r / (uint32_t)rpg];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * LPR + l) * 8;
      if (c < D && live) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs * (dv[i][j] - c1 - xh[i][j] * c2);
        if (g_in) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += gin[i][j];
        }
        store8<TG>(g_out + row * gstride + c, o);
        if (gb_scale) {
          float sc[8];
          load8<float>(gb_scale + c, sc);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] *= sc[j];
        }
        if (gb_rowscale) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] *= rsc;
        }
        if (gb_out) store8<TGB>(gb_out + row * gbs + c, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) gs[i][j] += o[j];
      }
    }
  }
  // block reduction over the 4 * RPW row slots, one quantity at a time
  float* prow = part + (int64_t)blockIdx.x * 3 * D;
  const int slot = w * RPW + sub;
#pragma unroll 1
  for (int pass = 0; pass < (want_gsum ? 3 : 2); ++pass) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * LPR + l) * 8;
      float t[8];                                    // selected per element: a reference chosen at run time would send the arrays to scratch
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = pass == 0 ? dg[i][j] : (pass == 1 ? db[i][j] : gs[i][j]);
      if (c < D) store8<float>(red + slot * D + c, t);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 4 * RPW; ++t) a += red[t * D + c];
      prow[pass * D + c] = a;
    }
    __syncthreads();
  }
}

// (LPR, NV) of the 8-element form for a row of D elements (D % 8 == 0): the narrowest lane group whose NV <= 4 chunks cover it
struct Ln8Cfg { int lpr, nv; };
inline Ln8Cfg ln8_cfg(int64_t D) {
  if (D <= 128) return {16, 1};
  if (D <= 256) return {16, 2};
  if (D <= 384) return {16, 3};
  if (D <= 512) return {32, 2};
  if (D <= 768) return {32, 3};
  if (D <= 1024) return {64, 2};
  if (D <= 1536) return {64, 3};
  return {64, 4};
}
static std::atomic<int> g_ln8{1};     // diagnostic hook (vitmi_debug_ln8): 0 = the 4-element kernels everywhere

inline int ln_nv(int64_t D) { return D <= 512 ? 2 : D <= 768 ? 3 : D <= 1024 ? 4 : LN_MAXV; }
// grid-stride blocks of the backward kernel: 4 per CU while the row fits few registers
inline int ln_bwd_blocks(int64_t M, int64_t D) {
  // resident blocks per CU by the kernels' register counts (102 / 132 / 162 / 256+ VGPRs)
  const int nv = ln_nv(D);
  const int64_t cap = nv == 2 ? 1024 : nv <= 4 ? 768 : 512;
  int64_t b = (M + 3) / 4;
  return (int)(b < cap ? b : cap);
}

inline bool ln8_ok(int64_t D, std::initializer_list<int64_t> strides, std::initializer_list<const void*> ptrs) {
  if (!g_ln8 || D % 8 != 0 || D < 8) return false;
  for (int64_t s : strides) if (s % 8 != 0) return false;
  for (const void* p : ptrs) if (p && !is_aligned(p, 16)) return false;
  return true;
}
inline int ln_bwd8_blocks(int64_t M, int64_t D) {
  const Ln8Cfg c = ln8_cfg(D);
  const int64_t groups = (M + 64 / c.lpr - 1) / (64 / c.lpr);
  // resident 4-wave blocks on 256 CUs by the kernels' VGPR counts (100 / 156 / 214 / 256: 4 / 3 / 2 / 1 waves per SIMD)
  const int64_t cap = c.nv == 1 ? 1024 : c.nv == 2 ? 768 : c.nv == 3 ? 512 : 256;
  const int64_t b = (groups + 3) / 4;
  return (int)(b < cap ? b : cap);
}

}  // namespace

extern "C" void vitmi_debug_ln8(int on) { g_ln8 = on != 0; }

static int check_ln_common(const void* x, int x_dtype, int64_t x_stride, int64_t M, int64_t D, const char* who) {
  VITMI_REQUIRE(x && M > 0 && D > 0, VITMI_E_BADARG, "%s: null input or empty shape", who);
  VITMI_REQUIRE(D % 4 == 0 && D <= LN_MAXV * 256, VITMI_E_SHAPE, "%s: D=%lld must be a multiple of 4 and <= %d", who, (long long)D, LN_MAXV * 256);
  VITMI_REQUIRE(x_stride % 4 == 0 && is_aligned(x, 4 * dtype_size(x_dtype)), VITMI_E_ALIGN, "%s: rows must be 4-element aligned", who);
  return 0;
}

extern "C" int vitmi_layernorm_fwd(const void* x, int x_dtype, int64_t x_stride, const float* gamma,
                                   const float* beta, void* y, int y_dtype, int64_t y_stride,
                                   float* mean, float* rstd, int64_t M, int64_t D, float eps,
                                   void* stream_) {
  int rc = check_ln_common(x, x_dtype, x_stride, M, D, "layernorm_fwd");
  if (rc) return rc;
  VITMI_REQUIRE(y && gamma && beta, VITMI_E_BADARG, "layernorm_fwd: null y/gamma/beta");
  VITMI_REQUIRE(y_stride % 4 == 0 && is_aligned(y, 4 * dtype_size(y_dtype)) && is_aligned(gamma, 16) && is_aligned(beta, 16),
                VITMI_E_ALIGN, "layernorm_fwd: y/gamma/beta alignment");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (ln8_ok(D, {x_stride, y_stride}, {x, y, gamma, beta})) {
    const Ln8Cfg c = ln8_cfg(D);
    const int rpw = 64 / c.lpr;
    int64_t nb = ((M + rpw - 1) / rpw + 3) / 4;
    if (nb > 2048) nb = 2048;
#define LN_FWD8_T(L, N, TX, TY)                                                                                  \
    hipLaunchKernelGGL((ln_fwd8_kernel<L, N, TX, TY>), dim3((unsigned)nb), dim3(256), 0, stream, (const TX*)x, x_stride, \
                       gamma, beta, (TY*)y, y_stride, mean, rstd, M, (int)D, eps)
#define LN_FWD8(TX, TY)                                                                  \
    do {                                                                                  \
      switch (c.lpr * 8 + c.nv) {                                                         \
        case 16 * 8 + 1: LN_FWD8_T(16, 1, TX, TY); break;                                 \
        case 16 * 8 + 2: LN_FWD8_T(16, 2, TX, TY); break;                                 \
        case 16 * 8 + 3: LN_FWD8_T(16, 3, TX, TY); break;                                 \
        case 32 * 8 + 2: LN_FWD8_T(32, 2, TX, TY); break;                                 \
        case 32 * 8 + 3: LN_FWD8_T(32, 3, TX, TY); break;                                 \
        case 64 * 8 + 2: LN_FWD8_T(64, 2, TX, TY); break;                                 \
        case 64 * 8 + 3: LN_FWD8_T(64, 3, TX, TY); break;                                 \
        default: LN_FWD8_T(64, 4, TX, TY); break;                                         \
      }                                                                                   \
    } while (0)
    if (x_dtype == VITMI_F32 && y_dtype == VITMI_F32) LN_FWD8(float, float);
    else if (x_dtype == VITMI_F32 && y_dtype == VITMI_BF16) LN_FWD8(float, bf16);
    else if (x_dtype == VITMI_BF16 && y_dtype == VITMI_BF16) LN_FWD8(bf16, bf16);
    else if (x_dtype == VITMI_BF16 && y_dtype == VITMI_F32) LN_FWD8(bf16, float);
    else return vitmi_fail(VITMI_E_DTYPE, "layernorm_fwd: dtype combination");
#undef LN_FWD8
#undef LN_FWD8_T
    return vitmi_check_launch("ln_fwd8_kernel");
  }
  // grid-stride over rows: 8 blocks per CU keep the loads of several rows in flight per
  // SIMD without paying a workgroup launch per 4 rows
  int64_t nblk = (M + 3) / 4;
  if (nblk > 2048) nblk = 2048;
  dim3 grid((unsigned)nblk), block(256);
#define LN_FWD_NV(NVV, TX, TY)                                                                     \
  hipLaunchKernelGGL((ln_fwd_kernel<NVV, TX, TY>), grid, block, 0, stream, (const TX*)x, x_stride, \
                     gamma, beta, (TY*)y, y_stride, mean, rstd, M, (int)D, eps)
#define LN_FWD(TX, TY)                                                                        \
  do {                                                                                        \
    switch (ln_nv(D)) {                                                                       \
      case 2: LN_FWD_NV(2, TX, TY); break;                                                    \
      case 3: LN_FWD_NV(3, TX, TY); break;                                                    \
      case 4: LN_FWD_NV(4, TX, TY); break;                                                    \
      default: LN_FWD_NV(LN_MAXV, TX, TY); break;                                             \
    }                                                                                         \
  } while (0)
  if (x_dtype == VITMI_F32 && y_dtype == VITMI_F32) LN_FWD(float, float);
  else if (x_dtype == VITMI_F32 && y_dtype == VITMI_BF16) LN_FWD(float, bf16);
  else if (x_dtype == VITMI_BF16 && y_dtype == VITMI_BF16) LN_FWD(bf16, bf16);
  else if (x_dtype == VITMI_BF16 && y_dtype == VITMI_F32) LN_FWD(bf16, float);
  else return vitmi_fail(VITMI_E_DTYPE, "layernorm_fwd: dtype combination");
#undef LN_FWD
#undef LN_FWD_NV
  return vitmi_check_launch("ln_fwd_kernel");
}

extern "C" size_t vitmi_layernorm_bwd_workspace(int64_t M, int64_t D) {
  int nb = ln_bwd_blocks(M, D);
  if (D % 8 == 0 && D >= 8) nb = nb > ln_bwd8_blocks(M, D) ? nb : ln_bwd8_blocks(M, D);   // whichever form the call takes
  return (size_t)nb * 3 * (size_t)D * sizeof(float);
}

extern "C" int vitmi_layernorm_bwd(const void* dy, int dy_dtype, int64_t dy_stride, const void* x,
                                   int x_dtype, int64_t x_stride, const float* mean,
                                   const float* rstd, const float* gamma, const void* g_in,
                                   void* g_out, int g_dtype, int64_t g_stride, void* gb_out,
                                   int gb_dtype, int64_t gb_stride, float* dgamma, float* dbeta,
                                   float* gsum, const float* gb_scale, const float* gb_rowscale,
                                   int64_t rows_per_group, int64_t M, int64_t D,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
  vitmi_fold_desc fold;
  int rc = vitmi_layernorm_bwd_deferred(dy, dy_dtype, dy_stride, x, x_dtype, x_stride, mean, rstd, gamma, g_in, g_out, g_dtype,
                                        g_stride, gb_out, gb_dtype, gb_stride, dgamma, dbeta, gsum, gb_scale, gb_rowscale,
                                        rows_per_group, M, D, workspace, workspace_bytes, &fold, stream_);
  if (rc) return rc;
  return vitmi_fold_many(&fold, 1, stream_);
}

static void ln_fold_desc(vitmi_fold_desc* f, const float* part, int S, int64_t D, float* dgamma, float* dbeta, float* gsum) {
  f->struct_size = (int64_t)sizeof(vitmi_fold_desc);
  f->part = part; f->S = S; f->nseg = gsum ? 3 : 2; f->N = D; f->ld = 3 * D;
  f->out[0] = dgamma; f->out[1] = dbeta; f->out[2] = gsum;
}

extern "C" int vitmi_layernorm_bwd_deferred(const void* dy, int dy_dtype, int64_t dy_stride, const void* x,
                                            int x_dtype, int64_t x_stride, const float* mean,
                                            const float* rstd, const float* gamma, const void* g_in,
                                            void* g_out, int g_dtype, int64_t g_stride, void* gb_out,
                                            int gb_dtype, int64_t gb_stride, float* dgamma, float* dbeta,
                                            float* gsum, const float* gb_scale, const float* gb_rowscale,
                                            int64_t rows_per_group, int64_t M, int64_t D,
                                            void* workspace, size_t workspace_bytes, vitmi_fold_desc* fold, void* stream_) {
  VITMI_REQUIRE(fold, VITMI_E_BADARG, "layernorm_bwd_deferred: null fold descriptor");
  int rc = check_ln_common(x, x_dtype, x_stride, M, D, "layernorm_bwd");
  if (rc) return rc;
  VITMI_REQUIRE(dy && mean && rstd && gamma && g_out && dgamma && dbeta, VITMI_E_BADARG, "layernorm_bwd: null argument");
  VITMI_REQUIRE(!gb_rowscale || (M < (1ll << 32) && rows_per_group < (1ll << 32)), VITMI_E_SHAPE, "layernorm_bwd: gb_rowscale needs M < 2^32");
  VITMI_REQUIRE(dy_stride % 4 == 0 && g_stride % 4 == 0 && (!gb_out || gb_stride % 4 == 0), VITMI_E_ALIGN, "layernorm_bwd: strides must be multiples of 4");
  VITMI_REQUIRE(is_aligned(dy, 4 * dtype_size(dy_dtype)) && is_aligned(g_out, 4 * dtype_size(g_dtype)) &&
                    (!g_in || is_aligned(g_in, 4 * dtype_size(g_dtype))) &&
                    (!gb_out || is_aligned(gb_out, 4 * dtype_size(gb_dtype))) && is_aligned(gamma, 16),
                VITMI_E_ALIGN, "layernorm_bwd: pointer alignment");
  VITMI_REQUIRE(workspace && workspace_bytes >= vitmi_layernorm_bwd_workspace(M, D), VITMI_E_WORKSPACE, "layernorm_bwd: workspace too small");
  VITMI_REQUIRE(is_aligned(workspace, 16), VITMI_E_ALIGN, "layernorm_bwd: workspace alignment");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  float* part = reinterpret_cast<float*>(workspace);
  if (ln8_ok(D, {dy_stride, x_stride, g_stride, gb_out ? gb_stride : 0}, {dy, x, g_in, g_out, gb_out, gamma, gb_scale})) {
    const Ln8Cfg c = ln8_cfg(D);
    const int nb8 = ln_bwd8_blocks(M, D);
    const size_t lds8 = (size_t)4 * (64 / c.lpr) * D * sizeof(float);
    const int gbd8 = gb_out ? gb_dtype : dy_dtype;
#define LN_BWD8_T(L, N, TDY, TX, TG, TGB)                                                                            \
    hipLaunchKernelGGL((ln_bwd8_kernel<L, N, TDY, TX, TG, TGB>), dim3(nb8), dim3(256), lds8, stream, (const TDY*)dy,  \
                       dy_stride, (const TX*)x, x_stride, mean, rstd, gamma, (const TG*)g_in, (TG*)g_out, g_stride,   \
                       (TGB*)gb_out, gb_stride, part, gsum ? 1 : 0, gb_scale, gb_rowscale,                            \
                       rows_per_group > 0 ? rows_per_group : 1, M, (int)D)
#define LN_BWD8(TDY, TX, TG, TGB)                                                              \
    do {                                                                                        \
      switch (c.lpr * 8 + c.nv) {                                                               \
        case 16 * 8 + 1: LN_BWD8_T(16, 1, TDY, TX, TG, TGB); break;                             \
        case 16 * 8 + 2: LN_BWD8_T(16, 2, TDY, TX, TG, TGB); break;                             \
        case 16 * 8 + 3: LN_BWD8_T(16, 3, TDY, TX, TG, TGB); break;                             \
        case 32 * 8 + 2: LN_BWD8_T(32, 2, TDY, TX, TG, TGB); break;                             \
        case 32 * 8 + 3: LN_BWD8_T(32, 3, TDY, TX, TG, TGB); break;                             \
        case 64 * 8 + 2: LN_BWD8_T(64, 2, TDY, TX, TG, TGB); break;                             \
        case 64 * 8 + 3: LN_BWD8_T(64, 3, TDY, TX, TG, TGB); break;                             \
        default: LN_BWD8_T(64, 4, TDY, TX, TG, TGB); break;                                     \
      }                                                                                         \
    } while (0)
    if (dy_dtype == VITMI_F32 && x_dtype == VITMI_F32 && g_dtype == VITMI_F32 && gbd8 == VITMI_F32) LN_BWD8(float, float, float, float);
    else if (dy_dtype == VITMI_BF16 && x_dtype == VITMI_F32 && g_dtype == VITMI_F32 && gbd8 == VITMI_BF16) LN_BWD8(bf16, float, float, bf16);
    else if (dy_dtype == VITMI_BF16 && x_dtype == VITMI_BF16 && g_dtype == VITMI_BF16 && gbd8 == VITMI_BF16) LN_BWD8(bf16, bf16, bf16, bf16);
    else if (dy_dtype == VITMI_F32 && x_dtype == VITMI_BF16 && g_dtype == VITMI_BF16 && gbd8 == VITMI_F32) LN_BWD8(float, bf16, bf16, float);
    else if (dy_dtype == VITMI_F32 && x_dtype == VITMI_F32 && g_dtype == VITMI_F32 && gbd8 == VITMI_BF16) LN_BWD8(float, float, float, bf16);
    else if (dy_dtype == VITMI_F32 && x_dtype == VITMI_BF16 && g_dtype == VITMI_BF16 && gbd8 == VITMI_BF16) LN_BWD8(float, bf16, bf16, bf16);
    else return vitmi_fail(VITMI_E_DTYPE, "layernorm_bwd: dtype combination (dy=%d x=%d g=%d gb=%d)", dy_dtype, x_dtype, g_dtype, gbd8);
#undef LN_BWD8
#undef LN_BWD8_T
    rc = vitmi_check_launch("ln_bwd8_kernel");
    if (rc) return rc;
    ln_fold_desc(fold, part, nb8, D, dgamma, dbeta, gsum);
    return 0;
  }
  const int nblk = ln_bwd_blocks(M, D);
  const size_t lds = 4 * (size_t)D * sizeof(float);
  // activation dtype T (dy, gb) and residual dtype R (x, g) combinations built:
  //   (T,R) = (f32,f32), (bf16,f32), (bf16,bf16)
#define LN_BWD_NV(NVV, TDY, TX, TG, TGB)                                                       \
  hipLaunchKernelGGL((ln_bwd_kernel<NVV, TDY, TX, TG, TGB>), dim3(nblk), dim3(256), lds, stream, \
                     (const TDY*)dy, dy_stride, (const TX*)x, x_stride, mean, rstd, gamma,     \
                     (const TG*)g_in, (TG*)g_out, g_stride, (TGB*)gb_out, gb_stride, part,     \
                     gsum ? 1 : 0, gb_scale, gb_rowscale, rows_per_group > 0 ? rows_per_group : 1, M, (int)D)
#define LN_BWD(TDY, TX, TG, TGB)                                                               \
  do {                                                                                         \
    switch (ln_nv(D)) {                                                                        \
      case 2: LN_BWD_NV(2, TDY, TX, TG, TGB); break;                                           \
      case 3: LN_BWD_NV(3, TDY, TX, TG, TGB); break;                                           \
      case 4: LN_BWD_NV(4, TDY, TX, TG, TGB); break;                                           \
      default: LN_BWD_NV(LN_MAXV, TDY, TX, TG, TGB); break;                                    \
    }                                                                                          \
  } while (0)
  const int gbd = gb_out ? gb_dtype : dy_dtype;
  if (dy_dtype == VITMI_F32 && x_dtype == VITMI_F32 && g_dtype == VITMI_F32 && gbd == VITMI_F32) LN_BWD(float, float, float, float);
  else if (dy_dtype == VITMI_BF16 && x_dtype == VITMI_F32 && g_dtype == VITMI_F32 && gbd == VITMI_BF16) LN_BWD(bf16, float, float, bf16);
  else if (dy_dtype == VITMI_BF16 && x_dtype == VITMI_BF16 && g_dtype == VITMI_BF16 && gbd == VITMI_BF16) LN_BWD(bf16, bf16, bf16, bf16);
  else if (dy_dtype == VITMI_F32 && x_dtype == VITMI_BF16 && g_dtype == VITMI_BF16 && gbd == VITMI_F32) LN_BWD(float, bf16, bf16, float);
  else if (dy_dtype == VITMI_F32 && x_dtype == VITMI_F32 && g_dtype == VITMI_F32 && gbd == VITMI_BF16) LN_BWD(float, float, float, bf16);
  else if (dy_dtype == VITMI_F32 && x_dtype == VITMI_BF16 && g_dtype == VITMI_BF16 && gbd == VITMI_BF16) LN_BWD(float, bf16, bf16, bf16);
  else return vitmi_fail(VITMI_E_DTYPE, "layernorm_bwd: dtype combination (dy=%d x=%d g=%d gb=%d)", dy_dtype, x_dtype, g_dtype, gbd);
#undef LN_BWD
#undef LN_BWD_NV
  rc = vitmi_check_launch("ln_bwd_kernel");
  if (rc) return rc;
  ln_fold_desc(fold, part, nblk, D, dgamma, dbeta, gsum);
  return 0;
}

// every diagnostic switch of this file back to its default (vitmi_debug_reset, core.cpp)
void vitmi_debug_reset_layernorm() {
  g_ln8 = 1;
}
