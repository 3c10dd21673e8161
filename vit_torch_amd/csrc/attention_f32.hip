// fp32 attention (the parity mode): plain fp32 FMAs on the vector units, same
// data layouts and the same fwd / dq / dkdv split as attention.hip, no MFMA —
// every product and every exp is an fp32 operation, as in the reference's
// `(q @ k.transpose(-2,-1))`, `softmax`, `attn @ v` (models/swin.py:124-142).
// One wave owns RPW rows (queries, or keys in dkdv); the other side streams
// through LDS in tiles of 64 rows, one row per lane.  hd <= 64.
#include "common.h"

namespace {

constexpr int RPW = 4;           // rows per wave
constexpr int RPB = 4 * RPW;     // rows per 256-thread block

// stage 64 rows x hd floats (row stride hd+1 in LDS), zero beyond N
__device__ __forceinline__ void stage_f32(float* lds, const float* g, int64_t ts, int row0, int N,
                                          int hd, int tid) {
  for (int idx = tid; idx < 64 * hd; idx += 256) {
    const int r = idx / hd, d = idx % hd;
    const int gr = row0 + r;
    lds[r * (hd + 1) + d] = gr < N ? g[(int64_t)gr * ts + d] : 0.f;
  }
}
// stage this block's RPB own rows (stride hd), zero beyond N
__device__ __forceinline__ void stage_own(float* lds, const float* g, int64_t ts, int row0, int N,
                                          int hd, int tid) {
  for (int idx = tid; idx < RPB * hd; idx += 256) {
    const int r = idx / hd, d = idx % hd;
    const int gr = row0 + r;
    lds[idx] = gr < N ? g[(int64_t)gr * ts + d] : 0.f;
  }
}

__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ qkv,
                                                           float* __restrict__ out,
                                                           float* __restrict__ lse, int N, int H,
                                                           int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;                       // [64][hd+1]
  float* Vs = Ks + 64 * (hd + 1);       // [64][hd+1]
  float* Qs = Vs + 64 * (hd + 1);       // [RPB][hd]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int64_t ts = (int64_t)3 * H * hd;
  const float* qb = qkv + (int64_t)b * N * ts + h * hd;
  const float* kb = qb + H * hd;
  const float* vb = qb + 2 * H * hd;
  const int r0 = blockIdx.x * RPB;
  stage_own(Qs, qb, ts, r0, N, hd, tid);
  float m[RPW], l[RPW], o[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) { m[i] = -INFINITY; l[i] = 0.f; o[i] = 0.f; }
  const int nkt = (N + 63) / 64;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    stage_f32(Ks, kb, ts, kt * 64, N, hd, tid);
    stage_f32(Vs, vb, ts, kt * 64, N, hd, tid);
    __syncthreads();
    const int key = kt * 64 + lane;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const float* qi = Qs + (w * RPW + i) * hd;
      float s = 0.f;
      for (int d = 0; d < hd; ++d) s = fmaf(qi[d], Ks[lane * (hd + 1) + d], s);
      s = key < N ? s * scale : -INFINITY;
      const float m_new = fmaxf(m[i], wave_max(s));
      const float alpha = expf(m[i] - m_new);
      const float p = expf(s - m_new);
      l[i] = l[i] * alpha + wave_sum(p);
      m[i] = m_new;
      float acc = o[i] * alpha;
      const int dl = lane < hd ? lane : 0;
      for (int j = 0; j < 64; ++j) acc = fmaf(__shfl(p, j), Vs[j * (hd + 1) + dl], acc);
      o[i] = acc;
    }
  }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int q = r0 + w * RPW + i;
    if (q < N) {
      if (lane < hd) out[((int64_t)(b * (int64_t)N + q) * H + h) * hd + lane] = o[i] / l[i];
      if (lane == 0) lse[(int64_t)bh * N + q] = m[i] + logf(l[i]);
    }
  }
}

__global__ void attn_delta_f32_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                      float* __restrict__ delta, int64_t rows, int N, int H, int hd) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  float s = 0.f;
  for (int d = 0; d < hd; ++d) s = fmaf(out[row * hd + d], dout[row * hd + d], s);
  const int h = (int)(row % H);
  const int64_t bn = row / H;
  const int n = (int)(bn % N);
  const int64_t b = bn / N;
  delta[(b * H + h) * N + n] = s;
}

// dQ: wave owns RPW queries, keys stream through LDS
__global__ __launch_bounds__(256) void attn_bwd_dq_f32_kernel(const float* __restrict__ qkv,
                                                              const float* __restrict__ dout,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ delta,
                                                              float* __restrict__ dqkv, int N, int H,
                                                              int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = Ks + 64 * (hd + 1);
  float* Qs = Vs + 64 * (hd + 1);
  float* dOs = Qs + RPB * hd;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int64_t ts = (int64_t)3 * H * hd, os = (int64_t)H * hd;
  const float* qb = qkv + (int64_t)b * N * ts + h * hd;
  const float* kb = qb + H * hd;
  const float* vb = qb + 2 * H * hd;
  const float* dob = dout + (int64_t)b * N * os + h * hd;
  const int r0 = blockIdx.x * RPB;
  stage_own(Qs, qb, ts, r0, N, hd, tid);
  stage_own(dOs, dob, os, r0, N, hd, tid);
  float lse_i[RPW], del_i[RPW], dq[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int q = min(r0 + w * RPW + i, N - 1);
    lse_i[i] = lse[(int64_t)bh * N + q];
    del_i[i] = delta[(int64_t)bh * N + q];
    dq[i] = 0.f;
  }
  const int nkt = (N + 63) / 64;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    stage_f32(Ks, kb, ts, kt * 64, N, hd, tid);
    stage_f32(Vs, vb, ts, kt * 64, N, hd, tid);
    __syncthreads();
    const int key = kt * 64 + lane;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const float* qi = Qs + (w * RPW + i) * hd;
      const float* doi = dOs + (w * RPW + i) * hd;
      float s = 0.f, dp = 0.f;
      for (int d = 0; d < hd; ++d) {
        s = fmaf(qi[d], Ks[lane * (hd + 1) + d], s);
        dp = fmaf(doi[d], Vs[lane * (hd + 1) + d], dp);
      }
      const float p = key < N ? expf(s * scale - lse_i[i]) : 0.f;
      const float ds = p * (dp - del_i[i]);
      float acc = dq[i];
      const int dl = lane < hd ? lane : 0;
      for (int j = 0; j < 64; ++j) acc = fmaf(__shfl(ds, j), Ks[j * (hd + 1) + dl], acc);
      dq[i] = acc;
    }
  }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int q = r0 + w * RPW + i;
    if (q < N && lane < hd) dqkv[(int64_t)(b * (int64_t)N + q) * ts + h * hd + lane] = dq[i] * scale;
  }
}

// dK, dV: wave owns RPW keys, queries stream through LDS
__global__ __launch_bounds__(256) void attn_bwd_dkdv_f32_kernel(const float* __restrict__ qkv,
                                                                const float* __restrict__ dout,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ delta,
                                                                float* __restrict__ dqkv, int N,
                                                                int H, int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Qs = sm;                         // [64][hd+1]
  float* dOs = Qs + 64 * (hd + 1);        // [64][hd+1]
  float* Kown = dOs + 64 * (hd + 1);      // [RPB][hd]
  float* Vown = Kown + RPB * hd;          // [RPB][hd]
  float* lse_s = Vown + RPB * hd;         // [64]
  float* del_s = lse_s + 64;              // [64]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int64_t ts = (int64_t)3 * H * hd, os = (int64_t)H * hd;
  const float* qb = qkv + (int64_t)b * N * ts + h * hd;
  const float* kb = qb + H * hd;
  const float* vb = qb + 2 * H * hd;
  const float* dob = dout + (int64_t)b * N * os + h * hd;
  const int r0 = blockIdx.x * RPB;
  stage_own(Kown, kb, ts, r0, N, hd, tid);
  stage_own(Vown, vb, ts, r0, N, hd, tid);
  float dk[RPW], dv[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
  const int nqt = (N + 63) / 64;
  for (int qt = 0; qt < nqt; ++qt) {
    __syncthreads();
    stage_f32(Qs, qb, ts, qt * 64, N, hd, tid);
    stage_f32(dOs, dob, os, qt * 64, N, hd, tid);
    if (tid < 64) {
      const int q = qt * 64 + tid;
      lse_s[tid] = q < N ? lse[(int64_t)bh * N + q] : 0.f;
      del_s[tid] = q < N ? delta[(int64_t)bh * N + q] : 0.f;
    }
    __syncthreads();
    const int q = qt * 64 + lane;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const float* kj = Kown + (w * RPW + i) * hd;
      const float* vj = Vown + (w * RPW + i) * hd;
      float s = 0.f, dp = 0.f;
      for (int d = 0; d < hd; ++d) {
        s = fmaf(Qs[lane * (hd + 1) + d], kj[d], s);
        dp = fmaf(dOs[lane * (hd + 1) + d], vj[d], dp);
      }
      const float p = q < N ? expf(s * scale - lse_s[lane]) : 0.f;
      const float ds = p * (dp - del_s[lane]);
      float ak = dk[i], av = dv[i];
      const int dl = lane < hd ? lane : 0;
      for (int j = 0; j < 64; ++j) {
        ak = fmaf(__shfl(ds, j), Qs[j * (hd + 1) + dl], ak);
        av = fmaf(__shfl(p, j), dOs[j * (hd + 1) + dl], av);
      }
      dk[i] = ak; dv[i] = av;
    }
  }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int key = r0 + w * RPW + i;
    if (key < N && lane < hd) {
      float* row = dqkv + (int64_t)(b * (int64_t)N + key) * ts + h * hd;
      row[H * hd + lane] = dk[i] * scale;
      row[2 * H * hd + lane] = dv[i];
    }
  }
}

}  // namespace

int attn_fwd_f32(const float* qkv, float* out, float* lse, int64_t B, int64_t N, int64_t H,
                 int64_t hd, float scale, hipStream_t stream) {
  VITMI_REQUIRE(qkv && out && lse && B > 0 && N > 0 && H > 0, VITMI_E_BADARG, "attn_fwd(f32): bad argument");
  VITMI_REQUIRE(hd > 0 && hd <= 64, VITMI_E_SHAPE, "attn_fwd(f32): head dim %lld > 64", (long long)hd);
  VITMI_REQUIRE(B * H <= 65535, VITMI_E_SHAPE, "attn_fwd(f32): B*H exceeds grid limit");
  dim3 grid((unsigned)((N + RPB - 1) / RPB), (unsigned)(B * H));
  const size_t lds = (2 * 64 * (hd + 1) + RPB * hd) * sizeof(float);
  hipLaunchKernelGGL(attn_fwd_f32_kernel, grid, dim3(256), lds, stream, qkv, out, lse, (int)N, (int)H, (int)hd, scale);
  return vitmi_check_launch("attn_fwd_f32_kernel");
}

int attn_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse,
                 float* dqkv, int64_t B, int64_t N, int64_t H, int64_t hd, float scale,
                 float* delta, hipStream_t stream) {
  VITMI_REQUIRE(qkv && out && dout && lse && dqkv && delta && B > 0 && N > 0 && H > 0, VITMI_E_BADARG, "attn_bwd(f32): bad argument");
  VITMI_REQUIRE(hd > 0 && hd <= 64, VITMI_E_SHAPE, "attn_bwd(f32): head dim %lld > 64", (long long)hd);
  VITMI_REQUIRE(B * H <= 65535, VITMI_E_SHAPE, "attn_bwd(f32): B*H exceeds grid limit");
  const int64_t rows = B * N * H;
  hipLaunchKernelGGL(attn_delta_f32_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, out, dout, delta, rows, (int)N, (int)H, (int)hd);
  int rc = vitmi_check_launch("attn_delta_f32_kernel");
  if (rc) return rc;
  dim3 grid((unsigned)((N + RPB - 1) / RPB), (unsigned)(B * H));
  const size_t lds_q = (2 * 64 * (hd + 1) + 2 * RPB * hd) * sizeof(float);
  hipLaunchKernelGGL(attn_bwd_dq_f32_kernel, grid, dim3(256), lds_q, stream, qkv, dout, lse, delta, dqkv, (int)N, (int)H, (int)hd, scale);
  rc = vitmi_check_launch("attn_bwd_dq_f32_kernel");
  if (rc) return rc;
  const size_t lds_k = (2 * 64 * (hd + 1) + 2 * RPB * hd + 128) * sizeof(float);
  hipLaunchKernelGGL(attn_bwd_dkdv_f32_kernel, grid, dim3(256), lds_k, stream, qkv, dout, lse, delta, dqkv, (int)N, (int)H, (int)hd, scale);
  return vitmi_check_launch("attn_bwd_dkdv_f32_kernel");
}
