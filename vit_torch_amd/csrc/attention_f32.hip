// fp32 attention (the parity mode): plain fp32 FMAs on the vector units, same
// data layouts and the same fwd / dq / dkdv split as attention.hip, no MFMA —
// every product and every exp is an fp32 operation, as in the reference's
// `(q @ k.transpose(-2,-1))`, `softmax`, `attn @ v` (models/swin.py:124-142).
// One wave owns RPW rows (queries, or keys in dkdv); the other side streams
// through LDS in tiles of 64 rows, one row per lane.  hd <= 64.
//
// Round 5: the same three kernels on the fp32 MATRIX pipe (`v_mfma_f32_32x32x2_f32`: exact fp32 products, fp32
// accumulation) for hd in {32, 64} and N <= 256 — the VALU forms above ran the ViT-B/16 step's attention in 1.67 s
// (attn_bwd_dkdv 105 ms per layer) and were 92 % of the fp32 / bf16x3 parity modes' step.  One workgroup per (image,
// head), one wave per block of 32 queries (keys in the dK / dV kernel); the other side's rows live in LDS as fp32 images
// [row][hd + 4].  The products are computed TRANSPOSED (S^T = K Q^T, O^T = V^T P^T, ...) so that the lane owns the
// wave's own row: the softmax statistics, lse and delta are lane-local, and an accumulator block of S^T IS the B operand
// of the next product — the MFMA contraction index may be enumerated in any order as long as both operands agree, and
// the C layout's row order (r&3) + 8 (r>>2) + 4 (lane>>5) is used as that order.
#include <atomic>
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

// ---------------------------------------------------------------------------------------------------------------
// fp32 MFMA forms (round 5)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int HD> struct F32M {
  static constexpr int P = HD + 4;       // LDS row pitch in floats: 16-B aligned rows, b128 reads of 16 rows cover all banks
  static constexpr int T = HD / 8;       // b128 operand reads per 32-row block (4 contraction steps each)
  static constexpr int DB = HD / 32;     // 32-wide blocks of the head dimension
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// row index inside a 32-row accumulator block that register r of a lane in half hf holds
__device__ __forceinline__ int crow(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// rows [0, rows_pad) x HD of a strided fp32 matrix -> LDS image [row][P]; rows >= N are zero (loads clamped, unconditional)
template <int HD>
__device__ __forceinline__ void stage_rows_m(float* lds, const float* g, int64_t ts, int N, int rows_pad, int tid, int nthr) {
  constexpr int PPR = HD / 4;
  for (int idx = tid; idx < rows_pad * PPR; idx += nthr) {
    const int r = idx / PPR, c = (idx % PPR) * 4;
    const int rr = r < N ? r : N - 1;
    f32x4 v = *reinterpret_cast<const f32x4*>(g + (int64_t)rr * ts + c);
    if (r >= N) v = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(lds + r * F32M<HD>::P + c) = v;
  }
}
// this lane's operand fragments of its own row: 4 consecutive floats at d = 8 t + 4 hf, t = 0 .. T-1
template <int HD>
__device__ __forceinline__ void load_frag_m(f32x4 (&f)[F32M<HD>::T], const float* row, int hf) {
#pragma unroll
  for (int t = 0; t < F32M<HD>::T; ++t) f[t] = *reinterpret_cast<const f32x4*>(row + 8 * t + 4 * hf);
}
// acc[32 x 32] = LDS rows (blk*32 ..) x fragments^T : contraction over d
template <int HD>
__device__ __forceinline__ f32x16 rows_times_frag(const float* img, int blk, int l32, int hf, const f32x4 (&f)[F32M<HD>::T]) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const float* row = img + (blk * 32 + l32) * F32M<HD>::P + 4 * hf;
#pragma unroll
  for (int t = 0; t < F32M<HD>::T; ++t) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + 8 * t);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = mfma32(a[j], f[t][j], acc);
  }
  return acc;
}
// out^T[d][own row] += sum over the block's 32 rows of img[row][d] * w[row] (w = accumulator block, C row order)
template <int HD>
__device__ __forceinline__ void img_t_times_acc(f32x16 (&o)[F32M<HD>::DB], const float* img, int blk, int l32, int hf, const f32x16& w) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* row = img + (blk * 32 + crow(r, hf)) * F32M<HD>::P + l32;
#pragma unroll
    for (int db = 0; db < F32M<HD>::DB; ++db) o[db] = mfma32(row[32 * db], w[r], o[db]);
  }
}
// store a transposed accumulator (lane = own row, registers = d) as 16-byte pieces of the row
template <int HD>
__device__ __forceinline__ void store_row_m(float* row, const f32x16 (&o)[F32M<HD>::DB], int hf, float mul) {
#pragma unroll
  for (int db = 0; db < F32M<HD>::DB; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 v = {o[db][4 * g4] * mul, o[db][4 * g4 + 1] * mul, o[db][4 * g4 + 2] * mul, o[db][4 * g4 + 3] * mul};
      *reinterpret_cast<f32x4*>(row + db * 32 + 8 * g4 + 4 * hf) = v;
    }
}

template <int HD, int NB>
__global__ __launch_bounds__(64 * NB) void attn_fwd_f32m_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                float* __restrict__ lse, int N, int H, float scale) {
  using C = F32M<HD>;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = sm + NB * 32 * C::P;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hf = lane >> 5, l32 = lane & 31;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int64_t ts = (int64_t)3 * H * HD;
  const float* qb = qkv + (int64_t)b * N * ts + h * HD;
  stage_rows_m<HD>(Ks, qb + H * HD, ts, N, NB * 32, tid, 64 * NB);
  stage_rows_m<HD>(Vs, qb + 2 * H * HD, ts, N, NB * 32, tid, 64 * NB);
  const int q = w * 32 + l32, qc = q < N ? q : N - 1;
  f32x4 qf[C::T];
  load_frag_m<HD>(qf, qb + (int64_t)qc * ts, hf);
  __syncthreads();
  // S^T[key][q] for every key block; softmax exactly as the reference writes it: max, exp, sum, divide
  f32x16 s[NB];
  float m = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    s[kb] = rows_times_frag<HD>(Ks, kb, l32, hf, qf);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = kb * 32 + crow(r, hf) < N ? s[kb][r] * scale : -INFINITY;
      s[kb][r] = v;
      m = fmaxf(m, v);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32));
  float l = 0.f;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = expf(s[kb][r] - m);
      s[kb][r] = p;
      l += p;
    }
  l += __shfl_xor(l, 32);
  f32x16 o[C::DB];
#pragma unroll
  for (int db = 0; db < C::DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kb][r] = s[kb][r] / l;
    img_t_times_acc<HD>(o, Vs, kb, l32, hf, s[kb]);
  }
  if (q < N) {
    store_row_m<HD>(out + ((int64_t)(b * (int64_t)N + q) * H + h) * HD, o, hf, 1.f);
    if (hf == 0) lse[(int64_t)bh * N + q] = m + logf(l);
  }
}

// dQ (and delta = rowsum(dO * O), which the dK / dV kernel reads): wave = 32 queries, K and V in LDS
template <int HD, int NB>
__global__ __launch_bounds__(64 * NB) void attn_bwd_dq_f32m_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                                   const float* __restrict__ dout, const float* __restrict__ lse,
                                                                   float* __restrict__ delta, float* __restrict__ dqkv, int N,
                                                                   int H, float scale) {
  using C = F32M<HD>;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = sm + NB * 32 * C::P;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hf = lane >> 5, l32 = lane & 31;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int64_t ts = (int64_t)3 * H * HD, os = (int64_t)H * HD;
  const float* qb = qkv + (int64_t)b * N * ts + h * HD;
  stage_rows_m<HD>(Ks, qb + H * HD, ts, N, NB * 32, tid, 64 * NB);
  stage_rows_m<HD>(Vs, qb + 2 * H * HD, ts, N, NB * 32, tid, 64 * NB);
  const int q = w * 32 + l32, qc = q < N ? q : N - 1;
  f32x4 qf[C::T], dof[C::T];
  load_frag_m<HD>(qf, qb + (int64_t)qc * ts, hf);
  load_frag_m<HD>(dof, dout + (int64_t)(b * (int64_t)N + qc) * os + h * HD, hf);
  float del = 0.f;
  {
    f32x4 of[C::T];
    load_frag_m<HD>(of, out + (int64_t)(b * (int64_t)N + qc) * os + h * HD, hf);
#pragma unroll
    for (int t = 0; t < C::T; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) del = fmaf(dof[t][j], of[t][j], del);
  }
  del += __shfl_xor(del, 32);
  const float lse_q = lse[(int64_t)bh * N + qc];
  if (q < N && hf == 0) delta[(int64_t)bh * N + q] = del;
  __syncthreads();
  f32x16 dq[C::DB];
#pragma unroll
  for (int db = 0; db < C::DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[db][i] = 0.f;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    const f32x16 st = rows_times_frag<HD>(Ks, kb, l32, hf, qf);      // S^T block
    const f32x16 dpt = rows_times_frag<HD>(Vs, kb, l32, hf, dof);    // dP^T block = V dO^T
    f32x16 ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = kb * 32 + crow(r, hf) < N ? expf(st[r] * scale - lse_q) : 0.f;
      ds[r] = p * (dpt[r] - del);
    }
    img_t_times_acc<HD>(dq, Ks, kb, l32, hf, ds);                     // dQ^T += K^T dS^T
  }
  if (q < N) store_row_m<HD>(dqkv + (int64_t)(b * (int64_t)N + q) * ts + h * HD, dq, hf, scale);
}

// dK, dV: wave = 32 keys, Q and dO (and lse, delta) in LDS
template <int HD, int NB>
__global__ __launch_bounds__(64 * NB) void attn_bwd_dkdv_f32m_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                     const float* __restrict__ lse, const float* __restrict__ delta,
                                                                     float* __restrict__ dqkv, int N, int H, float scale) {
  using C = F32M<HD>;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Qs = sm;
  float* dOs = sm + NB * 32 * C::P;
  float* lse_s = dOs + NB * 32 * C::P;
  float* del_s = lse_s + NB * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hf = lane >> 5, l32 = lane & 31;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int64_t ts = (int64_t)3 * H * HD, os = (int64_t)H * HD;
  const float* qb = qkv + (int64_t)b * N * ts + h * HD;
  stage_rows_m<HD>(Qs, qb, ts, N, NB * 32, tid, 64 * NB);
  stage_rows_m<HD>(dOs, dout + (int64_t)b * N * os + h * HD, os, N, NB * 32, tid, 64 * NB);
  for (int i = tid; i < NB * 32; i += 64 * NB) {
    const int ic = i < N ? i : N - 1;
    lse_s[i] = lse[(int64_t)bh * N + ic];
    del_s[i] = delta[(int64_t)bh * N + ic];
  }
  const int key = w * 32 + l32, kc = key < N ? key : N - 1;
  f32x4 kf[C::T], vf[C::T];
  load_frag_m<HD>(kf, qb + H * HD + (int64_t)kc * ts, hf);
  load_frag_m<HD>(vf, qb + 2 * H * HD + (int64_t)kc * ts, hf);
  __syncthreads();
  f32x16 dk[C::DB], dv[C::DB];
#pragma unroll
  for (int db = 0; db < C::DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[db][i] = 0.f; dv[db][i] = 0.f; }
#pragma unroll
  for (int qblk = 0; qblk < NB; ++qblk) {
    const f32x16 sq = rows_times_frag<HD>(Qs, qblk, l32, hf, kf);     // S block: rows = queries, lane = key
    const f32x16 dp = rows_times_frag<HD>(dOs, qblk, l32, hf, vf);    // dP block = dO V^T
    f32x16 p, ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = qblk * 32 + crow(r, hf);
      const float pv = qq < N ? expf(sq[r] * scale - lse_s[qq]) : 0.f;
      p[r] = pv;
      ds[r] = pv * (dp[r] - del_s[qq]);
    }
    img_t_times_acc<HD>(dv, dOs, qblk, l32, hf, p);                   // dV^T += dO^T P
    img_t_times_acc<HD>(dk, Qs, qblk, l32, hf, ds);                   // dK^T += Q^T dS
  }
  if (key < N) {
    float* row = dqkv + (int64_t)(b * (int64_t)N + key) * ts + h * HD;
    store_row_m<HD>(row + H * HD, dk, hf, scale);
    store_row_m<HD>(row + 2 * H * HD, dv, hf, 1.f);
  }
}

std::atomic<int> g_f32_valu{0};        // diagnostic / test hook: 1 = keep the VALU forms (A/B, and the reference for the MFMA forms' test)

template <int HD> size_t f32m_lds(int nb, bool dkdv) {
  return ((size_t)2 * nb * 32 * F32M<HD>::P + (dkdv ? 2 * nb * 32 : 0)) * sizeof(float);
}
bool f32m_ok(int64_t N, int64_t hd, const void* a, const void* b, const void* c) {
  return !g_f32_valu && N >= 1 && N <= 256 && (hd == 32 || hd == 64) && is_aligned(a, 16) && is_aligned(b, 16) && is_aligned(c, 16);
}

#define F32M_NB(HDV, KERN, nb, ...)                                     \
  switch (nb) {                                                         \
    case 1: F32M_GO(HDV, KERN, 1, __VA_ARGS__); break;                  \
    case 2: F32M_GO(HDV, KERN, 2, __VA_ARGS__); break;                  \
    case 3: F32M_GO(HDV, KERN, 3, __VA_ARGS__); break;                  \
    case 4: F32M_GO(HDV, KERN, 4, __VA_ARGS__); break;                  \
    case 5: F32M_GO(HDV, KERN, 5, __VA_ARGS__); break;                  \
    case 6: F32M_GO(HDV, KERN, 6, __VA_ARGS__); break;                  \
    case 7: F32M_GO(HDV, KERN, 7, __VA_ARGS__); break;                  \
    default: F32M_GO(HDV, KERN, 8, __VA_ARGS__); break;                 \
  }
#define F32M_GO(HDV, KERN, NBV, lds, grid, ...)                                                                      \
  do {                                                                                                               \
    auto kern_ = KERN<HDV, NBV>;                                                                                     \
    if (int rc_ = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(kern_), 160 * 1024, #KERN)) return rc_;      \
    hipLaunchKernelGGL(kern_, dim3((unsigned)(grid)), dim3(64 * NBV), lds, stream, __VA_ARGS__);                     \
  } while (0)

}  // namespace

extern "C" void vitmi_debug_attn_f32_valu(int on) { g_f32_valu = on != 0; }
void vitmi_debug_reset_attention_f32() { g_f32_valu = 0; }

int attn_fwd_f32(const float* qkv, float* out, float* lse, int64_t B, int64_t N, int64_t H,
                 int64_t hd, float scale, hipStream_t stream) {
  VITMI_REQUIRE(qkv && out && lse && B > 0 && N > 0 && H > 0, VITMI_E_BADARG, "attn_fwd(f32): bad argument");
  VITMI_REQUIRE(hd > 0 && hd <= 64, VITMI_E_SHAPE, "attn_fwd(f32): head dim %lld > 64", (long long)hd);
  if (f32m_ok(N, hd, qkv, out, out)) {
    const int nb = (int)((N + 31) / 32);
    if (hd == 64) { F32M_NB(64, attn_fwd_f32m_kernel, nb, f32m_lds<64>(nb, false), B * H, qkv, out, lse, (int)N, (int)H, scale); }
    else { F32M_NB(32, attn_fwd_f32m_kernel, nb, f32m_lds<32>(nb, false), B * H, qkv, out, lse, (int)N, (int)H, scale); }
    return vitmi_check_launch("attn_fwd_f32m_kernel");
  }
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
  if (f32m_ok(N, hd, qkv, out, dout) && is_aligned(dqkv, 16)) {
    const int nb = (int)((N + 31) / 32);
    if (hd == 64) { F32M_NB(64, attn_bwd_dq_f32m_kernel, nb, f32m_lds<64>(nb, false), B * H, qkv, out, dout, lse, delta, dqkv, (int)N, (int)H, scale); }
    else { F32M_NB(32, attn_bwd_dq_f32m_kernel, nb, f32m_lds<32>(nb, false), B * H, qkv, out, dout, lse, delta, dqkv, (int)N, (int)H, scale); }
    if (int rc_ = vitmi_check_launch("attn_bwd_dq_f32m_kernel")) return rc_;
    if (hd == 64) { F32M_NB(64, attn_bwd_dkdv_f32m_kernel, nb, f32m_lds<64>(nb, true), B * H, qkv, dout, lse, delta, dqkv, (int)N, (int)H, scale); }
    else { F32M_NB(32, attn_bwd_dkdv_f32m_kernel, nb, f32m_lds<32>(nb, true), B * H, qkv, dout, lse, delta, dqkv, (int)N, (int)H, scale); }
    return vitmi_check_launch("attn_bwd_dkdv_f32m_kernel");
  }
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
