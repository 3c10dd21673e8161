// Swin-specific kernels:
//  * (shifted-)window attention, models/swin.py:113-144 + :241-261: the cyclic shift
//    (torch.roll), window_partition and window_reverse are NOT materialised — every row
//    access goes through the window->token map below, so qkv, the attention output and
//    their gradients stay in token order and the reference's four permute/roll copies per
//    block disappear.  Relative-position bias [H,N,N] and the -100 shift mask [nW,N,N] are
//    added to the scores in registers.
//  * relative-position-bias gather / gradient scatter (:126-129), PatchMerging gather /
//    scatter (:317-323), token mean (AdaptiveAvgPool1d, :584).
// Two implementations: fp32 vector kernels (any hd <= 64, both dtypes: the parity mode) with
// one workgroup per (window, head), and the bf16 MFMA kernels for hd = 32 (every Swin variant
// at window 7) with one WAVE per (window, head) — "MFMA path" below.
#include <atomic>
#include "common.h"

namespace {

struct WinGeom { int Himg, Wimg, ws, shift, nWx, nW; };

// global token row of local position i of window bw (bw = image * nW + window)
__device__ __forceinline__ int64_t win_token(const WinGeom& g, int64_t bw, int i) {
  const int64_t b = bw / g.nW;
  const int w = (int)(bw % g.nW);
  const int wy = w / g.nWx, wx = w % g.nWx;
  int y = wy * g.ws + i / g.ws + g.shift;
  int x = wx * g.ws + i % g.ws + g.shift;
  if (y >= g.Himg) y -= g.Himg;
  if (x >= g.Wimg) x -= g.Wimg;
  return b * (int64_t)g.Himg * g.Wimg + (int64_t)y * g.Wimg + x;
}

// stage rows [N][hd] of one head (through the window map) into LDS, row stride hd+1
template <typename T>
__device__ __forceinline__ void stage_win(float* lds, const T* base, int64_t ts, const WinGeom& g,
                                          int64_t bw, int N, int hd, int tid) {
  for (int idx = tid; idx < 64 * hd; idx += 256) {
    const int r = idx / hd, d = idx % hd;
    lds[r * (hd + 1) + d] = r < N ? to_f32(base[win_token(g, bw, r) * ts + d]) : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void win_attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                          float* __restrict__ lse, const float* __restrict__ bias,
                                                          const float* __restrict__ mask, WinGeom g, int H, int N,
                                                          int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Qs = sm;
  float* Ks = Qs + 64 * (hd + 1);
  float* Vs = Ks + 64 * (hd + 1);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t bw = blockIdx.x;
  const int h = blockIdx.y;
  const int64_t ts = (int64_t)3 * H * hd;
  stage_win(Qs, qkv + h * hd, ts, g, bw, N, hd, tid);
  stage_win(Ks, qkv + (H + h) * hd, ts, g, bw, N, hd, tid);
  stage_win(Vs, qkv + (2 * H + h) * hd, ts, g, bw, N, hd, tid);
  __syncthreads();
  const float* mrow = mask ? mask + (bw % g.nW) * (int64_t)N * N : nullptr;
  for (int i = w; i < N; i += 4) {
    float s = -INFINITY;
    if (lane < N) {
      s = 0.f;
      for (int d = 0; d < hd; ++d) s = fmaf(Qs[i * (hd + 1) + d] * scale, Ks[lane * (hd + 1) + d], s);
      s += bias[((int64_t)h * N + i) * N + lane];
      if (mrow) s += mrow[i * N + lane];
    }
    const float mx = wave_max(s);
    const float p = expf(s - mx);
    const float sum = wave_sum(p);
    float acc = 0.f;
    const int dl = lane < hd ? lane : 0;
    for (int j = 0; j < N; ++j) acc = fmaf(__shfl(p, j), Vs[j * (hd + 1) + dl], acc);
    const int64_t tok = win_token(g, bw, i);
    if (lane < hd) out[tok * H * hd + h * hd + lane] = from_f32<T>(acc / sum);
    if (lane == 0) lse[(bw * H + h) * N + i] = mx + logf(sum);
  }
}


// ------------------------------------------------------------ MFMA path ---
// bf16, hd = 32, N <= 64 (every Swin variant at window 7): ONE WAVE per (window, head),
// v_mfma_f32_16x16x32_bf16 with K = 32 = the head dim, so a 16x16 score tile is one MFMA.
//   S^T[key][q] = K Q^T      A = K rows, B = Q rows, both straight from global memory
//                            (a token row of one head is 64 contiguous bytes);
//                            the lane owns query q = 16*qb + (lane & 15) and keys
//                            16*kb + 4*(lane >> 4) + r: softmax statistics are lane-local
//                            plus two cross-group shuffles
//   O^T[d][q]  = V^T P^T     B = the score accumulators themselves: k-step s takes the
//                            tiles kb = 2s, 2s+1, so the k slot order is (kb&1)*16 + 4g + r,
//                            and V^T is read from a [key][d] LDS image (96-B pitch) with
//                            ds_read_b64_tr_b16 in that same order
struct __attribute__((packed, aligned(4))) F4U { float v[4]; };   // 4-B aligned 16-B load

constexpr int WP = 96;                 // LDS pitch of a staged [64 rows][32 bf16] image
constexpr float NEG_BIG = -1e30f;

// operand fragment (A or B) of rows = tokens: lane (i = lane&15, g = lane>>4) gets
// X[token(16*blk + i)][8g .. 8g+7]; rows >= N are clamped (their results are masked/unused)
__device__ __forceinline__ bf16x8 win_row_frag(const bf16* base, int64_t ts, const WinGeom& geo, int64_t bw,
                                               int blk, int N, int lane) {
  const int r = min(blk * 16 + (lane & 15), N - 1);
  return *reinterpret_cast<const bf16x8*>(base + win_token(geo, bw, r) * ts + 8 * (lane >> 4));
}
// stage [64 rows][32] bf16 of one head into a wave-private LDS image (rows >= N zero)
__device__ __forceinline__ void win_stage(char* img, const bf16* base, int64_t ts, const WinGeom& geo, int64_t bw,
                                          int N, int lane) {
  bf16x8 v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {                    // unconditional loads from clamped rows (a branch
    const int p = lane + 64 * t, row = min(p >> 2, N - 1), c = p & 3;   // around a load serialises them)
    v[t] = *reinterpret_cast<const bf16x8*>(base + win_token(geo, bw, row) * ts + 8 * c);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int p = lane + 64 * t, row = p >> 2, c = p & 3;
    if (row >= N) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[t][e] = (bf16)0.f;
    }
    *reinterpret_cast<bf16x8*>(img + row * WP + c * 16) = v[t];
  }
}
// A/B fragment whose k runs over the ROWS of a staged image, in the slot order
// k(g, j) = k0 + 16*(j >> 2) + 4*g + (j & 3) (the order accumulator tiles present their rows)
__device__ __forceinline__ bf16x8 win_tr_frag(const char* img, int pitch, int k0, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const char* p = img + (k0 + 4 * g + (i >> 2)) * pitch + (c0 + 4 * (i & 3)) * 2;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p + 16 * pitch));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
__device__ __forceinline__ bf16x8 pack_tiles(const f32x4& a, const f32x4& b) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = (bf16)a[e]; r[4 + e] = (bf16)b[e]; }
  return r;
}
// scaled scores + relative-position bias + shift mask of tile (kb, qb) for this lane's query;
// keys >= N get a large negative value
__device__ __forceinline__ f32x4 win_bias_tile(const f32x4& st, float scale, const float* __restrict__ brow,
                                               const float* __restrict__ mrow, int kb, int g, int N) {
  f32x4 o;
  const int key0 = kb * 16 + 4 * g;
  if (key0 + 4 <= N) {
    const F4U b = *reinterpret_cast<const F4U*>(brow + key0);
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = fmaf(st[r], scale, b.v[r]);
    if (mrow) {
      const F4U m = *reinterpret_cast<const F4U*>(mrow + key0);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] += m.v[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = key0 + r;
      o[r] = key < N ? fmaf(st[r], scale, brow[key]) + (mrow ? mrow[key] : 0.f) : NEG_BIG;
    }
  }
  return o;
}

__global__ __launch_bounds__(256) void win_attn_fwd_mfma_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                               float* __restrict__ lse, const float* __restrict__ bias,
                                                               const float* __restrict__ mask, WinGeom geo, int H, int N,
                                                               float scale, int64_t tasks) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 64 * WP];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t task = (int64_t)blockIdx.x * 4 + w;
  if (task >= tasks) return;                       // no barriers below: waves are independent
  const int64_t bw = task / H;
  const int h = (int)(task % H);
  const int i = lane & 15, g = lane >> 4;
  const int64_t ts = (int64_t)3 * H * 32;
  const bf16* qb_ = qkv + h * 32;
  const bf16* kb_ = qkv + (H + h) * 32;
  const bf16* vb_ = qkv + (2 * H + h) * 32;
  char* Vs = smem + w * 64 * WP;
  win_stage(Vs, vb_, ts, geo, bw, N, lane);
  bf16x8 qf[4], kf[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    qf[b] = win_row_frag(qb_, ts, geo, bw, b, N, lane);
    kf[b] = win_row_frag(kb_, ts, geo, bw, b, N, lane);
  }
  f32x4 st[4][4];                                  // [kb][qb]
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      st[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb], qf[qb], z, 0, 0, 0);
    }
  const float* mwin = mask ? mask + (bw % geo.nW) * (int64_t)N * N : nullptr;
  float rinv[4];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const int q = min(qb * 16 + i, N - 1);
    const float* brow = bias + ((int64_t)h * N + q) * N;
    const float* mrow = mwin ? mwin + (int64_t)q * N : nullptr;
    float mx = NEG_BIG;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      st[kb][qb] = win_bias_tile(st[kb][qb], scale, brow, mrow, kb, g, N);
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kb][qb][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(st[kb][qb][r] - mx);
        st[kb][qb][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    rinv[qb] = 1.f / sum;
    if (g == 0 && qb * 16 + i < N) lse[(bw * H + h) * N + qb * 16 + i] = mx + __logf(sum);
  }
  // O^T[d][q] = sum_key V^T[d][key] P^T[key][q]; the probabilities are normalised at the end
  f32x4 o[2][4];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 vt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) vt[db] = win_tr_frag(Vs, WP, 32 * s, 16 * db, lane);
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      const bf16x8 pf = pack_tiles(st[2 * s][qb], st[2 * s + 1][qb]);
#pragma unroll
      for (int db = 0; db < 2; ++db) o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[db], pf, o[db][qb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const int q = qb * 16 + i;
    if (q < N) {
      bf16* orow = out + win_token(geo, bw, q) * (int64_t)H * 32 + h * 32;
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        bf16x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (bf16)(o[db][qb][r] * rinv[qb]);
        *reinterpret_cast<bf16x4*>(orow + db * 16 + 4 * g) = v;
      }
    }
  }
}


// Backward, same mapping (one wave per (window, head)); a wave keeps ONE head and walks
// windows, so the relative-position-bias gradient (a sum over all windows of a head)
// accumulates in registers and is written once per wave as a partial [N][N] tile.
//   S^T = K Q^T, dP^T = V dO^T           operands straight from global rows
//   P^T = exp(S^T - lse), delta = sum_key P dP, dS^T = P (dP - delta)     lane-local (+2 shuffles)
//   dV^T += dO^T P, dK^T += Q^T dS        contraction over queries: P / dS go through a
//                                         wave-private LDS tile [q][key] (160-B pitch) and come
//                                         back as B operands by ds_read_b64_tr_b16; dO^T / Q^T
//                                         from [row][d] images the same way
//   dQ^T += K^T dS^T                      B operand = the dS accumulators themselves
constexpr int TP = 160;                // pitch of the [64 q][64 key] bf16 tile
static std::atomic<int> g_win_bwd_prefetch{1};     // diagnostic hook (vitmi_debug_win_bwd_prefetch): L2 prefetch of a wave's next window
constexpr int WIN_BWD_LDS = 3 * 64 * WP + 64 * TP;     // Q, K, dO images + tile, per wave

__device__ __forceinline__ void win_store_T(bf16* dst, const f32x4& acc, float mul, int g) {
  bf16x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (bf16)(acc[r] * mul);
  *reinterpret_cast<bf16x4*>(dst + 4 * g) = v;
}

__global__ __launch_bounds__(256) void win_attn_bwd_mfma_kernel(const bf16* __restrict__ qkv,
                                                               const bf16* __restrict__ dout,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ mask,
                                                               bf16* __restrict__ dqkv, float* __restrict__ dbias_part,
                                                               float* __restrict__ qkvb_part,
                                                               WinGeom geo, int H, int N, float scale, int64_t Bw,
                                                               int nwaves, int pf_dump) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wid = blockIdx.x * 4 + w;
  if (wid >= nwaves) return;                       // waves are independent: no barriers below
  const int h = wid % H;
  const int wstep = nwaves / H;
  const int i = lane & 15, g = lane >> 4;
  const int64_t ts = (int64_t)3 * H * 32, os = (int64_t)H * 32;
  char* Qs = smem + w * WIN_BWD_LDS;
  char* Ks = Qs + 64 * WP;
  char* dOs = Ks + 64 * WP;
  char* T = dOs + 64 * WP;
  const bf16* qb_ = qkv + h * 32;
  const bf16* kb_ = qkv + (H + h) * 32;
  const bf16* vb_ = qkv + (2 * H + h) * 32;
  const bf16* dob = dout + h * 32;

  // column sums of dQ / dK / dV over this wave's windows (the qkv Linear's bias gradient);
  // padded queries / keys contribute exact zeros, so no masking is needed
  f32x4 cq[2], ck[2], cv[2];
#pragma unroll
  for (int db = 0; db < 2; ++db) { cq[db] = f32x4{0.f, 0.f, 0.f, 0.f}; ck[db] = cq[db]; cv[db] = cq[db]; }
  f32x4 dsum[4][4];                                // [kb][qb] running d(score) of this head
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) dsum[kb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int64_t bw = wid / H; bw < Bw; bw += wstep) {
    win_stage(Qs, qb_, ts, geo, bw, N, lane);
    win_stage(Ks, kb_, ts, geo, bw, N, lane);
    win_stage(dOs, dob, os, geo, bw, N, lane);
    f32x4 st[4][4], dpt[4][4];
    {
      bf16x8 qf[4], kf[4], vf[4], dof[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        qf[b] = win_row_frag(qb_, ts, geo, bw, b, N, lane);
        kf[b] = win_row_frag(kb_, ts, geo, bw, b, N, lane);
        vf[b] = win_row_frag(vb_, ts, geo, bw, b, N, lane);
        dof[b] = win_row_frag(dob, os, geo, bw, b, N, lane);
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          st[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb], qf[qb], z, 0, 0, 0);
          dpt[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kb], dof[qb], z, 0, 0, 0);
        }
    }
    const float* mwin = mask ? mask + (bw % geo.nW) * (int64_t)N * N : nullptr;
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      const int q = min(qb * 16 + i, N - 1);
      const bool qok = qb * 16 + i < N;
      const float* brow = bias + ((int64_t)h * N + q) * N;
      const float* mrow = mwin ? mwin + (int64_t)q * N : nullptr;
      const float l = lse[(bw * H + h) * N + q];
      float del = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const f32x4 sc = win_bias_tile(st[kb][qb], scale, brow, mrow, kb, g, N);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = qok ? __expf(sc[r] - l) : 0.f;     // keys >= N: exp(-1e30 - l) = 0
          st[kb][qb][r] = p;
          del = fmaf(p, dpt[kb][qb][r], del);
        }
      }
      del += __shfl_xor(del, 16, 64);
      del += __shfl_xor(del, 32, 64);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ds = st[kb][qb][r] * (dpt[kb][qb][r] - del);
          dpt[kb][qb][r] = ds;                     // now dS^T
          dsum[kb][qb][r] += ds;
        }
    }
    // ---- L2 prefetch of this wave's NEXT window (round 3).  A wave walks its windows serially — load, wait, compute,
    // store — at one wave per SIMD, so every HBM round trip of a window is exposed (two per window: the staging batch and
    // the V fragments).  All loads of the current window have been consumed at this point and ~60 % of its arithmetic is
    // still ahead: each 64-B row piece (Q, K, V, dO of every token) of the next window is touched by a 4-byte LDS-DMA
    // (4 N lanes = 4 instructions, no VGPR destination, the dwords land in a 256-B dump behind the workgroup's LDS), so
    // the next iteration's loads find their lines in this XCD's L2.  Inline asm: hipcc need not know about them — they
    // are older than every load whose data is used and vmcnt retires in order.
    if (pf_dump >= 0 && bw + wstep < Bw) {
      const int64_t bwn = bw + wstep;
      int lane_p = lane;
      asm volatile("" : "+v"(lane_p));
#pragma unroll 1
      for (int p = lane_p; p < 4 * N; p += 64) {
        const int m = p / N, r = p - m * N;
        const int64_t tok = win_token(geo, bwn, r);
        const char* src = m < 3 ? reinterpret_cast<const char*>(qkv + tok * ts + (m * H + h) * 32)
                                : reinterpret_cast<const char*>(dout + tok * os + h * 32);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(pf_dump) : "memory");
      }
    }
    // ---- P -> tile [q][key]; dV^T[d][key] = sum_q dO^T[d][q] P[q][key]
#pragma unroll
    for (int qb = 0; qb < 4; ++qb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        win_store_T(reinterpret_cast<bf16*>(T + (qb * 16 + i) * TP) + kb * 16, st[kb][qb], 1.f, g);
    {
      f32x4 dv[2][4];
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) dv[db][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 a[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) a[db] = win_tr_frag(dOs, WP, 32 * s, 16 * db, lane);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const bf16x8 b = win_tr_frag(T, TP, 32 * s, 16 * kb, lane);
#pragma unroll
          for (int db = 0; db < 2; ++db) dv[db][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[db], b, dv[db][kb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int key = kb * 16 + i;
        if (key < N) {
          bf16* row = dqkv + win_token(geo, bw, key) * ts + (2 * H + h) * 32;
#pragma unroll
          for (int db = 0; db < 2; ++db) win_store_T(row + db * 16, dv[db][kb], 1.f, g);
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) cv[db] += dv[db][kb];
      }
    }
    // ---- dS -> the same tile; dK^T[d][key] = scale * sum_q Q^T[d][q] dS[q][key]
#pragma unroll
    for (int qb = 0; qb < 4; ++qb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        win_store_T(reinterpret_cast<bf16*>(T + (qb * 16 + i) * TP) + kb * 16, dpt[kb][qb], 1.f, g);
    {
      f32x4 dk[2][4];
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) dk[db][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 a[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) a[db] = win_tr_frag(Qs, WP, 32 * s, 16 * db, lane);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const bf16x8 b = win_tr_frag(T, TP, 32 * s, 16 * kb, lane);
#pragma unroll
          for (int db = 0; db < 2; ++db) dk[db][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[db], b, dk[db][kb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int key = kb * 16 + i;
        if (key < N) {
          bf16* row = dqkv + win_token(geo, bw, key) * ts + (H + h) * 32;
#pragma unroll
          for (int db = 0; db < 2; ++db) win_store_T(row + db * 16, dk[db][kb], scale, g);
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) ck[db] += dk[db][kb];
      }
    }
    // ---- dQ^T[d][q] = scale * sum_key K^T[d][key] dS^T[key][q]   (B = the dS accumulators)
    {
      f32x4 dq[2][4];
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) dq[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 a[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) a[db] = win_tr_frag(Ks, WP, 32 * s, 16 * db, lane);
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
          const bf16x8 b = pack_tiles(dpt[2 * s][qb], dpt[2 * s + 1][qb]);
#pragma unroll
          for (int db = 0; db < 2; ++db) dq[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[db], b, dq[db][qb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int qb = 0; qb < 4; ++qb) {
        const int q = qb * 16 + i;
        if (q < N) {
          bf16* row = dqkv + win_token(geo, bw, q) * ts + h * 32;
#pragma unroll
          for (int db = 0; db < 2; ++db) win_store_T(row + db * 16, dq[db][qb], scale, g);
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) cq[db] += dq[db][qb];
      }
    }
  }
  if (qkvb_part) {
    // fold the 16 lanes of a group (they hold different rows of the same d), lane i == 0 stores
    // d = 16*db + 4g + r; row wid / H of [nwaves / H][3][H][32]
    float* brow = qkvb_part + (int64_t)(wid / H) * 3 * H * 32 + h * 32;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = cq[db][r] * scale, b = ck[db][r] * scale, c = cv[db][r];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
          a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); c += __shfl_xor(c, off, 64);
        }
        if (i == 0) {
          brow[db * 16 + 4 * g + r] = a;
          brow[H * 32 + db * 16 + 4 * g + r] = b;
          brow[2 * H * 32 + db * 16 + 4 * g + r] = c;
        }
      }
  }
  // partial d(bias): row wid / H of [nwaves / H][H][N][N]
  float* prow = dbias_part + ((int64_t)(wid / H) * H + h) * N * N;
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const int q = qb * 16 + i;
    if (q >= N) continue;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb * 16 + 4 * g + r;
        if (key < N) prow[q * N + key] = dsum[kb][qb][r];
      }
  }
}

// dQ + delta + dBias (per-window partial): wave per query row
template <typename T>
__global__ __launch_bounds__(256) void win_attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                             const float* __restrict__ lse, const float* __restrict__ bias,
                                                             const float* __restrict__ mask, T* __restrict__ dqkv,
                                                             float* __restrict__ delta, float* __restrict__ dbias_part,
                                                             WinGeom g, int H, int N, int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Qs = sm;
  float* Ks = Qs + 64 * (hd + 1);
  float* Vs = Ks + 64 * (hd + 1);
  float* dOs = Vs + 64 * (hd + 1);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t bw = blockIdx.x;
  const int h = blockIdx.y;
  const int64_t ts = (int64_t)3 * H * hd, os = (int64_t)H * hd;
  stage_win(Qs, qkv + h * hd, ts, g, bw, N, hd, tid);
  stage_win(Ks, qkv + (H + h) * hd, ts, g, bw, N, hd, tid);
  stage_win(Vs, qkv + (2 * H + h) * hd, ts, g, bw, N, hd, tid);
  stage_win(dOs, dout + h * hd, os, g, bw, N, hd, tid);
  __syncthreads();
  const float* mrow = mask ? mask + (bw % g.nW) * (int64_t)N * N : nullptr;
  for (int i = w; i < N; i += 4) {
    float s = 0.f, dp = 0.f, p = 0.f;
    if (lane < N) {
      for (int d = 0; d < hd; ++d) {
        s = fmaf(Qs[i * (hd + 1) + d] * scale, Ks[lane * (hd + 1) + d], s);
        dp = fmaf(dOs[i * (hd + 1) + d], Vs[lane * (hd + 1) + d], dp);
      }
      s += bias[((int64_t)h * N + i) * N + lane];
      if (mrow) s += mrow[i * N + lane];
      p = expf(s - lse[(bw * H + h) * N + i]);
    }
    const float del = wave_sum(p * dp);
    const float ds = p * (dp - del);
    if (lane < N) dbias_part[((bw * H + h) * N + i) * N + lane] = ds;
    if (lane == 0) delta[(bw * H + h) * N + i] = del;
    float acc = 0.f;
    const int dl = lane < hd ? lane : 0;
    for (int j = 0; j < N; ++j) acc = fmaf(__shfl(ds, j), Ks[j * (hd + 1) + dl], acc);
    if (lane < hd) dqkv[win_token(g, bw, i) * ts + h * hd + lane] = from_f32<T>(acc * scale);
  }
}

// dK, dV: wave per key row, lanes = queries
template <typename T>
__global__ __launch_bounds__(256) void win_attn_bwd_dkdv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               const float* __restrict__ bias, const float* __restrict__ mask,
                                                               T* __restrict__ dqkv, WinGeom g, int H, int N, int hd,
                                                               float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Qs = sm;
  float* Ks = Qs + 64 * (hd + 1);
  float* Vs = Ks + 64 * (hd + 1);
  float* dOs = Vs + 64 * (hd + 1);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t bw = blockIdx.x;
  const int h = blockIdx.y;
  const int64_t ts = (int64_t)3 * H * hd, os = (int64_t)H * hd;
  stage_win(Qs, qkv + h * hd, ts, g, bw, N, hd, tid);
  stage_win(Ks, qkv + (H + h) * hd, ts, g, bw, N, hd, tid);
  stage_win(Vs, qkv + (2 * H + h) * hd, ts, g, bw, N, hd, tid);
  stage_win(dOs, dout + h * hd, os, g, bw, N, hd, tid);
  __syncthreads();
  const float* mrow = mask ? mask + (bw % g.nW) * (int64_t)N * N : nullptr;
  for (int j = w; j < N; j += 4) {
    float p = 0.f, ds = 0.f;
    if (lane < N) {                                  // lane = query i
      float s = 0.f, dp = 0.f;
      for (int d = 0; d < hd; ++d) {
        s = fmaf(Qs[lane * (hd + 1) + d] * scale, Ks[j * (hd + 1) + d], s);
        dp = fmaf(dOs[lane * (hd + 1) + d], Vs[j * (hd + 1) + d], dp);
      }
      s += bias[((int64_t)h * N + lane) * N + j];
      if (mrow) s += mrow[lane * N + j];
      p = expf(s - lse[(bw * H + h) * N + lane]);
      ds = p * (dp - delta[(bw * H + h) * N + lane]);
    }
    float ak = 0.f, av = 0.f;
    const int dl = lane < hd ? lane : 0;
    for (int i = 0; i < N; ++i) {
      ak = fmaf(__shfl(ds, i), Qs[i * (hd + 1) + dl], ak);
      av = fmaf(__shfl(p, i), dOs[i * (hd + 1) + dl], av);
    }
    if (lane < hd) {
      T* row = dqkv + win_token(g, bw, j) * ts + h * hd + lane;
      row[H * hd] = from_f32<T>(ak * scale);
      row[2 * H * hd] = from_f32<T>(av);
    }
  }
}

// bias[h][i][j] = table[index[i*N + j]][h]
__global__ void relpos_gather_kernel(const float* __restrict__ table, const int64_t* __restrict__ index,
                                     float* __restrict__ bias, int H, int NN) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= H * NN) return;
  const int h = t / NN, ij = t % NN;
  bias[t] = table[index[ij] * H + h];
}
// dtable[t][h] = sum_{ij: index[ij]==t} dbias[h][ij]: one workgroup per table row t.  Phase 1: the threads scan
// the index ONCE and compact the matching positions (at most N of the N*N) into LDS in a FIXED slot order
// ((iteration, wave) slots, ballot-ranked inside a slot); phase 2: one thread per head sums its <= N values in that
// order (deterministic, no atomics).  (Round 2 scanned the whole index once per head and wave: 33 us per call.)
__global__ __launch_bounds__(256) void relpos_scatter_kernel(const float* __restrict__ dbias,
                                                            const int64_t* __restrict__ index,
                                                            float* __restrict__ dtable, int T, int H, int NN) {
  __shared__ int list[64 * 64];        // NN <= 4096: 16 iterations x 4 waves slots of up to 64 hits
  __shared__ int cnt[64];
  const int t = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int iters = (NN + 255) / 256;
  for (int it = 0; it < iters; ++it) {
    const int ij = it * 256 + threadIdx.x;
    const bool hit = ij < NN && index[ij] == t;
    const unsigned long long m = __ballot(hit);
    const int slot = it * 4 + w;
    if (hit) list[slot * 64 + __popcll(m & ((1ull << lane) - 1ull))] = ij;
    if (lane == 0) cnt[slot] = __popcll(m);
  }
  __syncthreads();
  for (int h = threadIdx.x; h < H; h += 256) {
    float s = 0.f;
    for (int slot = 0; slot < iters * 4; ++slot) {
      const int c = cnt[slot];
      for (int e = 0; e < c; ++e) s += dbias[(int64_t)h * NN + list[slot * 64 + e]];
    }
    dtable[(int64_t)t * H + h] = s;
  }
}

// PatchMerging gather (inverse = 0): out[b,(i,j), k*C + c] = x[b,(2i+dy_k, 2j+dx_k), c],
// k = 0..3 <-> (dy,dx) = (0,0),(1,0),(0,1),(1,1); inverse = 1 scatters back (a permutation)
template <typename T>
__global__ void patch_merge_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t B, int Hh, int Ww, int C,
                                   int inverse) {
  const int c4 = C / 4;
  const int64_t total = B * Hh * Ww * c4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(idx % c4);
    int64_t t = idx / c4;
    const int x = (int)(t % Ww); t /= Ww;
    const int y = (int)(t % Hh);
    const int64_t b = t / Hh;
    const int k = (y & 1) + 2 * (x & 1);
    const int64_t tok_full = (b * Hh + y) * Ww + x;
    const int64_t tok_half = (b * (Hh / 2) + (y >> 1)) * (Ww / 2) + (x >> 1);
    const T* s = inverse ? src + tok_half * 4 * C + k * C + cq * 4 : src + tok_full * C + cq * 4;
    T* d = inverse ? dst + tok_full * C + cq * 4 : dst + tok_half * 4 * C + k * C + cq * 4;
    store4<T>(d, load4<T>(s));
  }
}

// out[b][c] = mean_l x[b][l][c]  /  dx[b][l][c] = dout[b][c] / L
template <typename T>
__global__ void token_mean_fwd_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t B, int L, int C) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * C) return;
  const int64_t b = idx / C;
  const int c = (int)(idx % C);
  float s = 0.f;
  for (int l = 0; l < L; ++l) s += to_f32(x[(b * L + l) * C + c]);
  out[idx] = s / (float)L;
}
template <typename T>
__global__ void token_mean_bwd_kernel(const float* __restrict__ dout, T* __restrict__ dx, int64_t B, int L, int C) {
  const int64_t total = B * L * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const int64_t b = idx / ((int64_t)L * C);
    dx[idx] = from_f32<T>(dout[b * C + c] / (float)L);
  }
}

}  // namespace

static std::atomic<int> g_win_mfma{-1};   // diagnostic / test hook: 0 = fp32 vector kernels only, else MFMA where it applies
extern "C" void vitmi_debug_win_attn_mfma(int mode) { g_win_mfma = mode; }

static int win_check(int64_t Bw, int64_t H, int64_t N, int64_t hd, int64_t Himg, int64_t Wimg, int64_t ws, int64_t shift, const char* who) {
  VITMI_REQUIRE(Bw > 0 && H > 0 && H <= 65535, VITMI_E_BADARG, "%s: bad batch / heads", who);
  VITMI_REQUIRE(N == ws * ws && N <= 64 && hd >= 1 && hd <= 64, VITMI_E_SHAPE, "%s: window tokens %lld (<=64) / head dim %lld (<=64)", who, (long long)N, (long long)hd);
  VITMI_REQUIRE(Himg % ws == 0 && Wimg % ws == 0 && shift >= 0 && shift < ws, VITMI_E_SHAPE, "%s: resolution %lldx%lld not divisible by window %lld or bad shift", who, (long long)Himg, (long long)Wimg, (long long)ws);
  VITMI_REQUIRE(Bw % ((Himg / ws) * (Wimg / ws)) == 0, VITMI_E_SHAPE, "%s: window count not a multiple of windows per image", who);
  return 0;
}

extern "C" int vitmi_win_attn_fwd(const void* qkv, void* out, float* lse, const float* bias, const float* mask,
                                  int dtype, int64_t Bw, int64_t H, int64_t N, int64_t hd, int64_t Himg,
                                  int64_t Wimg, int64_t ws, int64_t shift, float scale, void* stream_) {
  VITMI_REQUIRE(qkv && out && lse && bias, VITMI_E_BADARG, "win_attn_fwd: null argument");
  int rc = win_check(Bw, H, N, hd, Himg, Wimg, ws, shift, "win_attn_fwd");
  if (rc) return rc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  WinGeom g{(int)Himg, (int)Wimg, (int)ws, (int)shift, (int)(Wimg / ws), (int)((Himg / ws) * (Wimg / ws))};
  if (dtype == VITMI_BF16 && hd == 32 && g_win_mfma != 0 && is_aligned(qkv, 16) && is_aligned(out, 8)) {
    const int64_t tasks = Bw * H;
    hipLaunchKernelGGL(win_attn_fwd_mfma_kernel, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, stream,
                       (const bf16*)qkv, (bf16*)out, lse, bias, mask, g, (int)H, (int)N, scale, tasks);
    return vitmi_check_launch("win_attn_fwd_mfma_kernel");
  }
  dim3 grid((unsigned)Bw, (unsigned)H);
  const size_t lds = 3 * 64 * (hd + 1) * sizeof(float);
  if (dtype == VITMI_BF16)
    hipLaunchKernelGGL((win_attn_fwd_kernel<bf16>), grid, dim3(256), lds, stream, (const bf16*)qkv, (bf16*)out, lse, bias, mask, g, (int)H, (int)N, (int)hd, scale);
  else if (dtype == VITMI_F32)
    hipLaunchKernelGGL((win_attn_fwd_kernel<float>), grid, dim3(256), lds, stream, (const float*)qkv, (float*)out, lse, bias, mask, g, (int)H, (int)N, (int)hd, scale);
  else return vitmi_fail(VITMI_E_DTYPE, "win_attn_fwd: bad dtype");
  return vitmi_check_launch("win_attn_fwd_kernel");
}

extern "C" size_t vitmi_win_attn_bwd_workspace(int64_t Bw, int64_t H, int64_t N) {
  // delta [Bw,H,N] + per-window (vector kernels) or per-wave (MFMA kernel, <= 1024 waves)
  // partial d(bias) tiles
  int64_t rows = Bw > 1024 / (H > 0 ? H : 1) + 1 ? Bw : 1024 / (H > 0 ? H : 1) + 1;
  return (size_t)(Bw * H * N) * sizeof(float) + (size_t)(rows * H * N * N) * sizeof(float) +
         (size_t)(1024 / (H > 0 ? H : 1) + 1) * 3 * H * 64 * sizeof(float);   // + qkv-bias partials
}

// dbias [H,N,N] fp32 (overwritten) = sum over windows of d(score)
extern "C" int vitmi_win_attn_bwd_fuses_qkv_bias(int dtype, int64_t hd) {
  return dtype == VITMI_BF16 && hd == 32 && g_win_mfma != 0 ? 1 : 0;
}

extern "C" int vitmi_win_attn_bwd(const void* qkv, const void* dout, const float* lse, const float* bias,
                                  const float* mask, void* dqkv, float* dbias, float* dqkv_bias, int dtype,
                                  int64_t Bw, int64_t H, int64_t N, int64_t hd, int64_t Himg, int64_t Wimg,
                                  int64_t ws, int64_t shift, float scale, void* workspace, size_t workspace_bytes,
                                  void* stream_) {
  VITMI_REQUIRE(qkv && dout && lse && bias && dqkv && dbias, VITMI_E_BADARG, "win_attn_bwd: null argument");
  int rc = win_check(Bw, H, N, hd, Himg, Wimg, ws, shift, "win_attn_bwd");
  if (rc) return rc;
  VITMI_REQUIRE(workspace && workspace_bytes >= vitmi_win_attn_bwd_workspace(Bw, H, N) && is_aligned(workspace, 16), VITMI_E_WORKSPACE, "win_attn_bwd: workspace too small");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  WinGeom g{(int)Himg, (int)Wimg, (int)ws, (int)shift, (int)(Wimg / ws), (int)((Himg / ws) * (Wimg / ws))};
  float* delta = reinterpret_cast<float*>(workspace);
  float* part = delta + Bw * H * N;
  if (dtype == VITMI_BF16 && hd == 32 && g_win_mfma != 0 && is_aligned(qkv, 16) && is_aligned(dout, 16) && is_aligned(dqkv, 8)) {
    // one head per wave, ~4 waves per CU; nwaves a multiple of H, at most one wave per task
    int64_t per_head = 1024 / H;
    if (per_head > Bw) per_head = Bw;
    if (per_head < 1) per_head = 1;
    const int nwaves = (int)(per_head * H);
    auto kern = win_attn_bwd_mfma_kernel;
    if ((rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(kern), 4 * WIN_BWD_LDS + 256, "win_attn_bwd"))) return rc;
    float* qb_part = dqkv_bias ? part + per_head * H * N * N : nullptr;     // [per_head][3*H*32]
    const int pf_dump = g_win_bwd_prefetch ? 4 * WIN_BWD_LDS : -1;           // LDS offset of the prefetch dump, -1 = off
    hipLaunchKernelGGL(kern, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 4 * WIN_BWD_LDS + 256, stream, (const bf16*)qkv,
                       (const bf16*)dout, lse, bias, mask, (bf16*)dqkv, part, qb_part, g, (int)H, (int)N, scale, Bw, nwaves, pf_dump);
    if ((rc = vitmi_check_launch("win_attn_bwd_mfma_kernel"))) return rc;
    if ((rc = vitmi_reduce_rows(part, (int)per_head, H * N * N, H * N * N, dbias, stream))) return rc;
    if (dqkv_bias) return vitmi_reduce_rows(qb_part, (int)per_head, 3 * H * 32, 3 * H * 32, dqkv_bias, stream);
    return 0;
  }
  VITMI_REQUIRE(!dqkv_bias, VITMI_E_DTYPE, "win_attn_bwd: dqkv_bias is produced by the bf16 hd = 32 kernel only (ask vitmi_win_attn_bwd_fuses_qkv_bias)");
  dim3 grid((unsigned)Bw, (unsigned)H);
  const size_t lds = 4 * 64 * (hd + 1) * sizeof(float);
  if (dtype == VITMI_BF16) {
    hipLaunchKernelGGL((win_attn_bwd_dq_kernel<bf16>), grid, dim3(256), lds, stream, (const bf16*)qkv, (const bf16*)dout, lse, bias, mask, (bf16*)dqkv, delta, part, g, (int)H, (int)N, (int)hd, scale);
    if ((rc = vitmi_check_launch("win_attn_bwd_dq_kernel"))) return rc;
    hipLaunchKernelGGL((win_attn_bwd_dkdv_kernel<bf16>), grid, dim3(256), lds, stream, (const bf16*)qkv, (const bf16*)dout, lse, delta, bias, mask, (bf16*)dqkv, g, (int)H, (int)N, (int)hd, scale);
  } else if (dtype == VITMI_F32) {
    hipLaunchKernelGGL((win_attn_bwd_dq_kernel<float>), grid, dim3(256), lds, stream, (const float*)qkv, (const float*)dout, lse, bias, mask, (float*)dqkv, delta, part, g, (int)H, (int)N, (int)hd, scale);
    if ((rc = vitmi_check_launch("win_attn_bwd_dq_kernel"))) return rc;
    hipLaunchKernelGGL((win_attn_bwd_dkdv_kernel<float>), grid, dim3(256), lds, stream, (const float*)qkv, (const float*)dout, lse, delta, bias, mask, (float*)dqkv, g, (int)H, (int)N, (int)hd, scale);
  } else return vitmi_fail(VITMI_E_DTYPE, "win_attn_bwd: bad dtype");
  if ((rc = vitmi_check_launch("win_attn_bwd_dkdv_kernel"))) return rc;
  // deterministic reduction of the per-window d(score) tiles: rows = windows, cols = H*N*N
  return vitmi_reduce_rows(part, (int)Bw, H * N * N, H * N * N, dbias, stream);
}

extern "C" void vitmi_debug_win_bwd_prefetch(int on) { g_win_bwd_prefetch = on; }

extern "C" int vitmi_relpos_bias(const float* table, const int64_t* index, float* bias, const float* dbias,
                                 float* dtable, int64_t T, int64_t H, int64_t N, void* stream_) {
  VITMI_REQUIRE(index && T > 0 && H > 0 && N > 0, VITMI_E_BADARG, "relpos_bias: bad argument");
  VITMI_REQUIRE(N <= 64, VITMI_E_SHAPE, "relpos_bias: windows of at most 64 tokens (N = %lld)", (long long)N);
  VITMI_REQUIRE((table && bias) || (dbias && dtable), VITMI_E_BADARG, "relpos_bias: need (table,bias) and/or (dbias,dtable)");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int NN = (int)(N * N);
  if (table && bias) {
    hipLaunchKernelGGL(relpos_gather_kernel, dim3((unsigned)((H * NN + 255) / 256)), dim3(256), 0, stream, table, index, bias, (int)H, NN);
    int rc = vitmi_check_launch("relpos_gather_kernel");
    if (rc) return rc;
  }
  if (dbias && dtable) {
    hipLaunchKernelGGL(relpos_scatter_kernel, dim3((unsigned)T), dim3(256), 0, stream, dbias, index, dtable, (int)T, (int)H, NN);
    return vitmi_check_launch("relpos_scatter_kernel");
  }
  return 0;
}

extern "C" int vitmi_patch_merge(const void* src, void* dst, int dtype, int64_t B, int64_t Hh, int64_t Ww, int64_t C,
                                 int inverse, void* stream_) {
  VITMI_REQUIRE(src && dst && B > 0 && Hh > 0 && Ww > 0 && C > 0, VITMI_E_BADARG, "patch_merge: bad argument");
  VITMI_REQUIRE(Hh % 2 == 0 && Ww % 2 == 0 && C % 4 == 0, VITMI_E_SHAPE, "patch_merge: H, W must be even and C a multiple of 4");
  VITMI_REQUIRE(is_aligned(src, 4 * dtype_size(dtype)) && is_aligned(dst, 4 * dtype_size(dtype)), VITMI_E_ALIGN, "patch_merge: alignment");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  int64_t blocks = (B * Hh * Ww * (C / 4) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (dtype == VITMI_BF16)
    hipLaunchKernelGGL((patch_merge_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)src, (bf16*)dst, B, (int)Hh, (int)Ww, (int)C, inverse);
  else if (dtype == VITMI_F32)
    hipLaunchKernelGGL((patch_merge_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)src, (float*)dst, B, (int)Hh, (int)Ww, (int)C, inverse);
  else return vitmi_fail(VITMI_E_DTYPE, "patch_merge: bad dtype");
  return vitmi_check_launch("patch_merge_kernel");
}

extern "C" int vitmi_token_mean(const void* x, float* out, const float* dout, void* dx, int dtype, int64_t B,
                                int64_t L, int64_t C, void* stream_) {
  VITMI_REQUIRE(B > 0 && L > 0 && C > 0 && ((x && out) || (dout && dx)), VITMI_E_BADARG, "token_mean: bad argument");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (x && out) {
    dim3 grid((unsigned)((B * C + 255) / 256));
    if (dtype == VITMI_BF16) hipLaunchKernelGGL((token_mean_fwd_kernel<bf16>), grid, dim3(256), 0, stream, (const bf16*)x, out, B, (int)L, (int)C);
    else if (dtype == VITMI_F32) hipLaunchKernelGGL((token_mean_fwd_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, out, B, (int)L, (int)C);
    else return vitmi_fail(VITMI_E_DTYPE, "token_mean: bad dtype");
    int rc = vitmi_check_launch("token_mean_fwd_kernel");
    if (rc) return rc;
  }
  if (dout && dx) {
    int64_t blocks = (B * L * C + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (dtype == VITMI_BF16) hipLaunchKernelGGL((token_mean_bwd_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, dout, (bf16*)dx, B, (int)L, (int)C);
    else if (dtype == VITMI_F32) hipLaunchKernelGGL((token_mean_bwd_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, dout, (float*)dx, B, (int)L, (int)C);
    else return vitmi_fail(VITMI_E_DTYPE, "token_mean: bad dtype");
    return vitmi_check_launch("token_mean_bwd_kernel");
  }
  return 0;
}

// every diagnostic switch of this file back to its default (vitmi_debug_reset, core.cpp)
void vitmi_debug_reset_swin() {
  g_win_bwd_prefetch = 1;
  g_win_mfma = -1;
}
