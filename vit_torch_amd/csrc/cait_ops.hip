// CaiT-specific kernels:
//  * talking-heads softmax (models/cait.py:118-122): S' = proj_l(S) over the head axis,
//    P = softmax_j(S'), P' = proj_w(P) — one wave per (image, query row), all H heads of
//    that row in registers, so the two HxH mixes wrap the softmax without any of the
//    permute(0,2,3,1) round trips of the reference;
//  * class attention core (models/cait.py:38-52): one query (the CLS token) against all
//    tokens, one wave per (image, head);
//  * column sum of an element-wise product (LayerScale gradient).
// Correctness-first versions: the score tensors S, P, P' live in HBM between the batched
// MFMA products (gemm.hip, batched form) and these kernels.  H <= 8, row length <= 256.
#include <atomic>
#include "common.h"

namespace {

constexpr int TH_MAXH = 8;
constexpr int TH_MAXC = 4;     // columns per lane -> row length <= 256

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return to_f32(*p); }

// A lane owns TH_MAXC keys of a score row.  VEC: keys 4*lane .. 4*lane+3 (one 8/16-byte
// access per head and tensor; needs ld % 4 == 0 and 4-element aligned bases) — otherwise
// keys lane, lane+64, ... by element.  2-byte accesses made these kernels texture-address
// bound (~10x their arithmetic); the vector form is what the engine uses.
// The H x H mixing weights as wave-uniform values WITHOUT a memory access per use: lane k of
// one VGPR holds W[k] (H*H <= 64) and v_readlane brings an entry into a scalar register.
// (Indexing the global array inside the unrolled FMA nests put a scalar load and its wait in
// front of every FMA: the backward kernel ran at ~1/10 of its arithmetic.)
__device__ __forceinline__ float th_wlane(float v, int idx) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), idx));
}
// The holder VGPR keeps the matrix in the PADDED 8 x 8 layout (lane 8a + c = W[a][c], zero
// outside H x H), so every readlane has a compile-time lane and no branch on H.
__device__ __forceinline__ float th_wholder(const float* W, int lane, int H) {
  const int a = lane >> 3, c = lane & 7;
  const float v = W[min(a, H - 1) * H + min(c, H - 1)];        // unconditional, clamped
  return (a < H && c < H) ? v : 0.f;
}
__device__ __forceinline__ float th_bholder(const float* b, int lane, int H) {
  const float v = b[min(lane, H - 1)];
  return lane < H ? v : 0.f;
}
#define TH_LOAD_W(dst, src_vgpr, H_)                                                    \
  float dst[TH_MAXH][TH_MAXH];                                                          \
  _Pragma("unroll") for (int a_ = 0; a_ < TH_MAXH; ++a_)                                \
    _Pragma("unroll") for (int c_ = 0; c_ < TH_MAXH; ++c_)                              \
      dst[a_][c_] = th_wlane(src_vgpr, a_ * TH_MAXH + c_)

template <bool VEC> __device__ __forceinline__ int th_key(int lane, int c) { return VEC ? lane * 4 + c : c * 64 + lane; }

template <typename T> struct Keep4;                 // 4 keys of one head, compact in registers
template <> struct Keep4<bf16> {
  bf16x4 v;
  __device__ __forceinline__ float get(int c) const { return (float)v[c]; }
  __device__ __forceinline__ void set(int c, float x) { v[c] = (bf16)x; }
};
template <> struct Keep4<float> {
  f32x4 v;
  __device__ __forceinline__ float get(int c) const { return v[c]; }
  __device__ __forceinline__ void set(int c, float x) { v[c] = x; }
};
template <typename T, bool VEC>
__device__ __forceinline__ Keep4<T> th_load(const T* row, int lane, int Nk, bool head_ok) {
  Keep4<T> k;
  if constexpr (VEC) {
    // UNCONDITIONAL load from a clamped position, zeroed by selects afterwards: a branch
    // around a load makes hipcc wait for every load separately (48 HBM round trips per score
    // row; the caller already clamps the head index)
    const int lc = min(lane * 4, (Nk - 1) / 4 * 4);
    if constexpr (sizeof(T) == 2) k.v = *reinterpret_cast<const bf16x4*>(row + lc);
    else k.v = *reinterpret_cast<const f32x4*>(row + lc);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (!head_ok || lane * 4 + c >= Nk) k.set(c, 0.f);     // pad columns may hold anything
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) k.set(c, 0.f);
    if (!head_ok) return k;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c * 64 + lane < Nk) k.set(c, to_f32(row[c * 64 + lane]));
  }
  return k;
}
template <typename T, bool VEC>
__device__ __forceinline__ void th_store(T* row, int lane, int Nk, const Keep4<T>& k) {
  if constexpr (VEC) {
    if (lane * 4 < Nk) {                            // pad columns inside the vector get zeros
      Keep4<T> o = k;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (lane * 4 + c >= Nk) o.set(c, 0.f);
      if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x4*>(row + lane * 4) = o.v;
      else *reinterpret_cast<f32x4*>(row + lane * 4) = o.v;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c * 64 + lane < Nk) row[c * 64 + lane] = from_f32<T>(k.get(c));
  }
}

// ---- forward: S[B,H,N,ld] -> P (softmax of mixed scores), Pm (mixed probabilities)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void th_softmax_fwd_kernel(const T* __restrict__ S, const float* __restrict__ Wl,
                                                            const float* __restrict__ bl, const float* __restrict__ Ww,
                                                            const float* __restrict__ bw, T* __restrict__ P,
                                                            T* __restrict__ Pm, int64_t rows /*B*N*/, int H, int N,
                                                            int Nk, int ld) {
  static_assert(TH_MAXC == 4, "a lane owns four keys");
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = (int64_t)blockIdx.x * 4 + w;          // wave-uniform: the row arithmetic stays scalar
  if (row >= rows) return;
  const int64_t b = row / N, i = row % N;
  const float wl_v = th_wholder(Wl, lane, H), ww_v = th_wholder(Ww, lane, H);
  const float bl_v = th_bholder(bl, lane, H), bw_v = th_bholder(bw, lane, H);
  // bf16 scores: hardware exp2 / rcp (P is stored in bf16); the fp32 parity form keeps expf and /
  constexpr bool FAST = sizeof(T) == 2;
  float s[TH_MAXH][TH_MAXC], p[TH_MAXH][TH_MAXC];
#pragma unroll
  for (int h = 0; h < TH_MAXH; ++h) {
    const Keep4<T> k = th_load<T, VEC>(S + ((b * H + (h < H ? h : 0)) * N + i) * ld, lane, Nk, h < H);
#pragma unroll
    for (int c = 0; c < TH_MAXC; ++c) s[h][c] = k.get(c);
  }
  TH_LOAD_W(wl, wl_v, H);
  // All TH_MAXH heads go through the same straight-line code (rows of Wl / Ww beyond H are
  // zero, so a surplus head is a softmax of zeros that nothing reads): with a `break` at H
  // the eight dependent reduction chains ran one after the other behind branches; like this
  // the scheduler interleaves them.  Only the stores are guarded.
#pragma unroll
  for (int hp = 0; hp < TH_MAXH; ++hp) {
    const float blh = th_wlane(bl_v, hp);
    float v[TH_MAXC], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < TH_MAXC; ++c) {
      float a = blh;
#pragma unroll
      for (int h = 0; h < TH_MAXH; ++h) a = fmaf(wl[hp][h], s[h][c], a);
      v[c] = (th_key<VEC>(lane, c) < Nk) ? a : -INFINITY;
      mx = fmaxf(mx, v[c]);
    }
    mx = wave_max_dpp(mx);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < TH_MAXC; ++c) {
      v[c] = FAST ? __builtin_amdgcn_exp2f((v[c] - mx) * 1.4426950408889634f) : expf(v[c] - mx);
      sum += v[c];
    }
    sum = wave_sum_dpp(sum);
    const float inv = FAST ? __builtin_amdgcn_rcpf(sum) : 1.f / sum;
    Keep4<T> o;
#pragma unroll
    for (int c = 0; c < TH_MAXC; ++c) {
      o.set(c, v[c] * inv);
      p[hp][c] = o.get(c);                          // backward differentiates at the STORED probabilities
    }
    if (hp < H) th_store<T, VEC>(P + ((b * H + hp) * N + i) * ld, lane, Nk, o);
  }
  __builtin_amdgcn_sched_barrier(0);                // Wl's 64 scalars are dead before Ww's are fetched
  TH_LOAD_W(ww, ww_v, H);
#pragma unroll
  for (int ho = 0; ho < TH_MAXH; ++ho) {
    if (ho >= H) break;
    const float bwh = th_wlane(bw_v, ho);
    Keep4<T> o;
#pragma unroll
    for (int c = 0; c < TH_MAXC; ++c) {
      float a = bwh;
#pragma unroll
      for (int hp = 0; hp < TH_MAXH; ++hp) a = fmaf(ww[ho][hp], p[hp][c], a);
      o.set(c, a);
    }
    th_store<T, VEC>(Pm + ((b * H + ho) * N + i) * ld, lane, Nk, o);
  }
}

// ---- backward.  part layout per workgroup: [dWl H*H | dbl H | dWw H*H | dbw H].
// The four parameter gradients are sums over every score position of an outer product of two
// H-vectors: dWw[ho][hp] = sum dP'[ho] P[hp], dWl[hp][h] = sum dS'[hp] S[h].  Two forms:
//  * MF (bf16, 16-B-aligned rows): they are [H x keys] x [keys x H] products, so each row's
//    keys go through v_mfma_f32_16x16x32_bf16 — A = the 8 head rows of dP' (straight from
//    global memory) or dS' (through a wave-private LDS tile), B = the head rows of P / S plus
//    a row of ones whose product is the bias gradient.  Two 16x16 accumulator tiles replace
//    144 per-lane accumulators: the kernel drops from 256 VGPRs (two waves per SIMD) and
//    loses 512 FMAs per row and the whole cross-lane fold.
//  * otherwise (fp32 parity mode, unaligned rows): per-lane fp32 accumulators of every
//    element, folded at the end by a transposing butterfly.
constexpr int TH_DSP = 2 * 64 * TH_MAXC + 16;      // pitch of a dS' tile row (bytes): 8 rows on 8 bank groups
constexpr int TH_KSTEPS = 64 * TH_MAXC / 32;       // 32-key MFMA steps of the longest row
__device__ __forceinline__ bf16x8 th_frag(const bf16* row, int key0, int Nk, bool head_ok) {
  // 8 consecutive keys of one head row; keys >= Nk (pad columns may hold anything) and
  // surplus heads read as zero.  key0 % 8 == 0 and ld % 8 == 0, so key0 < Nk stays inside the row.
  bf16x8 v = *reinterpret_cast<const bf16x8*>(row + (key0 < Nk ? key0 : 0));
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (!head_ok || key0 + e >= Nk) v[e] = (bf16)0.f;
  return v;
}
__device__ __forceinline__ bf16x8 th_ones(int key0, int Nk) {
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = key0 + e < Nk ? (bf16)1.f : (bf16)0.f;
  return v;
}
template <typename T, bool VEC, bool MF>
__global__ __launch_bounds__(256) void th_softmax_bwd_kernel(const T* __restrict__ S, const T* __restrict__ P,
                                                            const T* __restrict__ dPm, const float* __restrict__ Wl,
                                                            const float* __restrict__ Ww, T* __restrict__ dS,
                                                            float* __restrict__ part, int64_t rows, int H, int N,
                                                            int Nk, int ld) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float wl_v = th_wholder(Wl, lane, H), ww_v = th_wholder(Ww, lane, H);
  constexpr int NA = MF ? 1 : TH_MAXH;             // the per-lane accumulators exist in the FMA form only
  float aWl[NA][NA], aWw[NA][NA], abl[NA], abw[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    abl[a] = 0.f; abw[a] = 0.f;
#pragma unroll
    for (int c = 0; c < NA; ++c) { aWl[a][c] = 0.f; aWw[a][c] = 0.f; }
  }
  f32x4 tWw = {0.f, 0.f, 0.f, 0.f}, tWl = {0.f, 0.f, 0.f, 0.f};   // MF: D[m][n], lane = (n = lane & 15, rows 4 (lane >> 4) + r)
  __shared__ __attribute__((aligned(16))) char dstile[MF ? 4 * TH_MAXH * TH_DSP : 16];
  const int fi = lane & 15, fg = lane >> 4;        // fragment role: head row fi, keys 32 t + 8 fg ..

  // Two passes per row with the row kept COMPACT (Keep4: the 96 values of S, P, dP' for 8
  // heads would otherwise push the 144 gradient accumulators out of the register file).
  // Pass 1: dP = Ww^T dP', the softmax row dots, dWw / dbw.  Pass 2 re-reads P (L2) and S:
  // dS' = P (dP - dot), dS = Wl^T dS', dWl / dbl.  dP stays in registers between the passes.
  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < rows; row += (int64_t)gridDim.x * 4) {
    const int64_t b = row / N, i = row % N;
    float dp[TH_MAXH][TH_MAXC], dot[TH_MAXH];
#pragma unroll
    for (int hp = 0; hp < TH_MAXH; ++hp) dot[hp] = 0.f;
    {
      Keep4<T> pk[TH_MAXH], gk[TH_MAXH];
#pragma unroll
      for (int h = 0; h < TH_MAXH; ++h) {
        const int64_t o = ((b * H + (h < H ? h : 0)) * N + i) * ld;
        pk[h] = th_load<T, VEC>(P + o, lane, Nk, h < H);
        gk[h] = th_load<T, VEC>(dPm + o, lane, Nk, h < H);      // dL/dP'
      }
      TH_LOAD_W(ww, ww_v, H);
#pragma unroll
      for (int c = 0; c < TH_MAXC; ++c) {
        float p[TH_MAXH], g[TH_MAXH];
#pragma unroll
        for (int h = 0; h < TH_MAXH; ++h) { p[h] = pk[h].get(c); g[h] = gk[h].get(c); }
        // through proj_w: dP[hp] = sum_ho Ww[ho][hp] dP'[ho];  dWw[ho][hp] += dP'[ho] * P[hp]
#pragma unroll
        for (int hp = 0; hp < TH_MAXH; ++hp) {
          float a = 0.f;
#pragma unroll
          for (int ho = 0; ho < TH_MAXH; ++ho) a = fmaf(ww[ho][hp], g[ho], a);
          dp[hp][c] = a;
          dot[hp] = fmaf(a, p[hp], dot[hp]);
        }
        if constexpr (!MF) {
#pragma unroll
          for (int ho = 0; ho < TH_MAXH; ++ho) {
            abw[ho] += g[ho];
#pragma unroll
            for (int hp = 0; hp < TH_MAXH; ++hp) aWw[ho][hp] = fmaf(g[ho], p[hp], aWw[ho][hp]);
          }
        }
      }
    }
    if constexpr (MF) {     // dWw[ho][hp] += sum_key dP'[ho][key] P[hp][key]; column 8 of B = ones -> dbw
      // every step's loads are unconditional (a step beyond Nk loads key 0 and zeroes it):
      // one batch of loads, not a round trip per step
      const int64_t o = ((b * H + (fi < H ? fi : 0)) * N + i) * ld;
      bf16x8 af[TH_KSTEPS], bf[TH_KSTEPS];
#pragma unroll
      for (int t = 0; t < TH_KSTEPS; ++t) {
        const int key0 = 32 * t + 8 * fg;
        af[t] = th_frag(reinterpret_cast<const bf16*>(dPm) + o, key0, Nk, fi < H);
        bf[t] = th_frag(reinterpret_cast<const bf16*>(P) + o, key0, Nk, fi < H);
        if (fi == TH_MAXH) bf[t] = th_ones(key0, Nk);
      }
#pragma unroll
      for (int t = 0; t < TH_KSTEPS; ++t) tWw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], bf[t], tWw, 0, 0, 0);
    }
#pragma unroll
    for (int hp = 0; hp < TH_MAXH; ++hp) dot[hp] = wave_sum_dpp(dot[hp]);
    {
      Keep4<T> pk[TH_MAXH], sk[TH_MAXH], ok[TH_MAXH];
      bf16x4 dsk[MF ? TH_MAXH : 1];                 // MF: dS' of this lane's four keys, per head
#pragma unroll
      for (int h = 0; h < TH_MAXH; ++h) {
        const int64_t o = ((b * H + (h < H ? h : 0)) * N + i) * ld;
        pk[h] = th_load<T, VEC>(P + o, lane, Nk, h < H);
        sk[h] = th_load<T, VEC>(S + o, lane, Nk, h < H);
      }
      TH_LOAD_W(wl, wl_v, H);
#pragma unroll
      for (int c = 0; c < TH_MAXC; ++c) {
        float sv[TH_MAXH], ds[TH_MAXH];
        // through the softmax: dS'[hp] = P[hp] * (dP[hp] - sum_j dP[hp] P[hp])
#pragma unroll
        for (int hp = 0; hp < TH_MAXH; ++hp) {
          ds[hp] = pk[hp].get(c) * (dp[hp][c] - dot[hp]);
          sv[hp] = sk[hp].get(c);
        }
        // through proj_l: dS[h] = sum_hp Wl[hp][h] dS'[hp];  dWl[hp][h] += dS'[hp] * S[h]
        if constexpr (!MF) {
#pragma unroll
          for (int hp = 0; hp < TH_MAXH; ++hp) {
            abl[hp] += ds[hp];
#pragma unroll
            for (int h = 0; h < TH_MAXH; ++h) aWl[hp][h] = fmaf(ds[hp], sv[h], aWl[hp][h]);
          }
        } else {
#pragma unroll
          for (int hp = 0; hp < TH_MAXH; ++hp) dsk[hp][c] = (bf16)ds[hp];
        }
#pragma unroll
        for (int h = 0; h < TH_MAXH; ++h) {
          float a = 0.f;
#pragma unroll
          for (int hp = 0; hp < TH_MAXH; ++hp) a = fmaf(wl[hp][h], ds[hp], a);
          ok[h].set(c, a);
        }
      }
#pragma unroll
      for (int h = 0; h < TH_MAXH; ++h)
        if (h < H) th_store<T, VEC>(dS + ((b * H + h) * N + i) * ld, lane, Nk, ok[h]);
      if constexpr (MF) {   // dS' -> the wave's tile [head][key] (VEC: this lane's keys are 4 lane .. 4 lane + 3)
#pragma unroll
        for (int hp = 0; hp < TH_MAXH; ++hp)
          *reinterpret_cast<bf16x4*>(dstile + (w * TH_MAXH + hp) * TH_DSP + lane * 8) = dsk[hp];
      }
    }
    if constexpr (MF) {     // dWl[hp][h] += sum_key dS'[hp][key] S[h][key]
      asm volatile("" ::: "memory");                // the tile stores above are not to move below these reads
      const int64_t o = ((b * H + (fi < H ? fi : 0)) * N + i) * ld;      // (same-wave LDS operations execute in order)
      bf16x8 af[TH_KSTEPS], bf[TH_KSTEPS];
#pragma unroll
      for (int t = 0; t < TH_KSTEPS; ++t) {
        const int key0 = 32 * t + 8 * fg;
        af[t] = *reinterpret_cast<const bf16x8*>(dstile + (w * TH_MAXH + (fi & (TH_MAXH - 1))) * TH_DSP + key0 * 2);
        if (fi >= TH_MAXH) {
#pragma unroll
          for (int e = 0; e < 8; ++e) af[t][e] = (bf16)0.f;
        }
        bf[t] = th_frag(reinterpret_cast<const bf16*>(S) + o, key0, Nk, fi < H);
        if (fi == TH_MAXH) bf[t] = th_ones(key0, Nk);
      }
#pragma unroll
      for (int t = 0; t < TH_KSTEPS; ++t) tWl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], bf[t], tWl, 0, 0, 0);
      asm volatile("" ::: "memory");                // ... nor the next row's stores above these reads
    }
  }
  // ---- 144 per-lane accumulators -> one partial row per wave.  A plain wave_sum per value
  // is 6 dependent shuffles each (864 ds_bpermute, ~45 us per wave: more than the rows
  // themselves); the transposing butterfly halves the value count at each of the first four
  // lane bits (72 + 36 + 18 + 9 shuffles), then two plain steps fold the last 9 values.
  constexpr int NV = 2 * TH_MAXH * TH_MAXH + 2 * TH_MAXH;     // 144, padded layout
  __shared__ float wred[4][NV];
  if constexpr (MF) {
    // the accumulator tiles ARE the sums over the wave's keys: rows m = 4 fg + r (heads of the
    // A operand), column fi (heads of B, column 8 = the ones row)
    constexpr int HH = TH_MAXH * TH_MAXH;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * fg + r;
      if (m < TH_MAXH && fi < TH_MAXH) {
        wred[w][m * TH_MAXH + fi] = tWl[r];
        wred[w][HH + TH_MAXH + m * TH_MAXH + fi] = tWw[r];
      } else if (m < TH_MAXH && fi == TH_MAXH) {
        wred[w][HH + m] = tWl[r];
        wred[w][2 * HH + TH_MAXH + m] = tWw[r];
      }
    }
  } else {
  float v[NV];
#pragma unroll
  for (int a = 0; a < TH_MAXH; ++a) {
#pragma unroll
    for (int c = 0; c < TH_MAXH; ++c) {
      v[a * TH_MAXH + c] = aWl[a][c];
      v[TH_MAXH * TH_MAXH + TH_MAXH + a * TH_MAXH + c] = aWw[a][c];
    }
    v[TH_MAXH * TH_MAXH + a] = abl[a];
    v[2 * TH_MAXH * TH_MAXH + TH_MAXH + a] = abw[a];
  }
  int base = 0;
#pragma unroll
  for (int sbit = 0; sbit < 4; ++sbit) {
    const int cnt = NV >> (sbit + 1);                // 72, 36, 18, 9
    const bool up = (lane >> sbit) & 1;
#pragma unroll
    for (int k = 0; k < cnt; ++k) {
      float lo = v[k], hi = v[k + cnt];
      asm volatile("" : "+v"(lo), "+v"(hi));         // keep the selects from becoming a dynamic index
      const float keep = up ? hi : lo, send = up ? lo : hi;
      v[k] = keep + __shfl_xor(send, 1 << sbit, 64);
    }
    base += up ? cnt : 0;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    v[k] += __shfl_xor(v[k], 16, 64);
    v[k] += __shfl_xor(v[k], 32, 64);
  }
  if (lane < 16) {
#pragma unroll
    for (int k = 0; k < 9; ++k) wred[w][base + k] = v[k];       // index in the padded layout
  }
  }
  // the four waves of the workgroup fold through LDS (fixed order) into ONE partial row per
  // workgroup: the row fold that follows then walks 512 rows, not 2048
  __syncthreads();
  const int stride = 2 * H * H + 2 * H;
  float* prow = part + (int64_t)blockIdx.x * stride;
  const int idx = threadIdx.x;
  if (idx < NV) {
    const float t = (wred[0][idx] + wred[1][idx]) + (wred[2][idx] + wred[3][idx]);
    constexpr int HH = TH_MAXH * TH_MAXH;
    if (idx < HH) {
      const int a = idx / TH_MAXH, c = idx % TH_MAXH;
      if (a < H && c < H) prow[a * H + c] = t;
    } else if (idx < HH + TH_MAXH) {
      const int a = idx - HH;
      if (a < H) prow[H * H + a] = t;
    } else if (idx < 2 * HH + TH_MAXH) {
      const int a = (idx - HH - TH_MAXH) / TH_MAXH, c = (idx - HH - TH_MAXH) % TH_MAXH;
      if (a < H && c < H) prow[H * H + H + a * H + c] = t;
    } else {
      const int a = idx - 2 * HH - TH_MAXH;
      if (a < H) prow[2 * H * H + H + a] = t;
    }
  }
}

inline int th_bwd_blocks(int64_t rows) {
  int64_t b = (rows + 3) / 4;
  return (int)(b < 512 ? b : 512);      // one partial row per workgroup: keep the fold short (768 and 1024 workgroups with the
                                        // register bound lowered to three / four waves per SIMD measured slower: 86 / 99 vs 71 us)
}

// ---- class attention: one wave per (b, h); q [B, H*hd] (already scaled by the caller via
// `scale`), k/v rows at token stride ts, p_save [B,H,N] fp32
template <typename T>
__global__ __launch_bounds__(256) void class_attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                            const T* __restrict__ v, int64_t ts, T* __restrict__ out,
                                                            float* __restrict__ psave, int64_t BH, int H, int N,
                                                            int hd, float scale) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t bh = (int64_t)blockIdx.x * 4 + w;
  if (bh >= BH) return;
  const int64_t b = bh / H;
  const int h = (int)(bh % H);
  const T* qv = q + b * H * hd + h * hd;
  float sc[TH_MAXC], mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < TH_MAXC; ++c) {
    const int j = c * 64 + lane;
    float a = -INFINITY;
    if (j < N) {
      const T* kr = k + (b * N + j) * ts + h * hd;
      a = 0.f;
      for (int d = 0; d < hd; ++d) a = fmaf(to_f32(qv[d]) * scale, to_f32(kr[d]), a);
    }
    sc[c] = a;
    mx = fmaxf(mx, a);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < TH_MAXC; ++c) { sc[c] = expf(sc[c] - mx); sum += sc[c]; }
  sum = wave_sum(sum);
#pragma unroll
  for (int c = 0; c < TH_MAXC; ++c) {
    sc[c] /= sum;
    const int j = c * 64 + lane;
    if (j < N) psave[bh * N + j] = sc[c];
  }
  // out[d] = sum_j p_j v[j][d]: lane = d (hd <= 64), p_j broadcast
  float acc = 0.f;
  const int dl = lane < hd ? lane : 0;
#pragma unroll
  for (int c = 0; c < TH_MAXC; ++c)
    for (int jj = 0; jj < 64; ++jj) {
      const int j = c * 64 + jj;
      if (j >= N) break;
      acc = fmaf(__shfl(sc[c], jj), to_f32(v[(b * N + j) * ts + h * hd + dl]), acc);
    }
  if (lane < hd) out[b * H * hd + h * hd + lane] = from_f32<T>(acc);
}

// backward: dq [B,H*hd] (fp32), dk/dv rows [B,N,*] at stride ts
template <typename T>
__global__ __launch_bounds__(256) void class_attn_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                            const T* __restrict__ v, int64_t ts,
                                                            const T* __restrict__ dout, const float* __restrict__ psave,
                                                            T* __restrict__ dq, T* __restrict__ dk, T* __restrict__ dv,
                                                            int64_t dts, int64_t BH, int H, int N, int hd, float scale) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t bh = (int64_t)blockIdx.x * 4 + w;
  if (bh >= BH) return;
  const int64_t b = bh / H;
  const int h = (int)(bh % H);
  const T* qv = q + b * H * hd + h * hd;
  const T* dov = dout + b * H * hd + h * hd;
  // dp_j = dout . v_j ; ds_j = p_j (dp_j - sum p dp)
  float p[TH_MAXC], ds[TH_MAXC], dot = 0.f;
#pragma unroll
  for (int c = 0; c < TH_MAXC; ++c) {
    const int j = c * 64 + lane;
    p[c] = 0.f; ds[c] = 0.f;
    if (j < N) {
      p[c] = psave[bh * N + j];
      const T* vr = v + (b * N + j) * ts + h * hd;
      float a = 0.f;
      for (int d = 0; d < hd; ++d) a = fmaf(to_f32(dov[d]), to_f32(vr[d]), a);
      ds[c] = a;
      dot = fmaf(p[c], a, dot);
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int c = 0; c < TH_MAXC; ++c) ds[c] = p[c] * (ds[c] - dot);
  // dv_j = p_j dout ; dk_j = ds_j * scale * q   (rows written by the lane that owns key j)
#pragma unroll
  for (int c = 0; c < TH_MAXC; ++c) {
    const int j = c * 64 + lane;
    if (j < N) {
      T* dvr = dv + (b * N + j) * dts + h * hd;
      T* dkr = dk + (b * N + j) * dts + h * hd;
      for (int d = 0; d < hd; ++d) {
        dvr[d] = from_f32<T>(p[c] * to_f32(dov[d]));
        dkr[d] = from_f32<T>(ds[c] * scale * to_f32(qv[d]));
      }
    }
  }
  // dq[d] = scale * sum_j ds_j k[j][d]
  float acc = 0.f;
  const int dl = lane < hd ? lane : 0;
#pragma unroll
  for (int c = 0; c < TH_MAXC; ++c)
    for (int jj = 0; jj < 64; ++jj) {
      const int j = c * 64 + jj;
      if (j >= N) break;
      acc = fmaf(__shfl(ds[c], jj), to_f32(k[(b * N + j) * ts + h * hd + dl]), acc);
    }
  if (lane < hd) dq[b * H * hd + h * hd + lane] = from_f32<T>(acc * scale);
}

// ---- class attention, bf16 with hd % 8 == 0 (round 4): the kernels above walk a key row element by element (96 two-byte
// loads and stores per lane and row: 127 / 248 us per cait_S24_224 layer for 77 MB of k / v).  Here EIGHT lanes share a key
// row, lane & 7 = its 16-byte piece of the head's hd elements (pieces >= hd / 8 idle), lane >> 3 = one of 8 rows per step:
// every access of k, v, dk, dv is a 16-byte piece of a contiguous hd x 2-byte run; a row's dot product is three xor-shuffles.
constexpr int CA_STEPS = 64 * TH_MAXC / 8;         // 32 steps of 8 rows: N <= 256
__device__ __forceinline__ void ca_fence() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void ca_unpack(const bf16x8& v, float (&f)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
}
__device__ __forceinline__ float ca_row_sum(float a) {          // over the 8 lanes of a row
  a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
  return a;
}
__device__ __forceinline__ float ca_rows_sum(float a) {         // over the 8 rows of a step (same piece)
  a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
  return a;
}
__global__ __launch_bounds__(256) void class_attn_fwd_vec_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                                const bf16* __restrict__ v, int64_t ts, bf16* __restrict__ out,
                                                                float* __restrict__ psave, int64_t BH, int H, int N, int hd, float scale) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t bh = (int64_t)blockIdx.x * 4 + w;
  if (bh >= BH) return;
  const int64_t b = bh / H;
  const int h = (int)(bh % H);
  const int piece = lane & 7, rsub = lane >> 3;
  const bool pok = piece * 8 < hd;
  const int pc = pok ? piece : 0;                                // idle pieces read piece 0 and contribute zero
  float qf[8];
  ca_unpack(*reinterpret_cast<const bf16x8*>(q + b * H * hd + h * hd + pc * 8), qf);
#pragma unroll
  for (int e = 0; e < 8; ++e) qf[e] = pok ? qf[e] * scale : 0.f;
  const bf16* kb = k + b * N * ts + h * hd + pc * 8;
  const bf16* vb = v + b * N * ts + h * hd + pc * 8;
  // a batch = 8 steps whose 8 loads are issued together, unconditionally, from clamped rows (one round trip per batch: a
  // branch around each load would cost one per step); whole batches beyond N are skipped (wave-uniform)
  float sc[CA_STEPS], mx = -INFINITY;
#pragma unroll
  for (int s0 = 0; s0 < CA_STEPS; s0 += 8) {
    ca_fence();                                                  // keep the batches apart (hipcc hoists all 32 loads otherwise)
    bf16x8 raw[8];
    if (s0 * 8 < N) {
#pragma unroll
      for (int u = 0; u < 8; ++u) raw[u] = *reinterpret_cast<const bf16x8*>(kb + (int64_t)min((s0 + u) * 8 + rsub, N - 1) * ts);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u, j = s * 8 + rsub;
      sc[s] = -INFINITY;
      if (s0 * 8 < N) {
        float kf[8];
        ca_unpack(raw[u], kf);
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a = fmaf(qf[e], kf[e], a);
        a = ca_row_sum(a);
        sc[s] = j < N ? a : -INFINITY;
        mx = fmaxf(mx, sc[s]);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 8, 64)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < CA_STEPS; ++s) { sc[s] = expf(sc[s] - mx); sum += sc[s]; }     // exp(-inf) = 0 for the padding
  sum = ca_rows_sum(sum);
  const float rs = 1.f / sum;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int s0 = 0; s0 < CA_STEPS; s0 += 8) {
    ca_fence();                                                  // keep the batches apart (hipcc hoists all 32 loads otherwise)
    if (s0 * 8 < N) {
      bf16x8 raw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) raw[u] = *reinterpret_cast<const bf16x8*>(vb + (int64_t)min((s0 + u) * 8 + rsub, N - 1) * ts);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + u, j = s * 8 + rsub;
        const float p = sc[s] * rs;                                // 0 for j >= N
        if (j < N && piece == 0) psave[bh * N + j] = p;
        float vf[8];
        ca_unpack(raw[u], vf);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, vf[e], acc[e]);
      }
    }
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)ca_rows_sum(acc[e]);
  if (rsub == 0 && pok) *reinterpret_cast<bf16x8*>(out + b * H * hd + h * hd + piece * 8) = o;
}

__global__ __launch_bounds__(256) void class_attn_bwd_vec_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                                const bf16* __restrict__ v, int64_t ts,
                                                                const bf16* __restrict__ dout, const float* __restrict__ psave,
                                                                bf16* __restrict__ dq, bf16* __restrict__ dk, bf16* __restrict__ dv,
                                                                int64_t dts, int64_t BH, int H, int N, int hd, float scale) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t bh = (int64_t)blockIdx.x * 4 + w;
  if (bh >= BH) return;
  const int64_t b = bh / H;
  const int h = (int)(bh % H);
  const int piece = lane & 7, rsub = lane >> 3;
  const bool pok = piece * 8 < hd;
  const int pc = pok ? piece : 0;
  float qf[8], dof[8];
  ca_unpack(*reinterpret_cast<const bf16x8*>(q + b * H * hd + h * hd + pc * 8), qf);
  ca_unpack(*reinterpret_cast<const bf16x8*>(dout + b * H * hd + h * hd + pc * 8), dof);
#pragma unroll
  for (int e = 0; e < 8; ++e) { qf[e] = pok ? qf[e] * scale : 0.f; dof[e] = pok ? dof[e] : 0.f; }
  const bf16* kb = k + b * N * ts + h * hd + pc * 8;
  const bf16* vb = v + b * N * ts + h * hd + pc * 8;
  // dp_j = dout . v_j ; ds_j = p_j (dp_j - sum p dp)
  float p[CA_STEPS], ds[CA_STEPS], dot = 0.f;
#pragma unroll
  for (int s0 = 0; s0 < CA_STEPS; s0 += 8) {
    ca_fence();                                                  // keep the batches apart (hipcc hoists all 32 loads otherwise)
    bf16x8 raw[8];
    float pr[8];
    if (s0 * 8 < N) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int jc = min((s0 + u) * 8 + rsub, N - 1);
        raw[u] = *reinterpret_cast<const bf16x8*>(vb + (int64_t)jc * ts);
        pr[u] = psave[bh * N + jc];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u, j = s * 8 + rsub;
      p[s] = 0.f; ds[s] = 0.f;
      if (s0 * 8 < N) {
        float vf[8];
        ca_unpack(raw[u], vf);
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a = fmaf(dof[e], vf[e], a);
        a = ca_row_sum(a);
        p[s] = j < N ? pr[u] : 0.f;
        ds[s] = a;
        dot = fmaf(p[s], a, dot);
      }
    }
  }
  dot = ca_rows_sum(dot);                                        // every lane of a row carries the row's term once
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int s0 = 0; s0 < CA_STEPS; s0 += 8) {
    ca_fence();                                                  // keep the batches apart (hipcc hoists all 32 loads otherwise)
    if (s0 * 8 < N) {
      bf16x8 raw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) raw[u] = *reinterpret_cast<const bf16x8*>(kb + (int64_t)min((s0 + u) * 8 + rsub, N - 1) * ts);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + u, j = s * 8 + rsub;
        const float d = p[s] * (ds[s] - dot);                    // 0 for j >= N
        float kf[8];
        ca_unpack(raw[u], kf);
        bf16x8 ov, ok;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ov[e] = (bf16)(p[s] * dof[e]);                         // dv_j = p_j dout
          ok[e] = (bf16)(d * qf[e]);                             // dk_j = ds_j scale q   (qf carries the scale)
          acc[e] = fmaf(d, kf[e], acc[e]);                       // dq = scale sum_j ds_j k_j
        }
        if (j < N && pok) {
          *reinterpret_cast<bf16x8*>(dv + (b * N + j) * dts + h * hd + piece * 8) = ov;
          *reinterpret_cast<bf16x8*>(dk + (b * N + j) * dts + h * hd + piece * 8) = ok;
        }
      }
    }
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)(ca_rows_sum(acc[e]) * scale);
  if (rsub == 0 && pok) *reinterpret_cast<bf16x8*>(dq + b * H * hd + h * hd + piece * 8) = o;
}

// ---- out[n] = sum_m x[m][n] * y[m][n]
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void colsum_mul_partial_kernel(const TX* __restrict__ x, int64_t ldx,
                                                                const TY* __restrict__ y, int64_t ldy, int64_t M,
                                                                int64_t N, float* __restrict__ part) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t c0 = ((int64_t)blockIdx.x * 64 + lane) * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c0 < N)
#pragma unroll 4
    for (int64_t r = (int64_t)blockIdx.y * 4 + w; r < M; r += (int64_t)gridDim.y * 4) {
      const f32x4 a = load4<TX>(x + r * ldx + c0), b = load4<TY>(y + r * ldy + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(a[e], b[e], acc[e]);
    }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[w][lane * 4 + e] = acc[e];
  __syncthreads();
  if (w == 0 && c0 < N)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = lane * 4 + e;
      part[(int64_t)blockIdx.y * N + c0 + e] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
}

inline int cm_splits(int64_t M) { int64_t s = (M + 3) / 4; return (int)(s < 512 ? s : 512); }

}  // namespace

static int th_check(int H, int Nk, int ld, const char* who) {
  VITMI_REQUIRE(H >= 1 && H <= TH_MAXH, VITMI_E_SHAPE, "%s: %d heads, at most %d supported", who, H, TH_MAXH);
  VITMI_REQUIRE(Nk >= 1 && Nk <= 64 * TH_MAXC && ld >= Nk, VITMI_E_SHAPE, "%s: row length %d not in [1, %d]", who, Nk, 64 * TH_MAXC);
  return 0;
}

extern "C" int vitmi_th_softmax_fwd(const void* S, const float* Wl, const float* bl, const float* Ww,
                                    const float* bw, void* P, void* Pm, int dtype, int64_t B, int64_t H,
                                    int64_t N, int64_t Nk, int64_t ld, void* stream_) {
  VITMI_REQUIRE(S && Wl && bl && Ww && bw && P && Pm && B > 0 && N > 0, VITMI_E_BADARG, "th_softmax_fwd: bad argument");
  int rc = th_check((int)H, (int)Nk, (int)ld, "th_softmax_fwd");
  if (rc) return rc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int64_t rows = B * N;
  dim3 grid((unsigned)((rows + 3) / 4));
  // four keys per lane as one vector when the rows allow it
  const size_t va = dtype == VITMI_BF16 ? 8 : 16;
  const bool vec = ld % 4 == 0 && ld >= (Nk + 3) / 4 * 4 && is_aligned(S, va) && is_aligned(P, va) && is_aligned(Pm, va);
#define TH_FWD(T, V) hipLaunchKernelGGL((th_softmax_fwd_kernel<T, V>), grid, dim3(256), 0, stream, (const T*)S, Wl, bl, Ww, bw, (T*)P, (T*)Pm, rows, (int)H, (int)N, (int)Nk, (int)ld)
  if (dtype == VITMI_BF16) { if (vec) TH_FWD(bf16, true); else TH_FWD(bf16, false); }
  else if (dtype == VITMI_F32) { if (vec) TH_FWD(float, true); else TH_FWD(float, false); }
  else return vitmi_fail(VITMI_E_DTYPE, "th_softmax_fwd: bad dtype");
#undef TH_FWD
  return vitmi_check_launch("th_softmax_fwd_kernel");
}

extern "C" size_t vitmi_th_softmax_bwd_workspace(int64_t B, int64_t H, int64_t N) {
  return (size_t)th_bwd_blocks(B * N) * (size_t)(2 * H * H + 2 * H) * sizeof(float);
}

static std::atomic<int> g_th_mfma{1};   // diagnostic / test hook: 0 = parameter gradients on per-lane FMAs only
extern "C" void vitmi_debug_th_mfma(int on) { g_th_mfma = on; }

extern "C" int vitmi_th_softmax_bwd(const void* S, const void* P, const void* dPm, const float* Wl,
                                    const float* Ww, void* dS, float* dWl, float* dbl, float* dWw,
                                    float* dbw, int dtype, int64_t B, int64_t H, int64_t N, int64_t Nk,
                                    int64_t ld, void* workspace, size_t workspace_bytes, void* stream_) {
  VITMI_REQUIRE(S && P && dPm && Wl && Ww && dS && dWl && dbl && dWw && dbw && B > 0 && N > 0, VITMI_E_BADARG, "th_softmax_bwd: bad argument");
  int rc = th_check((int)H, (int)Nk, (int)ld, "th_softmax_bwd");
  if (rc) return rc;
  VITMI_REQUIRE(workspace && workspace_bytes >= vitmi_th_softmax_bwd_workspace(B, H, N), VITMI_E_WORKSPACE, "th_softmax_bwd: workspace too small");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int64_t rows = B * N;
  const int nblk = th_bwd_blocks(rows);
  float* part = reinterpret_cast<float*>(workspace);
  const size_t va = dtype == VITMI_BF16 ? 8 : 16;
  const bool vec = ld % 4 == 0 && ld >= (Nk + 3) / 4 * 4 && is_aligned(S, va) && is_aligned(P, va) && is_aligned(dPm, va) && is_aligned(dS, va);
#define TH_BWD(T, V, MFV) hipLaunchKernelGGL((th_softmax_bwd_kernel<T, V, MFV>), dim3(nblk), dim3(256), 0, stream, (const T*)S, (const T*)P, (const T*)dPm, Wl, Ww, (T*)dS, part, rows, (int)H, (int)N, (int)Nk, (int)ld)
  const bool mf = g_th_mfma != 0 && vec && dtype == VITMI_BF16 && ld % 8 == 0 && is_aligned(S, 16) && is_aligned(P, 16) && is_aligned(dPm, 16);
  if (dtype == VITMI_BF16) { if (mf) TH_BWD(bf16, true, true); else if (vec) TH_BWD(bf16, true, false); else TH_BWD(bf16, false, false); }
  else if (dtype == VITMI_F32) { if (vec) TH_BWD(float, true, false); else TH_BWD(float, false, false); }
  else return vitmi_fail(VITMI_E_DTYPE, "th_softmax_bwd: bad dtype");
#undef TH_BWD
  rc = vitmi_check_launch("th_softmax_bwd_kernel");
  if (rc) return rc;
  const int64_t stride = 2 * H * H + 2 * H;
  const int nrows = nblk;                          // one partial row per workgroup
  float* const outs[4] = {dWl, dbl, dWw, dbw};
  const int widths[4] = {(int)(H * H), (int)H, (int)(H * H), (int)H};
  return vitmi_reduce_rows_segs(part, nrows, stride, outs, widths, stream);
}

extern "C" int vitmi_class_attn_fwd(const void* q, const void* k, const void* v, int64_t kv_token_stride,
                                    void* out, float* p_save, int dtype, int64_t B, int64_t H, int64_t N,
                                    int64_t hd, float scale, void* stream_) {
  VITMI_REQUIRE(q && k && v && out && p_save && B > 0 && H > 0 && N > 0, VITMI_E_BADARG, "class_attn_fwd: bad argument");
  VITMI_REQUIRE(hd >= 1 && hd <= 64 && N <= 64 * TH_MAXC, VITMI_E_SHAPE, "class_attn_fwd: hd <= 64 and N <= %d required", 64 * TH_MAXC);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int64_t BH = B * H;
  dim3 grid((unsigned)((BH + 3) / 4));
  if (dtype == VITMI_BF16 && hd % 8 == 0 && kv_token_stride % 8 == 0 && is_aligned(q, 16) && is_aligned(k, 16) && is_aligned(v, 16) && is_aligned(out, 16))
    hipLaunchKernelGGL(class_attn_fwd_vec_kernel, grid, dim3(256), 0, stream, (const bf16*)q, (const bf16*)k, (const bf16*)v, kv_token_stride, (bf16*)out, p_save, BH, (int)H, (int)N, (int)hd, scale);
  else if (dtype == VITMI_BF16)
    hipLaunchKernelGGL((class_attn_fwd_kernel<bf16>), grid, dim3(256), 0, stream, (const bf16*)q, (const bf16*)k, (const bf16*)v, kv_token_stride, (bf16*)out, p_save, BH, (int)H, (int)N, (int)hd, scale);
  else if (dtype == VITMI_F32)
    hipLaunchKernelGGL((class_attn_fwd_kernel<float>), grid, dim3(256), 0, stream, (const float*)q, (const float*)k, (const float*)v, kv_token_stride, (float*)out, p_save, BH, (int)H, (int)N, (int)hd, scale);
  else return vitmi_fail(VITMI_E_DTYPE, "class_attn_fwd: bad dtype");
  return vitmi_check_launch("class_attn_fwd_kernel");
}

extern "C" int vitmi_class_attn_bwd(const void* q, const void* k, const void* v, int64_t kv_token_stride,
                                    const void* dout, const float* p_save, void* dq, void* dk, void* dv,
                                    int64_t dkv_token_stride, int dtype, int64_t B, int64_t H, int64_t N,
                                    int64_t hd, float scale, void* stream_) {
  VITMI_REQUIRE(q && k && v && dout && p_save && dq && dk && dv && B > 0 && H > 0 && N > 0, VITMI_E_BADARG, "class_attn_bwd: bad argument");
  VITMI_REQUIRE(hd >= 1 && hd <= 64 && N <= 64 * TH_MAXC, VITMI_E_SHAPE, "class_attn_bwd: hd <= 64 and N <= %d required", 64 * TH_MAXC);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int64_t BH = B * H;
  dim3 grid((unsigned)((BH + 3) / 4));
  if (dtype == VITMI_BF16 && hd % 8 == 0 && kv_token_stride % 8 == 0 && dkv_token_stride % 8 == 0 && is_aligned(q, 16) && is_aligned(k, 16) &&
      is_aligned(v, 16) && is_aligned(dout, 16) && is_aligned(dq, 16) && is_aligned(dk, 16) && is_aligned(dv, 16))
    hipLaunchKernelGGL(class_attn_bwd_vec_kernel, grid, dim3(256), 0, stream, (const bf16*)q, (const bf16*)k, (const bf16*)v, kv_token_stride, (const bf16*)dout, p_save, (bf16*)dq, (bf16*)dk, (bf16*)dv, dkv_token_stride, BH, (int)H, (int)N, (int)hd, scale);
  else if (dtype == VITMI_BF16)
    hipLaunchKernelGGL((class_attn_bwd_kernel<bf16>), grid, dim3(256), 0, stream, (const bf16*)q, (const bf16*)k, (const bf16*)v, kv_token_stride, (const bf16*)dout, p_save, (bf16*)dq, (bf16*)dk, (bf16*)dv, dkv_token_stride, BH, (int)H, (int)N, (int)hd, scale);
  else if (dtype == VITMI_F32)
    hipLaunchKernelGGL((class_attn_bwd_kernel<float>), grid, dim3(256), 0, stream, (const float*)q, (const float*)k, (const float*)v, kv_token_stride, (const float*)dout, p_save, (float*)dq, (float*)dk, (float*)dv, dkv_token_stride, BH, (int)H, (int)N, (int)hd, scale);
  else return vitmi_fail(VITMI_E_DTYPE, "class_attn_bwd: bad dtype");
  return vitmi_check_launch("class_attn_bwd_kernel");
}

extern "C" size_t vitmi_colsum_mul_workspace(int64_t M, int64_t N) { return (size_t)cm_splits(M) * (size_t)N * sizeof(float); }

extern "C" int vitmi_colsum_mul(const void* x, int x_dtype, int64_t ldx, const void* y, int y_dtype, int64_t ldy,
                                int64_t M, int64_t N, float* out, void* workspace, size_t workspace_bytes,
                                void* stream_) {
  VITMI_REQUIRE(x && y && out && M > 0 && N > 0 && ldx >= N && ldy >= N, VITMI_E_BADARG, "colsum_mul: bad argument");
  VITMI_REQUIRE(N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && is_aligned(x, 4 * dtype_size(x_dtype)) && is_aligned(y, 4 * dtype_size(y_dtype)),
                VITMI_E_ALIGN, "colsum_mul: N and leading dimensions must be multiples of 4, pointers 4-element aligned");
  VITMI_REQUIRE(workspace && workspace_bytes >= vitmi_colsum_mul_workspace(M, N), VITMI_E_WORKSPACE, "colsum_mul: workspace too small");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int S = cm_splits(M);
  float* part = reinterpret_cast<float*>(workspace);
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)S);
#define CM(TX, TY) hipLaunchKernelGGL((colsum_mul_partial_kernel<TX, TY>), grid, dim3(256), 0, stream, (const TX*)x, ldx, (const TY*)y, ldy, M, N, part)
  if (x_dtype == VITMI_F32 && y_dtype == VITMI_F32) CM(float, float);
  else if (x_dtype == VITMI_F32 && y_dtype == VITMI_BF16) CM(float, bf16);
  else if (x_dtype == VITMI_BF16 && y_dtype == VITMI_BF16) CM(bf16, bf16);
  else if (x_dtype == VITMI_BF16 && y_dtype == VITMI_F32) CM(bf16, float);
  else return vitmi_fail(VITMI_E_DTYPE, "colsum_mul: bad dtypes");
#undef CM
  int rc = vitmi_check_launch("colsum_mul_partial_kernel");
  if (rc) return rc;
  return vitmi_reduce_rows(part, S, N, N, out, stream);
}

// every diagnostic switch of this file back to its default (vitmi_debug_reset, core.cpp)
void vitmi_debug_reset_cait() {
  g_th_mfma = 1;
}
