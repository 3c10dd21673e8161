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
#include "common.h"

namespace {

constexpr int TH_MAXH = 8;
constexpr int TH_MAXC = 4;     // columns per lane -> row length <= 256

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return to_f32(*p); }

// ---- forward: S[B,H,N,ld] -> P (softmax of mixed scores), Pm (mixed probabilities)
template <typename T>
__global__ __launch_bounds__(256) void th_softmax_fwd_kernel(const T* __restrict__ S, const float* __restrict__ Wl,
                                                            const float* __restrict__ bl, const float* __restrict__ Ww,
                                                            const float* __restrict__ bw, T* __restrict__ P,
                                                            T* __restrict__ Pm, int64_t rows /*B*N*/, int H, int N,
                                                            int Nk, int ld) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + w;
  if (row >= rows) return;
  const int64_t b = row / N, i = row % N;
  float s[TH_MAXH][TH_MAXC], p[TH_MAXH][TH_MAXC];
#pragma unroll
  for (int h = 0; h < TH_MAXH; ++h)
#pragma unroll
    for (int c = 0; c < TH_MAXC; ++c) {
      const int j = c * 64 + lane;
      s[h][c] = (h < H && j < Nk) ? ldf(S + ((b * H + h) * N + i) * ld + j) : 0.f;
    }
#pragma unroll
  for (int hp = 0; hp < TH_MAXH; ++hp) {
    if (hp >= H) break;
    float v[TH_MAXC], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < TH_MAXC; ++c) {
      float a = bl[hp];
#pragma unroll
      for (int h = 0; h < TH_MAXH; ++h)
        if (h < H) a = fmaf(Wl[hp * H + h], s[h][c], a);
      v[c] = (c * 64 + lane < Nk) ? a : -INFINITY;
      mx = fmaxf(mx, v[c]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < TH_MAXC; ++c) { v[c] = expf(v[c] - mx); sum += v[c]; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < TH_MAXC; ++c) {
      // backward differentiates at the STORED probabilities
      p[hp][c] = to_f32(from_f32<T>(v[c] * inv));
      const int j = c * 64 + lane;
      if (j < Nk) P[((b * H + hp) * N + i) * ld + j] = from_f32<T>(p[hp][c]);
    }
  }
#pragma unroll
  for (int ho = 0; ho < TH_MAXH; ++ho) {
    if (ho >= H) break;
#pragma unroll
    for (int c = 0; c < TH_MAXC; ++c) {
      float a = bw[ho];
#pragma unroll
      for (int hp = 0; hp < TH_MAXH; ++hp)
        if (hp < H) a = fmaf(Ww[ho * H + hp], p[hp][c], a);
      const int j = c * 64 + lane;
      if (j < Nk) Pm[((b * H + ho) * N + i) * ld + j] = from_f32<T>(a);
    }
  }
}

// ---- backward.  Per-lane accumulators of the four parameter gradients; part layout per
// wave: [dWl H*H | dbl H | dWw H*H | dbw H] (2*H*H + 2*H floats)
template <typename T>
__global__ __launch_bounds__(256) void th_softmax_bwd_kernel(const T* __restrict__ S, const T* __restrict__ P,
                                                            const T* __restrict__ dPm, const float* __restrict__ Wl,
                                                            const float* __restrict__ Ww, T* __restrict__ dS,
                                                            float* __restrict__ part, int64_t rows, int H, int N,
                                                            int Nk, int ld) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float aWl[TH_MAXH][TH_MAXH], aWw[TH_MAXH][TH_MAXH], abl[TH_MAXH], abw[TH_MAXH];
#pragma unroll
  for (int a = 0; a < TH_MAXH; ++a) {
    abl[a] = 0.f; abw[a] = 0.f;
#pragma unroll
    for (int c = 0; c < TH_MAXH; ++c) { aWl[a][c] = 0.f; aWw[a][c] = 0.f; }
  }
  // Two passes per row, one 64-key chunk at a time (the whole row of S, P, dP' for 8 heads
  // is 96 values per lane: holding it beside the 144 gradient accumulators leaves one wave
  // per SIMD).  Pass 1: dP = Ww^T dP', the softmax row dots, dWw / dbw.  Pass 2 re-reads P
  // (L2) and S: dS' = P (dP - dot), dS = Wl^T dS', dWl / dbl.  dP of all chunks stays in
  // registers between the passes.
  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < rows; row += (int64_t)gridDim.x * 4) {
    const int64_t b = row / N, i = row % N;
    float dp[TH_MAXH][TH_MAXC], dot[TH_MAXH];
#pragma unroll
    for (int hp = 0; hp < TH_MAXH; ++hp) dot[hp] = 0.f;
#pragma unroll
    for (int c = 0; c < TH_MAXC; ++c) {
      const int j = c * 64 + lane;
      if (c * 64 >= Nk) {                          // wave-uniform: chunk beyond the row
#pragma unroll
        for (int hp = 0; hp < TH_MAXH; ++hp) dp[hp][c] = 0.f;
        continue;
      }
      float p[TH_MAXH], g[TH_MAXH];
#pragma unroll
      for (int h = 0; h < TH_MAXH; ++h) {
        const bool ok = h < H && j < Nk;
        const int64_t o = ((b * H + h) * N + i) * ld + j;
        p[h] = ok ? ldf(P + o) : 0.f;
        g[h] = ok ? ldf(dPm + o) : 0.f;            // dL/dP'
      }
      // through proj_w: dP[hp] = sum_ho Ww[ho][hp] dP'[ho];  dWw[ho][hp] += dP'[ho] * P[hp]
#pragma unroll
      for (int hp = 0; hp < TH_MAXH; ++hp) {
        float a = 0.f;
#pragma unroll
        for (int ho = 0; ho < TH_MAXH; ++ho)
          if (ho < H && hp < H) a = fmaf(Ww[ho * H + hp], g[ho], a);
        dp[hp][c] = a;
        dot[hp] = fmaf(a, p[hp], dot[hp]);
      }
#pragma unroll
      for (int ho = 0; ho < TH_MAXH; ++ho) {
        abw[ho] += g[ho];
#pragma unroll
        for (int hp = 0; hp < TH_MAXH; ++hp) aWw[ho][hp] = fmaf(g[ho], p[hp], aWw[ho][hp]);
      }
    }
#pragma unroll
    for (int hp = 0; hp < TH_MAXH; ++hp) dot[hp] = wave_sum(dot[hp]);
#pragma unroll
    for (int c = 0; c < TH_MAXC; ++c) {
      const int j = c * 64 + lane;
      if (c * 64 >= Nk) continue;
      float p[TH_MAXH], sv[TH_MAXH], ds[TH_MAXH];
#pragma unroll
      for (int h = 0; h < TH_MAXH; ++h) {
        const bool ok = h < H && j < Nk;
        const int64_t o = ((b * H + h) * N + i) * ld + j;
        p[h] = ok ? ldf(P + o) : 0.f;
        sv[h] = ok ? ldf(S + o) : 0.f;
      }
      // through the softmax: dS'[hp] = P[hp] * (dP[hp] - sum_j dP[hp] P[hp])
#pragma unroll
      for (int hp = 0; hp < TH_MAXH; ++hp) ds[hp] = p[hp] * (dp[hp][c] - dot[hp]);
      // through proj_l: dS[h] = sum_hp Wl[hp][h] dS'[hp];  dWl[hp][h] += dS'[hp] * S[h]
#pragma unroll
      for (int hp = 0; hp < TH_MAXH; ++hp) {
        abl[hp] += ds[hp];
#pragma unroll
        for (int h = 0; h < TH_MAXH; ++h) aWl[hp][h] = fmaf(ds[hp], sv[h], aWl[hp][h]);
      }
#pragma unroll
      for (int h = 0; h < TH_MAXH; ++h) {
        if (h >= H) break;
        float a = 0.f;
#pragma unroll
        for (int hp = 0; hp < TH_MAXH; ++hp)
          if (hp < H) a = fmaf(Wl[hp * H + h], ds[hp], a);
        if (j < Nk) dS[((b * H + h) * N + i) * ld + j] = from_f32<T>(a);
      }
    }
  }
  const int stride = 2 * H * H + 2 * H;
  float* prow = part + ((int64_t)blockIdx.x * 4 + w) * stride;
#pragma unroll
  for (int a = 0; a < TH_MAXH; ++a) {
    if (a >= H) break;
#pragma unroll
    for (int c = 0; c < TH_MAXH; ++c) {
      if (c >= H) break;
      const float x = wave_sum(aWl[a][c]), y = wave_sum(aWw[a][c]);
      if (lane == 0) { prow[a * H + c] = x; prow[H * H + H + a * H + c] = y; }
    }
    const float x = wave_sum(abl[a]), y = wave_sum(abw[a]);
    if (lane == 0) { prow[H * H + a] = x; prow[2 * H * H + H + a] = y; }
  }
}

inline int th_bwd_blocks(int64_t rows) {
  int64_t b = (rows + 3) / 4;
  return (int)(b < 512 ? b : 512);      // one partial row per wave: keep the fold short
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

inline int cm_splits(int64_t M) { int64_t s = (M + 3) / 4; return (int)(s < 128 ? s : 128); }

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
  if (dtype == VITMI_BF16)
    hipLaunchKernelGGL((th_softmax_fwd_kernel<bf16>), grid, dim3(256), 0, stream, (const bf16*)S, Wl, bl, Ww, bw, (bf16*)P, (bf16*)Pm, rows, (int)H, (int)N, (int)Nk, (int)ld);
  else if (dtype == VITMI_F32)
    hipLaunchKernelGGL((th_softmax_fwd_kernel<float>), grid, dim3(256), 0, stream, (const float*)S, Wl, bl, Ww, bw, (float*)P, (float*)Pm, rows, (int)H, (int)N, (int)Nk, (int)ld);
  else return vitmi_fail(VITMI_E_DTYPE, "th_softmax_fwd: bad dtype");
  return vitmi_check_launch("th_softmax_fwd_kernel");
}

extern "C" size_t vitmi_th_softmax_bwd_workspace(int64_t B, int64_t H, int64_t N) {
  return (size_t)th_bwd_blocks(B * N) * 4 * (size_t)(2 * H * H + 2 * H) * sizeof(float);
}

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
  if (dtype == VITMI_BF16)
    hipLaunchKernelGGL((th_softmax_bwd_kernel<bf16>), dim3(nblk), dim3(256), 0, stream, (const bf16*)S, (const bf16*)P, (const bf16*)dPm, Wl, Ww, (bf16*)dS, part, rows, (int)H, (int)N, (int)Nk, (int)ld);
  else if (dtype == VITMI_F32)
    hipLaunchKernelGGL((th_softmax_bwd_kernel<float>), dim3(nblk), dim3(256), 0, stream, (const float*)S, (const float*)P, (const float*)dPm, Wl, Ww, (float*)dS, part, rows, (int)H, (int)N, (int)Nk, (int)ld);
  else return vitmi_fail(VITMI_E_DTYPE, "th_softmax_bwd: bad dtype");
  rc = vitmi_check_launch("th_softmax_bwd_kernel");
  if (rc) return rc;
  const int64_t stride = 2 * H * H + 2 * H;
  const int nrows = nblk * 4;
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
  if (dtype == VITMI_BF16)
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
  if (dtype == VITMI_BF16)
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
