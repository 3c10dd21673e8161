// HBM-bound helpers of the path: dtype cast, patch gather (im2col), column
// sums (bias / pos_embed gradients), softmax cross-entropy, SGD-momentum.
// All grid-stride with 16-B vector accesses where alignment allows.
#include "common.h"

namespace {

constexpr int EW_BLOCK = 256;
inline unsigned ew_grid(int64_t work_items) {
  int64_t b = (work_items + EW_BLOCK - 1) / EW_BLOCK;
  const int64_t cap = 256 * 8;   // 8 blocks per CU
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ------------------------------------------------------------------ cast --
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    store4<TD>(dst + i * 4, load4<TS>(src + i * 4));
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = from_f32<TD>(to_f32(src[i]));
}

// ------------------------------------------------------------------ axpy --
// y += a * x (fp32): gradient accumulation over the flat gradient buffer when backward() runs a
// second time before zero_grad() (torch's contract; the engines overwrite their buffer)
__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, float a, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4), w = *reinterpret_cast<const f32x4*>(y + i * 4);
    *reinterpret_cast<f32x4*>(y + i * 4) = w + a * v;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] += a * x[i];
}

// ------------------------------------------------------------ scale_cast --
// out[m][n] = cast(x[m][n] * scale[n])  (scale may be NULL): the GEMM-operand copy of a
// residual-stream gradient entering a LayerScale branch
template <typename TS, typename TD>
__global__ void scale_cast_kernel(const TS* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                  const float* __restrict__ rowscale, int64_t rpg,
                                  TD* __restrict__ out, int64_t ldo, int64_t M, int64_t N) {
  const int64_t n4 = N / 4, total = M * n4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / n4, n = (i % n4) * 4;
    f32x4 v = load4<TS>(x + m * ldx + n);
    if (scale) v *= *reinterpret_cast<const f32x4*>(scale + n);
    if (rowscale) v *= rowscale[m / rpg];
    store4<TD>(out + m * ldo + n, v);
  }
}

// -------------------------------------------------------------- patchify --
// one thread = 4 consecutive k of one output row (k = c*p*p + i*p + j; j runs
// along W, so 4 consecutive j are contiguous in NCHW when p % 4 == 0)
template <typename TD>
__global__ void patchify_kernel(const float* __restrict__ x, int64_t sb, int64_t sc, int64_t sh,
                                int64_t sw, TD* __restrict__ out, int64_t out_ld, int64_t B, int C, int H, int W,
                                int p, int cls_rows) {
  const int gh = H / p, gw = W / p;
  const int ntok = cls_rows + gh * gw;
  const int Kp = C * p * p;
  const int kq = (int)(out_ld / 4);                // columns [Kp, out_ld) are written as zeros (K padding)
  const int64_t total = B * ntok * (int64_t)kq;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int k4 = (int)(idx % kq);
    const int64_t row = idx / kq;
    const int t = (int)(row % ntok);
    const int64_t b = row / ntok;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t >= cls_rows && k4 * 4 < Kp) {
      const int pt = t - cls_rows;
      const int py = pt / gw, px = pt % gw;
      const int k = k4 * 4;
      const int c = k / (p * p), rem = k % (p * p);
      const int i = rem / p, j = rem % p;
      const float* src = x + b * sb + c * sc + (int64_t)(py * p + i) * sh + (int64_t)(px * p + j) * sw;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = src[e * sw];
    }
    store4<TD>(out + row * out_ld + k4 * 4, v);
  }
}

// ---------------------------------------------------------------- colsum --
// partial[s][n] = sum over rows r = s*4+w, step 4*S of x[r][n]
template <typename T, int VEC>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int64_t M,
                                                             int64_t N, int64_t ld,
                                                             float* __restrict__ part) {
  __shared__ float red[4][64 * VEC];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t c0 = ((int64_t)blockIdx.x * 64 + lane) * VEC;
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (c0 < N) {
    for (int64_t r = (int64_t)blockIdx.y * 4 + w; r < M; r += (int64_t)gridDim.y * 4) {
      if constexpr (VEC == 4) {
        const f32x4 v = load4<T>(x + r * ld + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
      } else {
        acc[0] += to_f32(x[r * ld + c0]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[w][lane * VEC + e] = acc[e];
  __syncthreads();
  if (w == 0 && c0 < N) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int i = lane * VEC + e;
      part[(int64_t)blockIdx.y * N + c0 + e] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
  }
}

// The same for NARROW matrices (N <= 512, 4-element vectors): with one row per wave a
// 96-column row is a 192-B load by 24 of 64 lanes and every wave walks hundreds of them one
// latency at a time (Swin stage 1: 401 408 x 96 in 93 us = 0.8 TB/s).  Here the 256 threads
// of a workgroup tile R = 256 / (N / 4) consecutive rows per step (thread t: row t / gpr,
// column group t % gpr, so a step is one contiguous run of R rows when ld == N); a thread
// always meets the same column group, the R row classes meet in LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_rows_kernel(const T* __restrict__ x, int64_t M, int gpr,
                                                                  int64_t ld, float* __restrict__ part) {
  __shared__ f32x4 red[256];
  const int R = 256 / gpr;
  const int rr = threadIdx.x / gpr, cg = threadIdx.x - rr * gpr;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (rr < R) {
    const T* p = x + cg * 4;
#pragma unroll 4
    for (int64_t r = (int64_t)blockIdx.x * R + rr; r < M; r += (int64_t)gridDim.x * R) acc += load4<T>(p + r * ld);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < gpr) {
    f32x4 t = red[threadIdx.x];
    for (int k = 1; k < R; ++k) t += red[k * gpr + threadIdx.x];
    *reinterpret_cast<f32x4*>(part + (int64_t)blockIdx.x * gpr * 4 + threadIdx.x * 4) = t;
  }
}

// out[c] = sum_{r<S} part[r*ld + c], fixed summation order (deterministic).
// Column sums of S partial rows.  The partial buffers are narrow (144 .. 2304 columns) and
// tall (256 .. 2048 rows): with 64 columns per workgroup the grid was 3 .. 36 workgroups of
// 64 .. 128 dependent-latency loads per thread (15 .. 36 us for a few MB).  A workgroup now
// takes FOLD_COLS columns and spreads the rows over 1024 / FOLD_COLS row groups (64-B row
// pieces, from L2 / MALL: the producer has just written them); the groups' sums meet in LDS
// and are added in a fixed order.
constexpr int FOLD_COLS = 16;
constexpr int FOLD_GROUPS = 1024 / FOLD_COLS;
__device__ __forceinline__ float fold_rows(const float* __restrict__ part, int S, int64_t ld, int64_t col, bool ok,
                                           float (*red)[FOLD_COLS]) {
  const int c = threadIdx.x % FOLD_COLS, rg = threadIdx.x / FOLD_COLS;
  float s = 0.f;
  if (ok) {
#pragma unroll 4
    for (int r = rg; r < S; r += FOLD_GROUPS) s += part[(int64_t)r * ld + col];
  }
  red[rg][c] = s;
  __syncthreads();
  // 64 groups -> 4 partial sums per column (lanes 0..63 of wave 0), then a fixed 4-term sum
  float t = 0.f;
  if (threadIdx.x < 4 * FOLD_COLS) {
    const int q = threadIdx.x / FOLD_COLS;
#pragma unroll
    for (int i = 0; i < FOLD_GROUPS / 4; ++i) t += red[q * (FOLD_GROUPS / 4) + i][c];
  }
  __syncthreads();
  if (threadIdx.x < 4 * FOLD_COLS) red[threadIdx.x / FOLD_COLS][c] = t;
  __syncthreads();
  return (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);     // valid in every thread of column c
}

__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ part, int S,
                                                           int64_t N, int64_t ld,
                                                           float* __restrict__ out) {
  __shared__ float red[FOLD_GROUPS][FOLD_COLS];
  const int64_t c = (int64_t)blockIdx.x * FOLD_COLS + threadIdx.x % FOLD_COLS;
  const float t = fold_rows(part, S, ld, c, c < N, red);
  if (threadIdx.x < FOLD_COLS && c < N) out[c] = t;
}

// the same for up to three N-wide column segments of one partial buffer, each with its own
// destination (LayerNorm backward: dgamma | dbeta | column sum), in one launch
__global__ __launch_bounds__(1024) void reduce_rows3_kernel(const float* __restrict__ part, int S,
                                                            int64_t N, int64_t ld, float* out0,
                                                            float* out1, float* out2) {
  __shared__ float red[FOLD_GROUPS][FOLD_COLS];
  const int seg = blockIdx.y;
  float* out = seg == 0 ? out0 : (seg == 1 ? out1 : out2);
  const int64_t c = (int64_t)blockIdx.x * FOLD_COLS + threadIdx.x % FOLD_COLS;
  const float t = fold_rows(part, S, ld, seg * N + c, c < N, red);
  if (threadIdx.x < FOLD_COLS && c < N) out[c] = t;
}

// one partial buffer whose row is up to four consecutive segments of different widths, each
// with its own destination (talking-heads backward: dWl | dbl | dWw | dbw), in one launch
struct RowSegs { float* out[4]; int end[4]; };
__global__ __launch_bounds__(1024) void reduce_rows_segs_kernel(const float* __restrict__ part, int S,
                                                                int64_t ld, RowSegs sg) {
  __shared__ float red[FOLD_GROUPS][FOLD_COLS];
  const int c = blockIdx.x * FOLD_COLS + threadIdx.x % FOLD_COLS;
  const int n = sg.end[3];
  const float t = fold_rows(part, S, ld, c, c < n, red);
  if (threadIdx.x < FOLD_COLS && c < n) {
    const int k = c < sg.end[0] ? 0 : (c < sg.end[1] ? 1 : (c < sg.end[2] ? 2 : 3));
    sg.out[k][c - (k == 0 ? 0 : sg.end[k - 1])] = t;
  }
}

// Many folds in ONE launch (round 3).  A training step folds ~50 small partial buffers (LayerNorm dgamma | dbeta | bias
// sums, the fc1 / qkv bias partials of the GEMM and attention epilogues): each is 0.2-3 MB and a launch of its own costs
// 5 us of ramp for 0.5 us of traffic.  The engines queue the folds of a backward pass (or of a gradient-bucket section)
// and run them together: block b finds its fold in a prefix table of block counts (kernel argument, <= FOLD_MAX entries);
// every output is summed by fold_rows exactly as in the single launches, so results do not change by a bit.
constexpr int FOLD_MAX = 32;
struct FoldTable {
  const float* part[FOLD_MAX];
  float* out[FOLD_MAX][3];
  int S[FOLD_MAX], N[FOLD_MAX], ld[FOLD_MAX], blk_end[FOLD_MAX];
  int n;
};
__global__ __launch_bounds__(1024) void fold_many_kernel(FoldTable t) {
  __shared__ float red[FOLD_GROUPS][FOLD_COLS];
  const int b = blockIdx.x;
  int i = 0;
  while (i < t.n - 1 && b >= t.blk_end[i]) ++i;                  // workgroup-uniform
  const int lb = b - (i ? t.blk_end[i - 1] : 0);
  const int N = t.N[i], cb = (N + FOLD_COLS - 1) / FOLD_COLS;
  const int seg = lb / cb;
  const int64_t c = (int64_t)(lb % cb) * FOLD_COLS + threadIdx.x % FOLD_COLS;
  const float tot = fold_rows(t.part[i], t.S[i], t.ld[i], (int64_t)seg * N + c, c < N, red);
  if (threadIdx.x < FOLD_COLS && c < N) t.out[i][seg][c] = tot;
}

inline int colsum_splits(int64_t M) {
  int64_t s = (M + 3) / 4;
  return (int)(s < 512 ? s : 512);
}

// ------------------------------------------------------------------ xent --
// one wave per sample; K classes strided over lanes
__global__ __launch_bounds__(256) void xent_kernel(const float* __restrict__ logits,
                                                   const int64_t* __restrict__ labels,
                                                   float* __restrict__ loss_rows,
                                                   float* __restrict__ dlogits,
                                                   int32_t* __restrict__ correct_rows,
                                                   int64_t B, int K) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  if (b >= B) return;
  const float* lr = logits + b * K;
  float mx = -INFINITY;
  int amax = 0;
  for (int k = lane; k < K; k += 64) {
    const float v = lr[k];
    if (v > mx) { mx = v; amax = k; }     // first maximum within the lane
  }
  // wave argmax, ties -> lowest index (torch.argmax on CPU returns the first)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o);
    const int oa = __shfl_xor(amax, o);
    if (om > mx || (om == mx && oa < amax)) { mx = om; amax = oa; }
  }
  float se = 0.f;
  for (int k = lane; k < K; k += 64) se += __expf(lr[k] - mx);
  se = wave_sum(se);
  const float lse = mx + __logf(se);
  const int64_t lab = labels[b];
  const float invB = 1.f / (float)B;
  if (dlogits) {
    for (int k = lane; k < K; k += 64) {
      const float p = __expf(lr[k] - lse);
      dlogits[b * K + k] = (p - (k == lab ? 1.f : 0.f)) * invB;
    }
  }
  if (lane == 0) {
    // a label outside [0, K) must not become an out-of-bounds read: its sample's loss (and
    // with it the mean) is NaN, which the host sees at the next .item()
    const bool ok = lab >= 0 && lab < (int64_t)K;
    loss_rows[b] = ok ? lse - lr[ok ? lab : 0] : __builtin_nanf("");
    correct_rows[b] = (ok && amax == (int)lab) ? 1 : 0;
  }
}

// deterministic final reduction of the per-sample losses (single block)
__global__ __launch_bounds__(256) void xent_reduce_kernel(const float* __restrict__ loss_rows,
                                                          const int32_t* __restrict__ correct_rows,
                                                          float* __restrict__ loss,
                                                          int32_t* __restrict__ correct, int64_t B) {
  __shared__ float sl[256];
  __shared__ int sc[256];
  float l = 0.f;
  int c = 0;
  for (int64_t i = threadIdx.x; i < B; i += 256) { l += loss_rows[i]; c += correct_rows[i]; }
  sl[threadIdx.x] = l; sc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { loss[0] = sl[0] / (float)B; if (correct) correct[0] = sc[0]; }
}

// ------------------------------------------------------------------- sgd --
__global__ void sgd_momentum_kernel(float* __restrict__ p, const float* __restrict__ g,
                                    float* __restrict__ buf, bf16* __restrict__ shadow, int64_t n,
                                    float lr, float momentum, float gscale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    // 1.7 GB streamed once per step (ViT-B/16), far beyond the 256-MB Infinity Cache: non-temporal on every access except
    // the bf16 shadow, which the next forward's first GEMMs read
    const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i * 4));
    f32x4 bv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(buf + i * 4));
    f32x4 pv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i * 4));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bv[e] = momentum * bv[e] + gscale * gv[e];
      pv[e] = pv[e] - lr * bv[e];
    }
    __builtin_nontemporal_store(bv, reinterpret_cast<f32x4*>(buf + i * 4));
    __builtin_nontemporal_store(pv, reinterpret_cast<f32x4*>(p + i * 4));
    if (shadow) store4<bf16>(shadow + i * 4, pv);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float b = momentum * buf[i] + gscale * g[i];
    buf[i] = b;
    const float pn = p[i] - lr * b;
    p[i] = pn;
    if (shadow) shadow[i] = (bf16)pn;
  }
}

// ------------------------------------------------------------------ adam --
// torch.optim.Adam / AdamW single-tensor update (decoupled decay when `decoupled`):
//   p *= 1 - lr*wd (AdamW)  |  g += wd*p (Adam);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
//   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// The step count t lives on the DEVICE (state[0]) and is advanced by a one-thread kernel
// before the update, so a captured HIP graph replays the right bias corrections.
__global__ void adam_tick_kernel(float* state) { state[0] += 1.f; }

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, bf16* __restrict__ shadow, const float* __restrict__ state,
                            int64_t n, float lr, float b1, float b2, float eps, float wd, int decoupled,
                            float gscale) {
  const float t = state[0];
  const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
  const float step = lr / bc1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pi = p[i], gi = g[i] * gscale;
    if (decoupled) pi *= 1.f - lr * wd;
    else gi = fmaf(wd, pi, gi);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= step * mi / (sqrtf(vi) / bc2s + eps);
    p[i] = pi;
    if (shadow) shadow[i] = (bf16)pi;
  }
}

}  // namespace

int vitmi_reduce_rows3(const float* part, int S, int64_t N, int64_t ld, float* out0, float* out1, float* out2,
                       hipStream_t stream) {
  const unsigned segs = out2 ? 3 : 2;
  hipLaunchKernelGGL(reduce_rows3_kernel, dim3((unsigned)((N + FOLD_COLS - 1) / FOLD_COLS), segs), dim3(1024), 0, stream, part, S,
                     N, ld, out0, out1, out2);
  return vitmi_check_launch("reduce_rows3_kernel");
}

int vitmi_reduce_rows_segs(const float* part, int S, int64_t ld, float* const out[4], const int width[4],
                           hipStream_t stream) {
  RowSegs sg;
  int e = 0;
  for (int i = 0; i < 4; ++i) { sg.out[i] = out[i]; e += width[i]; sg.end[i] = e; }
  hipLaunchKernelGGL(reduce_rows_segs_kernel, dim3((unsigned)((e + FOLD_COLS - 1) / FOLD_COLS)), dim3(1024), 0, stream, part, S, ld, sg);
  return vitmi_check_launch("reduce_rows_segs_kernel");
}

extern "C" int vitmi_fold_many(const vitmi_fold_desc* descs, int n, void* stream_) {
  VITMI_REQUIRE(descs && n > 0, VITMI_E_BADARG, "fold_many: null descriptors or n <= 0");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  for (int i0 = 0; i0 < n; i0 += FOLD_MAX) {
    FoldTable t;
    t.n = n - i0 < FOLD_MAX ? n - i0 : FOLD_MAX;
    int blocks = 0;
    for (int i = 0; i < FOLD_MAX; ++i) {
      const vitmi_fold_desc& d = descs[i0 + (i < t.n ? i : 0)];    // unused entries repeat the first (never selected)
      if (i < t.n) {
        VITMI_REQUIRE(d.struct_size == (int64_t)sizeof(vitmi_fold_desc), VITMI_E_BADARG, "fold_many: descriptor size %lld, this library expects %zu", (long long)d.struct_size, sizeof(vitmi_fold_desc));
        VITMI_REQUIRE(d.part && d.S > 0 && d.nseg >= 1 && d.nseg <= 3 && d.N > 0 && d.N < (1ll << 30) && d.ld >= d.nseg * d.N && d.ld < (1ll << 31),
                      VITMI_E_SHAPE, "fold_many: descriptor %d: S %d, nseg %d, N %lld, ld %lld", i0 + i, d.S, d.nseg, (long long)d.N, (long long)d.ld);
        for (int k = 0; k < d.nseg; ++k) VITMI_REQUIRE(d.out[k], VITMI_E_BADARG, "fold_many: descriptor %d: null output %d", i0 + i, k);
        blocks += d.nseg * (int)((d.N + FOLD_COLS - 1) / FOLD_COLS);
      }
      t.part[i] = d.part; t.S[i] = d.S; t.N[i] = (int)d.N; t.ld[i] = (int)d.ld; t.blk_end[i] = blocks;
      for (int k = 0; k < 3; ++k) t.out[i][k] = d.out[k < d.nseg ? k : 0];
    }
    hipLaunchKernelGGL(fold_many_kernel, dim3((unsigned)blocks), dim3(1024), 0, stream, t);
    if (int rc = vitmi_check_launch("fold_many_kernel")) return rc;
  }
  return 0;
}

int vitmi_reduce_rows(const float* part, int S, int64_t N, int64_t ld, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((N + FOLD_COLS - 1) / FOLD_COLS)), dim3(1024), 0, stream, part, S, N, ld, out);
  return vitmi_check_launch("reduce_rows_kernel");
}

extern "C" int vitmi_cast(const void* src, int sd, void* dst, int dd, int64_t n, void* stream_) {
  VITMI_REQUIRE(src && dst && n > 0, VITMI_E_BADARG, "cast: null pointer or n <= 0");
  VITMI_REQUIRE(is_aligned(src, 4 * dtype_size(sd)) && is_aligned(dst, 4 * dtype_size(dd)), VITMI_E_ALIGN, "cast: pointers must be 4-element aligned");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const unsigned grid = ew_grid(n / 4 + 1);
  if (sd == VITMI_F32 && dd == VITMI_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(grid), dim3(EW_BLOCK), 0, stream, (const float*)src, (bf16*)dst, n);
  else if (sd == VITMI_BF16 && dd == VITMI_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(grid), dim3(EW_BLOCK), 0, stream, (const bf16*)src, (float*)dst, n);
  else if (sd == VITMI_F32 && dd == VITMI_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(EW_BLOCK), 0, stream, (const float*)src, (float*)dst, n);
  else if (sd == VITMI_BF16 && dd == VITMI_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), dim3(grid), dim3(EW_BLOCK), 0, stream, (const bf16*)src, (bf16*)dst, n);
  else return vitmi_fail(VITMI_E_DTYPE, "cast: bad dtypes %d -> %d", sd, dd);
  return vitmi_check_launch("cast_kernel");
}

extern "C" int vitmi_axpy(const float* x, float* y, float a, int64_t n, void* stream_) {
  VITMI_REQUIRE(x && y && n > 0, VITMI_E_BADARG, "axpy: null pointer or n <= 0");
  VITMI_REQUIRE(is_aligned(x, 16) && is_aligned(y, 16), VITMI_E_ALIGN, "axpy: pointers must be 16-byte aligned");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_BLOCK), 0, stream, x, y, a, n);
  return vitmi_check_launch("axpy_kernel");
}

extern "C" int vitmi_scale_cast(const void* x, int x_dtype, int64_t ldx, const float* scale,
                                const float* rowscale, int64_t rows_per_group, void* out,
                                int out_dtype, int64_t ldo, int64_t M, int64_t N, void* stream_) {
  VITMI_REQUIRE(x && out && M > 0 && N > 0 && ldx >= N && ldo >= N, VITMI_E_BADARG, "scale_cast: bad argument");
  VITMI_REQUIRE(N % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && is_aligned(x, 4 * dtype_size(x_dtype)) &&
                    is_aligned(out, 4 * dtype_size(out_dtype)) && (!scale || is_aligned(scale, 16)),
                VITMI_E_ALIGN, "scale_cast: 4-element alignment required");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const unsigned grid = ew_grid(M * N / 4);
#define SC(TS, TD) hipLaunchKernelGGL((scale_cast_kernel<TS, TD>), dim3(grid), dim3(EW_BLOCK), 0, stream, (const TS*)x, ldx, scale, rowscale, rows_per_group > 0 ? rows_per_group : 1, (TD*)out, ldo, M, N)
  if (x_dtype == VITMI_F32 && out_dtype == VITMI_BF16) SC(float, bf16);
  else if (x_dtype == VITMI_F32 && out_dtype == VITMI_F32) SC(float, float);
  else if (x_dtype == VITMI_BF16 && out_dtype == VITMI_BF16) SC(bf16, bf16);
  else if (x_dtype == VITMI_BF16 && out_dtype == VITMI_F32) SC(bf16, float);
  else return vitmi_fail(VITMI_E_DTYPE, "scale_cast: bad dtypes");
#undef SC
  return vitmi_check_launch("scale_cast_kernel");
}

extern "C" int vitmi_patchify(const float* x, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                              void* out, int out_dtype, int64_t out_ld, int64_t B, int64_t C, int64_t H, int64_t W,
                              int64_t p, int cls_rows, void* stream_) {
  VITMI_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0 && p > 0, VITMI_E_BADARG, "patchify: null pointer or empty shape");
  VITMI_REQUIRE(H % p == 0 && W % p == 0, VITMI_E_SHAPE, "patchify: image %lldx%lld not divisible by patch %lld", (long long)H, (long long)W, (long long)p);
  VITMI_REQUIRE(p % 4 == 0, VITMI_E_SHAPE, "patchify: patch size must be a multiple of 4");
  VITMI_REQUIRE(cls_rows == 0 || cls_rows == 1, VITMI_E_BADARG, "patchify: cls_rows must be 0 or 1");
  if (out_ld <= 0) out_ld = C * p * p;
  VITMI_REQUIRE(out_ld >= C * p * p && out_ld % 4 == 0, VITMI_E_BADARG, "patchify: out_ld must be >= C*p*p and a multiple of 4");
  VITMI_REQUIRE(is_aligned(out, 4 * dtype_size(out_dtype)), VITMI_E_ALIGN, "patchify: out alignment");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int64_t ntok = cls_rows + (H / p) * (W / p);
  const int64_t total = B * ntok * (out_ld / 4);
  const unsigned grid = ew_grid(total);
  if (out_dtype == VITMI_BF16)
    hipLaunchKernelGGL((patchify_kernel<bf16>), dim3(grid), dim3(EW_BLOCK), 0, stream, x, sb, sc, sh, sw, (bf16*)out, out_ld, B, (int)C, (int)H, (int)W, (int)p, cls_rows);
  else if (out_dtype == VITMI_F32)
    hipLaunchKernelGGL((patchify_kernel<float>), dim3(grid), dim3(EW_BLOCK), 0, stream, x, sb, sc, sh, sw, (float*)out, out_ld, B, (int)C, (int)H, (int)W, (int)p, cls_rows);
  else return vitmi_fail(VITMI_E_DTYPE, "patchify: bad out dtype");
  return vitmi_check_launch("patchify_kernel");
}

extern "C" size_t vitmi_colsum_workspace(int64_t M, int64_t N) {
  return (size_t)colsum_splits(M) * (size_t)N * sizeof(float);
}

extern "C" int vitmi_colsum(const void* x, int dtype, int64_t M, int64_t N, int64_t ld, float* out,
                            void* workspace, size_t workspace_bytes, void* stream_) {
  VITMI_REQUIRE(x && out && M > 0 && N > 0 && ld >= N, VITMI_E_BADARG, "colsum: bad argument");
  VITMI_REQUIRE(workspace && workspace_bytes >= vitmi_colsum_workspace(M, N), VITMI_E_WORKSPACE, "colsum: workspace too small");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  // a short fp32 matrix (the per-row-block partial sums the GEMM / attention epilogues leave: 256-400 rows)
  // IS a partial buffer: fold it directly, one launch instead of two
  if (dtype == VITMI_F32 && M <= 2048) return vitmi_reduce_rows(reinterpret_cast<const float*>(x), (int)M, N, ld, out, stream);
  const int S = colsum_splits(M);
  float* part = reinterpret_cast<float*>(workspace);
  const bool vec = (N % 4 == 0) && (ld % 4 == 0) && is_aligned(x, 4 * dtype_size(dtype));
  if (vec && N <= 512 && (dtype == VITMI_BF16 || dtype == VITMI_F32)) {      // >= 2 rows per workgroup step
    const int gpr = (int)(N / 4);
    if (dtype == VITMI_BF16) hipLaunchKernelGGL((colsum_partial_rows_kernel<bf16>), dim3((unsigned)S), dim3(256), 0, stream, (const bf16*)x, M, gpr, ld, part);
    else hipLaunchKernelGGL((colsum_partial_rows_kernel<float>), dim3((unsigned)S), dim3(256), 0, stream, (const float*)x, M, gpr, ld, part);
    int rc = vitmi_check_launch("colsum_partial_rows_kernel");
    if (rc) return rc;
    return vitmi_reduce_rows(part, S, N, N, out, stream);
  }
  const int cols_per_block = vec ? 256 : 64;
  dim3 grid((unsigned)((N + cols_per_block - 1) / cols_per_block), (unsigned)S);
  if (dtype == VITMI_BF16) {
    if (vec) hipLaunchKernelGGL((colsum_partial_kernel<bf16, 4>), grid, dim3(256), 0, stream, (const bf16*)x, M, N, ld, part);
    else hipLaunchKernelGGL((colsum_partial_kernel<bf16, 1>), grid, dim3(256), 0, stream, (const bf16*)x, M, N, ld, part);
  } else if (dtype == VITMI_F32) {
    if (vec) hipLaunchKernelGGL((colsum_partial_kernel<float, 4>), grid, dim3(256), 0, stream, (const float*)x, M, N, ld, part);
    else hipLaunchKernelGGL((colsum_partial_kernel<float, 1>), grid, dim3(256), 0, stream, (const float*)x, M, N, ld, part);
  } else return vitmi_fail(VITMI_E_DTYPE, "colsum: bad dtype");
  int rc = vitmi_check_launch("colsum_partial_kernel");
  if (rc) return rc;
  return vitmi_reduce_rows(part, S, N, N, out, stream);
}

extern "C" int vitmi_softmax_xent(const float* logits, const int64_t* labels, float* loss,
                                  float* dlogits, int32_t* correct, int64_t B, int64_t K,
                                  void* stream_) {
  VITMI_REQUIRE(logits && labels && loss && B > 0 && K > 0, VITMI_E_BADARG, "softmax_xent: bad argument");
  VITMI_REQUIRE(dlogits, VITMI_E_BADARG, "softmax_xent: dlogits buffer required (its tail is used as scratch)");
  // scratch: per-sample loss / correct live behind the caller's loss pointer?
  // No: keep the ABI allocation-free by requiring loss to have room for 1+B
  // floats and correct for 1+B ints (documented in INTEGRATION.md).
  VITMI_REQUIRE(correct, VITMI_E_BADARG, "softmax_xent: correct buffer required");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(xent_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, logits, labels, loss + 1, dlogits, correct + 1, B, (int)K);
  int rc = vitmi_check_launch("xent_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(xent_reduce_kernel, dim3(1), dim3(256), 0, stream, loss + 1, correct + 1, loss, correct, B);
  return vitmi_check_launch("xent_reduce_kernel");
}

extern "C" int vitmi_adam(float* p, const float* g, float* m, float* v, void* shadow, float* state, int64_t n,
                          float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled,
                          float grad_scale, void* stream_) {
  VITMI_REQUIRE(p && g && m && v && state && n > 0, VITMI_E_BADARG, "adam: bad argument");
  VITMI_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, VITMI_E_BADARG, "adam: bad hyper-parameter");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, stream, state);
  int rc = vitmi_check_launch("adam_tick_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, stream, p, g, m, v, (bf16*)shadow, state, n, lr,
                     beta1, beta2, eps, weight_decay, decoupled, grad_scale);
  return vitmi_check_launch("adam_kernel");
}

extern "C" int vitmi_sgd_momentum(float* p, const float* g, float* buf, void* shadow, int64_t n,
                                  float lr, float momentum, float grad_scale, void* stream_) {
  VITMI_REQUIRE(p && g && buf && n > 0, VITMI_E_BADARG, "sgd_momentum: bad argument");
  VITMI_REQUIRE(is_aligned(p, 16) && is_aligned(g, 16) && is_aligned(buf, 16) && (!shadow || is_aligned(shadow, 8)), VITMI_E_ALIGN, "sgd_momentum: buffers must be 16-B aligned");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(sgd_momentum_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_BLOCK), 0, stream, p, g, buf, (bf16*)shadow, n, lr, momentum, grad_scale);
  return vitmi_check_launch("sgd_momentum_kernel");
}
