// vitmi_gemm: argument checking, dispatch, and the GENERIC strided MFMA kernel.
//
// The generic kernel takes any M/N/K, any strides and both operand
// orientations, for bf16 (v_mfma_f32_32x32x16_bf16) and fp32
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain in k order — the parity mode).
// It is the correctness workhorse (classifier head, ragged shapes, fp32 mode);
// the aligned hot shapes go to gemm_fast.hip.
#include <atomic>
#include "epilogue.h"

bool gemm_fast_supported(const GemmArgs& g, int in_bf16);
int gemm_fast_launch(const GemmArgs& g, hipStream_t stream);
size_t gemm_fast_workspace(const GemmArgs& g);

// diagnostic only (tools/gemm_phases.py): block 0 of the PIPE=1 kernel accumulates
// s_memtime stamps of its R / M phases and barrier waits into this buffer
// debug hook: prefetched tiles of the persistent walk wait for EVERYTHING (vmcnt(0)) before their first slab is read,
// instead of the counted wait that relies on the epilogue issuing at least E_MIN stores; results must be bit-identical
static std::atomic<int> g_gemm_strict_wait{0};
extern "C" void vitmi_debug_gemm_strict_wait(int on) { g_gemm_strict_wait = on != 0; }
static std::atomic<unsigned long long*> g_gemm_dbg{nullptr};
static std::atomic<int> g_gemm_dbg_blocks{64};
static std::atomic<int> g_gemm_alias{0};
// diagnostic hook (tools/r04_gemm_probe.py): every tile of the 256x256 kernel stages operand panel 0 (bit 0: A, bit 1: B) —
// WRONG results on purpose; it takes the streamed operand's HBM / Infinity-Cache misses out of the main loop
extern "C" void vitmi_debug_gemm_alias(int bits) { g_gemm_alias = bits; }
extern "C" void vitmi_debug_gemm_stamps(unsigned long long* buf) { g_gemm_dbg = buf; g_gemm_dbg_blocks = 64; }
// timeline of the first `blocks` workgroups: buf holds 64 + 4 * blocks entries
extern "C" void vitmi_debug_gemm_timeline(unsigned long long* buf, int blocks) { g_gemm_dbg = buf; g_gemm_dbg_blocks = blocks; }

namespace {

constexpr int GBM = 64, GBN = 64, GBK = 32;

template <typename T> struct GenericTraits;
template <> struct GenericTraits<bf16> { static constexpr int LDK = 40; };   // 80-B rows
template <> struct GenericTraits<float> { static constexpr int LDK = 33; };

// stage a [64 rows][32 k] tile of op(X) into LDS as [row][k], zero-filling
// everything outside (rows_total, K).  `vec`: rows start 16-B aligned and the leading
// dimension keeps them so -> 16-B global loads along the contiguous dimension for the
// interior of the matrix (edges fall back to the element-wise path).
template <typename T>
__device__ __forceinline__ void stage_tile(T* lds, const T* X, int64_t ld, int kmajor,
                                           int64_t row0, int64_t rows_total, int64_t k0,
                                           int64_t K, int tid, int vec) {
  constexpr int LDK = GenericTraits<T>::LDK;
  constexpr int V = 16 / sizeof(T);                 // elements per 16-B vector
  if (vec) {
    const bool interior = kmajor ? (k0 + GBK <= K) : (row0 + GBM <= rows_total);
    if (interior) {
      constexpr int NVEC = GBM * GBK / V;           // 256 (bf16) or 512 (fp32) vectors
#pragma unroll
      for (int i = 0; i < NVEC / 256; ++i) {
        const int idx = tid + i * 256;
        if (kmajor) {                               // vector along k
          const int r = idx / (GBK / V), kc = (idx % (GBK / V)) * V;
          const int64_t gr = row0 + r;
          if (gr < rows_total) {
            const auto v = *reinterpret_cast<const f32x4*>(X + gr * ld + k0 + kc);
            const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int j = 0; j < V; ++j) lds[r * LDK + kc + j] = e[j];
          } else {
#pragma unroll
            for (int j = 0; j < V; ++j) lds[r * LDK + kc + j] = from_f32<T>(0.f);
          }
        } else {                                    // vector along the row index
          const int kk = idx / (GBM / V), rc = (idx % (GBM / V)) * V;
          const int64_t gk = k0 + kk;
          if (gk < K) {
            const auto v = *reinterpret_cast<const f32x4*>(X + gk * ld + row0 + rc);
            const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int j = 0; j < V; ++j) lds[(rc + j) * LDK + kk] = e[j];
          } else {
#pragma unroll
            for (int j = 0; j < V; ++j) lds[(rc + j) * LDK + kk] = from_f32<T>(0.f);
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < (GBM * GBK) / 256; ++i) {
    const int idx = tid + i * 256;
    int r, kk;
    if (kmajor) { r = idx >> 5; kk = idx & 31; }   // consecutive lanes walk k
    else        { r = idx & 63; kk = idx >> 6; }   // consecutive lanes walk rows
    const int64_t gr = row0 + r, gk = k0 + kk;
    T v = from_f32<T>(0.f);
    if (gr < rows_total && gk < K) v = kmajor ? X[gr * ld + gk] : X[gk * ld + gr];
    lds[r * LDK + kk] = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmArgs g) {
  constexpr int LDK = GenericTraits<T>::LDK;
  __shared__ __attribute__((aligned(16))) T As[GBM * LDK];
  __shared__ __attribute__((aligned(16))) T Bs[GBN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.y * GBM, n0 = (int64_t)blockIdx.x * GBN;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);
  if (g.batch > 1) {
    const int64_t z = blockIdx.z, zo = z / g.batch_inner, zi = z % g.batch_inner;
    A += zo * g.a_bs[0] + zi * g.a_bs[1];
    B += zo * g.b_bs[0] + zi * g.b_bs[1];
    const int64_t co = zo * g.c_bs[0] + zi * g.c_bs[1];
    g.e.C = reinterpret_cast<char*>(g.e.C) + co * (g.e.c_bf16 ? 2 : 4);
  }

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  for (int64_t k0 = 0; k0 < g.K; k0 += GBK) {
    stage_tile<T>(As, A, g.lda, g.a_km, m0, g.M, k0, g.K, tid, g.vec_a);
    stage_tile<T>(Bs, B, g.ldb, g.b_km, n0, g.N, k0, g.K, tid, g.vec_b);
    __syncthreads();
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(&As[(wm * 32 + lr) * LDK + s * 16 + 8 * lh]);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bs[(wn * 32 + lr) * LDK + s * 16 + 8 * lh]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float a = As[(wm * 32 + lr) * LDK + 2 * s + lh];
        const float b = Bs[(wn * 32 + lr) * LDK + 2 * s + lh];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }

  const int64_t n = n0 + wn * 32 + lr;
  if (n >= g.N) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + wm * 32 + mfma32_row(r, lh);
    if (m < g.M) epi_store1_rt(g.e, m, n, acc[r]);
  }
}

// ---- skinny fp32 products (the classifier head: 256 x 10 x 768 and its two gradients).  The 64x64-tile kernel above runs
// them on 4-12 workgroups as a staged loop of K / 32 dependent global -> LDS -> MFMA steps: 39 us per launch for 4 MFLOP,
// three launches per step.  Plain FMAs with every lane on its own part of the contraction take a few us (round 3).
//   FORM 0  C[m][n] = sum_k A[m][k] B[n][k],  N <= 16   (logits = feat W^T): one workgroup per row m, a quarter of k per wave,
//           lanes stride over k, the N partial sums folded across the wave in a fixed butterfly order
//   FORM 1  C[m][n] = sum_k A[m][k] B[k][n],  K <= 32   (d feat = d logits W): one thread per output
//   FORM 2  C[m][n] = sum_k A[k][m] B[k][n],  M <= 32   (d W = d logits^T feat): 32 outputs x 8 k-slices per workgroup
// Every sum has a fixed order (deterministic); the epilogue is the generic kernel's (epi_store1_rt).
constexpr int SKINNY_MAXN = 16;
template <int FORM>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs g) {
  const float* __restrict__ A = reinterpret_cast<const float*>(g.A);
  const float* __restrict__ B = reinterpret_cast<const float*>(g.B);
  if constexpr (FORM == 0) {
    // a workgroup per row m: its four waves take a quarter of the contraction each (lanes stride over k), fold their N
    // partial sums across the wave, and wave 0 adds the four quarters in a fixed order
    __shared__ float quarter[4][SKINNY_MAXN];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t m = blockIdx.x;
    float acc[SKINNY_MAXN];
#pragma unroll
    for (int n = 0; n < SKINNY_MAXN; ++n) acc[n] = 0.f;
    const int N = (int)g.N;
    // rows of B beyond N re-read the last one, unconditionally: a branch around a load would give each its own round trip
    const float* Bn[SKINNY_MAXN];
#pragma unroll
    for (int n = 0; n < SKINNY_MAXN; ++n) Bn[n] = B + (int64_t)(n < N ? n : N - 1) * g.ldb;
    const float* Am = A + m * g.lda;
    const int64_t per = ((g.K + 3) / 4 + 63) / 64 * 64;                      // a quarter, in whole 64-lane strides
    const int64_t k0 = w * per, k1 = k0 + per < g.K ? k0 + per : g.K;
#pragma unroll 4
    for (int64_t kb = k0; kb < k1; kb += 64) {
      const int64_t k = kb + lane, kc = k < k1 ? k : k1 - 1;               // clamped loads, zeroed factor at the ragged end
      const float a = k < k1 ? Am[kc] : 0.f;
#pragma unroll
      for (int n = 0; n < SKINNY_MAXN; ++n) acc[n] = fmaf(a, Bn[n][kc], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < SKINNY_MAXN; ++n) {
      float v = acc[n];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == n) quarter[w][n] = v;
    }
    __syncthreads();
    if (w == 0 && lane < N) epi_store1_rt(g.e, m, lane, (quarter[0][lane] + quarter[1][lane]) + (quarter[2][lane] + quarter[3][lane]));
  } else if constexpr (FORM == 1) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.M * g.N) return;
    const int64_t m = i / g.N, n = i % g.N;
    float acc = 0.f;
    for (int64_t k = 0; k < g.K; ++k) acc = fmaf(A[m * g.lda + k], B[k * g.ldb + n], acc);
    epi_store1_rt(g.e, m, n, acc);
  } else {
    // a workgroup = 32 consecutive outputs (n fastest) x 8 slices of the contraction; the slices are added in a fixed order
    __shared__ float part[8][32];
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + o, tot = g.M * g.N;
    const int64_t ic = i < tot ? i : tot - 1;        // clamped: every thread loads, only valid ones store
    const int64_t m = ic / g.N, n = ic % g.N;
    const int64_t per = (g.K + 7) / 8, k0 = sl * per, k1 = k0 + per < g.K ? k0 + per : g.K;
    float acc = 0.f;
#pragma unroll 4
    for (int64_t k = k0; k < k1; ++k) acc = fmaf(A[k * g.lda + m], B[k * g.ldb + n], acc);
    part[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && i < tot) {
      float v = part[0][o];
#pragma unroll
      for (int q = 1; q < 8; ++q) v += part[q][o];
      epi_store1_rt(g.e, m, n, v);
    }
  }
}

}  // namespace

static std::atomic<int> g_skinny{1};             // diagnostic / test hook: 0 = the head's products stay on the 64x64-tile kernel
extern "C" void vitmi_debug_gemm_skinny(int on) { g_skinny = on != 0; }
// which fp32 problems take the skinny kernel: -1 = none
static int gemm_skinny_form(const GemmArgs& g, int in_bf16) {
  if (!g_skinny || in_bf16 || g.batch != 1 || g.M * g.N > (1ll << 24)) return -1;
  if (g.a_km && g.b_km) return (g.N <= SKINNY_MAXN && g.K >= 64) ? 0 : -1;
  if (g.a_km && !g.b_km) return g.K <= 32 ? 1 : -1;
  if (!g.a_km && !g.b_km) return (g.M <= 32 && g.K <= 8192) ? 2 : -1;
  return -1;
}
static int gemm_skinny_launch(const GemmArgs& g, int form, hipStream_t stream) {
  if (form == 0) hipLaunchKernelGGL(gemm_skinny_kernel<0>, dim3((unsigned)g.M), dim3(256), 0, stream, g);
  else {
    if (form == 1) hipLaunchKernelGGL(gemm_skinny_kernel<1>, dim3((unsigned)((g.M * g.N + 255) / 256)), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL(gemm_skinny_kernel<2>, dim3((unsigned)((g.M * g.N + 31) / 32)), dim3(256), 0, stream, g);
  }
  return vitmi_check_launch("gemm_skinny_kernel");
}

// Cache policy of the tile kernels' output stores (storev_pol): -1 = automatic, else forced (diagnostic hook).
// Automatic: `nt` for the wide activation outputs that are written once and read by a LATER kernel
// (qkv, fc1's two outputs, the data gradients: >= 64 MB), plain for everything else (the residual
// stream's new rows are read back at once by the LayerNorm that follows, weight-gradient tiles are small).
static std::atomic<int> g_c_policy{-1};
extern "C" void vitmi_debug_gemm_store_policy(int p) { g_c_policy = p; }
static std::atomic<int> g_nt_min_mb{64};         // diagnostic hook: outputs / side inputs of at least this many MB take the nt policy
extern "C" void vitmi_debug_gemm_nt_min_mb(int mb) { g_nt_min_mb = mb > 0 ? mb : 64; }
static std::atomic<int> g_side_nt{-1};           // diagnostic hook: -1 = automatic (wide side inputs), 0 / 1 forced
extern "C" void vitmi_debug_gemm_side_nt(int v) { g_side_nt = v; }
static std::atomic<int> g_side_depth{3};         // diagnostic hook: strips of the epilogue's side input in flight (bf16 outputs): 1 or 3
extern "C" void vitmi_debug_gemm_side_depth(int d) { g_side_depth = d >= 3 ? 3 : 1; }
static int side_policy(const vitmi_gemm_desc* d) {
  if (g_side_nt >= 0) return g_side_nt;
  return d->M * d->N * 2 >= ((int64_t)g_nt_min_mb << 20) ? 1 : 0;
}
static int store_policy(const vitmi_gemm_desc* d) {
  if (g_c_policy >= 0) return g_c_policy;
  if (d->epilogue == VITMI_EPI_RESIDUAL || d->epilogue == VITMI_EPI_PATCH_POS) return 0;
  const int64_t bytes = d->M * d->N * (d->c_dtype == VITMI_BF16 ? 2 : 4);
  return bytes >= ((int64_t)g_nt_min_mb << 20) ? 2 : 0;
}
static int build_args(const vitmi_gemm_desc* d, GemmArgs* out) {
  VITMI_REQUIRE(d, VITMI_E_BADARG, "gemm: null descriptor");
  VITMI_REQUIRE(d->struct_size == (int64_t)sizeof(vitmi_gemm_desc), VITMI_E_BADARG,
                "gemm: descriptor struct_size %lld != %lld (caller built against another vitmi.h?)",
                (long long)d->struct_size, (long long)sizeof(vitmi_gemm_desc));
  VITMI_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, VITMI_E_BADARG, "gemm: M,N,K must be > 0 (got %lld,%lld,%lld)",
                (long long)d->M, (long long)d->N, (long long)d->K);
  VITMI_REQUIRE(d->A && d->B && d->C, VITMI_E_BADARG, "gemm: A, B, C must be non-null");
  VITMI_REQUIRE(d->in_dtype == VITMI_F32 || d->in_dtype == VITMI_BF16, VITMI_E_DTYPE, "gemm: bad in_dtype %d", d->in_dtype);
  VITMI_REQUIRE(d->c_dtype == VITMI_F32 || d->c_dtype == VITMI_BF16, VITMI_E_DTYPE, "gemm: bad c_dtype %d", d->c_dtype);
  VITMI_REQUIRE(d->epilogue >= VITMI_EPI_STORE && d->epilogue <= VITMI_EPI_PATCH_POS, VITMI_E_BADARG, "gemm: bad epilogue %d", d->epilogue);
  VITMI_REQUIRE(d->lda >= (d->a_kmajor ? d->K : d->M), VITMI_E_BADARG, "gemm: lda too small");
  VITMI_REQUIRE(d->ldb >= (d->b_kmajor ? d->K : d->N), VITMI_E_BADARG, "gemm: ldb too small");
  VITMI_REQUIRE(d->ldc >= d->N, VITMI_E_BADARG, "gemm: ldc too small");
  GemmArgs g;
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.A = d->A; g.lda = d->lda; g.a_km = d->a_kmajor ? 1 : 0;
  g.B = d->B; g.ldb = d->ldb; g.b_km = d->b_kmajor ? 1 : 0;
  g.ws = d->workspace; g.ws_bytes = d->workspace_bytes;
  g.rfold = 0;
  g.band = 0;
  g.stag_cycles = 0; g.stag_phases = 1;
  g.side_depth = g_side_depth;
  g.launch_flags = d->launch_flags;
  g.strict_wait = g_gemm_strict_wait;
  g.zero_cnt = nullptr; g.zero_n = 0; g.fix_cnt = nullptr;
  g.A2 = g.B2 = nullptr; g.lda2 = g.ldb2 = g.M2 = g.N2 = 0; g.tiles1 = g.tiles_n2 = 0; g.ws2 = nullptr;
  g.dbg = g_gemm_dbg;
  g.dbg_blocks = g_gemm_dbg_blocks;
  g.dbg_alias = g_gemm_alias;
  g.batch = d->batch > 1 ? d->batch : 1;
  g.batch_inner = d->batch_inner > 0 ? d->batch_inner : 1;
  for (int i = 0; i < 2; ++i) { g.a_bs[i] = d->a_bs[i]; g.b_bs[i] = d->b_bs[i]; g.c_bs[i] = d->c_bs[i]; }
  {
    const size_t es = dtype_size(d->in_dtype);
    const int64_t v = 16 / (int64_t)es;
    auto ok = [&](const void* p, int64_t ld, const int64_t* bs) {
      return is_aligned(p, 16) && ld % v == 0 && (g.batch == 1 || (bs[0] % v == 0 && bs[1] % v == 0));
    };
    g.vec_a = ok(d->A, d->lda, d->a_bs) ? 1 : 0;
    g.vec_b = ok(d->B, d->ldb, d->b_bs) ? 1 : 0;
  }
  if (g.ws && !is_aligned(g.ws, 16)) { g.ws = nullptr; g.ws_bytes = 0; }
  EpiArgs& e = g.e;
  e.mode = d->epilogue;
  e.C = d->C; e.ldc = d->ldc; e.c_bf16 = d->c_dtype == VITMI_BF16;
  e.C2 = d->C2; e.ldc2 = d->ldc2;
  // second output: pre-activation in C's dtype (BIAS_GELU) / un-scaled branch output in the
  // operand dtype (RESIDUAL with LayerScale, kept for d gamma)
  e.c2_bf16 = d->epilogue == VITMI_EPI_RESIDUAL ? (d->in_dtype == VITMI_BF16) : (d->c_dtype == VITMI_BF16);
  e.bias = d->bias;
  e.R = d->R; e.ldr = d->ldr; e.r_bf16 = d->r_dtype == VITMI_BF16;
  e.gamma = d->gamma;
  e.AUX = d->AUX; e.ldaux = d->ldaux; e.aux_bf16 = d->in_dtype == VITMI_BF16;
  e.aux_deriv = d->aux_is_derivative != 0;
  e.c_policy = store_policy(d);
  e.side_nt = side_policy(d);
  e.pos = d->pos; e.n_tok = d->n_tok; e.ldpos = d->N; e.cls = d->cls;
  e.alpha = d->alpha == 0.f ? 1.f : d->alpha;
  e.accumulate = d->accumulate;
  e.rowscale = d->epilogue == VITMI_EPI_RESIDUAL ? d->rowscale : nullptr;
  e.rpg = d->rows_per_group > 0 ? d->rows_per_group : 1;
  e.colsum_part = d->colsum_part;
  VITMI_REQUIRE(!e.rowscale || (d->M < (1ll << 32) && e.rpg < (1ll << 32)), VITMI_E_SHAPE, "gemm: rowscale needs M < 2^32");
  VITMI_REQUIRE(g.batch == 1 || (d->epilogue == VITMI_EPI_STORE && !d->accumulate && g.batch % g.batch_inner == 0 && g.batch <= 65535),
                VITMI_E_BADARG, "gemm: batched form supports EPI_STORE without accumulate, batch %% batch_inner == 0, batch <= 65535");
  switch (d->epilogue) {
    case VITMI_EPI_STORE:
      VITMI_REQUIRE(!d->accumulate || d->c_dtype == VITMI_F32, VITMI_E_DTYPE, "gemm: accumulate needs an fp32 C");
      break;
    case VITMI_EPI_BIAS_GELU:
      VITMI_REQUIRE(!d->C2 || d->ldc2 >= d->N, VITMI_E_BADARG, "gemm: ldc2 too small");
      break;
    case VITMI_EPI_RESIDUAL:
      VITMI_REQUIRE(!d->C2 || d->ldc2 >= d->N, VITMI_E_BADARG, "gemm: ldc2 too small");
      VITMI_REQUIRE(d->R && d->ldr >= d->N, VITMI_E_BADARG, "gemm: EPI_RESIDUAL needs R with ldr >= N");
      VITMI_REQUIRE(d->r_dtype == d->c_dtype, VITMI_E_DTYPE, "gemm: EPI_RESIDUAL needs r_dtype == c_dtype");
      break;
    case VITMI_EPI_DGELU:
      VITMI_REQUIRE(d->AUX && d->ldaux >= d->N, VITMI_E_BADARG, "gemm: EPI_DGELU needs AUX with ldaux >= N");
      break;
    case VITMI_EPI_PATCH_POS:
      VITMI_REQUIRE(d->pos && d->n_tok > 0, VITMI_E_BADARG, "gemm: EPI_PATCH_POS needs pos and n_tok");
      break;
  }
  *out = g;
  return 0;
}

int gemm_small_form(const GemmArgs& g, int in_bf16);
int gemm_small_launch(const GemmArgs& g, int form, hipStream_t stream);
static std::atomic<int> g_small_override{-1};    // diagnostic / test hook: 0 = never, 1 = also when impl == GENERIC, -1 = default
extern "C" void vitmi_debug_gemm_small(int mode) { g_small_override = mode; }
static bool small_lds_ok(const GemmArgs& g, int form) {
  if (form == 0) return ((g.N + 15) / 16 * 16) * (64 * 2 + 16) <= 96 * 1024;
  const int64_t Kp = (g.K + 31) / 32 * 32, Mp = (g.M + 15) / 16 * 16;
  const int64_t ry = g.N * 2 + (((g.N * 2) % 128 == 0) ? 32 : 0), rs = Mp * 2 + (((Mp * 2) % 128 == 0) ? 32 : 0);
  return Kp * ry + (form == 2 ? 64 * rs : 0) <= 96 * 1024;
}

extern "C" int vitmi_gemm_uses_fast(const vitmi_gemm_desc* d) {
  GemmArgs g;
  if (build_args(d, &g) != 0) return 0;
  if (d->impl == VITMI_GEMM_GENERIC || g.batch > 1) return 0;
  return gemm_fast_supported(g, d->in_dtype == VITMI_BF16) ? 1 : 0;
}

extern "C" size_t vitmi_gemm_workspace(const vitmi_gemm_desc* d) {
  GemmArgs g;
  if (build_args(d, &g) != 0 || d->impl == VITMI_GEMM_GENERIC) return 0;
  if (g.batch > 1 || !gemm_fast_supported(g, d->in_dtype == VITMI_BF16)) return 0;
  return gemm_fast_workspace(g);
}

size_t gemm_fast_pair_workspace(const GemmArgs& a, const GemmArgs& b);
int gemm_fast_pair_launch(const GemmArgs& a, const GemmArgs& b, void* ws, size_t ws_bytes, hipStream_t stream);

static std::atomic<int> g_pair{1};               // diagnostic hook: 0 = never pair (two launches), for A/B inside the step
extern "C" void vitmi_debug_gemm_pair(int on) { g_pair = on != 0; }
static bool pair_args(const vitmi_gemm_desc* d0, const vitmi_gemm_desc* d1, GemmArgs* a, GemmArgs* b) {
  if (!g_pair) return false;
  if (build_args(d0, a) != 0 || build_args(d1, b) != 0) return false;
  if (d0->in_dtype != VITMI_BF16 || d1->in_dtype != VITMI_BF16) return false;
  if (d0->impl == VITMI_GEMM_GENERIC || d1->impl == VITMI_GEMM_GENERIC) return false;
  return gemm_fast_supported(*a, true) && gemm_fast_supported(*b, true);
}
extern "C" size_t vitmi_gemm_pair_workspace(const vitmi_gemm_desc* d0, const vitmi_gemm_desc* d1) {
  GemmArgs a, b;
  if (!pair_args(d0, d1, &a, &b)) return 0;
  return gemm_fast_pair_workspace(a, b);
}
extern "C" int vitmi_gemm_pair(const vitmi_gemm_desc* d0, const vitmi_gemm_desc* d1, void* workspace, size_t workspace_bytes,
                               void* stream_) {
  GemmArgs a, b;
  if (pair_args(d0, d1, &a, &b)) {
    const int rc = gemm_fast_pair_launch(a, b, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream_));
    if (rc != -1000) return rc;
  }
  if (int rc = vitmi_gemm(d0, stream_)) return rc;      // not pairable: one after the other, same results
  return vitmi_gemm(d1, stream_);
}

extern "C" int vitmi_gemm(const vitmi_gemm_desc* d, void* stream_) {
  GemmArgs g;
  int rc = build_args(d, &g);
  if (rc) return rc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const bool in_bf16 = d->in_dtype == VITMI_BF16;
  const bool fast_ok = g.batch == 1 && gemm_fast_supported(g, in_bf16);
  if (d->impl == VITMI_GEMM_FAST)
    VITMI_REQUIRE(fast_ok, VITMI_E_SHAPE, "gemm: VITMI_GEMM_FAST requested but shape/dtype/alignment unsupported (M=%lld N=%lld K=%lld)",
                  (long long)d->M, (long long)d->N, (long long)d->K);
  if (fast_ok && d->impl != VITMI_GEMM_GENERIC) return gemm_fast_launch(g, stream);
  VITMI_REQUIRE(!d->colsum_part, VITMI_E_BADARG,
                "gemm: colsum_part is produced only by the aligned bf16 path with EPI_DGELU (ask vitmi_gemm_uses_fast)");

  if (d->impl != VITMI_GEMM_GENERIC || g_small_override == 1) {     // batched form: one workgroup per small problem
    const int form = g_small_override == 0 ? -1 : gemm_small_form(g, in_bf16 ? 1 : 0);
    if (form >= 0 && small_lds_ok(g, form)) return gemm_small_launch(g, form, stream);
  }
  if (d->impl != VITMI_GEMM_GENERIC) {                              // the classifier head's fp32 products
    const int sform = gemm_skinny_form(g, in_bf16 ? 1 : 0);
    if (sform >= 0) return gemm_skinny_launch(g, sform, stream);
  }
  dim3 grid((unsigned)((g.N + GBN - 1) / GBN), (unsigned)((g.M + GBM - 1) / GBM), (unsigned)g.batch);
  VITMI_REQUIRE(grid.y <= 65535u, VITMI_E_SHAPE, "gemm: M too large for the generic kernel grid");
  if (in_bf16) hipLaunchKernelGGL(gemm_generic_kernel<bf16>, grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL(gemm_generic_kernel<float>, grid, dim3(256), 0, stream, g);
  return vitmi_check_launch("gemm_generic_kernel");
}

// every diagnostic switch of this file back to its default (vitmi_debug_reset, core.cpp)
void vitmi_debug_reset_gemm() {
  g_gemm_strict_wait = 0;
  g_gemm_dbg = nullptr;
  g_gemm_dbg_blocks = 64;
  g_gemm_alias = 0;
  g_skinny = 1;
  g_c_policy = -1;
  g_nt_min_mb = 64;
  g_side_nt = -1;
  g_side_depth = 3;
  g_small_override = -1;
  g_pair = 1;
}
