// Batched small GEMMs: one workgroup per problem, for the per-(image, head) products of an
// attention whose scores must live in HBM (CaiT talking-heads, models/cait.py:113-127:
// q k^T, P' v and their four gradients).  A problem is ~196 x 196 x 48: the generic
// 64x64-tile kernel spends its time in per-tile staging and barriers (~95 us per batch of
// 512); here a workgroup of four waves owns the whole problem, k-contiguous operands go
// from global memory straight into MFMA fragments, k-strided operands are staged once in
// LDS in their natural row-major form and read transposed with ds_read_b64_tr_b16.
//   FORM 0  C[i][j] = alpha sum_k X[i][k] Y[j][k]      (a_kmajor, b_kmajor)   K = head dim
//   FORM 1  C[i][n] = alpha sum_k S[i][k] Y[k][n]      (a_kmajor, !b_kmajor)  K = tokens
//   FORM 2  C[j][n] = alpha sum_k S[k][j] Y[k][n]      (!a_kmajor, !b_kmajor) K = tokens
// v_mfma_f32_16x16x32_bf16, D[m][n]: the lane owns output column n = lane & 15 (consecutive
// lanes write consecutive elements of a C row), rows 4*(lane >> 4) + r.
#include <atomic>
#include "common.h"
#include "epilogue.h"

namespace {

constexpr int SB_MAXN = 64;      // FORM 1/2: columns of Y (head dim)
constexpr int SB_MAXK = 256;     // FORM 1/2: contraction length staged in LDS (tokens)
constexpr int SB_CHUNK = 64;     // FORM 2: rows of S resident per pass

struct SmallArgs {
  const bf16* A; const bf16* B; bf16* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int64_t batch_inner, a_bs[2], b_bs[2], c_bs[2];
  float alpha;
};

__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
  return z;
}
// 8 consecutive k of row `row` (clamped by the caller), elements at k >= K read as zero.
// The load is unconditional (a piece wholly beyond K re-reads the row's first piece and is
// zeroed): a lane-dependent branch around it would give every fragment its own round trip.
__device__ __forceinline__ bf16x8 row_frag(const bf16* X, int64_t ld, int row, int k, int K) {
  bf16x8 v = *reinterpret_cast<const bf16x8*>(X + (int64_t)row * ld + (k < K ? k : 0));
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (k + e >= K) v[e] = (bf16)0.f;
  return v;
}
// operand fragment whose k runs along the ROWS of an LDS image [k][c] (row pitch RS bytes):
// lane (g = lane >> 4, i = lane & 15) receives k = k0 + 8g .. +7 of column c0 + i
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int RS, int k0, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const char* p = img + (k0 + 8 * g + (i >> 2)) * RS + (c0 + 4 * (i & 3)) * 2;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p + 4 * RS));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
__device__ __forceinline__ void store_tile(bf16* C, int64_t ldc, int m0, int n0, int M, int N, const f32x4& acc,
                                           float alpha, int lane) {
  const int n = n0 + (lane & 15);
  if (n >= N) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + 4 * (lane >> 4) + r;
    if (m < M) C[(int64_t)m * ldc + n] = (bf16)(acc[r] * alpha);
  }
}
// transposed accumulator: lane owns row m0 + (lane & 15), columns n0 + 4*(lane >> 4) + r
__device__ __forceinline__ void store_tile_T(bf16* C, int64_t ldc, int m0, int n0, int M, int N, const f32x4& acc,
                                             float alpha, int lane) {
  const int m = m0 + (lane & 15), n = n0 + 4 * (lane >> 4);
  if (m >= M || n >= N) return;
  bf16* p = C + (int64_t)m * ldc + n;
  if (n + 4 <= N && (ldc & 3) == 0 && ((uintptr_t)C & 7) == 0) {
    bf16x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (bf16)(acc[r] * alpha);
    *reinterpret_cast<bf16x4*>(p) = v;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < N) p[r] = (bf16)(acc[r] * alpha);
  }
}
__device__ __forceinline__ int y_pitch(int N) { return N * 2 + (((N * 2) % 128 == 0) ? 32 : 0); }

// rows [r0, r0 + rows) x [0, cols) of X -> LDS image (pitch RS), zero outside (R_total, cols)
__device__ __forceinline__ void stage_rows(char* img, int RS, const bf16* X, int64_t ld, int r0, int rows,
                                           int R_total, int cols, int cols_pad, int tid, int nthr) {
  const int cpr = cols_pad / 8;                    // 16-B pieces per staged row
  const int last_pc = (cols - 1) / 8 * 8;
  for (int c = tid; c < rows * cpr; c += nthr) {
    const int r = c / cpr, pc = (c % cpr) * 8;
    // unconditional load from a clamped position, zeroed afterwards (a branch around the
    // load would make every piece wait for its own round trip)
    const bool ok = r0 + r < R_total && pc < cols;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(X + (int64_t)min(r0 + r, R_total - 1) * ld + min(pc, last_pc));
    if (!ok) v = zero8();
    *reinterpret_cast<bf16x8*>(img + r * RS + pc * 2) = v;
  }
}

// the same in two halves, so that a chunk's loads can fly while the previous chunk is
// contracted: NP pieces of 16 B per thread (rows * cols_pad / 8 <= 256 * NP)
template <int NP>
__device__ __forceinline__ void load_rows(bf16x8 (&v)[NP], const bf16* X, int64_t ld, int r0, int rows, int R_total,
                                          int cols, int cols_pad, int tid) {
  const int cpr = cols_pad / 8, n = rows * cpr;
  const int last_pc = (cols - 1) / 8 * 8;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = tid + i * 256, cc = min(c, n - 1);
    const int r = cc / cpr, pc = (cc - r * cpr) * 8;
    v[i] = *reinterpret_cast<const bf16x8*>(X + (int64_t)min(r0 + r, R_total - 1) * ld + min(pc, last_pc));   // unconditional
    if (!(r0 + r < R_total && pc < cols)) v[i] = zero8();
  }
}
template <int NP>
__device__ __forceinline__ void store_rows(char* img, int RS, const bf16x8 (&v)[NP], int rows, int cols_pad, int tid) {
  const int cpr = cols_pad / 8, n = rows * cpr;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = tid + i * 256;
    if (c < n) {
      const int r = c / cpr, pc = (c - r * cpr) * 8;
      *reinterpret_cast<bf16x8*>(img + r * RS + pc * 2) = v[i];
    }
  }
}

template <int FORM>
__global__ __launch_bounds__(256) void gemm_small_kernel(SmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t z = blockIdx.x, zo = z / a.batch_inner, zi = z % a.batch_inner;
  const bf16* A = a.A + zo * a.a_bs[0] + zi * a.a_bs[1];
  const bf16* B = a.B + zo * a.b_bs[0] + zi * a.b_bs[1];
  bf16* C = a.C + zo * a.c_bs[0] + zi * a.c_bs[1];
  const int M = a.M, N = a.N, K = a.K;
  const int tiles_m = (M + 15) / 16, tiles_n = (N + 15) / 16;
  const int g = lane >> 4, li = lane & 15;
  // gridDim.y workgroups share one problem (small batches: 512 problems are two workgroups per
  // CU, too few waves to hide the staging round trips): part p takes row tiles 4p + w, stepping
  // by 4 * parts; every part stages the shared operand for itself (19-28 KB, from L2)
  const int part = blockIdx.y, parts = gridDim.y;

  if constexpr (FORM == 0) {
    const int ksteps = (K + 31) / 32;              // <= 2 (host checks K <= 64)
    // Y (the B rows) is staged once in LDS: every wave walks all of its column tiles, and an
    // L2 round trip per tile (two waves per SIMD cannot hide it) was the whole run time
    const int RSB = 64 * 2 + 16;                   // 64 k (zero beyond K) + pad: conflict-free row reads
    stage_rows(smem, RSB, B, a.ldb, 0, tiles_n * 16, N, K, 64, tid, 256);
    __syncthreads();
    // the NEXT row tile's A fragments are loaded while this one's column tiles are contracted and stored (round 3: the
    // load of a row tile used to sit at the top of its iteration, one exposed global round trip per 16 rows and wave;
    // q k^T 73.7 -> 69.1 us, dO v^T 77.7 -> 70.3 us at CaiT-S24's 2048 problems, tools/cait_bgemm_bench.py)
    bf16x8 afn[2];
    {
      const int row0 = min((w + 4 * part) * 16 + li, M - 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) afn[ks] = ks < ksteps ? row_frag(A, a.lda, row0, ks * 32 + 8 * g, K) : zero8();
    }
    for (int rt = w + 4 * part; rt < tiles_m; rt += 4 * parts) {
      bf16x8 af[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[ks] = afn[ks];
      {
        const int rown = min(min(rt + 4 * parts, tiles_m - 1) * 16 + li, M - 1);       // unconditional (the last one re-reads)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) afn[ks] = ks < ksteps ? row_frag(A, a.lda, rown, ks * 32 + 8 * g, K) : zero8();
      }
      for (int ct = 0; ct < tiles_n; ++ct) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (ks < ksteps) {
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(smem + (ct * 16 + li) * RSB + (ks * 32 + 8 * g) * 2);
            // operands swapped: D^T[j][i], so the lane owns row i of C and 4 CONSECUTIVE
            // columns j: one 8-byte store instead of four 2-byte ones (this form writes
            // N x N scores per problem and is store-issue bound)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, af[ks], acc, 0, 0, 0);
          }
        }
        store_tile_T(C, a.ldc, rt * 16, ct * 16, M, N, acc, a.alpha, lane);
      }
    }
  } else {
    const int Kp = (K + 31) / 32 * 32;
    const int RSY = y_pitch(N);
    const int Np = tiles_n * 16;
    char* Ys = smem;                               // [Kp][N] row-major, rows >= K zero
    stage_rows(Ys, RSY, B, a.ldb, 0, Kp, K, N, Np, tid, 256);
    if constexpr (FORM == 1) {
      __syncthreads();
      for (int rt = w + 4 * part; rt < tiles_m; rt += 4 * parts) {
        const int row = min(rt * 16 + li, M - 1);
        f32x4 acc[SB_MAXN / 16];
#pragma unroll
        for (int ct = 0; ct < SB_MAXN / 16; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (all k-steps' fragments in one batch of loads was tried: 38 -> 144 VGPRs, three
        // instead of eight waves per SIMD, 16.7 -> 28 us; occupancy hides this chain better.  Round 3: only the NEXT
        // k-step's fragment in flight, +4 VGPRs: 73.5 / 76.6 -> 73.8 / 79.5 us at 2048 problems — no gain either)
        for (int ks = 0; ks < Kp / 32; ++ks) {
          const bf16x8 af = row_frag(A, a.lda, row, ks * 32 + 8 * g, K);
#pragma unroll
          for (int ct = 0; ct < SB_MAXN / 16; ++ct) {
            if (ct < tiles_n) {
              const bf16x8 bf = tr_frag(Ys, RSY, ks * 32, ct * 16, lane);
              acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[ct], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int ct = 0; ct < SB_MAXN / 16; ++ct)
          if (ct < tiles_n) store_tile(C, a.ldc, rt * 16, ct * 16, M, N, acc[ct], a.alpha, lane);
      }
    } else {
      // FORM 2: S = A is [K][M]; its rows (k) stream through LDS SB_CHUNK at a time
      const int Mp = tiles_m * 16;
      const int RSS = Mp * 2 + (((Mp * 2) % 128 == 0) ? 32 : 0);
      char* Ss = smem + Kp * RSY;
      constexpr int RT_MAX = 4;                    // row tiles per wave: M <= 256
      f32x4 acc[RT_MAX][SB_MAXN / 16];
#pragma unroll
      for (int q = 0; q < RT_MAX; ++q)
#pragma unroll
        for (int ct = 0; ct < SB_MAXN / 16; ++ct) acc[q][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      // S streams through LDS one chunk at a time, the NEXT chunk's loads in registers meanwhile
      constexpr int NP = SB_CHUNK * (256 / 8) / 256;           // M <= 256: <= 32 pieces per row, 64 rows, 256 threads
      bf16x8 nxt[NP];
      load_rows<NP>(nxt, A, a.lda, 0, min(SB_CHUNK, Kp), K, M, Mp, tid);
      for (int kc = 0; kc < Kp; kc += SB_CHUNK) {
        __syncthreads();                           // previous chunk consumed (and Ys staged)
        const int rows = min(SB_CHUNK, Kp - kc);
        store_rows<NP>(Ss, RSS, nxt, rows, Mp, tid);
        if (kc + SB_CHUNK < Kp) load_rows<NP>(nxt, A, a.lda, kc + SB_CHUNK, min(SB_CHUNK, Kp - kc - SB_CHUNK), K, M, Mp, tid);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RT_MAX; ++q) {
          const int rt = w + 4 * (part + parts * q);
          if (rt < tiles_m) {
            for (int ks = 0; ks < rows / 32; ++ks) {
              const bf16x8 af = tr_frag(Ss, RSS, ks * 32, rt * 16, lane);
#pragma unroll
              for (int ct = 0; ct < SB_MAXN / 16; ++ct) {
                if (ct < tiles_n) {
                  const bf16x8 bf = tr_frag(Ys, RSY, kc + ks * 32, ct * 16, lane);
                  acc[q][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[q][ct], 0, 0, 0);
                }
              }
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < RT_MAX; ++q) {
        const int rt = w + 4 * (part + parts * q);
        if (rt < tiles_m) {
#pragma unroll
          for (int ct = 0; ct < SB_MAXN / 16; ++ct)
            if (ct < tiles_n) store_tile(C, a.ldc, rt * 16, ct * 16, M, N, acc[q][ct], a.alpha, lane);
        }
      }
    }
  }
}

inline size_t small_lds(int form, int M, int N, int K) {
  if (form == 0) return (size_t)((N + 15) / 16 * 16) * (64 * 2 + 16);
  const int Kp = (K + 31) / 32 * 32;
  const int RSY = N * 2 + (((N * 2) % 128 == 0) ? 32 : 0);
  size_t b = (size_t)Kp * RSY;
  if (form == 2) {
    const int Mp = (M + 15) / 16 * 16;
    b += (size_t)SB_CHUNK * (Mp * 2 + (((Mp * 2) % 128 == 0) ? 32 : 0));
  }
  return b;
}

}  // namespace

static std::atomic<int> g_small_parts{0};     // diagnostic hook: workgroups per problem, 0 = heuristic
extern "C" void vitmi_debug_gemm_small_parts(int n) { g_small_parts = n; }

// which batched bf16 problems take this kernel (plain store epilogue only)
int gemm_small_form(const GemmArgs& g, int in_bf16) {
  if (!in_bf16 || g.batch <= 1 || g.e.mode != VITMI_EPI_STORE || !g.e.c_bf16 || g.e.bias || g.e.accumulate) return -1;
  if (g.lda % 8 || g.ldb % 8 || !is_aligned(g.A, 16) || !is_aligned(g.B, 16)) return -1;
  for (int i = 0; i < 2; ++i)
    if (g.a_bs[i] % 8 || g.b_bs[i] % 8) return -1;
  if (g.a_km && g.b_km) return (g.K % 8 == 0 && g.K <= 64) ? 0 : -1;
  if (g.N % 8 || g.N > SB_MAXN || g.K > SB_MAXK) return -1;
  // S is read in 16-B pieces along its rows: the row pitch must cover the last (partly
  // padded) piece; what lies beyond K (form 1) / M (form 2) is masked, never used
  if (g.a_km && !g.b_km) return g.lda >= (g.K + 7) / 8 * 8 ? 1 : -1;
  if (!g.a_km && !g.b_km) return (g.lda >= (g.M + 7) / 8 * 8 && g.M <= 256) ? 2 : -1;
  return -1;
}

int gemm_small_launch(const GemmArgs& g, int form, hipStream_t stream) {
  SmallArgs a;
  a.A = reinterpret_cast<const bf16*>(g.A); a.B = reinterpret_cast<const bf16*>(g.B);
  a.C = reinterpret_cast<bf16*>(g.e.C);
  a.M = (int)g.M; a.N = (int)g.N; a.K = (int)g.K;
  a.lda = g.lda; a.ldb = g.ldb; a.ldc = g.e.ldc;
  a.batch_inner = g.batch_inner;
  for (int i = 0; i < 2; ++i) { a.a_bs[i] = g.a_bs[i]; a.b_bs[i] = g.b_bs[i]; a.c_bs[i] = g.c_bs[i]; }
  a.alpha = g.e.alpha;
  const size_t lds = small_lds(form, a.M, a.N, a.K);
  // fewer than four workgroups per CU: split the row tiles of a problem over two workgroups
  const int cus = vitmi_cu_count();
  const int tiles_m = (a.M + 15) / 16;
  // (measured at 512 problems of 196 x 196 x 48: form 0 22.3 -> 20.3 us, form 1 22.1 -> 16.7;
  // form 2 streams S through LDS once per workgroup, so a second one doubles that: 24.5 -> 35.9)
  const unsigned parts = (g_small_parts > 0) ? (unsigned)g_small_parts
                         : ((form != 2 && g.batch < 4 * (int64_t)cus && tiles_m >= 8) ? 2u : 1u);
  const dim3 grid((unsigned)g.batch, parts);
#define SMALL_GO(F)                                                                                   \
  do {                                                                                                \
    auto kern = gemm_small_kernel<F>;                                                                 \
    if (int rc_ = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(kern), 96 * 1024, "gemm_small")) return rc_; \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);                                        \
  } while (0)
  if (form == 0) SMALL_GO(0);
  else if (form == 1) SMALL_GO(1);
  else SMALL_GO(2);
#undef SMALL_GO
  return vitmi_check_launch("gemm_small_kernel");
}

// every diagnostic switch of this file back to its default (vitmi_debug_reset, core.cpp)
void vitmi_debug_reset_gemm_small() {
  g_small_parts = 0;
}
