// GEMM epilogues shared by the generic and the fast kernels (include/vitmi.h
// VITMI_EPI_*).  All arithmetic in fp32; one rounding at the store.
#pragma once
#include "common.h"

struct EpiArgs {
  int mode;
  void* C; int64_t ldc; int c_bf16;
  void* C2; int64_t ldc2; int c2_bf16;
  const float* bias;
  const void* R; int64_t ldr; int r_bf16;
  const float* gamma;
  const void* AUX; int64_t ldaux; int aux_bf16;
  const float* pos; int64_t n_tok; int64_t ldpos; const float* cls;
  float alpha;
  int accumulate;
  const float* rowscale; int64_t rpg;      // RESIDUAL: per-row-group branch scale (DropPath)
  float* colsum_part;                      // DGELU, fast path: [M/128][N] column sums of C
  int aux_deriv;                           // GELU: C2 = gelu'(pre) instead of pre; DGELU: C = acc * AUX
  int side_nt;                             // tile kernels: non-temporal loads of R / AUX
  int c_policy;                            // tile kernels: cache policy of the C / C2 stores (0 plain, 1 sc1, 2 nt)
};

struct GemmArgs {
  int64_t M, N, K;
  const void* A; int64_t lda; int a_km;
  const void* B; int64_t ldb; int b_km;
  void* ws; size_t ws_bytes;      // optional scratch (split-K partial tiles)
  unsigned long long* dbg;        // diagnostic phase stamps (NULL in production)
  int dbg_blocks;                 // workgroups whose timeline is stamped
  int64_t batch, batch_inner, a_bs[2], b_bs[2], c_bs[2];   // batched form (generic kernel)
  int vec_a, vec_b;               // operand rows are 16-B aligned: vector staging allowed
  int rfold;                      // gemm_fast: stream the fp32 residual through LDS during the main loop
  int band;                       // gemm_fast: column-band width of the tile order (0 = row-major)
  int launch_flags;               // VITMI_LAUNCH_*
  unsigned* zero_cnt; int zero_n;   // gemm_fast full-rounds launch: counters of the tail launch that follows, zeroed by workgroup 0
  unsigned* fix_cnt;              // gemm_fast tail slices: arrival counters (one per tail tile); the last slice to arrive applies the epilogue
  int strict_wait;                // gemm_fast debug: a prefetched tile waits vmcnt(0) instead of the counted wait (vitmi_debug_gemm_strict_wait)
  // gemm_fast split-K, PAIRED launch (vitmi_gemm_pair): a second product with the same K / layouts shares the grid;
  // tiles [0, tiles1) belong to this problem, [tiles1, tiles1 + tiles2) to the second one (tiles1 == 0: no pair)
  const void* A2; const void* B2; int64_t lda2, ldb2, M2, N2; int tiles1, tiles_n2; float* ws2;
  int stag_cycles, stag_phases;   // gemm_fast, persistent walk: start delay of the workgroups with the shorter tile list
  int dbg_alias;                  // gemm_fast diagnostic: bit 0 = every tile stages A panel 0, bit 1 = B panel 0 (operand streams out of L2: tools/r04_gemm_probe.py)
  int side_depth;                 // gemm_fast, bf16 epilogues with a side input: strips of it in flight (1 = round 2's one ahead, 3)
  EpiArgs e;
};

__device__ __forceinline__ float ld_any(const void* p, int64_t i, int is_bf16) {
  return is_bf16 ? (float)reinterpret_cast<const bf16*>(p)[i]
                 : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st_any(void* p, int64_t i, int is_bf16, float v) {
  if (is_bf16) reinterpret_cast<bf16*>(p)[i] = (bf16)v;
  else reinterpret_cast<float*>(p)[i] = v;
}

// value-level epilogue: returns the value to store in C; *v2 = value for C2
template <int MODE>
__device__ __forceinline__ float epi_value(const EpiArgs& e, int64_t m, int64_t n, float acc,
                                           float* v2) {
  if constexpr (MODE == VITMI_EPI_STORE) {
    float v = acc * e.alpha;
    if (e.bias) v += e.bias[n];
    if (e.accumulate) v += reinterpret_cast<const float*>(e.C)[m * e.ldc + n];
    return v;
  } else if constexpr (MODE == VITMI_EPI_BIAS_GELU) {
    float pre = acc + (e.bias ? e.bias[n] : 0.f);
    // the backward pass differentiates at the STORED pre-activation, so
    // round it first when C2 is bf16 (keeps fwd and bwd consistent)
    if (e.aux_deriv) {
      *v2 = dgelu_erf(pre);
      return gelu_erf(pre);
    }
    if (e.c_bf16) pre = (float)(bf16)pre;
    *v2 = pre;
    return gelu_erf(pre);
  } else if constexpr (MODE == VITMI_EPI_RESIDUAL) {
    float v = acc + (e.bias ? e.bias[n] : 0.f);
    *v2 = v;                               // branch output before LayerScale (for d gamma)
    if (e.gamma) v *= e.gamma[n];
    if (e.rowscale) v *= e.rowscale[m / e.rpg];
    return ld_any(e.R, m * e.ldr + n, e.r_bf16) + v;
  } else if constexpr (MODE == VITMI_EPI_DGELU) {
    const float aux = ld_any(e.AUX, m * e.ldaux + n, e.aux_bf16);
    return acc * (e.aux_deriv ? aux : dgelu_erf(aux));
  } else {  // VITMI_EPI_PATCH_POS
    const int64_t t = m % e.n_tok;
    if (t == 0 && e.cls) return e.cls[n] + e.pos[n];
    return acc + (e.bias ? e.bias[n] : 0.f) + e.pos[t * e.ldpos + n];
  }
}

// scalar store of one output element (generic kernel)
template <int MODE>
__device__ __forceinline__ void epi_store1(const EpiArgs& e, int64_t m, int64_t n, float acc) {
  float v2 = 0.f;
  const float v = epi_value<MODE>(e, m, n, acc, &v2);
  st_any(e.C, m * e.ldc + n, e.c_bf16, v);
  if constexpr (MODE == VITMI_EPI_BIAS_GELU || MODE == VITMI_EPI_RESIDUAL) {
    if (e.C2) st_any(e.C2, m * e.ldc2 + n, e.c2_bf16, v2);
  }
}

__device__ __forceinline__ void epi_store1_rt(const EpiArgs& e, int64_t m, int64_t n, float acc) {
  switch (e.mode) {
    case VITMI_EPI_STORE: epi_store1<VITMI_EPI_STORE>(e, m, n, acc); break;
    case VITMI_EPI_BIAS_GELU: epi_store1<VITMI_EPI_BIAS_GELU>(e, m, n, acc); break;
    case VITMI_EPI_RESIDUAL: epi_store1<VITMI_EPI_RESIDUAL>(e, m, n, acc); break;
    case VITMI_EPI_DGELU: epi_store1<VITMI_EPI_DGELU>(e, m, n, acc); break;
    default: epi_store1<VITMI_EPI_PATCH_POS>(e, m, n, acc); break;
  }
}
