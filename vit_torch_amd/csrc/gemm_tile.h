// Pieces shared by the fast GEMM kernels (gemm_fast.hip: 256x256 tiles, one workgroup per
// CU; gemm_fast2.hip: 256x128 tiles, two workgroups per CU): LDS-DMA, in-flight
// fragments, counted waits, and the row-wise epilogue.
#pragma once
#include <cstdlib>
#include "epilogue.h"

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}


// A fragment in flight: for the k-major image one ds_read_b128 the compiler tracks; for
// the k-row image two ds_read_b64_tr_b16 issued from INLINE ASM.  The builtin form makes
// hipcc put `s_waitcnt vmcnt(0)` in front of every transposed read while an LDS-DMA is
// outstanding (it cannot prove the read does not alias the DMA's LDS destination), which
// drains the ring every slab; asm reads are invisible to that pass.  Their results are
// only touched after frag_wait() names them (guide §5.7 item 1, form (ii)).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool KM> struct Frag;
template <> struct Frag<true> {
  bf16x8 v;
  __device__ __forceinline__ void load(const char* slab, uint32_t off) {
    v = *reinterpret_cast<const bf16x8*>(slab + off);
  }
  __device__ __forceinline__ bf16x8 get() const { return v; }
};
template <> struct Frag<false> {
  u32x2 lo, hi;
  __device__ __forceinline__ void load(const char* slab, uint32_t off) {
    const uint32_t a = (uint32_t)(uintptr_t)LDS_PTR(char, slab) + off;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048"
                 : "=&v"(lo), "=&v"(hi) : "v"(a) : "memory");
  }
  __device__ __forceinline__ bf16x8 get() const {
    const u32x4 r = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, r);
  }
};
__device__ __forceinline__ void frag_wait4(Frag<true>&, Frag<true>&, Frag<true>&, Frag<true>&) {}
__device__ __forceinline__ void frag_wait4(Frag<false>& a, Frag<false>& b, Frag<false>& c, Frag<false>& d) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi)
               :: "memory");
}

// all but the `n` youngest of this wave's vector-memory ops (LDS-DMA included) are done
__device__ __forceinline__ void wait_vm(int n) {
  if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
__device__ __forceinline__ void raw_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}


// ---- fast GELU for the bf16 epilogues: erf by Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7, far below bf16 resolution); ONE exp serves both the erf
// tail and the Gaussian pdf, so gelu' costs no second transcendental.  The fp32
// parity mode (generic kernel) keeps erff.
__device__ __forceinline__ void gelu_parts(float x, float* cdf, float* pdf) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
  const float ex = __expf(-ax * ax);                      // exp(-x^2/2)
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f +
                     t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.f - poly * ex;
  *cdf = 0.5f * (1.f + copysignf(erf_abs, x));
  *pdf = 0.39894228040143267794f * ex;
}

// ---- W-wide row vectors (W = 8 for bf16 outputs = 16 B, W = 4 for fp32 = 16 B)
template <typename T, int W>
__device__ __forceinline__ void loadv(const T* p, float (&o)[W]) {
  if constexpr (sizeof(T) == 2 && W == 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  } else {
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
      const f32x4 v = load4<T>(p + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) o[4 * q + i] = v[i];
    }
  }
}
template <typename T, int W>
__device__ __forceinline__ void storev(T* p, const float (&v)[W]) {
  if constexpr (sizeof(T) == 2 && W == 8) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = o;
  } else {
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
      const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      store4<T>(p + 4 * q, o);
    }
  }
}

// epilogue on W consecutive columns n.. of row m; v = raw accumulators in, stored out
// b / gm: bias and LayerScale of the lane's W columns (the same for every row, so the
// caller loads them ONCE: a load inside the row loop would put a vmcnt wait, which also
// counts the previous rows' stores, in front of every store)
// does this epilogue read a per-element side input (residual / saved pre-activation / C)?
template <int MODE, typename TC>
__device__ __forceinline__ bool epi_has_side(const EpiArgs& e) {
  if constexpr (MODE == VITMI_EPI_RESIDUAL || MODE == VITMI_EPI_DGELU) return true;
  else if constexpr (MODE == VITMI_EPI_STORE && sizeof(TC) == 4) return e.accumulate != 0;
  else return false;
}
// load it for W columns of row m: issued a whole strip ahead of its use, so the HBM
// latency of the 128-512 KiB a tile reads here is not paid row by row
template <int MODE, typename TC, int W>
__device__ __forceinline__ void epi_side(const EpiArgs& e, int64_t m, int64_t n, float (&x)[W]) {
  if constexpr (MODE == VITMI_EPI_RESIDUAL) loadv<TC, W>(reinterpret_cast<const TC*>(e.R) + m * e.ldr + n, x);
  else if constexpr (MODE == VITMI_EPI_DGELU) loadv<bf16, W>(reinterpret_cast<const bf16*>(e.AUX) + m * e.ldaux + n, x);
  else if constexpr (MODE == VITMI_EPI_STORE && sizeof(TC) == 4) loadv<float, W>(reinterpret_cast<const float*>(e.C) + m * e.ldc + n, x);
}

template <int MODE, typename TC, int W>
__device__ __forceinline__ void epi_row(const EpiArgs& e, int64_t m, int64_t n, float (&v)[W],
                                        const float (&b)[W], const float (&gm)[W], const float (&x)[W]) {
  TC* C = reinterpret_cast<TC*>(e.C);
  if constexpr (MODE == VITMI_EPI_STORE) {
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = v[i] * e.alpha + b[i];
    if constexpr (sizeof(TC) == 4) {
      if (e.accumulate) {
#pragma unroll
        for (int i = 0; i < W; ++i) v[i] += x[i];
      }
    }
  } else if constexpr (MODE == VITMI_EPI_BIAS_GELU) {
    float pre[W];
#pragma unroll
    for (int i = 0; i < W; ++i) {
      pre[i] = v[i] + b[i];
      if constexpr (sizeof(TC) == 2) pre[i] = (float)(bf16)pre[i];
      float cdf, pdf;
      gelu_parts(pre[i], &cdf, &pdf);
      v[i] = pre[i] * cdf;
    }
    if (e.C2) storev<TC, W>(reinterpret_cast<TC*>(e.C2) + m * e.ldc2 + n, pre);
  } else if constexpr (MODE == VITMI_EPI_RESIDUAL) {
    if (e.C2) {                              // un-scaled branch output, operand dtype (bf16 here)
      float f[W];
#pragma unroll
      for (int i = 0; i < W; ++i) f[i] = v[i] + b[i];
      storev<bf16, W>(reinterpret_cast<bf16*>(e.C2) + m * e.ldc2 + n, f);
    }
    float rs = 1.f;
    if (e.rowscale) rs = e.rowscale[(uint32_t)m / (uint32_t)e.rpg];      // uniform branch; M < 2^32
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = x[i] + rs * gm[i] * (v[i] + b[i]);
  } else if constexpr (MODE == VITMI_EPI_DGELU) {
#pragma unroll
    for (int i = 0; i < W; ++i) {
      float cdf, pdf;
      gelu_parts(x[i], &cdf, &pdf);
      v[i] *= cdf + x[i] * pdf;
    }
  } else {  // PATCH_POS
    const int64_t t = m % e.n_tok;
    float ps[W];
    loadv<float, W>(e.pos + t * e.ldpos + n, ps);
    if (t == 0 && e.cls) {
      float c[W];
      loadv<float, W>(e.cls + n, c);
#pragma unroll
      for (int i = 0; i < W; ++i) v[i] = c[i] + ps[i];
    } else {
#pragma unroll
      for (int i = 0; i < W; ++i) v[i] = v[i] + b[i] + ps[i];
    }
  }
  storev<TC, W>(C + m * e.ldc + n, v);
}

constexpr int TRS = 68;                       // floats per row of the transpose strip (64 + pad)
constexpr int TR_BYTES = 16 * TRS * 4;        // one wave's 16-row strip
