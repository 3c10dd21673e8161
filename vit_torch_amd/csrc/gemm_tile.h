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

// The same LDS-DMA issued from inline asm: hipcc does not know of it.  Used for the NEXT tile's prologue stages, which are
// issued in front of an epilogue: with the builtin form outstanding, hipcc answers the first use of the epilogue's ordinary
// side-input loads (residual rows, saved gelu') with `s_waitcnt vmcnt(0)` — the whole prologue plus every side load issued
// so far — instead of the counted wait it emits when it only sees loads and stores.  The prologue is waited for by the
// counted `wait_vm_c` of the next tile's first slab, never by the compiler.  M0 (the DMA's LDS base) is compiler-reserved:
// saved, written and restored inside the one statement (guide §5.7).
__device__ __forceinline__ void glds16_hidden(const void* gsrc, char* lds_dst) {
  unsigned keep;
  const unsigned l = (unsigned)(uintptr_t)LDS_PTR(char, lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(l) : "memory");
}
// ... and in the SGPR-base + 32-bit-VGPR-offset form: the staging plans keep a wave-uniform base pointer and a per-lane
// 32-bit offset for exactly this, but through the builtin hipcc adds them into a 64-bit VGPR address for every DMA
// (a v_lshl_add_u64 per instruction in the R phase of every slab)
__device__ __forceinline__ void glds16_sbase(const char* sbase, uint32_t voff, char* lds_dst) {
  unsigned keep;
  const unsigned l = (unsigned)(uintptr_t)LDS_PTR(char, lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(l) : "memory");
}
// ... with M0 left as written: only for kernels in which EVERY LDS-DMA is one of these statements (the phased GEMM loops:
// hipcc then has no use of M0 of its own to protect; two scalar moves less per DMA in the R phase)
__device__ __forceinline__ void glds16_sbase_m0(const char* sbase, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
template <bool HID> __device__ __forceinline__ void glds16x(const void* gsrc, char* lds_dst) {
  if constexpr (HID) glds16_hidden(gsrc, lds_dst);
  else glds16(gsrc, lds_dst);
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
  // the same read at LDS byte address `addr` + IMM, IMM a compile-time constant that goes into the instruction's 16-bit
  // offset field (the stage of an unrolled ring iteration: no per-fragment address add)
  template <int IMM> __device__ __forceinline__ void load_imm(uint32_t addr) {
    static_assert(IMM >= 0 && IMM < 65536, "DS offset field");
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;
    v = *reinterpret_cast<const lds_bf16x8*>(reinterpret_cast<const lds_char*>((uintptr_t)addr) + IMM);
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
  template <int IMM> __device__ __forceinline__ void load_imm(uint32_t addr) {
    static_assert(IMM >= 0 && IMM + 2048 < 65536, "DS offset field");
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(lo), "=&v"(hi) : "v"(addr), "n"(IMM), "n"(IMM + 2048) : "memory");
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


// ---- fast GELU for the bf16 epilogues, two elements at a time so that the polynomial
// runs on v_pk_fma_f32 / v_pk_mul_f32 (the GELU / gelu' epilogues are VALU-issue bound).
// erfc by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below bf16 resolution):
//   h(x) = Phi(-|x|) = 0.5 erfc(|x| / sqrt 2) = e * t (b1 + t (b2 + ... t b5)),
//   t = 1 / (1 + 0.3275911 |x| / sqrt 2),  e = exp(-x^2 / 2) = exp2(x^2 * -0.5 log2 e),
// with b_i = a_i / 2.  ONE exp serves the tail and the Gaussian pdf, so gelu' costs no
// second transcendental.  Then
//   gelu(x)  = relu(x) - |x| h              (x >= 0: x (1 - h);  x < 0: x h)
//   gelu'(x) = 0.5 + copysign(0.5 - h, x) + x e / sqrt(2 pi).
// The fp32 parity mode (generic kernel) keeps erff.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 fma2(f32x2 a, float b, float c) { return __builtin_elementwise_fma(a, f32x2{b, b}, f32x2{c, c}); }
__device__ __forceinline__ void gelu_tail2(f32x2 x, f32x2* h, f32x2* e) {
  const f32x2 arg = (x * x) * -0.72134752044f;
  f32x2 ee, t;
  ee[0] = __builtin_amdgcn_exp2f(arg[0]);
  ee[1] = __builtin_amdgcn_exp2f(arg[1]);
  t[0] = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x[0]), 0.2316419f, 1.f));
  t[1] = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x[1]), 0.2316419f, 1.f));
  // explicit fmas: left to -ffp-contract the same source contracted differently in different epilogue variants
  // (results one fp32 ulp apart, an occasional bf16 rounding flipped between two builds of the same formula)
  f32x2 p = fma2(t, 0.5307027145f, -0.7265760135f);
  p = fma2(p, t, f32x2{0.7107068705f, 0.7107068705f});
  p = fma2(p, t, f32x2{-0.142248368f, -0.142248368f});
  p = fma2(p, t, f32x2{0.127414796f, 0.127414796f});
  *h = (p * t) * ee;
  *e = ee;
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
  f32x2 h, e, r;
  gelu_tail2(x, &h, &e);
  r[0] = __builtin_fmaf(-__builtin_fabsf(x[0]), h[0], __builtin_fmaxf(x[0], 0.f));
  r[1] = __builtin_fmaf(-__builtin_fabsf(x[1]), h[1], __builtin_fmaxf(x[1], 0.f));
  return r;
}
// gelu and gelu' of the same argument from one tail evaluation (aux_is_derivative forward)
__device__ __forceinline__ void gelu_both2(f32x2 x, f32x2* gl, f32x2* dg) {
  f32x2 h, e, r;
  gelu_tail2(x, &h, &e);
  r[0] = __builtin_fmaf(-__builtin_fabsf(x[0]), h[0], __builtin_fmaxf(x[0], 0.f));
  r[1] = __builtin_fmaf(-__builtin_fabsf(x[1]), h[1], __builtin_fmaxf(x[1], 0.f));
  f32x2 d = 0.5f - h;
  d[0] = __builtin_copysignf(d[0], x[0]);
  d[1] = __builtin_copysignf(d[1], x[1]);
  *gl = r;
  *dg = fma2(x * e, f32x2{0.39894228040143267794f, 0.39894228040143267794f}, d + 0.5f);
}
__device__ __forceinline__ f32x2 dgelu2(f32x2 x) {
  f32x2 h, e;
  gelu_tail2(x, &h, &e);
  f32x2 d = 0.5f - h;
  d[0] = __builtin_copysignf(d[0], x[0]);
  d[1] = __builtin_copysignf(d[1], x[1]);
  return fma2(x * e, f32x2{0.39894228040143267794f, 0.39894228040143267794f}, d + 0.5f);
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

// Side inputs of an epilogue (residual rows, saved GELU derivative) are read exactly once: the
// non-temporal form keeps them from taking L2 lines the operand panels would re-use.
template <typename T, int W>
__device__ __forceinline__ void loadv_nt(const T* p, float (&o)[W]) {
  if constexpr (sizeof(T) == 2 && W == 8) {
    const bf16x8 v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  } else if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
      const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4 * q));
#pragma unroll
      for (int i = 0; i < 4; ++i) o[4 * q + i] = v[i];
    }
  } else {
    loadv<T, W>(p, o);
  }
}

// The same store with a cache policy for output tiles: 1 = `sc1` (the line is written through and
// DROPPED from the XCD's L2), 2 = `nt`.  A GEMM's C / C2 tiles are never read again by the launch
// that writes them, and left in L2 they evict the operand panels the next tiles of the XCD share.
template <typename T, int W>
__device__ __forceinline__ void storev_pol(T* p, const float (&v)[W], int policy) {
  if (policy == 0) { storev<T, W>(p, v); return; }
  if constexpr (sizeof(T) == 2 && W == 8) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
    // `nt` through the builtin, NOT inline asm (round 3): an asm store is absent from hipcc's vmcnt bookkeeping, so the
    // counted wait it emits for the NEXT strip's side-input loads (`vmcnt(2)` = "my two younger loads may fly") also
    // drained the asm stores issued in between — every strip of the gelu'-multiply epilogue waited for the previous
    // strip's stores to be acknowledged by memory.  (`sc1`, a diagnostic policy, has no builtin and stays asm.)
    if (policy == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(o) : "memory");
    else __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(p));
  } else if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
      const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      T* pq = p + 4 * q;
      if (policy == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(pq), "v"(o) : "memory");
      else __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(pq));
    }
  } else {
    storev<T, W>(p, v);
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
// The row index comes in two parts, m = mu + ml: `mu` wave-uniform (tile / strip / row
// group), `ml` the lane's row within the group.  mu * ld is then scalar arithmetic and
// ml * ld + n is the same for every row a lane handles, so a row costs one 64-bit add
// instead of a 64-bit vector multiply per array (those were 13 % of the GELU epilogue).
__device__ __forceinline__ int64_t row_off(int64_t mu, int64_t ml, int64_t ld, int64_t n) {
  return mu * ld + (ml * ld + n);
}
template <int MODE, typename TC, int W>
__device__ __forceinline__ void epi_side(const EpiArgs& e, int64_t mu, int64_t ml, int64_t n, float (&x)[W]) {
  if constexpr (MODE == VITMI_EPI_RESIDUAL) {
    if (e.side_nt) loadv_nt<TC, W>(reinterpret_cast<const TC*>(e.R) + row_off(mu, ml, e.ldr, n), x);
    else loadv<TC, W>(reinterpret_cast<const TC*>(e.R) + row_off(mu, ml, e.ldr, n), x);
  } else if constexpr (MODE == VITMI_EPI_DGELU) {
    if (e.side_nt) loadv_nt<bf16, W>(reinterpret_cast<const bf16*>(e.AUX) + row_off(mu, ml, e.ldaux, n), x);
    else loadv<bf16, W>(reinterpret_cast<const bf16*>(e.AUX) + row_off(mu, ml, e.ldaux, n), x);
  }
  else if constexpr (MODE == VITMI_EPI_STORE && sizeof(TC) == 4) loadv<float, W>(reinterpret_cast<const float*>(e.C) + row_off(mu, ml, e.ldc, n), x);
}
template <int MODE, typename TC, int W>
__device__ __forceinline__ void epi_side(const EpiArgs& e, int64_t m, int64_t n, float (&x)[W]) {
  epi_side<MODE, TC, W>(e, 0, m, n, x);
}

template <int MODE, typename TC, int W>
__device__ __forceinline__ void epi_row(const EpiArgs& e, int64_t mu, int64_t ml, int64_t n, float (&v)[W],
                                        const float (&b)[W], const float (&gm)[W], const float (&x)[W]) {
  TC* C = reinterpret_cast<TC*>(e.C);
  const int64_t m = mu + ml;
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
    if (e.aux_deriv) {                       // wave-uniform: C2 = gelu'(acc + bias)
#pragma unroll
      for (int i = 0; i < W; i += 2) {
        f32x2 r, d;
        gelu_both2(f32x2{v[i] + b[i], v[i + 1] + b[i + 1]}, &r, &d);
        pre[i] = d[0]; pre[i + 1] = d[1];
        v[i] = r[0]; v[i + 1] = r[1];
      }
    } else {
#pragma unroll
      for (int i = 0; i < W; i += 2) {
        f32x2 p = {v[i] + b[i], v[i + 1] + b[i + 1]};
        if constexpr (sizeof(TC) == 2) { p[0] = (float)(bf16)p[0]; p[1] = (float)(bf16)p[1]; }
        const f32x2 r = gelu2(p);
        pre[i] = p[0]; pre[i + 1] = p[1];
        v[i] = r[0]; v[i + 1] = r[1];
      }
    }
    if (e.C2) storev_pol<TC, W>(reinterpret_cast<TC*>(e.C2) + row_off(mu, ml, e.ldc2, n), pre, e.c_policy);
  } else if constexpr (MODE == VITMI_EPI_RESIDUAL) {
    if (e.C2) {                              // un-scaled branch output, operand dtype (bf16 here)
      float f[W];
#pragma unroll
      for (int i = 0; i < W; ++i) f[i] = v[i] + b[i];
      storev_pol<bf16, W>(reinterpret_cast<bf16*>(e.C2) + row_off(mu, ml, e.ldc2, n), f, e.c_policy);
    }
    float rs = 1.f;
    if (e.rowscale) rs = e.rowscale[(uint32_t)m / (uint32_t)e.rpg];      // uniform branch; M < 2^32
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = x[i] + rs * gm[i] * (v[i] + b[i]);
  } else if constexpr (MODE == VITMI_EPI_DGELU) {
    if (e.aux_deriv) {                       // wave-uniform: AUX is gelu'(pre) already
#pragma unroll
      for (int i = 0; i < W; ++i) v[i] *= x[i];
    } else {
#pragma unroll
      for (int i = 0; i < W; i += 2) {
        const f32x2 d = dgelu2(f32x2{x[i], x[i + 1]});
        v[i] *= d[0]; v[i + 1] *= d[1];
      }
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
  storev_pol<TC, W>(C + row_off(mu, ml, e.ldc, n), v, e.c_policy);
}
template <int MODE, typename TC, int W>
__device__ __forceinline__ void epi_row(const EpiArgs& e, int64_t m, int64_t n, float (&v)[W],
                                        const float (&b)[W], const float (&gm)[W], const float (&x)[W]) {
  epi_row<MODE, TC, W>(e, 0, m, n, v, b, gm, x);
}

constexpr int TRS = 68;                       // floats per row of the transpose strip (64 + pad)
constexpr int TR_BYTES = 16 * TRS * 4;        // one wave's 16-row strip
