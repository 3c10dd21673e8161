// bf16 GEMM, 256x128 output tiles, FOUR waves (2 x 2, each 128x64) per workgroup and
// TWO workgroups per CU.
//
// Why: the GEMMs of a ViT-B step have K = 768..3072, so a 256x256 tile spends 12-48
// k-steps in its main loop and then a comparable time in an epilogue that streams
// 128-512 KiB to/from HBM (bias+GELU with two outputs, fp32 residual read-modify-write,
// gelu' of a saved pre-activation).  With one 8-wave workgroup per CU (gemm_fast.hip) the
// matrix pipes idle during that epilogue, and all CUs reach it at about the same time.
// Two independent 4-wave workgroups per CU drift apart (different tiles), so one's
// epilogue, LDS-DMA issue and fragment reads run under the other's MFMAs: each SIMD hosts
// one wave of each workgroup.  Price: a 256x128 tile moves 1.5x the operand bytes per
// FLOP of a 256x256 tile through L2 (24 KiB per 32-deep slab, 6 LDS-DMA per wave).
//
// Main loop: ring of 3 slabs (72 KiB; 2 workgroups = 144 of the CU's 160 KiB); per slab
//   wait (counted vmcnt) for my LDS-DMA of slab j, barrier      -> slab j visible, and
//                                                                  slab j-1 fully read
//   issue LDS-DMA of slab j+2 into the stage slab j-1 occupied
//   12 fragment reads of slab j, 32 MFMAs
// LDS images, swizzles, epilogue: as gemm_fast.hip (gemm_tile.h).
#include <atomic>
#include <type_traits>
#include "gemm_tile.h"

namespace {

constexpr int BM2 = 256, BN2 = 128, BK2 = 64;    // split-K plans count 64-deep steps; K % 32 == 0
// Ragged shapes (M, N not multiples of the tile; Swin's C = 96 / 192 stages, odd batch
// sizes): operand rows / column chunks beyond the matrix are CLAMPED to the last valid
// one when the per-lane source offsets are built (once, outside the main loop), so the
// LDS-DMA never leaves the buffers and the surplus rows/columns of the tile hold
// duplicates; the epilogue masks their stores.  An output column depends only on its own
// B column and an output row only on its own A row, so duplicates cannot leak.
constexpr int A_SLAB = 256 * 32 * 2;             // 16 KiB
constexpr int B_SLAB = 128 * 32 * 2;             //  8 KiB
constexpr int STAGE2 = A_SLAB + B_SLAB;          // 24 KiB
constexpr int RING2 = 3;
constexpr int LDS2 = RING2 * STAGE2;             // 72 KiB
constexpr int NT2 = 256;

// staging plan of one operand slab [ROWS (m or n)] x [32 k], NP LDS-DMA per wave
template <bool KM, int ROWS> struct Plan2 {
  static constexpr int NP = ROWS / 64;           // pieces (1 KiB) per wave: 4 (A) or 2 (B)
  const char* base[NP];
  uint32_t off[NP];
  int64_t step;
  // R = rows (k-major) / columns (k-minor) the matrix really has along the tile's dimension
  __device__ __forceinline__ void init(const bf16* X, int64_t ld, int64_t r0, int64_t k0, int64_t R, int wave, int lane) {
    if constexpr (KM) {    // subtiles of 16 rows x 32 k (64-B rows), XOR byte bit5 ^= bit9
      const int pb = 16 * lane;
      const int lb = pb ^ (((pb >> 9) & 1) << 5);
      const int row = lb >> 6, ch = (lb & 63) >> 4;
      const int last = (int)(R - 1 - r0);          // >= 0: the tile starts inside the matrix
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int st = wave * NP + i;
        const int rel = min(st * 16 + row, last);
        base[i] = reinterpret_cast<const char*>(X + r0 * ld + k0);
        off[i] = (uint32_t)((rel * ld + ch * 8) * 2);
      }
      step = 64;
    } else {               // 32 k-rows of ROWS*2 bytes; one piece = 1024/(ROWS*2) k-rows
      constexpr int RB = ROWS * 2;               // bytes per k-row: 512 or 256
      constexpr int RPP = 1024 / RB;             // k-rows per piece: 2 or 4
      constexpr int LPR = RB / 16;               // lanes per k-row: 32 or 16
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int pc = wave * NP + i;
        const int row = pc * RPP + lane / LPR;
        const int pc16 = lane % LPR;
        const int key = (row & 3) | (((row >> 3) & 1) << 2);
        const int c32 = (pc16 >> 1) ^ key;
        const int col = min(c32 * 16 + (pc16 & 1) * 8, (int)(R - 8 - r0));   // R % 8 == 0
        base[i] = reinterpret_cast<const char*>(X + (k0 + pc * RPP) * ld + r0);
        off[i] = (uint32_t)(((lane / LPR) * ld + col) * 2);
      }
      step = 64 * ld;
    }
  }
  // SGPR base + 32-bit lane offset, from inline asm (round 3, as in gemm_fast.hip): no 64-bit vector add per DMA, and the
  // DMAs stay out of hipcc's vmcnt bookkeeping — every wait of the main loop is an explicit counted one
  __device__ __forceinline__ void issue(char* slab, int j, int wave) const {
#pragma unroll
    for (int i = 0; i < NP; ++i)
      glds16_sbase(base[i] + (int64_t)j * step, off[i], slab + (wave * NP + i) * 1024);
  }
};

template <bool KM, int ROWS>
__device__ __forceinline__ uint32_t frag_off2(int rb, int lane) {
  if constexpr (KM) {
    int pb = (lane & 15) * 64 + (lane >> 4) * 16;
    pb ^= ((pb >> 9) & 1) << 5;
    return (uint32_t)(rb * 1024 + pb);
  } else {
    const int g = lane >> 4, i = lane & 15;
    const int row = 8 * g + (i >> 2);
    const int key = (row & 3) | (((row >> 3) & 1) << 2);
    return (uint32_t)(row * (ROWS * 2) + ((rb ^ key) * 32) + 8 * (i & 3));
  }
}

// in-flight fragment whose transposed reads use a k-row pitch of PITCH bytes
template <bool KM, int PITCH> struct Frag2;
template <int PITCH> struct Frag2<true, PITCH> {
  bf16x8 v;
  __device__ __forceinline__ void load(const char* slab, uint32_t off) { v = *reinterpret_cast<const bf16x8*>(slab + off); }
  __device__ __forceinline__ bf16x8 get() const { return v; }
};
template <int PITCH> struct Frag2<false, PITCH> {
  u32x2 lo, hi;
  __device__ __forceinline__ void load(const char* slab, uint32_t off) {
    const uint32_t a = (uint32_t)(uintptr_t)LDS_PTR(char, slab) + off;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                 : "=&v"(lo), "=&v"(hi) : "v"(a), "i"(4 * PITCH) : "memory");
  }
  __device__ __forceinline__ bf16x8 get() const {
    const u32x4 r = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, r);
  }
};
template <int P>
__device__ __forceinline__ void fwait4(Frag2<true, P>&, Frag2<true, P>&, Frag2<true, P>&, Frag2<true, P>&) {}
template <int P>
__device__ __forceinline__ void fwait4(Frag2<false, P>& a, Frag2<false, P>& b, Frag2<false, P>& c, Frag2<false, P>& d) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi)
               :: "memory");
}

__device__ __forceinline__ void wait_vm6(bool next_in_flight) {
  if (next_in_flight) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// (the DGELU column sums of gemm_fast.hip, for ragged shapes: rows >= M were never added)
template <int MODE, bool SPLITK, int W, int LPR>
__device__ __forceinline__ void colsum_part2(const GemmArgs& g, float (&cs)[W], int64_t m0, int wm, int64_t ncol,
                                             bool col_ok, int rr) {
  if constexpr (MODE == VITMI_EPI_DGELU && !SPLITK) {
    if (g.e.colsum_part) {
#pragma unroll
      for (int i = 0; i < W; ++i) {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) cs[i] += __shfl_xor(cs[i], off, 64);
      }
      if (rr == 0 && col_ok && m0 + wm * 128 < g.M) storev<float, W>(g.e.colsum_part + ((m0 >> 7) + wm) * g.N + ncol, cs);
    }
  }
}

template <bool A_KM, bool B_KM, int MODE, typename TC, bool SPLITK>
__global__ __launch_bounds__(NT2, 2) void gemm_fast2_kernel(GemmArgs g, int tiles_n, int nwg, int ntiles,
                                                            int ksps, float* ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int split = SPLITK ? wg / ntiles : 0;
  const int tile = SPLITK ? wg % ntiles : wg;
  const int64_t m0 = (int64_t)(tile / tiles_n) * BM2;
  const int64_t n0 = (int64_t)(tile % tiles_n) * BN2;

  const bf16* A = reinterpret_cast<const bf16*>(g.A);
  const bf16* B = reinterpret_cast<const bf16*>(g.B);

  f32x4 acc[4][8];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) { acc[ni][mi][0] = 0.f; acc[ni][mi][1] = 0.f; acc[ni][mi][2] = 0.f; acc[ni][mi][3] = 0.f; }

  // k range of this block in 32-deep slabs (ksps counts 64-deep steps)
  const int ns_all = (int)(g.K / 32);
  const int ks0 = SPLITK ? split * ksps * 2 : 0;
  const int nsplit = SPLITK ? nwg / ntiles : 1;    // the last split also takes an odd final slab
  const int ns = !SPLITK ? ns_all : (split == nsplit - 1 ? ns_all - ks0 : ksps * 2);
  const int64_t kb0 = (int64_t)ks0 * 32;

  Plan2<A_KM, 256> pa;
  Plan2<B_KM, 128> pb_;
  pa.init(A, g.lda, m0, kb0, g.M, wave, lane);
  pb_.init(B, g.ldb, n0, kb0, g.N, wave, lane);
  uint32_t fa[8], fb[4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) fa[mi] = frag_off2<A_KM, 256>(wm * 8 + mi, lane);
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) fb[ni] = frag_off2<B_KM, 128>(wn * 4 + ni, lane) + A_SLAB;
  auto issue = [&](int j, int st_i) {
    char* st = smem + st_i * STAGE2;
    pa.issue(st, j, wave);
    pb_.issue(st + A_SLAB, j, wave);
  };
  issue(0, 0);
  if (ns > 1) issue(1, 1);
  int stage = 0;
  // STEADY (j + 2 < ns): slab j + 1 is always in flight behind slab j (vmcnt(6)) and slab j + 2 is always issued — as
  // compile-time facts, so the loop carries no branch around the wait and the issue; the last two slabs take the general form
  auto slab = [&](int j, auto steady_tag) {
    constexpr bool STEADY = decltype(steady_tag)::value;
    if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else wait_vm6(j + 1 < ns);                     // my 6 LDS-DMA of slab j are done (slab j+1's may fly)
    raw_barrier();                                 // slab j visible to all; slab j-1 no longer read
    if (STEADY || j + 2 < ns) issue(j + 2, stage == 0 ? RING2 - 1 : stage - 1);     // into the stage of slab j-1
    const char* As = smem + stage * STAGE2;
    Frag2<B_KM, 256> fbv[4];
    Frag2<A_KM, 512> fav[8];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) fbv[ni].load(As, fb[ni]);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) fav[mi].load(As, fa[mi]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    fwait4(fbv[0], fbv[1], fbv[2], fbv[3]);
    fwait4(fav[0], fav[1], fav[2], fav[3]);
    fwait4(fav[4], fav[5], fav[6], fav[7]);
    bf16x8 bf[4], af[8];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bf[ni] = fbv[ni].get();
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) af[mi] = fav[mi].get();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    stage = stage == RING2 - 1 ? 0 : stage + 1;
  };
  int j = 0;
#pragma unroll 1
  for (; j + 2 < ns; ++j) slab(j, std::true_type{});
#pragma unroll 1
  for (; j < ns; ++j) slab(j, std::false_type{});
  raw_barrier();                                   // every wave is done reading the ring

  // ---- epilogue: wave-private LDS transpose strips, full-line row accesses (gemm_tile.h)
  constexpr int W = sizeof(TC) == 2 ? 8 : 4;
  constexpr int LPR = 64 / W;
  constexpr int RPI = 64 / LPR;
  float* tr = reinterpret_cast<float*>(smem + wave * TR_BYTES);
  const int lr = lane & 15, lg = lane >> 4;
  const int rr = lane / LPR, rc = (lane % LPR) * W;
  float bias_r[W], gamma_r[W];
#pragma unroll
  for (int i = 0; i < W; ++i) { bias_r[i] = 0.f; gamma_r[i] = 1.f; }
  const int64_t ncol = n0 + wn * 64 + rc;
  const bool col_ok = ncol < g.N;                  // N % 8 == 0: a lane's W columns are all in or all out
  if constexpr (!SPLITK) {
    const int64_t nc = col_ok ? ncol : g.N - W;
    if (g.e.bias) loadv<float, W>(g.e.bias + nc, bias_r);
    if (MODE == VITMI_EPI_RESIDUAL && g.e.gamma) loadv<float, W>(g.e.gamma + nc, gamma_r);
  }
  constexpr int NJ = 16 / RPI;                     // row groups per strip
  const bool side = !SPLITK && epi_has_side<MODE, TC>(g.e);
  float cs[W];                                     // DGELU: column sums of this lane's rows
#pragma unroll
  for (int i = 0; i < W; ++i) cs[i] = 0.f;
  float sx[2][NJ][W];                              // side inputs: this strip and the next
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < W; ++i) { sx[0][j][i] = 0.f; sx[1][j][i] = 0.f; }
  // side inputs are loaded UNCONDITIONALLY from clamped rows / columns (a per-lane branch
  // around a load makes hipcc wait for each one); what a masked lane loads is never used
  const int64_t ncol_c = col_ok ? ncol : g.N - W;
  if (side) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int64_t ms = m0 + wm * 128 + j * RPI + rr;
      epi_side<MODE, TC, W>(g.e, ms < g.M ? ms : g.M - 1, ncol_c, sx[0][j]);
    }
  }
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    if (side && mi + 1 < 8) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int64_t ms = m0 + wm * 128 + (mi + 1) * 16 + j * RPI + rr;
        epi_side<MODE, TC, W>(g.e, ms < g.M ? ms : g.M - 1, ncol_c, sx[(mi + 1) & 1][j]);
      }
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      *reinterpret_cast<f32x4*>(tr + lr * TRS + ni * 16 + lg * 4) = acc[ni][mi];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = j * RPI + rr;
      float v[W];
#pragma unroll
      for (int qq = 0; qq < W / 4; ++qq) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(tr + row * TRS + rc + 4 * qq);
        v[4 * qq] = t4[0]; v[4 * qq + 1] = t4[1]; v[4 * qq + 2] = t4[2]; v[4 * qq + 3] = t4[3];
      }
      const int64_t m = m0 + wm * 128 + mi * 16 + row;
      if (col_ok && m < g.M) {                     // surplus rows / columns of a ragged tile
        if constexpr (SPLITK)
          storev<float, W>(ws + ((int64_t)split * g.M + m) * g.N + ncol, v);
        else {
          epi_row<MODE, TC, W>(g.e, m, ncol, v, bias_r, gamma_r, sx[mi & 1][j]);
          if constexpr (MODE == VITMI_EPI_DGELU) {
#pragma unroll
            for (int i = 0; i < W; ++i) cs[i] += v[i];
          }
        }
      }
    }
  }
  colsum_part2<MODE, SPLITK, W, LPR>(g, cs, m0, wm, ncol, col_ok, rr);
}

// C = epilogue(sum_s ws[s]).  A small output with many slices (Swin stage 1: 96 x 384 from
// 256 slices) would leave a one-thread-per-output kernel with 36 workgroups walking 256
// dependent-latency loads each, so the slices of one output are spread over the P waves of a
// workgroup (wave p sums s = p, p + P, ...; lane = output, so every load is a full 1-KiB
// row piece) and wave 0 adds the P partial sums in a fixed order: deterministic for a shape.
__global__ __launch_bounds__(1024) void splitk_reduce2_kernel(const float* __restrict__ ws, int splits, EpiArgs e,
                                                             int64_t M, int64_t N) {
  __shared__ f32x4 part[15 * 64];
  const int lane = threadIdx.x & 63;
  const int p = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), P = blockDim.x >> 6;
  const int64_t n4 = N / 4;
  const int64_t total = M * n4, slab = M * N;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {   // workgroup-uniform
    const int64_t i = base + lane;
    const bool ok = i < total;
    const float* src = ws + (ok ? i : total - 1) * 4;            // clamped: no branch around the loads
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = p; s < splits; s += P) acc += *reinterpret_cast<const f32x4*>(src + (int64_t)s * slab);
    if (P > 1) {
      if (p > 0) part[(p - 1) * 64 + lane] = acc;
      __syncthreads();
    }
    if (p == 0 && ok) {
      for (int q = 1; q < P; ++q) acc += part[(q - 1) * 64 + lane];
      const int64_t m = i / n4, n = (i % n4) * 4;
      float v[4] = {acc[0], acc[1], acc[2], acc[3]};
      float b[4] = {0.f, 0.f, 0.f, 0.f};
      const float one[4] = {1.f, 1.f, 1.f, 1.f};
      if (e.bias) loadv<float, 4>(e.bias + n, b);
      float x[4] = {0.f, 0.f, 0.f, 0.f};
      if (e.accumulate) epi_side<VITMI_EPI_STORE, float, 4>(e, m, n, x);
      epi_row<VITMI_EPI_STORE, float, 4>(e, m, n, v, b, one, x);
    }
    if (P > 1) __syncthreads();
  }
}
// waves per output group: enough waves to fill the chip, at least two slices per wave
inline int splitk_reduce2_waves(int64_t total4, int splits) {
  const int64_t groups = (total4 + 63) / 64;
  int P = 1;
  while (P < 16 && groups * P < 2048 && 2 * (2 * P) <= splits) P *= 2;
  return P;
}

// Split the contraction when the output has fewer tiles than the chip has CUs.  The target is
// ONE workgroup per CU, not the two that fit: every slice writes and the reduction re-reads a
// full fp32 output, and at the small outputs this path serves (CaiT D = 384, Swin C = 96..768
// weight gradients) that traffic outweighs the second workgroup's latency hiding (measured,
// whole step: target 512 / 384 / 256 / 192 / 128 -> CaiT-S24 19.74 / 19.59 / 19.37 / 19.59 /
// 20.43 ms, Swin-T 16.33 / - / 16.05 / 16.29 / 17.04 ms).
static std::atomic<int> g_splitk2_target{256};
inline void splitk_plan2(int tiles, int nt, int* splits, int* ksps) {
  int s = 1;
  if (tiles <= 256 && nt >= 16) {
    s = g_splitk2_target / tiles;
    if (s > nt / 8) s = nt / 8;
    if (s < 1) s = 1;
  }
  const int k = (nt + s - 1) / s;
  *ksps = k;
  *splits = (nt + k - 1) / k;
}

template <bool A_KM, bool B_KM, int MODE, typename TC>
int launch2(const GemmArgs& g, hipStream_t stream) {
  const int tiles_m = (int)((g.M + BM2 - 1) / BM2), tiles_n = (int)((g.N + BN2 - 1) / BN2);
  const int nwg = tiles_m * tiles_n;
  if constexpr (MODE == VITMI_EPI_STORE && sizeof(TC) == 4) {
    int splits, ksps;
    splitk_plan2(nwg, (int)(g.K / BK2), &splits, &ksps);
    if (splits > 1 && g.ws && g.ws_bytes >= (size_t)splits * g.M * g.N * sizeof(float)) {
      auto kern = gemm_fast2_kernel<A_KM, B_KM, MODE, TC, true>;
      if (int rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(kern), LDS2, "gemm_fast2(split-K)")) return rc;
      float* ws = reinterpret_cast<float*>(g.ws);
      hipLaunchKernelGGL(kern, dim3(nwg * splits), dim3(NT2), LDS2, stream, g, tiles_n, nwg * splits, nwg, ksps, ws);
      int rc = vitmi_check_launch("gemm_fast2_kernel(split-K)");
      if (rc) return rc;
      const int64_t total4 = g.M * g.N / 4;
      const int P = splitk_reduce2_waves(total4, splits);
      int64_t blocks = (total4 + 63) / 64;
      if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL(splitk_reduce2_kernel, dim3((unsigned)blocks), dim3(64 * P), 0, stream, ws, splits, g.e, g.M, g.N);
      return vitmi_check_launch("splitk_reduce2_kernel");
    }
  }
  auto kern = gemm_fast2_kernel<A_KM, B_KM, MODE, TC, false>;
  if (int rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(kern), LDS2, "gemm_fast2")) return rc;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT2), LDS2, stream, g, tiles_n, nwg, nwg, 0, (float*)nullptr);
  return vitmi_check_launch("gemm_fast2_kernel");
}

}  // namespace

// diagnostic hook: workgroups a split-K launch aims at (default 256 = one per CU)
extern "C" void vitmi_debug_gemm_splitk2_target(int n) { g_splitk2_target = n > 0 ? n : 256; }

// any M, N % 8 == 0 (a lane stores 4-8 consecutive columns), K a multiple of the 32-deep slab;
// a k-minor A ([K][M], the weight-gradient form) is staged in 8-column chunks: M % 8 == 0
bool gemm_fast2_shape_ok(const GemmArgs& g) {
  if (g.N % 8 != 0 || g.K % 32 != 0 || g.K < 64 || g.N < 8) return false;
  if (!g.a_km && (g.M % 8 != 0 || g.M < 8)) return false;
  const int64_t tiles = ((g.M + BM2 - 1) / BM2) * ((g.N + BN2 - 1) / BN2);
  return tiles < (1 << 30) && g.lda < (1 << 22) && g.ldb < (1 << 22);   // 32-bit per-lane offsets
}

size_t gemm_fast2_workspace(const GemmArgs& g) {
  if (g.e.mode != VITMI_EPI_STORE || g.e.c_bf16) return 0;
  int splits, ksps;
  splitk_plan2((int)(((g.M + BM2 - 1) / BM2) * ((g.N + BN2 - 1) / BN2)), (int)(g.K / BK2), &splits, &ksps);
  return splits > 1 ? (size_t)splits * g.M * g.N * sizeof(float) : 0;
}

int gemm_fast2_launch(const GemmArgs& g, hipStream_t s) {
  const EpiArgs& e = g.e;
  const bool nt = g.a_km && g.b_km, nn = g.a_km && !g.b_km;
#define GO(AKM, BKM, MODE) (e.c_bf16 ? launch2<AKM, BKM, MODE, bf16>(g, s) : launch2<AKM, BKM, MODE, float>(g, s))
  switch (e.mode) {
    case VITMI_EPI_STORE:
      if (nt) return GO(true, true, VITMI_EPI_STORE);
      if (nn) return GO(true, false, VITMI_EPI_STORE);
      return GO(false, false, VITMI_EPI_STORE);
    case VITMI_EPI_BIAS_GELU: return launch2<true, true, VITMI_EPI_BIAS_GELU, bf16>(g, s);
    case VITMI_EPI_RESIDUAL: return GO(true, true, VITMI_EPI_RESIDUAL);
    case VITMI_EPI_DGELU: return launch2<true, false, VITMI_EPI_DGELU, bf16>(g, s);
    case VITMI_EPI_PATCH_POS: return GO(true, true, VITMI_EPI_PATCH_POS);
  }
#undef GO
  return vitmi_fail(VITMI_E_SHAPE, "gemm_fast2: combination not built");
}

// every diagnostic switch of this file back to its default (vitmi_debug_reset, core.cpp)
void vitmi_debug_reset_gemm_fast2() {
  g_splitk2_target = 256;
}
