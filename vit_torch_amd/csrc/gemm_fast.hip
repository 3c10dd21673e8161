// Aligned-shape bf16 GEMM for gfx950: 256x256x64 tiles, 8 waves (2 x 4), each
// wave a 128x64 output block as 8x4 v_mfma_f32_16x16x32_bf16 accumulators,
// operands staged HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR
// round trip), double-buffered (2 x 64 KiB of the CU's 160 KiB LDS).
//
// Two LDS images, so that all three products of a Linear layer run without a
// transposed copy of anything in HBM:
//   * k-major operand  X[row][k]  (forward x and W, dgrad dy): "row image",
//     1-KiB subtiles of 16 rows x 32 k, XOR-swizzled (byte bit5 ^= bit9) so
//     the ds_read_b128 fragment reads are bank-conflict free;
//   * k-minor operand  X[k][col]  (dgrad W, wgrad dy and x): "k-row image",
//     512-B k-rows whose 32-B column chunks are XOR-permuted by
//     key(k) = (k&3) | ((k>>3)&1)<<2, read with ds_read_b64_tr_b16 (hardware
//     transpose) — conflict-free for the two 16-lane groups of a half-wave.
// LDS-DMA writes lane-linearly, so both swizzles are applied to the per-lane
// SOURCE address and again on the read (guide §5.4 rule 21).
//
// The MFMA is issued with the operands swapped (D^T = B^T-frag x A-frag) so a
// lane ends up with 4 CONSECUTIVE output columns of one row: epilogue loads
// (bias, residual, pre-activation) and stores are 8/16-byte vectors.
//
// Block -> tile map is XCD-aware (8 XCDs, private L2s): each XCD walks a
// contiguous run of tiles, n fastest, so the A panel of a row of tiles and the
// whole weight matrix stay in that XCD's L2.
#include <atomic>
#include <type_traits>
#include "gemm_tile.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = 256 * 64 * 2;       // one operand tile: 32 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + B
constexpr int NTHREADS = 512;

// ---- staging: each wave issues 4 LDS-DMA instructions (1 KiB each) per tile
template <bool KM, bool HID = false>
__device__ __forceinline__ void stage_tile(char* tile, const bf16* __restrict__ X, int64_t ld,
                                           int64_t r0, int64_t k0, int wave, int lane, int rmax = 255) {
  if constexpr (KM) {
    const int pb = 16 * lane;
    const int lb = pb ^ (((pb >> 9) & 1) << 5);
    const int row = lb >> 6, ch = (lb & 63) >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int st = wave * 4 + i;          // subtile: 16 rows x 32 k
      const int sr = st >> 1, kh = st & 1;
      const bf16* src = X + (r0 + min(sr * 16 + row, rmax)) * ld + k0 + kh * 32 + ch * 8;
      glds16x<HID>(src, tile + st * 1024);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = wave * 4 + i;           // pair of k-rows
      const int row = 2 * j + (lane >> 5);
      const int pc16 = lane & 31;
      const int key = (row & 3) | (((row >> 3) & 1) << 2);
      const int c32 = (pc16 >> 1) ^ key;
      const bf16* src = X + (k0 + row) * ld + r0 + c32 * 16 + (pc16 & 1) * 8;
      glds16x<HID>(src, tile + j * 1024);
    }
  }
}

// ---- fragment reads: 16 rows (or cols) x 32 k of block rb, k-half kh
template <bool KM>
__device__ __forceinline__ bf16x8 load_frag(const char* tile, int rb, int kh, int lane) {
  if constexpr (KM) {
    int pb = (lane & 15) * 64 + (lane >> 4) * 16;
    pb ^= ((pb >> 9) & 1) << 5;
    return *reinterpret_cast<const bf16x8*>(tile + (rb * 2 + kh) * 1024 + pb);
  } else {
    const int g = lane >> 4, i = lane & 15;
    const int row = 32 * kh + 8 * g + (i >> 2);
    const int key = (row & 3) | (((row >> 3) & 1) << 2);
    const char* p = tile + row * 512 + ((rb ^ key) * 32) + 8 * (i & 3);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p + 4 * 512));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

// ---- PIPE=1 building blocks: 32-deep k-slabs (A slab 16 KiB + B slab 16 KiB)
constexpr int SLAB_BYTES = 256 * 32 * 2;      // one operand slab
constexpr int RING_STAGE = 2 * SLAB_BYTES;    // A slab | B slab
// Per-wave staging plan of one operand: two LDS-DMA instructions per slab.  Everything
// that depends on the lane is folded into a 32-bit byte offset computed ONCE, everything
// else into a wave-uniform base pointer that advances by a constant per slab, so issuing
// a slab costs scalar adds + 2 global_load_lds per operand (no per-slab VALU address math:
// the R phase must stay shorter than the partner's 32-MFMA M phase).
template <bool KM> struct SlabPlan {
  const char* base[2];     // wave-uniform
  uint32_t off[2];         // per lane
  int64_t step;            // bytes per 32-deep slab
  // rmax: last valid row of the tile relative to r0 (255 for a whole tile).  A ragged last row tile (M % 256 != 0,
  // VITMI_LAUNCH_ROWS_PADDED) stages row rmax in place of the rows beyond the matrix: they only feed output rows that
  // land in the caller's row padding.  The clamp lives in the per-lane offset, computed once per tile.
  __device__ __forceinline__ void init(const bf16* X, int64_t ld, int64_t r0, int64_t k0, int wave, int lane, int rmax = 255) {
    if constexpr (KM) {    // 16 subtiles of 16 rows x 32 k; this wave fills subtiles 2w, 2w+1
      const int pb = 16 * lane;
      const int lb = pb ^ (((pb >> 9) & 1) << 5);
      const int row = lb >> 6, ch = (lb & 63) >> 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int st = wave * 2 + i;
        base[i] = reinterpret_cast<const char*>(X + r0 * ld + k0);
        off[i] = (uint32_t)(((int64_t)min(st * 16 + row, rmax) * ld + ch * 8) * 2);
      }
      step = 64;
    } else {               // 32 k-rows of 512 B; this wave fills row pairs 2w, 2w+1
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int jj = wave * 2 + i;
        const int row = 2 * jj + (lane >> 5);
        const int pc16 = lane & 31;
        const int key = (row & 3) | (((row >> 3) & 1) << 2);
        const int c32 = (pc16 >> 1) ^ key;
        base[i] = reinterpret_cast<const char*>(X + (k0 + 2 * jj) * ld + r0);
        off[i] = (uint32_t)(((lane >> 5) * ld + c32 * 16 + (pc16 & 1) * 8) * 2);
      }
      step = 64 * ld;
    }
  }
  // the same from a wave-uniform LDS byte address (ring iterations unrolled over the stages)
  __device__ __forceinline__ void issue_addr(uint32_t lds_addr, int j) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_sbase_m0(base[i] + (int64_t)j * step, off[i], lds_addr + i * 1024);
  }
  template <bool HID = false>
  __device__ __forceinline__ void issue(char* slab, int j, int wave) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (HID) glds16_sbase(base[i] + (int64_t)j * step, off[i], slab + (wave * 2 + i) * 1024);
      else glds16(base[i] + (int64_t)j * step + off[i], slab + (wave * 2 + i) * 1024);
    }
  }
};

// per-lane LDS byte offset of the fragment of block rb inside a slab (loop invariant)
template <bool KM>
__device__ __forceinline__ uint32_t frag_off(int rb, int lane) {
  if constexpr (KM) {
    int pb = (lane & 15) * 64 + (lane >> 4) * 16;
    pb ^= ((pb >> 9) & 1) << 5;
    return (uint32_t)(rb * 1024 + pb);
  } else {
    const int g = lane >> 4, i = lane & 15;
    const int row = 8 * g + (i >> 2);
    const int key = (row & 3) | (((row >> 3) & 1) << 2);
    return (uint32_t)(row * 512 + ((rb ^ key) * 32) + 8 * (i & 3));
  }
}
// ---- PIPE=2 building blocks: 64-deep stages whose k-major image is filled with FULL
// 128-B lines.  With 32-deep slabs a k-major operand is fetched as 64-B row pieces (16
// half lines per LDS-DMA instruction, every line touched twice, a slab apart): the
// texture-address path, not the MFMA, then paces the R phase (measured: R 750 cycles
// for NT against 620 for TN at 8192^3).  Here one instruction moves 8 rows x 128 B.
// Row image: [256 rows][128 B], 16-B chunk c of row r stored at chunk c ^ ((r>>1)&7)
// (conflict-free for the ds_read_b128 lane groups: within a group the 16 rows at chunk
// c / c+1 land on 16 distinct 16-B slots).
template <bool KM> struct StagePlan {
  const char* base[4];     // wave-uniform
  uint32_t off[4];         // per lane
  int64_t step;            // bytes per 64-deep stage
  __device__ __forceinline__ void init(const bf16* X, int64_t ld, int64_t r0, int64_t k0, int wave, int lane, int rmax = 255) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int st = wave * 4 + i;
      if constexpr (KM) {  // 32 subtiles of 8 rows x 64 k
        const int row = st * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);          // LDS position from the UNCLAMPED row (the image's swizzle)
        base[i] = reinterpret_cast<const char*>(X + r0 * ld + k0);
        off[i] = (uint32_t)(((int64_t)min(row, rmax) * ld + c * 8) * 2);
      } else {             // 32 pairs of 512-B k-rows
        const int row = 2 * st + (lane >> 5);
        const int pc16 = lane & 31;
        const int key = (row & 3) | (((row >> 3) & 1) << 2);
        const int c32 = (pc16 >> 1) ^ key;
        base[i] = reinterpret_cast<const char*>(X + (k0 + 2 * st) * ld + r0);
        off[i] = (uint32_t)(((lane >> 5) * ld + c32 * 16 + (pc16 & 1) * 8) * 2);
      }
    }
    step = KM ? 128 : 128 * ld;
  }
  template <bool HID = false>
  __device__ __forceinline__ void issue(char* tile, int t, int wave) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (HID) glds16_sbase(base[i] + (int64_t)t * step, off[i], tile + (wave * 4 + i) * 1024);
      else glds16(base[i] + (int64_t)t * step + off[i], tile + (wave * 4 + i) * 1024);
    }
  }
};
// per-lane LDS byte offset (k-half 0) of the fragment of block rb in a 64-deep stage
// image; k-half 1 is `off ^ 64` (k-major) / `off + 16384` (k-rows)
template <bool KM>
__device__ __forceinline__ uint32_t frag_off64(int rb, int lane) {
  if constexpr (KM) {
    const int row = rb * 16 + (lane & 15);
    const int pc = (lane >> 4) ^ ((row >> 1) & 7);
    return (uint32_t)(row * 128 + pc * 16);
  } else {
    const int g = lane >> 4, i = lane & 15;
    const int row = 8 * g + (i >> 2);
    const int key = (row & 3) | (((row >> 3) & 1) << 2);
    return (uint32_t)(row * 512 + ((rb ^ key) * 32) + 8 * (i & 3));
  }
}
template <bool KM>
__device__ __forceinline__ uint32_t frag_half(uint32_t off, int kh) {
  if constexpr (KM) return kh ? (off ^ 64u) : off;
  else return off + (uint32_t)kh * 16384u;
}


// ---- residual fold (EPI_RESIDUAL, fp32 stream, no LayerScale / DropPath: every ViT block).
// The epilogue x_new = R + (acc + b) used to read R (256 KiB of fp32 per tile) AFTER the main
// loop, all CUs at the same moment: 37-41 k cycles of HBM-bound epilogue during which no MFMA
// runs, against a 42 k-cycle main loop at K = 768 (tools/gemm_timeline.py).  Instead R streams
// HBM -> LDS by LDS-DMA DURING the main loop (no registers needed, HBM otherwise idle there),
// one wave-private 16-row x 64-column strip (4 KiB) per k-step, and is ADDED INTO THE
// ACCUMULATORS two k-steps later (4 ds_read_b128 + 16 adds per strip): the epilogue is then a
// bias-add + store.  An R strip comes from HBM (~6 k cycles under load), so TWO strips per wave
// must be in flight: 64 KiB of LDS.  The loop that carries it is therefore PIPE 3 = the
// PIPE 1 loop on a ring of THREE 32-deep slabs (96 KiB) instead of four.  (A first version kept
// PIPE 2's two 64-KiB stages and one 4-KiB strip per wave: its main loop went 42 k -> 65 k
// cycles waiting for each strip within its own k-step.)
// Ordering: a strip's four DMAs are issued AFTER the slab DMAs of the same iteration, so the
// counted vmcnt that retires slab j+1 (everything older than slab j+2) leaves the newest
// strip in flight; a strip issued in iteration j is complete after the wait of iteration
// j+2 and is read in iteration j+4 by the wave that issued it (no barrier involved).
// Strip image: 16 rows x 256 B, 16-B chunk c of row r stored at chunk c ^ r (on the DMA's
// per-lane SOURCE address, as always), so the accumulator-shaped reads (row = lane & 15,
// chunk 4 ni + lane >> 4) are bank-conflict free.
constexpr int RING3_BYTES = 3 * (2 * 256 * 32 * 2);   // three slabs of A|B: 96 KiB
constexpr int RF_BASE = RING3_BYTES;             // first byte above the ring
constexpr int RF_STRIP = 16 * 64 * 4;            // one strip: 4 KiB
constexpr int RF_LDS = RF_BASE + 8 * 2 * RF_STRIP;   // 160 KiB: the CU's whole LDS
struct ResFold {
  const char* base;       // wave-uniform: &R[m0 + wm*128][n0 + wn*64]
  uint32_t off[4];        // per lane, one per 1-KiB DMA piece (4 rows x 256 B)
  int64_t strip_step;     // bytes between strips (16 rows)
  uint32_t rd;            // per lane LDS offset of chunk (ni = 0) in its row; ni adds (4*ni) ^ ... see read()
  int lr, lg;
  __device__ __forceinline__ void init(const float* R, int64_t ldr, int64_t m0, int64_t n0, int wm, int wn, int lane) {
    base = reinterpret_cast<const char*>(R + (m0 + wm * 128) * ldr + n0 + wn * 64);
    strip_step = 16 * ldr * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * i + (lane >> 4);
      const int c = (lane & 15) ^ row;           // LDS position (lane & 15) of row `row` holds chunk c
      off[i] = (uint32_t)((row * ldr + c * 4) * 4);
    }
    lr = lane & 15; lg = lane >> 4;
  }
  template <bool HID = false>
  __device__ __forceinline__ void issue(char* strip, int s) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (HID) glds16_sbase(base + (int64_t)s * strip_step, off[i], strip + i * 1024);
      else glds16(base + (int64_t)s * strip_step + off[i], strip + i * 1024);
    }
  }
  // the wave's accumulator-shaped view of the strip: f32x4 of (row lr, columns 16 ni + 4 lg ..)
  __device__ __forceinline__ void read(const char* strip, u32x4 (&r)[4]) const {
    const uint32_t a = (uint32_t)(uintptr_t)LDS_PTR(char, strip) + (uint32_t)lr * 256u;
    const uint32_t a0 = a + (uint32_t)(((0 + lg) ^ lr) * 16), a1 = a + (uint32_t)(((4 + lg) ^ lr) * 16);
    const uint32_t a2 = a + (uint32_t)(((8 + lg) ^ lr) * 16), a3 = a + (uint32_t)(((12 + lg) ^ lr) * 16);
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
  }
};

// SPLITK: the grid is (tiles x splits); block (tile, s) contracts k-steps
// [s*ksps, min((s+1)*ksps, nt)) and stores its raw fp32 partial tile into slab s
// of `ws` ([splits][M][N]); splitk_reduce_kernel sums the slabs in a fixed order
// (deterministic) and applies the epilogue.  Used for the weight gradients,
// whose output is only 9-48 tiles while the contraction runs over all M tokens.
//
// PIPE = 1 main loop: a 4-stage ring of 32-deep k-slabs and the 8 waves split into two
// groups (wm = 0 / 1; waves w and w+4 share a SIMD) that run ONE BARRIER APART: every
// barrier interval one group is in its R phase (12 fragment reads of slab j + the 4
// LDS-DMA instructions of slab j+3) while the other is in its M phase (32 MFMAs), so
// each SIMD's matrix pipe always has a wave issuing MFMAs while its partner moves
// data.  Slab j sits in stage j&3; group 0 reads it in interval 2j, group 1 in 2j+1,
// so stage (j-1)&3 is free again at R(j) and is refilled with slab j+3, which is first
// read three slabs (>= 5 intervals) later: the DMA never has to be waited for in
// steady state, and the counted vmcnt (8 = two younger slabs may still be in flight)
// before the barrier that precedes the first read orders it (guide: "Read a staged
// buffer one phase AFTER the wait that retires it").
__device__ __forceinline__ bool g_rfold_enabled(const GemmArgs& g) {
  return g.rfold != 0 && !g.e.gamma && !g.e.rowscale && !g.e.C2 && !g.e.r_bf16;
}

// s_waitcnt vmcnt(N) with a compile-time N (counted waits of the persistent prologue)
template <int N> __device__ __forceinline__ void wait_vm_c() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// the wave's 16-row x 64-column fp32 transposition strip of the epilogue: 4 KiB, 16-B chunk c of
// row r at chunk c ^ r (conflict-free for the accumulator-shaped ds_write_b128 and for the
// row-shaped ds_read_b128 of both output widths)
__device__ __forceinline__ float* strip_at(float* tr, int row, int col) {
  return tr + row * 64 + ((((col >> 2) ^ row) & 15) << 2);
}

// Linear tile index -> (row tile, column tile).  band == 0: row-major over all column tiles.  band > 0:
// column BANDS of `band` tiles, row-major inside a band, bands one after the other (the last may be
// narrower).  An XCD's contiguous run of the order is then (rows x band) instead of (few rows x all
// columns): the band's B panels (band x 256 x K x 2 B) stay in its 4-MiB L2 from round to round while
// the A panels stream, where the row-major order re-fetches a whole N x K weight matrix that does
// not fit (N = 3072, K = 768: 4.7 MB) every round.
__device__ __forceinline__ void tile_mn(int t, int tiles_m, int tiles_n, int band, int* mt, int* nt) {
  if (band <= 0 || band >= tiles_n) { *mt = t / tiles_n; *nt = t % tiles_n; return; }
  const int per_band = tiles_m * band;
  const int b = t / per_band, r = t - b * per_band;
  const int w = min(band, tiles_n - b * band);
  *mt = r / w;
  *nt = b * band + r % w;
}

// PERSISTENT tile walk (every launch but split-K): the grid is min(tiles, CUs) workgroups, one per
// CU (a workgroup takes the CU's whole register file and >= 128 KiB of LDS), and each walks the
// tiles slot, slot + G/8, ... of its XCD's contiguous run (the same tile -> XCD order as before).
// When the main loop of a tile ends the operand stages are free, so the LDS-DMA of the NEXT
// tile's first stages is issued BEFORE the epilogue: the ~5 k cycles a tile spent between its
// first DMA and its first MFMA (launch, plan set-up, HBM / L2 round trip) now run under the
// epilogue's stores.  The transposition strips therefore live in the LDS above the stages
// (bytes 128 Ki .. 160 Ki; PIPE 3: the second residual strip of each wave).
// vmcnt of the first wait of a prefetched tile: the epilogue's VMEM operations are YOUNGER than the
// prefetch, so they are added to the count of younger DMAs; only the stores to C are counted
// (a lower bound: fewer outstanding operations than allowed is always safe).
// One tile's epilogue from its k-slices' raw fp32 partial tiles in `ws` ([splits][M][N]): rows summed over the slices in the
// fixed order 0, 1, ... (deterministic; the same order and arithmetic as tail_epilogue_kernel), then epi_row.  Run by the
// whole workgroup (512 threads: a thread = W columns of a row, full lines).
template <int MODE, typename TC>
__device__ __forceinline__ void tile_fixup(const float* __restrict__ ws, int splits, const EpiArgs& e, int64_t M, int64_t N,
                                           int64_t m0, int64_t n0, int tid) {
  constexpr int W = sizeof(TC) == 2 ? 8 : 4;
  constexpr int TPR = 256 / W;                       // threads per 256-column row
  constexpr int RPP = NTHREADS / TPR;                // rows per pass
  const int64_t n = n0 + (tid % TPR) * W;
  float b[W], gm[W];
#pragma unroll
  for (int i = 0; i < W; ++i) { b[i] = 0.f; gm[i] = 1.f; }
  if (e.bias) loadv<float, W>(e.bias + n, b);
  if (MODE == VITMI_EPI_RESIDUAL && e.gamma) loadv<float, W>(e.gamma + n, gm);
#pragma unroll 2
  for (int r = tid / TPR; r < 256; r += RPP) {
    const int64_t m = m0 + r;
    float v[W];
    loadv<float, W>(ws + m * N + n, v);
    for (int s = 1; s < splits; ++s) {
      float t[W];
      loadv<float, W>(ws + ((int64_t)s * M + m) * N + n, t);
#pragma unroll
      for (int i = 0; i < W; ++i) v[i] += t[i];
    }
    float x[W];
#pragma unroll
    for (int i = 0; i < W; ++i) x[i] = 0.f;
    if (epi_has_side<MODE, TC>(e)) epi_side<MODE, TC, W>(e, m, n, x);
    epi_row<MODE, TC, W>(e, m, n, v, b, gm, x);
  }
}

// FIX (split-K instantiations of the split tail only): the k-slices of a tail tile store their raw partial tiles and take a
// ticket on the tile's arrival counter; the slice that arrives LAST applies the epilogue to the sum (tile_fixup) — the
// row-wise finisher kernel of round 2 (12-15 us per launch, 36 launches per ViT-B/16 step) is gone.  Inter-workgroup
// visibility as the guide's split-K recipe prescribes: every storing wave drains its stores (vmcnt(0)), the workgroup's
// barrier, lane 0's agent-scope RELEASE fence + its own vmcnt(0) (the compiler may drop the fence's), then the relaxed
// agent-scope ticket; the last arriver's lane 0 issues ONE agent-scope ACQUIRE + vmcnt(0), a barrier, and every wave reads
// the slabs with plain loads.  Nothing spins: a wrong counter could only leave a tile without its epilogue (the tests'
// NaN-filled outputs would show it), never hang.  The counters are zeroed by the full-rounds launch that precedes the
// slices on the same stream (GemmArgs::zero_cnt).
// DEEP: 0 = side inputs one strip ahead (every epilogue); 1 = three strips ahead, packed (bf16 RESIDUAL / DGELU);
// 2 = the same for the PLAIN residual epilogue (no second output, no row scale, plain stores) as straight-line code
template <bool A_KM, bool B_KM, int MODE, typename TC, bool SPLITK = false, int PIPE = 0, bool FIX = false, int DEEP = 0>
__global__ __launch_bounds__(NTHREADS) void gemm_fast_kernel(GemmArgs g, int tiles_n, int nwg,
                                                             int ntiles, int ksps, float* ws, int tile0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // stages | strips
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  constexpr bool PERSIST = !SPLITK;
  constexpr int W = sizeof(TC) == 2 ? 8 : 4;       // columns per lane in the row pass
  constexpr int LPR = 64 / W;                      // lanes per 64-column row
  constexpr int RPI = 64 / LPR;                    // rows per wave instruction
  constexpr int NJ = 16 / RPI;                     // row groups per strip
  // INVARIANT (the first wait of a prefetched tile depends on it): every epilogue variant issues AT LEAST E_MIN = 8 strips x NJ
  // row groups UNCONDITIONAL vector-memory stores to C per wave AFTER the next tile's prologue DMAs (epi_row / the raw split-K
  // store below: one storev per (mi, j), never predicated or merged).  `vmcnt(younger + E_MIN)` then leaves at most the
  // epilogue's own operations outstanding, i.e. the prologue slab it is about to read has landed.  An epilogue that issued
  // FEWER stores would under-wait.  Second outputs (C2), side loads and column sums only ADD operations (safe direction).
  // Guard: vitmi_debug_gemm_strict_wait(1) replaces the counted wait by vmcnt(0);
  // tests/test_ops_gpu.py::test_gemm_counted_prefetch_wait_equals_strict_wait compares both bit for bit on every epilogue.
  constexpr int E_MIN = 8 * NJ;                    // stores to C every epilogue issues

  // XCD-aware, bijective tile map (guide §5: blocks b and b+8 share an XCD): XCD x owns the
  // contiguous run [x_start, x_start + x_len) of the nwg tiles, n fastest inside it
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int x_start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int x_len = q + (xcd < r ? 1 : 0);
  const int slot = bid >> 3;
  const int gx = PERSIST ? (((int)gridDim.x - xcd + 7) >> 3) : 1;   // workgroups of this XCD
  // split-K: the grid is (tiles x splits), one (tile, split) per workgroup
  const int wg0 = x_start + slot;
  const int split = SPLITK ? wg0 / ntiles : 0;

  // the problem this workgroup works on: the descriptor's, or (paired split-K launch) the second one
  const bf16* A = reinterpret_cast<const bf16*>(g.A);
  const bf16* B = reinterpret_cast<const bf16*>(g.B);
  int64_t lda = g.lda, ldb = g.ldb, pM = g.M, pN = g.N;
  int idx = slot;                                  // position inside the XCD's run (PERSIST)
  int tile = tile0 + (SPLITK ? wg0 % ntiles : x_start + idx);
  if constexpr (SPLITK) {
    if (g.tiles1 > 0 && tile >= g.tiles1) {        // wave-uniform
      tile -= g.tiles1;
      A = reinterpret_cast<const bf16*>(g.A2); B = reinterpret_cast<const bf16*>(g.B2);
      lda = g.lda2; ldb = g.ldb2; pM = g.M2; pN = g.N2;
      tiles_n = g.tiles_n2;
      ws = g.ws2;
    }
  }
  const int nt_all = (int)(g.K / BK);
  const int kt0 = SPLITK ? split * ksps : 0;
  const int nt = SPLITK ? min(ksps, nt_all - kt0) : nt_all;
  const int64_t kb0 = (int64_t)kt0 * BK;
  // residual fold (see ResFold): compiled into the PIPE 3 loop of the fp32-stream residual
  // epilogue, taken when nothing scales the branch and the contraction has the 19 slabs the
  // eight strips need
  constexpr bool RFOLD_T = MODE == VITMI_EPI_RESIDUAL && sizeof(TC) == 4 && !SPLITK && PIPE == 3;
  const bool rfold = RFOLD_T && g_rfold_enabled(g) && nt >= 10;
  // diagnostic build of the timeline (armed by tools/gemm_phases.py only): entry,
  // epilogue start and end of the first tile of the first blocks, wave 0
  const bool dbg_tl = g.dbg != nullptr && (int)blockIdx.x < g.dbg_blocks && wave == 0;
  // per-phase stamps of the main loop (tools/gemm_phases.py): compiled in only with -DVITMI_GEMM_PHASE_STAMPS
  // (VITMI_EXTRA_FLAGS of vit_torch_amd/build.py).  As a run-time option they put four scalar branches around s_memtime
  // blocks into every slab iteration of every production launch (round 3: ISA audit of the loop).
#ifdef VITMI_GEMM_PHASE_STAMPS
  const bool dbg = g.dbg != nullptr && blockIdx.x == 0;   // wave-uniform
#else
  constexpr bool dbg = false;
#endif
  if constexpr (PERSIST) {
    if (g.zero_cnt && blockIdx.x == 0 && tid < g.zero_n) g.zero_cnt[tid] = 0u;     // the tail launch that follows counts from 0
  }
  unsigned long long tl0 = 0;
  if (dbg_tl) {
    tl0 = stamp();
    unsigned long long rt0;                        // 100 MHz wall clock at entry: with the pair at exit, the shader clock under THIS kernel
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0) :: "memory");
    if (lane == 0) g.dbg[64 + blockIdx.x * 8 + 4] = rt0;
  }

  // ---- per-PIPE staging plans and fragment offsets (lane parts are tile independent)
  constexpr bool RINGP = PIPE == 1 || PIPE == 3;
  constexpr int RING = PIPE == 3 ? 3 : 4;          // slabs resident; RING - 1 are prefetched ahead
  constexpr int AHEAD = RING - 1;
  const int ns = 2 * nt;                           // 32-deep slabs
  const int grp = wm;                              // 0: leads, 1: one barrier behind
  SlabPlan<A_KM> sa;
  SlabPlan<B_KM> sb;
  StagePlan<A_KM> ta;
  StagePlan<B_KM> tb;
  ResFold rf;
  uint32_t fa[8], fb[4];
  if constexpr (RINGP) {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) fa[mi] = frag_off<A_KM>(wm * 8 + mi, lane);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) fb[ni] = frag_off<B_KM>(wn * 4 + ni, lane) + SLAB_BYTES;
  } else if constexpr (PIPE == 2) {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) fa[mi] = frag_off64<A_KM>(wm * 8 + mi, lane);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) fb[ni] = frag_off64<B_KM>(wn * 4 + ni, lane);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  uint32_t fa_lo[8], fa_hi[8], fb_lo[4], fb_hi[4];   // PIPE 1: LDS byte addresses of the fragments in stage 0 / stage 2 (see slab_iter)
  constexpr bool STAGE_UNROLL = PIPE == 1 && SPLITK && !(A_KM && !B_KM);   // (the NN split-K form spills with the second address set)
  if constexpr (STAGE_UNROLL) {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) { fa_lo[mi] = lds0 + fa[mi]; fa_hi[mi] = fa_lo[mi] + 2 * RING_STAGE; }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) { fb_lo[ni] = lds0 + fb[ni]; fb_hi[ni] = fb_lo[ni] + 2 * RING_STAGE; }
  }
  char* rstrip = smem + RF_BASE + wave * 2 * RF_STRIP;   // PIPE 3: this wave's two residual strip buffers
  // epilogue strip: above the stages (the stages may already be receiving the next tile)
  // (split-K launches own only the two stages' 128 KiB and have no prefetch in flight after the loop: their strips alias the
  // dead stages whatever the loop.  Round 2 placed the PIPE 3 split-K strips above 128 KiB — outside the allocation, where LDS
  // writes are dropped: the k-sliced tail of the fp32-stream fc2 forward lost the rows of waves 4-7.  Found by
  // test_gemm_tail_fixup_by_the_last_arriving_slice[nt-res-fp32] in round 3.)
  float* tr = reinterpret_cast<float*>(SPLITK ? smem + wave * 4096
                                              : (PIPE == 3 ? rstrip + RF_STRIP : smem + 2 * STAGE_BYTES + wave * 4096));

  // ragged M (VITMI_LAUNCH_ROWS_PADDED, k-major A): last valid row of the tile at m0, relative to m0
  auto a_rmax = [&](int64_t m0_) { return A_KM ? (int)min((int64_t)255, pM - 1 - m0_) : 255; };
  // plans of a tile + the DMAs that precede its main loop
  // hid (a std::bool_constant): the DMAs go out from inline asm, unseen by hipcc — the prologue issued in front of an
  // epilogue (glds16_hidden)
  auto start_tile = [&](int64_t m0, int64_t n0, auto hid) {
    constexpr bool HID = decltype(hid)::value || PIPE != 0;     // the phased loops wait and order every DMA themselves: all of theirs go out unseen
    if (g.dbg_alias & 1) m0 = 0;                                // diagnostic (vitmi_debug_gemm_alias): operand panel 0 for every tile
    if (g.dbg_alias & 2) n0 = 0;
    if constexpr (RINGP) {
      sa.init(A, lda, m0, kb0, wave, lane, a_rmax(m0));
      sb.init(B, ldb, n0, kb0, wave, lane);
      if constexpr (RFOLD_T) { if (rfold) rf.init(reinterpret_cast<const float*>(g.e.R), g.e.ldr, m0, n0, wm, wn, lane); }
#pragma unroll
      for (int a = 0; a < AHEAD; ++a)
        if (a < ns) {
          char* st = smem + a * RING_STAGE;
          sa.template issue<HID>(st, a, wave);
          sb.template issue<HID>(st + SLAB_BYTES, a, wave);
        }
      if constexpr (RFOLD_T) { if (rfold) rf.template issue<HID>(rstrip, 0); }     // strip 0 rides behind the prologue slabs
    } else if constexpr (PIPE == 2) {
      ta.init(A, lda, m0, kb0, wave, lane, a_rmax(m0));
      tb.init(B, ldb, n0, kb0, wave, lane);
      ta.template issue<HID>(smem, 0, wave);
      tb.template issue<HID>(smem + TILE_BYTES, 0, wave);
      if (nt > 1) {
        ta.template issue<HID>(smem + STAGE_BYTES, 1, wave);
        tb.template issue<HID>(smem + STAGE_BYTES + TILE_BYTES, 1, wave);
      }
    } else {                       // PIPE 0: stage 0 (the loop issues stage t + 1 itself)
      stage_tile<A_KM, HID>(smem, A, lda, m0, kb0, wave, lane, a_rmax(m0));
      stage_tile<B_KM, HID>(smem + TILE_BYTES, B, ldb, n0, kb0, wave, lane);
    }
  };

  if (PERSIST && idx >= x_len) return;
  const int tiles_m = (int)((pM + BM - 1) / BM);
  int mt_, nt_;
  tile_mn(tile, tiles_m, tiles_n, g.band, &mt_, &nt_);
  int64_t m0 = (int64_t)mt_ * BM, n0 = (int64_t)nt_ * BN;
  start_tile(m0, n0, std::false_type{});
  bool prefetched = false;                         // the tile's prologue DMAs were issued before an epilogue
  // Start stagger: a launch of q full rounds plus a remainder leaves most workgroups one tile short of
  // the longest list, i.e. idle for a tile time at the end.  Those workgroups instead start late by a
  // phase-dependent part of that slack, so that their epilogues (HBM bursts, no MFMA) fall into the main
  // loops of the others instead of all 256 CUs hitting HBM at the same moment.
  if constexpr (PERSIST) {
    if (g.stag_cycles > 0) {
      const int mine = (x_len - slot + gx - 1) / gx, longest = (x_len + gx - 1) / gx;
      if (mine < longest) {
        const unsigned long long t0 = stamp();
        const unsigned long long d = (unsigned long long)g.stag_cycles * (unsigned)(1 + slot % g.stag_phases) / (unsigned)g.stag_phases;
        while (stamp() - t0 < d) __builtin_amdgcn_s_sleep(16);
      }
    }
  }

#pragma unroll 1
  for (;;) {
  f32x4 acc[4][8];   // [ni][mi]: D^T blocks, lane holds row m = l&15, cols n = 4*(l>>4)+r
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) { acc[ni][mi][0] = 0.f; acc[ni][mi][1] = 0.f; acc[ni][mi][2] = 0.f; acc[ni][mi][3] = 0.f; }

  if constexpr (PIPE == 0) {
  __syncthreads();                 // stage 0 (start_tile) has landed: vmcnt(0) + barrier

  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * STAGE_BYTES;
    if (t + 1 < nt) {
      char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
      stage_tile<A_KM>(nxt, A, lda, m0, (int64_t)(kt0 + t + 1) * BK, wave, lane, a_rmax(m0));
      stage_tile<B_KM>(nxt + TILE_BYTES, B, ldb, n0, (int64_t)(kt0 + t + 1) * BK, wave, lane);
    }
    const char* At = cur;
    const char* Bt = cur + TILE_BYTES;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      bf16x8 bf[4], af[8];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bf[ni] = load_frag<B_KM>(Bt, wn * 4 + ni, kh, lane);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) af[mi] = load_frag<A_KM>(At, wm * 8 + mi, kh, lane);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
    __syncthreads();   // next stage landed (vmcnt(0)) and everyone is done reading `cur`
  }

  } else if constexpr (RINGP) {
    int stage_i = 0;                               // j % RING without a division
    auto stage_of = [&](int ahead) { int x = stage_i + ahead; return x >= RING ? x - RING : x; };
    {   // slab 0 has landed: everything issued after it may still fly (later slabs, the residual
        // strip, and, behind a prefetch, the previous tile's epilogue)
      const int younger = 4 * (min(ns, AHEAD) - 1) + ((RFOLD_T && rfold) ? 4 : 0);
      if (prefetched && g.strict_wait) wait_vm_c<0>();
      else if (prefetched) {
        if (younger >= 12) wait_vm_c<12 + E_MIN>();
        else if (younger >= 8) wait_vm_c<8 + E_MIN>();
        else if (younger >= 4) wait_vm_c<4 + E_MIN>();
        else wait_vm_c<E_MIN>();
      } else {
        if (younger >= 12) wait_vm_c<12>();
        else wait_vm(younger);
      }
    }
    raw_barrier();                                 // b0
    if (grp == 1) raw_barrier();                   // stagger
    unsigned long long tR = 0, tWR = 0, tM = 0, tWM = 0, t0 = 0, t1 = 0;
    // One slab iteration.  STEADY (j + AHEAD < ns, i.e. every iteration but the last three): the DMA of slab j + AHEAD is
    // always issued, two younger slabs are always in flight (vmcnt(8)) and both barriers are unconditional — as
    // compile-time facts, so the steady-state loop carries no scalar branch ladder (wait_vm's three-way switch, the
    // issue / barrier conditions: ~10 branches per iteration in the round-2 loop).  The residual-fold loop (PIPE 3) keeps
    // the general form for every iteration.
    // STAGE >= 0 (steady iterations of the four-stage ring, unrolled over the stages): the stage of slab j is a
    // compile-time fact, so the fragment reads take it as an IMMEDIATE offset from one of two per-lane address sets
    // (stages 0 / 1: the address itself, + 32 KiB; stages 2 / 3: address + 64 KiB, + 32 KiB) and the DMA destination is a
    // constant: the twelve per-fragment address adds and the ring arithmetic leave the R phase (ISA audit, round 3).
    // RFS >= 0 (residual fold): this iteration adds strip RFS — a COMPILE-TIME index into the accumulators.  As a run-time
    // comparison chain (`if ((j >> 1) - 2 == S)` over the unrolled S) hipcc recognised acc[ni][(j >> 1) - 2], put 24
    // accumulator quads into scratch and indexed them dynamically: 2.4 ms per launch instead of 0.13-0.25.
    auto slab_iter = [&](int j, auto steady_tag, auto stage_tag, auto rf_tag) {
      constexpr bool STEADY = decltype(steady_tag)::value;
      constexpr int STAGE = decltype(stage_tag)::value;
      constexpr int RFS = decltype(rf_tag)::value;
      if (dbg) t0 = stamp();
      // ---- R(j)
      Frag<B_KM> fbv[4];
      Frag<A_KM> fav[8];
      if constexpr (STAGE >= 0) {
        constexpr int DST = (STAGE + AHEAD) % RING;
        const uint32_t da = lds0 + DST * RING_STAGE + wave * 2048;
        sa.issue_addr(da, j + AHEAD);
        sb.issue_addr(da + SLAB_BYTES, j + AHEAD);
        constexpr int IMM = (STAGE & 1) * RING_STAGE;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) fbv[ni].template load_imm<IMM>(STAGE < 2 ? fb_lo[ni] : fb_hi[ni]);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) fav[mi].template load_imm<IMM>(STAGE < 2 ? fa_lo[mi] : fa_hi[mi]);
      } else {
      if (STEADY || j + AHEAD < ns) {
        char* st = smem + stage_of(AHEAD) * RING_STAGE;
        sa.template issue<true>(st, j + AHEAD, wave);
        sb.template issue<true>(st + SLAB_BYTES, j + AHEAD, wave);
      }
      const char* As = smem + stage_i * RING_STAGE;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fbv[ni].load(As, fb[ni]);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) fav[mi].load(As, fa[mi]);
      }
      // residual fold: strip s is issued in iteration 2s (s = 0 in the prologue) and read in
      // iteration 2s + 4; buffer s & 1
      u32x4 rfv[4];
      const bool rf_even = RFOLD_T && rfold && (j & 1) == 0;                 // wave-uniform
      const bool rf_issue = rf_even && j >= 2 && j < 16;
      if constexpr (RFOLD_T && RFS >= 0) rf.read(rstrip + (RFS & 1) * RF_STRIP, rfv);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of this stage are complete
      frag_wait4(fbv[0], fbv[1], fbv[2], fbv[3]);
      frag_wait4(fav[0], fav[1], fav[2], fav[3]);
      frag_wait4(fav[4], fav[5], fav[6], fav[7]);
      if constexpr (RFOLD_T) {
        if constexpr (RFS >= 0) {
          asm volatile("" : "+v"(rfv[0]), "+v"(rfv[1]), "+v"(rfv[2]), "+v"(rfv[3]));   // named after the wait
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[ni][RFS] += __builtin_bit_cast(f32x4, rfv[ni]);
        }
        if (rf_issue) rf.template issue<true>(rstrip + ((j >> 1) & 1) * RF_STRIP, j >> 1);   // its buffer has just been read
      }
      bf16x8 bf[4], af[8];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bf[ni] = fbv[ni].get();
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) af[mi] = fav[mi].get();
      // slab j+1 must have landed before the next iteration reads it: everything issued after
      // it may still be in flight = the slabs ahead of it + the strip issued in this iteration
      // or the previous one (strips go out every other iteration, behind that iteration's slab)
      int nvm = 4 * (AHEAD - 1);
      if constexpr (!STEADY) {
        int fly = min(ns - 2 - j, AHEAD - 1);
        if (fly < 0) fly = 0;
        nvm = 4 * fly;
        // strips younger than slab j+1: the one issued in this iteration (even j in [2,16)) or the
        // previous one (odd j), or the prologue's strip 0 (younger than slab 1 only: j = 0)
        if constexpr (RFOLD_T) { if (rfold && (j == 0 || (j >= 2 && j < 16))) nvm += 4; }
      }
      if (grp == 1) { if constexpr (STEADY) wait_vm_c<4 * (AHEAD - 1)>(); else wait_vm(nvm); }
      if (dbg) { t1 = stamp(); tR += t1 - t0; }
      raw_barrier();
      if (dbg) { t0 = stamp(); tWR += t0 - t1; }
      // ---- M(j)
#ifdef VITMI_GEMM_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[ni][mi], 0, 0, 0);
#ifdef VITMI_GEMM_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      if (grp == 0) { if constexpr (STEADY) wait_vm_c<4 * (AHEAD - 1)>(); else wait_vm(nvm); }
      if (dbg) { t1 = stamp(); tM += t1 - t0; }
      if (STEADY || !(grp == 1 && j == ns - 1)) raw_barrier();
      if (dbg) { t0 = stamp(); tWM += t0 - t1; }
      if constexpr (STAGE < 0) stage_i = stage_i == RING - 1 ? 0 : stage_i + 1;
    };
    using RT = std::integral_constant<int, -1>;
    using NRF = std::integral_constant<int, -1>;
    int j = 0;
    if constexpr (!RFOLD_T && RING == 4 && STAGE_UNROLL) {
      // four steady iterations per trip (all of j .. j + 3 have j + AHEAD < ns); the ring is back at stage 0 afterwards.
      // Only the split-K instantiations (the weight gradients: 113-step slices, a raw-store epilogue): the twelve extra
      // address registers make the kernels with a real epilogue spill inside the loop.
#pragma unroll 1
      for (; j + 3 + AHEAD < ns; j += 4) {
        slab_iter(j, std::true_type{}, std::integral_constant<int, 0>{}, NRF{});
        slab_iter(j + 1, std::true_type{}, std::integral_constant<int, 1>{}, NRF{});
        slab_iter(j + 2, std::true_type{}, std::integral_constant<int, 2>{}, NRF{});
        slab_iter(j + 3, std::true_type{}, std::integral_constant<int, 3>{}, NRF{});
      }
    } else if constexpr (!RFOLD_T) {
#pragma unroll 1
      for (; j + AHEAD < ns; ++j) slab_iter(j, std::true_type{}, RT{}, NRF{});
    } else {
      // residual fold (ns >= 20): slabs 0-3 in the general form, then strip S is added in iteration 4 + 2 S
      if (rfold) {
#pragma unroll 1
        for (; j < 4; ++j) slab_iter(j, std::false_type{}, RT{}, NRF{});
        auto rf_pair = [&](auto s_tag) {
          constexpr int S = decltype(s_tag)::value;
          slab_iter(4 + 2 * S, std::false_type{}, RT{}, s_tag);
          slab_iter(5 + 2 * S, std::false_type{}, RT{}, NRF{});
        };
        rf_pair(std::integral_constant<int, 0>{}); rf_pair(std::integral_constant<int, 1>{});
        rf_pair(std::integral_constant<int, 2>{}); rf_pair(std::integral_constant<int, 3>{});
        rf_pair(std::integral_constant<int, 4>{}); rf_pair(std::integral_constant<int, 5>{});
        rf_pair(std::integral_constant<int, 6>{}); rf_pair(std::integral_constant<int, 7>{});
        j = 20;
      }
    }
#pragma unroll 1
    for (; j < ns; ++j) slab_iter(j, std::false_type{}, RT{}, NRF{});
    if (dbg && lane == 0 && !prefetched) {
      g.dbg[wave * 4 + 0] = tR; g.dbg[wave * 4 + 1] = tWR;
      g.dbg[wave * 4 + 2] = tM; g.dbg[wave * 4 + 3] = tWM;
      g.dbg[32 + wave] = ns;
    }
  } else {
    // PIPE == 2: two 64-deep stages (full-line LDS-DMA), the same two wave groups one
    // barrier apart; stage t+1 is issued at R(2t) into the buffer stage t-1 has just
    // vacated and waited for (vmcnt(0)) at the end of phase 2t+1.
    if (prefetched && g.strict_wait) wait_vm_c<0>();
    else if (prefetched) { if (nt > 1) wait_vm_c<8 + E_MIN>(); else wait_vm_c<E_MIN>(); }
    else wait_vm(nt > 1 ? 8 : 0);
    raw_barrier();
    if (grp == 1) raw_barrier();
    unsigned long long tR = 0, tWR = 0, tM = 0, tWM = 0, t0 = 0, t1 = 0;
    // One 32-deep half of stage t, KH a compile-time fact: the stage pair is unrolled, so the k-half selects, the waits of
    // the second half and the issue of the first are not branches on j & 1 any more (round 3).
    auto half_iter = [&](int t, auto kh_tag, auto steady_tag) {
      constexpr int kh = decltype(kh_tag)::value;
      constexpr bool STEADY = decltype(steady_tag)::value;
      if (dbg) t0 = stamp();
      if (kh == 0 && (STEADY || (t >= 1 && t + 1 < nt))) {
        char* st = smem + ((t + 1) & 1) * STAGE_BYTES;
        ta.template issue<true>(st, t + 1, wave);
        tb.template issue<true>(st + TILE_BYTES, t + 1, wave);
      }
      const char* At = smem + (t & 1) * STAGE_BYTES;
      const char* Bt = At + TILE_BYTES;
      Frag<B_KM> fbv[4];
      Frag<A_KM> fav[8];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fbv[ni].load(Bt, frag_half<B_KM>(fb[ni], kh));
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) fav[mi].load(At, frag_half<A_KM>(fa[mi], kh));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      frag_wait4(fbv[0], fbv[1], fbv[2], fbv[3]);
      frag_wait4(fav[0], fav[1], fav[2], fav[3]);
      frag_wait4(fav[4], fav[5], fav[6], fav[7]);
      bf16x8 bf[4], af[8];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bf[ni] = fbv[ni].get();
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) af[mi] = fav[mi].get();
      if (kh == 1 && grp == 1) wait_vm_c<0>();
      if (dbg) { t1 = stamp(); tR += t1 - t0; }
      raw_barrier();
      if (dbg) { t0 = stamp(); tWR += t0 - t1; }
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[ni][mi], 0, 0, 0);
      if (kh == 1 && grp == 0) wait_vm_c<0>();
      if (dbg) { t1 = stamp(); tM += t1 - t0; }
      if (STEADY || kh == 0 || !(grp == 1 && t == nt - 1)) raw_barrier();
      if (dbg) { t0 = stamp(); tWM += t0 - t1; }
    };
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {                   // (a steady-state copy of the pair, as in the ring loop, made hipcc spill here)
      half_iter(t, std::integral_constant<int, 0>{}, std::false_type{});
      half_iter(t, std::integral_constant<int, 1>{}, std::false_type{});
    }
    if (dbg && lane == 0 && !prefetched) {
      g.dbg[wave * 4 + 0] = tR; g.dbg[wave * 4 + 1] = tWR;
      g.dbg[wave * 4 + 2] = tM; g.dbg[wave * 4 + 3] = tWM;
      g.dbg[32 + wave] = ns;
    }
  }

  // ---- epilogue.  The accumulators hold 16x16 blocks with 4 columns per lane; stored
  // as they are, one wave instruction touches 16 rows x 32..64 B (16 partial lines) and
  // the tile's store tail is issue-bound (~8 B/clk/CU measured).  Instead each wave
  // transposes one 16-row x 64-column strip at a time through a PRIVATE 4-KiB LDS strip
  // (same-wave DS ops execute in order, so no barrier is needed) and then reads,
  // post-processes and stores whole rows: every global access of the epilogue (C, C2, R,
  // AUX) is 16 B per lane and covers full 128-B lines.
  unsigned long long tl1 = 0;
  if (dbg_tl && !prefetched) tl1 = stamp();
  const int lr = lane & 15, lg = lane >> 4;
  const int rr = lane / LPR, rc = (lane % LPR) * W;
  float bias_r[W], gamma_r[W];
#pragma unroll
  for (int i = 0; i < W; ++i) { bias_r[i] = 0.f; gamma_r[i] = 1.f; }
  if constexpr (!SPLITK) {
    if (g.e.bias) loadv<float, W>(g.e.bias + n0 + wn * 64 + rc, bias_r);
    if (MODE == VITMI_EPI_RESIDUAL && g.e.gamma) loadv<float, W>(g.e.gamma + n0 + wn * 64 + rc, gamma_r);
  }
  // ---- the next tile of this workgroup: its first stages start to fill now
  bool has_next = false;
  int64_t m0n = 0, n0n = 0;
  if constexpr (PERSIST) {
    const int nidx = idx + gx;
    has_next = nidx < x_len;
    // bias / LayerScale first and waited for with a wait the compiler SEES: while an LDS-DMA is
    // in flight hipcc answers the first use of any ordinary load with vmcnt(0), which would
    // put the whole prefetch in front of the epilogue.  On BOTH paths (round 3): the main loop's DMAs are retired by
    // asm waits hipcc does not see, so without this it still believes them pending when the last tile's epilogue
    // (no prefetch) meets the prefetching path at the join, and answers the first side-input use with vmcnt(0) again.
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): nothing but these two small loads is outstanding
    if (has_next) {
      const int tn = x_start + nidx;
      tile_mn(tn, tiles_m, tiles_n, g.band, &mt_, &nt_);
      m0n = (int64_t)mt_ * BM;
      n0n = (int64_t)nt_ * BN;
      // the DEEP epilogues wait for their side loads with hipcc's own counted vmcnt: their prologue goes out unseen
      // (a run-time choice between the two forms would leave the builtin's pending DMA in hipcc's state at the join)
      start_tile(m0n, n0n, std::bool_constant<DEEP != 0>{});
    }
  }
  // (residual fold: R is already inside the accumulators; x = 0, gamma = 1 -> v = acc + b)
  const bool side = !SPLITK && epi_has_side<MODE, TC>(g.e) && !rfold;
  const int64_t ncol = n0 + wn * 64 + rc;
  float cs[W];                                     // DGELU: column sums of this lane's rows
#pragma unroll
  for (int i = 0; i < W; ++i) cs[i] = 0.f;
  // DEEP (bf16 outputs with a per-element side input: the bf16 residual stream, the saved gelu'): THREE strips of the side
  // input in flight, kept packed (4 registers per 16-B piece: 24 registers, fewer than the two unpacked strips of round 2).
  // One strip ahead gives a load the ~1.9 k cycles a strip takes to process; under the epilogue burst of all CUs an HBM
  // round trip is 3-5 k, so every strip waited (ISA: `vmcnt(2)` in front of each, ~2 k cycles each = the 15 k-cycle
  // DGELU epilogue).  Three ahead cover it.
  constexpr bool DEEP_T = DEEP != 0 && !SPLITK && W == 8 && (MODE == VITMI_EPI_RESIDUAL || MODE == VITMI_EPI_DGELU);
  bool deep_done = false;
  if constexpr (DEEP_T) {
    if (side) {                                      // kernel-uniform
      deep_done = true;
      const bf16* sbase = MODE == VITMI_EPI_RESIDUAL ? reinterpret_cast<const bf16*>(g.e.R) : reinterpret_cast<const bf16*>(g.e.AUX);
      const int64_t sld = MODE == VITMI_EPI_RESIDUAL ? g.e.ldr : g.e.ldaux;
      bf16x8 sp[3][NJ];
      auto side_load = [&](int strip, bf16x8 (&dst)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const bf16x8* p = reinterpret_cast<const bf16x8*>(sbase + row_off(m0 + wm * 128 + strip * 16 + j * RPI, rr, sld, ncol));
          dst[j] = g.e.side_nt ? __builtin_nontemporal_load(p) : *p;
        }
      };
      side_load(0, sp[0]);
      asm volatile("" ::: "memory");                 // keep the issue order: the first strip's data must be the oldest
      side_load(1, sp[1]);
      asm volatile("" ::: "memory");
      side_load(2, sp[2]);
      asm volatile("" ::: "memory");
      // PLAIN (DEEP == 2): the residual epilogue of every ViT / Swin block without DropPath (no second output, no row
      // scale, plain stores) and the gelu'-multiply with `nt` stores, as straight-line code.  The general row function
      // branches on those run-time options around loads and stores; at every join hipcc must assume the conditional
      // store / load may not have been issued and its counted waits turn into drains (ISA: vmcnt(0) in front of every row
      // of the residual strip; vmcnt(5) where nine operations could have stayed in flight in the gelu' one).
      auto deep_loop = [&](auto plain_tag) {
        constexpr bool PLAIN = decltype(plain_tag)::value;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            *reinterpret_cast<f32x4*>(strip_at(tr, lr, ni * 16 + lg * 4)) = acc[ni][mi];
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int row = j * RPI + rr;
            float v[W], x[W];
#pragma unroll
            for (int qq = 0; qq < W / 4; ++qq) {
              const f32x4 t4 = *reinterpret_cast<const f32x4*>(strip_at(tr, row, rc + 4 * qq));
              v[4 * qq] = t4[0]; v[4 * qq + 1] = t4[1]; v[4 * qq + 2] = t4[2]; v[4 * qq + 3] = t4[3];
            }
#pragma unroll
            for (int i = 0; i < W; ++i) x[i] = (float)sp[mi % 3][j][i];
            if constexpr (PLAIN && MODE == VITMI_EPI_RESIDUAL) {
#pragma unroll
              for (int i = 0; i < W; ++i) v[i] = x[i] + gamma_r[i] * (v[i] + bias_r[i]);      // the arithmetic of epi_row with rs = 1
              storev<TC, W>(reinterpret_cast<TC*>(g.e.C) + row_off(m0 + wm * 128 + mi * 16 + j * RPI, rr, g.e.ldc, ncol), v);
            } else if constexpr (PLAIN && MODE == VITMI_EPI_DGELU) {
#pragma unroll
              for (int i = 0; i < W; ++i) v[i] *= x[i];                                       // AUX = gelu'(pre), `nt` store
              bf16x8 o;
#pragma unroll
              for (int i = 0; i < W; ++i) o[i] = (bf16)v[i];
              __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(reinterpret_cast<TC*>(g.e.C) + row_off(m0 + wm * 128 + mi * 16 + j * RPI, rr, g.e.ldc, ncol)));
            } else {
              epi_row<MODE, TC, W>(g.e, m0 + wm * 128 + mi * 16 + j * RPI, rr, ncol, v, bias_r, gamma_r, x);
            }
            if constexpr (MODE == VITMI_EPI_DGELU) {
              const bool in_m = m0 + wm * 128 + mi * 16 + j * RPI + rr < pM;      // ragged M: the padding rows stay out of the column sums
#pragma unroll
              for (int i = 0; i < W; ++i) cs[i] += in_m ? v[i] : 0.f;
            }
          }
          if (mi + 3 < 8) side_load(mi + 3, sp[mi % 3]);
        }
      };
      deep_loop(std::bool_constant<DEEP == 2>{});
    }
  }
  float sx[2][NJ][W];                              // side inputs: this strip and the next
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < W; ++i) { sx[0][j][i] = 0.f; sx[1][j][i] = 0.f; }
  if (side && !deep_done) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) epi_side<MODE, TC, W>(g.e, m0 + wm * 128 + j * RPI, rr, ncol, sx[0][j]);
  }
  if (!deep_done) {
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    if (side && mi + 1 < 8) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        epi_side<MODE, TC, W>(g.e, m0 + wm * 128 + (mi + 1) * 16 + j * RPI, rr, ncol, sx[(mi + 1) & 1][j]);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      *reinterpret_cast<f32x4*>(strip_at(tr, lr, ni * 16 + lg * 4)) = acc[ni][mi];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = j * RPI + rr;
      float v[W];
#pragma unroll
      for (int qq = 0; qq < W / 4; ++qq) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(strip_at(tr, row, rc + 4 * qq));
        v[4 * qq] = t4[0]; v[4 * qq + 1] = t4[1]; v[4 * qq + 2] = t4[2]; v[4 * qq + 3] = t4[3];
      }
      if constexpr (SPLITK)
        storev<float, W>(ws + row_off((int64_t)split * pM + m0 + wm * 128 + mi * 16 + j * RPI, rr, pN, ncol), v);
      else if constexpr (MODE == VITMI_EPI_BIAS_GELU && DEEP == 2 && W == 8) {
        // the fc1 epilogue of the bf16 step as straight-line code: gelu and gelu' of acc + bias, both stored `nt`
        // (the general row function branches on the saved-derivative / second-output / store-policy options per store)
        bf16x8 oc, od;
#pragma unroll
        for (int i = 0; i < W; i += 2) {
          f32x2 r, d;
          gelu_both2(f32x2{v[i] + bias_r[i], v[i + 1] + bias_r[i + 1]}, &r, &d);
          oc[i] = (bf16)r[0]; oc[i + 1] = (bf16)r[1];
          od[i] = (bf16)d[0]; od[i + 1] = (bf16)d[1];
        }
        const int64_t mu = m0 + wm * 128 + mi * 16 + j * RPI;
        __builtin_nontemporal_store(od, reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(g.e.C2) + row_off(mu, rr, g.e.ldc2, ncol)));
        __builtin_nontemporal_store(oc, reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(g.e.C) + row_off(mu, rr, g.e.ldc, ncol)));
      } else if constexpr (MODE == VITMI_EPI_STORE && DEEP == 2 && W == 8) {
        // plain bf16 output (alpha = 1, nothing accumulated), `nt` store: qkv forward and the data gradients of the step
        bf16x8 oc;
#pragma unroll
        for (int i = 0; i < W; ++i) oc[i] = (bf16)(v[i] + bias_r[i]);
        __builtin_nontemporal_store(oc, reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(g.e.C) + row_off(m0 + wm * 128 + mi * 16 + j * RPI, rr, g.e.ldc, ncol)));
      } else {
        epi_row<MODE, TC, W>(g.e, m0 + wm * 128 + mi * 16 + j * RPI, rr, ncol, v, bias_r, gamma_r, sx[mi & 1][j]);
        if constexpr (MODE == VITMI_EPI_DGELU) {
          const bool in_m = m0 + wm * 128 + mi * 16 + j * RPI + rr < pM;
#pragma unroll
          for (int i = 0; i < W; ++i) cs[i] += in_m ? v[i] : 0.f;
        }
      }
    }
  }
  }
  if constexpr (SPLITK && FIX) {
    if (g.fix_cnt) {           // kernel-uniform
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // this wave's partial rows have left
      __syncthreads();
      unsigned* flag = reinterpret_cast<unsigned*>(smem + 48 * 1024);     // the stages are dead; the strips end at 32 KiB
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *flag = __hip_atomic_fetch_add(g.fix_cnt + (wg0 % ntiles), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      const unsigned ticket = *flag;
      const int nsplit = nwg / ntiles;
      if (ticket == (unsigned)(nsplit - 1)) {                             // workgroup-uniform: this slice arrived last
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        tile_fixup<MODE, TC>(ws, nsplit, g.e, pM, pN, m0, n0, tid);
      }
    }
  }
  if constexpr (MODE == VITMI_EPI_DGELU && !SPLITK) {
    if (g.e.colsum_part) {     // wave-uniform
      // lanes rr = 0..RPI-1 hold the same W columns: fold them, lane rr = 0 stores the
      // sums of this wave's 128 rows
#pragma unroll
      for (int i = 0; i < W; ++i) {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) cs[i] += __shfl_xor(cs[i], off, 64);
      }
      // (ragged M: the second half of the last row tile may lie wholly in the padding: there is no partial row for it)
      if (rr == 0 && m0 + wm * 128 < pM) storev<float, W>(g.e.colsum_part + ((m0 >> 7) + wm) * g.N + ncol, cs);
    }
  }
  if (dbg_tl && !prefetched) {
    wait_vm(0);                                    // stores retired = the wave could end here
    const unsigned long long tl2 = stamp();
    if (lane == 0) {
      g.dbg[64 + blockIdx.x * 8 + 0] = tl0; g.dbg[64 + blockIdx.x * 8 + 1] = tl1;
      g.dbg[64 + blockIdx.x * 8 + 2] = tl2;
      unsigned long long rt;                       // 100 MHz wall clock: relates cycles to time
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt) :: "memory");
      g.dbg[64 + blockIdx.x * 8 + 3] = rt;
    }
  }
  // ---- next tile
  if constexpr (!PERSIST) break;
  else {
    if (!has_next) break;
    idx += gx;
    m0 = m0n;
    n0 = n0n;
    prefetched = true;
  }
  }
  if (dbg_tl) {                                    // exit of the workgroup (all its tiles): cycles and wall clock
    wait_vm(0);
    const unsigned long long tlE = stamp();
    unsigned long long rtE;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rtE) :: "memory");
    if (lane == 0) { g.dbg[64 + blockIdx.x * 8 + 5] = tlE; g.dbg[64 + blockIdx.x * 8 + 6] = rtE; }
  }
}

// C = epilogue(sum_s ws[s]) for EPI_STORE, fp32 C; one thread = 4 columns
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splits, EpiArgs e, int64_t M,
                                     int64_t N) {
  const int64_t n4 = N / 4;
  const int64_t total = M * n4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / n4, n = (i % n4) * 4;
    f32x4 acc = *reinterpret_cast<const f32x4*>(ws + m * N + n);
    for (int s = 1; s < splits; ++s) acc += *reinterpret_cast<const f32x4*>(ws + ((int64_t)s * M + m) * N + n);
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    const float one[4] = {1.f, 1.f, 1.f, 1.f};
    if (e.bias) loadv<float, 4>(e.bias + n, b);
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (e.accumulate) epi_side<VITMI_EPI_STORE, float, 4>(e, m, n, x);
    epi_row<VITMI_EPI_STORE, float, 4>(e, m, n, v, b, one, x);
  }
}

// split-K plan for a problem with `tiles` output tiles and `nt` k-steps
static std::atomic<int> g_splitk_target{256};
inline void splitk_plan(int tiles, int nt, int* splits, int* ksps) {
  int s = 1;
  if (tiles <= 128 && nt >= 16) {
    s = g_splitk_target / tiles;
    if (s > nt / 8) s = nt / 8;
    if (s < 1) s = 1;
  }
  const int k = (nt + s - 1) / s;
  *ksps = k;
  *splits = (nt + k - 1) / k;
}

// Tail tiles: C = epilogue(sum of the k-slices).  16 blocks of 256 threads per 256x256 tile:
// block b of a tile covers rows 16b..16b+15, a thread W columns of one row (full lines).
template <int MODE, typename TC>
__global__ __launch_bounds__(256) void tail_epilogue_kernel(const float* __restrict__ ws, int splits, EpiArgs e,
                                                           int64_t M, int64_t N, int tiles_n, int tile0, int band) {
  constexpr int W = sizeof(TC) == 2 ? 8 : 4;
  constexpr int TPR = 256 / W;                       // threads per 256-column row
  constexpr int RPB = 256 / TPR;                     // rows per pass of the block
  const int tile = tile0 + blockIdx.x / 16;
  int mt_, nt_;
  tile_mn(tile, (int)(M / BM), tiles_n, band, &mt_, &nt_);
  const int64_t m0 = (int64_t)mt_ * BM + (blockIdx.x % 16) * 16;
  const int64_t n = (int64_t)nt_ * BN + (threadIdx.x % TPR) * W;
  float b[W], gm[W];
#pragma unroll
  for (int i = 0; i < W; ++i) { b[i] = 0.f; gm[i] = 1.f; }
  if (e.bias) loadv<float, W>(e.bias + n, b);
  if (MODE == VITMI_EPI_RESIDUAL && e.gamma) loadv<float, W>(e.gamma + n, gm);
  for (int r = threadIdx.x / TPR; r < 16; r += RPB) {
    const int64_t m = m0 + r;
    float v[W];
    loadv<float, W>(ws + m * N + n, v);
    for (int s = 1; s < splits; ++s) {
      float t[W];
      loadv<float, W>(ws + ((int64_t)s * M + m) * N + n, t);
#pragma unroll
      for (int i = 0; i < W; ++i) v[i] += t[i];
    }
    float x[W];
#pragma unroll
    for (int i = 0; i < W; ++i) x[i] = 0.f;
    if (epi_has_side<MODE, TC>(e)) epi_side<MODE, TC, W>(e, m, n, x);
    epi_row<MODE, TC, W>(e, m, n, v, b, gm, x);
  }
}

static std::atomic<int> g_tail_override{-1};
// diagnostic / test hook: 0 = never split the tail, 1 = whenever the shape allows, -1 = default heuristic
extern "C" void vitmi_debug_gemm_tail(int mode) { g_tail_override = mode; }
constexpr int TAIL_SPLITS = 3;
// tail_plan: is the remainder round worth slicing?  Needs a workspace of TAIL_SPLITS full
// M x N fp32 slabs (only the tail tiles' part is touched), no fused column sums, and a
// contraction long enough that a third of it still amortises the partial-tile traffic.
static bool tail_plan_shape(const GemmArgs& g, int nwg, int* rem, int* splits, int* ksps, bool forced) {
  const int cus = vitmi_cu_count();
  const int nt = (int)(g.K / BK);
  const int r = nwg % cus;
  if (nwg <= cus || r == 0 || r * TAIL_SPLITS > cus) return false;
  if (!forced && nt < 36) return false;
  if (nt < 2 * TAIL_SPLITS) return false;
  *rem = r;
  *splits = TAIL_SPLITS;
  *ksps = (nt + TAIL_SPLITS - 1) / TAIL_SPLITS;
  if ((*splits - 1) * *ksps >= nt) return false;      // every slice must be non-empty
  return true;
}
// workspace of a split tail: the slices' partial slabs + one arrival counter per tile (rounded up to 4 KiB)
static size_t tail_ws_bytes(const GemmArgs& g) { return (size_t)TAIL_SPLITS * g.M * g.N * sizeof(float) + 4096; }
static bool tail_plan(const GemmArgs& g, int nwg, int* rem, int* splits, int* ksps) {
  if (g_tail_override == 0 || g.e.colsum_part || g.e.accumulate || (g.M % BM) != 0) return false;   // (the slices' workspace is M x N)
  if (!tail_plan_shape(g, nwg, rem, splits, ksps, g_tail_override == 1)) return false;
  return g.ws != nullptr && g.ws_bytes >= tail_ws_bytes(g) && *rem <= NTHREADS;
}
// 0 (default) = row-wise finisher kernel after the slices (all 256 CUs sum the 79 tiles' slabs: 60 MB in 12-15 us);
// 1 = the last slice of a tile to arrive applies the epilogue itself (FIX).  The in-kernel form was asked for by VERDICT r02
// (item 1a), is built and tested (bit-identical results), and measured SLOWER inside the ViT-B/16 step, one box, interleaved:
// 34.85 / 34.89 ms against 34.02 / 34.14 with the finisher.  Why: a tail tile's three slabs are 768 KB, which ONE workgroup
// then reads back at the 40-65 GB/s a single CU gets for freshly written lines (12-19 us, on 79 of the 256 CUs), behind an
// agent-scope release that writes back 256 KB of dirty lines per slice — the guide's own sizing rule (in-launch combine
// pays for a few tens of KB per tile, not hundreds).  The separate finisher is 1 264 blocks over the whole chip.
static std::atomic<int> g_tail_fixup{0};
extern "C" void vitmi_debug_gemm_tail_fixup(int on) { g_tail_fixup = on != 0; }

static std::atomic<int> g_rfold_override{-1};
// diagnostic / test hook: 0 = epilogue reads the residual (round-1 behaviour), 1 / -1 = fold it in
extern "C" void vitmi_debug_gemm_rfold(int mode) { g_rfold_override = mode; }

// diagnostic hook / default of the start stagger: permille of an estimated tile time (0 = off) and phases
static std::atomic<int> g_stagger_permille{700}, g_stagger_phases{4};     // swept inside the ViT-B/16 step (tools/sweep_bench.sh): 34.53 -> 33.98 ms
extern "C" void vitmi_debug_gemm_stagger(int permille) { g_stagger_permille = permille; }
extern "C" void vitmi_debug_gemm_stagger_phases(int n) { g_stagger_phases = n > 0 ? n : 1; }
static std::atomic<int> g_pipe_override{-1};
// diagnostic / test hook: force the main-loop variant (0, 1, 2) or -1 = automatic
extern "C" void vitmi_debug_gemm_pipe(int mode) { g_pipe_override = mode; }

static int pipe_mode() {   // -1 = automatic (per layout), else forced 0/1/2
  if (g_pipe_override >= 0) return g_pipe_override;
  static int mode = -2;
  if (mode == -2) {
    const char* e = getenv("VITMI_GEMM_PIPE");
    mode = (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : -1;
  }
  return mode;
}

template <bool A_KM, bool B_KM, int MODE, typename TC, int PIPE>
int launch_p(const GemmArgs& g, hipStream_t stream);

template <bool A_KM, bool B_KM, int MODE, typename TC>
int launch(const GemmArgs& g, hipStream_t stream) {
  int pm = pipe_mode();
  constexpr bool CAN_FOLD = MODE == VITMI_EPI_RESIDUAL && sizeof(TC) == 4;
  if (pm < 0) {
    pm = (A_KM && B_KM) ? 2 : 1;   // k-major operands want full-line DMA (PIPE 2)
    // the plain-store forward with a short contraction (qkv: K = 768) runs best on the simple two-stage
    // loop: 189 -> 175 us inside the ViT-B/16 step (A/B of the three loops in one process); with an
    // epilogue that reads or computes (GELU, residual) or K = 3072 the phased loops win
    // (round 3: with the k-half unrolled and the DMAs in the SGPR-base form the phased loop wins here too —
    // 160-166 us against 169-171 for the qkv forward, one process — so nothing selects the two-stage loop by default)
    // the fp32-stream residual epilogue takes the ring-of-three loop that streams R through LDS
    if (CAN_FOLD && g_rfold_override != 0 && !g.e.gamma && !g.e.rowscale && !g.e.C2 && !g.e.r_bf16 && g.K >= 640) pm = 3;
  }
  if (pm == 0) return launch_p<A_KM, B_KM, MODE, TC, 0>(g, stream);
  if (pm == 1) return launch_p<A_KM, B_KM, MODE, TC, 1>(g, stream);
  if constexpr (CAN_FOLD) { if (pm == 3) return launch_p<A_KM, B_KM, MODE, TC, 3>(g, stream); }
  else if (pm == 3) pm = 1;        // PIPE 3 exists only where the fold does: everything else keeps the ring of four
  if (pm == 1) return launch_p<A_KM, B_KM, MODE, TC, 1>(g, stream);
  return launch_p<A_KM, B_KM, MODE, TC, 2>(g, stream);
}

// split-K: two stages (strips alias them after the loop); persistent launches: stages + 32 KiB of
// strips above them = the CU's 160 KiB (PIPE 3: ring of three + residual strips, the same size)
template <int MODE, typename TC, bool SPLITK, int PIPE>
constexpr int lds_bytes() {
  return SPLITK ? 2 * STAGE_BYTES : RF_LDS;
}
// launch_flags & VITMI_LAUNCH_SHARED_DEVICE: one tile per workgroup (grid = tiles); else the persistent grid
static int persistent_grid(int nwg, int launch_flags) {
  if (!vitmi_persist_on(launch_flags)) return nwg;
  int cus = vitmi_cu_count();
  cus -= cus % 8;                                  // whole XCD rows: every XCD gets the same number of workgroups
  if (cus < 8) cus = 8;
  return nwg < cus ? nwg : cus;
}

// Column-band width of the tile order (tile_mn): bands only where the whole B matrix does not fit an
// XCD's L2 beside the streaming operands (N x K x 2 B > 3 MiB) and there are rows enough to fill the
// XCDs inside a band; the band is the widest divisor-free choice whose B panels take <= 2 MiB.
static std::atomic<int> g_band_override{-1};     // diagnostic hook: -1 = automatic, 0 = row-major, n = bands of n column tiles
extern "C" void vitmi_debug_gemm_band(int n) { g_band_override = n; }
static std::atomic<int> g_band_min_kb{3072};     // diagnostic hook: B matrices up to this size keep the row-major order
extern "C" void vitmi_debug_gemm_band_kb(int kb) { g_band_min_kb = kb > 0 ? kb : 3072; }
static int band_env_kb() {      // VITMI_GEMM_BAND_KB: the same threshold from the environment (PMC passes run one process per shape)
  static int kb = -2;
  if (kb == -2) {
    const char* e = getenv("VITMI_GEMM_BAND_KB");
    kb = e ? atoi(e) : -1;
  }
  return kb;
}
static int band_for(const GemmArgs& g, int tiles_m, int tiles_n) {
  if (g_band_override >= 0) return g_band_override;
  const int64_t b_bytes = g.N * g.K * 2;
  const int64_t min_kb = band_env_kb() > 0 ? band_env_kb() : (int64_t)g_band_min_kb;
  if (b_bytes <= (min_kb << 10) || tiles_n < 2 || tiles_m < 64) return 0;
  const int64_t panel = (int64_t)BN * g.K * 2;
  int band = (int)((2 << 20) / panel);
  if (band < 1) band = 1;
  if (band >= tiles_n) return 0;
  for (int d = band; 2 * d >= band && d >= 1; --d)      // a divisor of tiles_n close below, if there is one
    if (tiles_n % d == 0) { band = d; break; }
  // distinct operand panels an XCD's round of 32 tiles pulls from beyond L2: 32/band rows of A with
  // the band's B resident, against 32/tiles_n rows + all tiles_n columns in row-major order
  if (32.0 / band >= 32.0 / tiles_n + tiles_n) return 0;
  return band;
}

template <bool A_KM, bool B_KM, int MODE, typename TC, int PIPE>
int launch_p(const GemmArgs& g_in, hipStream_t stream) {
  GemmArgs g = g_in;
  g.rfold = g_rfold_override == 0 ? 0 : 1;
  const int tiles_m = (int)((g.M + BM - 1) / BM), tiles_n = (int)(g.N / BN);      // ragged M: see ragged_m_ok
  const int nwg = tiles_m * tiles_n;
  g.band = band_for(g, tiles_m, tiles_n);
  {   // tile time estimate in shader cycles: ~3.6 k per 64-deep k-step + an epilogue
    const int64_t est = (g.K / BK) * 3600 + 8000;
    g.stag_cycles = (int)(est * g_stagger_permille / 1000);
    g.stag_phases = g_stagger_phases;
  }
  if constexpr (MODE == VITMI_EPI_STORE && sizeof(TC) == 4) {
    int splits, ksps;
    splitk_plan(nwg, (int)(g.K / BK), &splits, &ksps);
    // (a ragged last row tile stores WHOLE 256-row tiles: with the slices strided by M rows its padding rows would land on
    // the first rows of the next slice and, for the last slice, beyond the workspace — ragged M never splits K)
    if (splits > 1 && (g.M % BM) == 0 && g.ws && g.ws_bytes >= (size_t)splits * g.M * g.N * sizeof(float)) {
      auto kern = gemm_fast_kernel<A_KM, B_KM, MODE, TC, true, PIPE>;
      if (int rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(kern), 2 * STAGE_BYTES, "gemm_fast(split-K)")) return rc;
      float* ws = reinterpret_cast<float*>(g.ws);
      hipLaunchKernelGGL(kern, dim3(nwg * splits), dim3(NTHREADS), 2 * STAGE_BYTES, stream, g, tiles_n, nwg * splits, nwg, ksps, ws, 0);
      int rc = vitmi_check_launch("gemm_fast_kernel(split-K)");
      if (rc) return rc;
      const int64_t work = g.M * g.N / 4;
      int64_t blocks = (work + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ws, splits, g.e, g.M, g.N);
      return vitmi_check_launch("splitk_reduce_kernel");
    }
  }
  // ---- split tail: a launch of q full rounds plus a short remainder leaves most CUs idle
  // for a whole tile time.  The remainder tiles are instead contracted in TAIL_SPLITS k-slices
  // (one short round on all CUs, raw fp32 partial tiles to the workspace) and finished by a
  // row-wise kernel that sums the slices in a fixed order and applies the epilogue.
  if constexpr (MODE == VITMI_EPI_STORE || MODE == VITMI_EPI_RESIDUAL) {
    int rem = 0, splits = 0, ksps = 0;
    if (tail_plan(g, nwg, &rem, &splits, &ksps)) {
      const int full = nwg - rem;
      constexpr bool CAN_DEEP = sizeof(TC) == 2 && (MODE == VITMI_EPI_RESIDUAL || MODE == VITMI_EPI_DGELU);
      auto kmain = gemm_fast_kernel<A_KM, B_KM, MODE, TC, false, PIPE>;
      if constexpr (MODE == VITMI_EPI_STORE && sizeof(TC) == 2) {
        if (g.side_depth >= 3 && g.e.alpha == 1.f && !g.e.accumulate && g.e.c_policy == 2) kmain = gemm_fast_kernel<A_KM, B_KM, MODE, TC, false, PIPE, false, 2>;
      }
      if constexpr (CAN_DEEP) {
        if (g.side_depth >= 3) {
          kmain = gemm_fast_kernel<A_KM, B_KM, MODE, TC, false, PIPE, false, 1>;
          if constexpr (MODE == VITMI_EPI_RESIDUAL) {
            if (!g.e.C2 && !g.e.rowscale && g.e.c_policy == 0) kmain = gemm_fast_kernel<A_KM, B_KM, MODE, TC, false, PIPE, false, 2>;
          } else {
            if (g.e.aux_deriv && g.e.c_policy == 2) kmain = gemm_fast_kernel<A_KM, B_KM, MODE, TC, false, PIPE, false, 2>;
          }
        }
      }
      constexpr int LDSM = lds_bytes<MODE, TC, false, PIPE>();
      if (int rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(kmain), LDSM, "gemm_fast")) return rc;
      float* ws = reinterpret_cast<float*>(g.ws);
      unsigned* cnt = reinterpret_cast<unsigned*>(ws + (size_t)TAIL_SPLITS * g.M * g.N);
      if (g_tail_fixup) {
        // the last slice of a tile to arrive applies the epilogue itself (FIX): two launches instead of three
        auto ktail = gemm_fast_kernel<A_KM, B_KM, MODE, TC, true, PIPE, true>;
        if (int rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(ktail), 2 * STAGE_BYTES, "gemm_fast(tail)")) return rc;
        GemmArgs gm = g, gt = g;
        gm.zero_cnt = cnt; gm.zero_n = rem;
        gt.fix_cnt = cnt;
        hipLaunchKernelGGL(kmain, dim3(persistent_grid(full, g.launch_flags)), dim3(NTHREADS), LDSM, stream, gm, tiles_n, full, full, 0, (float*)nullptr, 0);
        int rc = vitmi_check_launch("gemm_fast_kernel(full rounds)");
        if (rc) return rc;
        hipLaunchKernelGGL(ktail, dim3(rem * splits), dim3(NTHREADS), 2 * STAGE_BYTES, stream, gt, tiles_n, rem * splits, rem, ksps, ws, full);
        return vitmi_check_launch("gemm_fast_kernel(tail slices + fix-up)");
      }
      auto ktail = gemm_fast_kernel<A_KM, B_KM, VITMI_EPI_STORE, float, true, PIPE>;
      if (int rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(ktail), 2 * STAGE_BYTES, "gemm_fast(tail)")) return rc;
      hipLaunchKernelGGL(kmain, dim3(persistent_grid(full, g.launch_flags)), dim3(NTHREADS), LDSM, stream, g, tiles_n, full, full, 0, (float*)nullptr, 0);
      int rc = vitmi_check_launch("gemm_fast_kernel(full rounds)");
      if (rc) return rc;
      hipLaunchKernelGGL(ktail, dim3(rem * splits), dim3(NTHREADS), 2 * STAGE_BYTES, stream, g, tiles_n, rem * splits, rem, ksps, ws, full);
      if ((rc = vitmi_check_launch("gemm_fast_kernel(tail slices)"))) return rc;
      hipLaunchKernelGGL((tail_epilogue_kernel<MODE, TC>), dim3(rem * 16), dim3(256), 0, stream, ws, splits, g.e, g.M, g.N, tiles_n, full, g.band);
      return vitmi_check_launch("tail_epilogue_kernel");
    }
  }
  constexpr bool CAN_DEEP2 = sizeof(TC) == 2 && (MODE == VITMI_EPI_RESIDUAL || MODE == VITMI_EPI_DGELU);
  auto kern = gemm_fast_kernel<A_KM, B_KM, MODE, TC, false, PIPE>;
  if constexpr (MODE == VITMI_EPI_BIAS_GELU && sizeof(TC) == 2) {
    if (g.side_depth >= 3 && g.e.aux_deriv && g.e.C2 && g.e.c_policy == 2) kern = gemm_fast_kernel<A_KM, B_KM, MODE, TC, false, PIPE, false, 2>;
  }
  if constexpr (MODE == VITMI_EPI_STORE && sizeof(TC) == 2) {
    if (g.side_depth >= 3 && g.e.alpha == 1.f && !g.e.accumulate && g.e.c_policy == 2) kern = gemm_fast_kernel<A_KM, B_KM, MODE, TC, false, PIPE, false, 2>;
  }
  if constexpr (CAN_DEEP2) {
    if (g.side_depth >= 3) {
      kern = gemm_fast_kernel<A_KM, B_KM, MODE, TC, false, PIPE, false, 1>;
      if constexpr (MODE == VITMI_EPI_RESIDUAL) {
        if (!g.e.C2 && !g.e.rowscale && g.e.c_policy == 0) kern = gemm_fast_kernel<A_KM, B_KM, MODE, TC, false, PIPE, false, 2>;
      } else {
        if (g.e.aux_deriv && g.e.c_policy == 2) kern = gemm_fast_kernel<A_KM, B_KM, MODE, TC, false, PIPE, false, 2>;
      }
    }
  }
  constexpr int LDSK = lds_bytes<MODE, TC, false, PIPE>();
  if (int rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(kern), LDSK, "gemm_fast")) return rc;
  hipLaunchKernelGGL(kern, dim3(persistent_grid(nwg, g.launch_flags)), dim3(NTHREADS), LDSK, stream, g, tiles_n, nwg, nwg, 0, (float*)nullptr, 0);
  return vitmi_check_launch("gemm_fast_kernel");
}

enum { OUT_BF16 = 0, OUT_F32 = 1 };

}  // namespace

bool gemm_fast2_shape_ok(const GemmArgs& g);
size_t gemm_fast2_workspace(const GemmArgs& g);
int gemm_fast2_launch(const GemmArgs& g, hipStream_t s);

// diagnostic hook: workgroups a split-K launch of the 256x256 kernel aims at (default 256)
extern "C" void vitmi_debug_gemm_splitk_target(int n) { g_splitk_target = n > 0 ? n : 256; }
static std::atomic<int> g_tile_override{-1};
// diagnostic / test hook: 1 = 256x256 tiles (one 8-wave workgroup per CU),
// 2 = 256x128 tiles (two 4-wave workgroups per CU), -1 = default
extern "C" void vitmi_debug_gemm_tile(int mode) { g_tile_override = mode; }
static int tile_mode() {
  if (g_tile_override > 0) return g_tile_override;
  static int mode = -2;
  if (mode == -2) {
    const char* e = getenv("VITMI_GEMM_TILE");
    mode = (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 1;
  }
  return mode;
}
// 256x128 tiles (ragged-shape capable) are used when forced, or whenever the shape is not
// a whole number of 256x256x64 tiles (D = 96..384 models, odd batch sizes); measured
// equal-or-slower than 256x256 on the ViT-B shapes
// Ragged M on the 256x256 kernel (round 4): with VITMI_LAUNCH_ROWS_PADDED the caller vouches that C, C2, R and AUX are
// allocated up to the next multiple of 256 rows; a k-major A is then staged with its last row repeated (the per-lane row
// offsets are clamped once per tile) and the last row tile writes its surplus rows into that padding.  Every batch size of
// a ViT (M = 197 B is a multiple of 256 only for B % 256 == 0) and dino_vitb8 at 96x96 (M = 145 B) then run on the tile
// kernel the benchmark runs on, instead of the 7-30 % slower 256x128 form.
static bool ragged_m_ok(const GemmArgs& g) {
  return (g.launch_flags & VITMI_LAUNCH_ROWS_PADDED) != 0 && g.a_km && (g.M % BM) != 0 && g.M > BM && (g.N % BN) == 0 &&
         (g.K % BK) == 0 && g.batch == 1 && !g.e.accumulate && !g.e.rowscale;      // (rowscale is indexed by row: none for the padding)
}
static bool use_tile2(const GemmArgs& g) {
  if (!gemm_fast2_shape_ok(g)) return false;
  if (tile_mode() != 2 && ragged_m_ok(g)) return false;
  return tile_mode() == 2 || (g.M % BM) != 0 || (g.N % BN) != 0 || (g.K % BK) != 0;
}

// which (layout, epilogue, output dtype) combinations are instantiated
static bool combo_built(const GemmArgs& g) {
  const EpiArgs& e = g.e;
  const bool nt = g.a_km && g.b_km, nn = g.a_km && !g.b_km, tn = !g.a_km && !g.b_km;
  switch (e.mode) {
    case VITMI_EPI_STORE: return (nt || nn || tn);
    case VITMI_EPI_BIAS_GELU: return nt && e.c_bf16;
    case VITMI_EPI_RESIDUAL: return nt;
    case VITMI_EPI_DGELU: return nn && e.c_bf16;
    case VITMI_EPI_PATCH_POS: return nt;
  }
  return false;
}

bool gemm_fast_supported(const GemmArgs& g, int in_bf16) {
  if (!in_bf16) return false;
  if (!use_tile2(g) && ((g.M % BM && !ragged_m_ok(g)) || g.N % BN || g.K % BK)) return false;
  if (g.e.colsum_part && (g.e.mode != VITMI_EPI_DGELU || !is_aligned(g.e.colsum_part, 16))) return false;
  if ((g.M + BM - 1) / BM * (g.N / BN) > (1 << 30)) return false;
  if (!combo_built(g)) return false;
  const EpiArgs& e = g.e;
  if (g.lda % 8 || g.ldb % 8 || !is_aligned(g.A, 16) || !is_aligned(g.B, 16)) return false;
  const size_t cb = e.c_bf16 ? 8 : 16;
  if (e.ldc % 4 || !is_aligned(e.C, cb)) return false;
  if (e.C2 && (e.ldc2 % 8 || !is_aligned(e.C2, 16))) return false;
  if (e.bias && !is_aligned(e.bias, 16)) return false;
  if (e.gamma && !is_aligned(e.gamma, 16)) return false;
  if (e.mode == VITMI_EPI_RESIDUAL && (e.ldr % 4 || !is_aligned(e.R, cb))) return false;
  if (e.mode == VITMI_EPI_DGELU && (e.ldaux % 4 || !is_aligned(e.AUX, 8))) return false;
  if (e.mode == VITMI_EPI_PATCH_POS && (!is_aligned(e.pos, 16) || (e.cls && !is_aligned(e.cls, 16)))) return false;
  if (e.mode == VITMI_EPI_STORE && e.accumulate && e.c_bf16) return false;
  return true;
}

size_t gemm_fast_workspace(const GemmArgs& g) {
  if (use_tile2(g)) return gemm_fast2_workspace(g);
  const int tiles = (int)((g.M + BM - 1) / BM * (g.N / BN));
  size_t need = 0;
  if (g.e.mode == VITMI_EPI_STORE && !g.e.c_bf16 && (g.M % BM) == 0) {
    int splits, ksps;
    splitk_plan(tiles, (int)(g.K / BK), &splits, &ksps);
    if (splits > 1) need = (size_t)splits * g.M * g.N * sizeof(float);
  }
  if (need == 0 && (g.e.mode == VITMI_EPI_STORE || g.e.mode == VITMI_EPI_RESIDUAL) && g_tail_override != 0 &&
      !g.e.colsum_part && !g.e.accumulate && (g.M % BM) == 0) {
    int rem, splits, ksps;
    if (tail_plan_shape(g, tiles, &rem, &splits, &ksps, g_tail_override == 1)) need = tail_ws_bytes(g);
  }
  return need;
}

static int gemm_fast_launch_whole(const GemmArgs& g, hipStream_t s) {
  const EpiArgs& e = g.e;
  const bool nt = g.a_km && g.b_km, nn = g.a_km && !g.b_km;
#define GO(AKM, BKM, MODE) (e.c_bf16 ? launch<AKM, BKM, MODE, bf16>(g, s) : launch<AKM, BKM, MODE, float>(g, s))
  switch (e.mode) {
    case VITMI_EPI_STORE:
      if (nt) return GO(true, true, VITMI_EPI_STORE);
      if (nn) return GO(true, false, VITMI_EPI_STORE);
      return GO(false, false, VITMI_EPI_STORE);
    case VITMI_EPI_BIAS_GELU: return launch<true, true, VITMI_EPI_BIAS_GELU, bf16>(g, s);
    case VITMI_EPI_RESIDUAL: return GO(true, true, VITMI_EPI_RESIDUAL);
    case VITMI_EPI_DGELU: return launch<true, false, VITMI_EPI_DGELU, bf16>(g, s);
    case VITMI_EPI_PATCH_POS: return GO(true, true, VITMI_EPI_PATCH_POS);
  }
#undef GO
  return vitmi_fail(VITMI_E_SHAPE, "gemm_fast: combination not built");
}

// ---- paired split-K launch (vitmi_gemm_pair): two weight-gradient products dW = dY^T X with the same K (all
// tokens) and layouts share ONE grid.  Why: the 768 x 768 proj gradient has 9 output tiles, so alone it is cut
// into 28 k-slices of 29 k-steps (each workgroup pays a pipeline fill and a 256-KB partial tile for 29 steps,
// 66 MB of partials in all: 789 TFLOP/s against 1 085-1 150 on the wider gradients); beside the 27 tiles of the
// qkv gradient the 36 tiles take 7 slices of 113 steps each and a quarter of the partial traffic.
static bool pair_ok(const GemmArgs& a, const GemmArgs& b) {
  auto plain = [](const GemmArgs& g) {
    return !g.a_km && !g.b_km && g.e.mode == VITMI_EPI_STORE && !g.e.c_bf16 && !g.e.bias && !g.e.accumulate &&
           g.e.alpha == 1.f && g.batch == 1 && (g.M % BM) == 0 && (g.N % BN) == 0 && (g.K % BK) == 0 &&
           // nothing the paired launch would silently drop (its main kernel stores raw partial tiles, its reduce applies
           // the plain store): descriptors that carry any of these take the two-call fallback, which honours or rejects them
           !g.e.colsum_part && !g.e.C2 && !g.e.rowscale && !g.e.gamma && !g.e.R && !g.e.AUX;
  };
  return plain(a) && plain(b) && a.K == b.K && a.launch_flags == b.launch_flags;
}
size_t gemm_fast_pair_workspace(const GemmArgs& a, const GemmArgs& b) {
  if (!pair_ok(a, b)) return 0;
  const int tiles = (int)(a.M / BM * (a.N / BN) + b.M / BM * (b.N / BN));
  int splits, ksps;
  splitk_plan(tiles, (int)(a.K / BK), &splits, &ksps);
  if (splits <= 1) return 0;
  return (size_t)splits * (a.M * a.N + b.M * b.N) * sizeof(float);
}
// returns -1000 when the pair cannot share a launch (the caller then issues the two products one after the other)
int gemm_fast_pair_launch(const GemmArgs& a_in, const GemmArgs& b, void* ws_, size_t ws_bytes, hipStream_t stream) {
  const size_t need = gemm_fast_pair_workspace(a_in, b);
  if (need == 0 || !ws_ || ws_bytes < need || !is_aligned(ws_, 16) || pipe_mode() == 0 || pipe_mode() == 2) return -1000;
  GemmArgs g = a_in;
  const int t1 = (int)(g.M / BM * (g.N / BN)), t2 = (int)(b.M / BM * (b.N / BN));
  int splits, ksps;
  splitk_plan(t1 + t2, (int)(g.K / BK), &splits, &ksps);
  float* ws = reinterpret_cast<float*>(ws_);
  g.A2 = b.A; g.B2 = b.B; g.lda2 = b.lda; g.ldb2 = b.ldb; g.M2 = b.M; g.N2 = b.N;
  g.tiles1 = t1; g.tiles_n2 = (int)(b.N / BN);
  g.ws2 = ws + (size_t)splits * g.M * g.N;
  g.band = 0; g.stag_cycles = 0; g.stag_phases = 1; g.rfold = 0;
  auto kern = gemm_fast_kernel<false, false, VITMI_EPI_STORE, float, true, 1>;
  if (int rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(kern), 2 * STAGE_BYTES, "gemm_fast(pair)")) return rc;
  const int nwg = (t1 + t2) * splits;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), 2 * STAGE_BYTES, stream, g, (int)(g.N / BN), nwg, t1 + t2, ksps, ws, 0);
  if (int rc = vitmi_check_launch("gemm_fast_kernel(pair, split-K)")) return rc;
  for (int i = 0; i < 2; ++i) {
    const GemmArgs& p = i == 0 ? a_in : b;
    const int64_t work = p.M * p.N / 4;
    int64_t blocks = (work + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, i == 0 ? ws : g.ws2, splits, p.e, p.M, p.N);
    if (int rc = vitmi_check_launch("splitk_reduce_kernel(pair)")) return rc;
  }
  return 0;
}

int gemm_fast_launch(const GemmArgs& g, hipStream_t s) {
  if (use_tile2(g)) return gemm_fast2_launch(g, s);
  return gemm_fast_launch_whole(g, s);
}

// every diagnostic switch of this file back to its default (vitmi_debug_reset, core.cpp)
void vitmi_debug_reset_gemm_fast() {
  g_splitk_target = 256;
  g_tail_override = -1;
  g_tail_fixup = 0;
  g_rfold_override = -1;
  g_stagger_permille = 700;
  g_stagger_phases = 4;
  g_pipe_override = -1;
  g_band_override = -1;
  g_band_min_kb = 3072;
  g_tile_override = -1;
}
