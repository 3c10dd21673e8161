// The "bf16x3" parity mode (round 5): fp32 operands as sums of two bf16 numbers, three bf16 MFMA products per fp32 product.
//
// The reference computes in fp32 throughout (/root/reference/main.py:244, utils_network.py:120); bf16 operands cannot meet
// north_star's 1e-3 on the logits (2^-9 per operand), and the fp32-MFMA GEMM that can runs at 12 TFLOP/s.  Split
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits kept, error 2^-17 |x|): then
//     a * b  =  a_hi b_hi + a_lo b_hi + a_hi b_lo  +  O(2^-16 |a b|)
// and the three products are ONE bf16 GEMM over a contraction three times as long, on the tile kernel the benchmark
// runs on, with its fp32 accumulators and its fp32 epilogues unchanged:
//     A3 = [ A_hi | A_lo | A_hi ]   (the "A pattern")        B3 = [ B_hi | B_hi | B_lo ]   (the "B pattern")
//     A3 B3^T = A_hi B_hi^T + A_lo B_hi^T + A_hi B_lo^T.
// split3_kernel writes those images from an fp32 matrix in one pass (4 B in, 6 B out per element): side by side along
// the row for a k-major operand (k runs along the row), stacked row blocks for a k-minor one (k is the row index).
// gelu_fwd / gelu_bwd are the fp32 element-wise halves of the two GELU epilogues, which the tile kernel only builds
// for bf16 outputs.
#include "common.h"

namespace {

constexpr int BLOCK = 256;
inline unsigned grid_for(int64_t items) {
  int64_t b = (items + BLOCK - 1) / BLOCK;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// one thread = W (4 or 8) consecutive columns of one row; grid-stride over (row, chunk) with the index advanced by
// carry instead of a 64-bit i / cols per element (that division held the first version to 3 TB/s)
template <bool B_PATTERN, bool STACKED, int W>
__global__ __launch_bounds__(BLOCK) void split3_kernel(const float* __restrict__ x, int64_t ldx, bf16* __restrict__ out,
                                                       int64_t ldo, int64_t R, int64_t Cn) {
  typedef bf16 bfv __attribute__((ext_vector_type(W)));
  const int64_t cw = Cn / W, stride = (int64_t)gridDim.x * BLOCK;
  const int64_t i0 = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  int64_t r = i0 / cw, ch = i0 % cw;
  const int64_t dr = stride / cw, dch = stride % cw;
  for (; r < R; r += dr) {
    const int64_t c = ch * W;
    bfv hi, lo;
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[4 * q + e] = (bf16)v[e];
        lo[4 * q + e] = (bf16)(v[e] - (float)hi[4 * q + e]);
      }
    }
    const bfv p1 = B_PATTERN ? hi : lo, p2 = B_PATTERN ? lo : hi;      // A: hi lo hi   B: hi hi lo
    if (STACKED) {
      *reinterpret_cast<bfv*>(out + r * ldo + c) = hi;
      *reinterpret_cast<bfv*>(out + (R + r) * ldo + c) = p1;
      *reinterpret_cast<bfv*>(out + (2 * R + r) * ldo + c) = p2;
    } else {
      *reinterpret_cast<bfv*>(out + r * ldo + c) = hi;
      *reinterpret_cast<bfv*>(out + r * ldo + Cn + c) = p1;
      *reinterpret_cast<bfv*>(out + r * ldo + 2 * Cn + c) = p2;
    }
    ch += dch;
    if (ch >= cw) { ch -= cw; ++r; }
  }
}

__global__ __launch_bounds__(BLOCK) void gelu_fwd_kernel(const float* __restrict__ pre, int64_t ldp, float* __restrict__ out,
                                                         int64_t ldo, int64_t M, int64_t N) {
  const int64_t n4 = N / 4, total = M * n4;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t m = i / n4, n = (i % n4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(pre + m * ldp + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    *reinterpret_cast<f32x4*>(out + m * ldo + n) = v;
  }
}

__global__ __launch_bounds__(BLOCK) void gelu_bwd_kernel(const float* __restrict__ dh, int64_t ldd, const float* __restrict__ pre,
                                                         int64_t ldp, float* __restrict__ out, int64_t ldo, int64_t M,
                                                         int64_t N) {
  const int64_t n4 = N / 4, total = M * n4;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t m = i / n4, n = (i % n4) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dh + m * ldd + n);
    const f32x4 p = *reinterpret_cast<const f32x4*>(pre + m * ldp + n);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = g[e] * dgelu_erf(p[e]);
    *reinterpret_cast<f32x4*>(out + m * ldo + n) = v;
  }
}

}  // namespace

extern "C" int vitmi_split3(const float* x, int64_t ldx, int64_t rows, int64_t cols, void* out, int64_t ldo,
                            int b_pattern, int stacked, void* stream_) {
  VITMI_REQUIRE(x && out && rows > 0 && cols > 0, VITMI_E_BADARG, "split3: null pointer or empty shape");
  VITMI_REQUIRE(cols % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldx >= cols && ldo >= (stacked ? cols : 3 * cols),
                VITMI_E_SHAPE, "split3: cols / strides must be multiples of 4, ldo >= %s", stacked ? "cols" : "3 * cols");
  VITMI_REQUIRE(is_aligned(x, 16) && is_aligned(out, 8), VITMI_E_ALIGN, "split3: x needs 16-byte, out 8-byte alignment");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  bf16* o = reinterpret_cast<bf16*>(out);
  const bool wide = cols % 8 == 0 && ldo % 8 == 0 && is_aligned(out, 16);       // 16-byte stores
  const int w = wide ? 8 : 4;
  const dim3 grid(grid_for(rows * cols / w));
#define GO(BP, ST, WV) hipLaunchKernelGGL((split3_kernel<BP, ST, WV>), grid, dim3(BLOCK), 0, stream, x, ldx, o, ldo, rows, cols)
#define GO2(BP, ST) do { if (wide) GO(BP, ST, 8); else GO(BP, ST, 4); } while (0)
  if (b_pattern) { if (stacked) GO2(true, true); else GO2(true, false); }
  else { if (stacked) GO2(false, true); else GO2(false, false); }
#undef GO2
#undef GO
  return vitmi_check_launch("split3_kernel");
}

extern "C" int vitmi_gelu_fwd(const float* pre, int64_t ldp, float* out, int64_t ldo, int64_t M, int64_t N, void* stream_) {
  VITMI_REQUIRE(pre && out && M > 0 && N > 0, VITMI_E_BADARG, "gelu_fwd: null pointer or empty shape");
  VITMI_REQUIRE(N % 4 == 0 && ldp % 4 == 0 && ldo % 4 == 0 && ldp >= N && ldo >= N && is_aligned(pre, 16) && is_aligned(out, 16),
                VITMI_E_ALIGN, "gelu_fwd: 4-element alignment required");
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(M * N / 4)), dim3(BLOCK), 0, reinterpret_cast<hipStream_t>(stream_), pre,
                     ldp, out, ldo, M, N);
  return vitmi_check_launch("gelu_fwd_kernel");
}

extern "C" int vitmi_gelu_bwd(const float* dh, int64_t ldd, const float* pre, int64_t ldp, float* out, int64_t ldo, int64_t M,
                              int64_t N, void* stream_) {
  VITMI_REQUIRE(dh && pre && out && M > 0 && N > 0, VITMI_E_BADARG, "gelu_bwd: null pointer or empty shape");
  VITMI_REQUIRE(N % 4 == 0 && ldd % 4 == 0 && ldp % 4 == 0 && ldo % 4 == 0 && ldd >= N && ldp >= N && ldo >= N &&
                    is_aligned(dh, 16) && is_aligned(pre, 16) && is_aligned(out, 16),
                VITMI_E_ALIGN, "gelu_bwd: 4-element alignment required");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(M * N / 4)), dim3(BLOCK), 0, reinterpret_cast<hipStream_t>(stream_), dh, ldd,
                     pre, ldp, out, ldo, M, N);
  return vitmi_check_launch("gelu_bwd_kernel");
}
