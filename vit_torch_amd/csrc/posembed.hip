// Bicubic resize of the position embedding as a row-sparse product (fp32).
//
// Upstream DINO's interpolate_pos_encoding (the module the reference loads at
// /root/reference/models/vision_all.py:156; restated in oracle/vit_ref.py:110-127) resizes the
// stored [side x side] grid of pos_embed to the patch grid of the input with
// F.interpolate(mode="bicubic").  The interpolation weights depend only on the two grids, so the
// host builds them once per shape (vit_torch_amd/posembed.py: 16 taps per output position, the
// CLS row a single unit tap) and this kernel applies them:
//     dst[r, :] = sum_{e in [row_ptr[r], row_ptr[r+1])} w[e] * src[col[e], :]
// The backward pass is the same kernel on the transposed table (entries of a row in a fixed
// order: deterministic, no atomics).  HBM traffic is a few hundred KB; one workgroup per output row.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void pos_resample_kernel(const float* __restrict__ src, int64_t lds_,
                                                           const int32_t* __restrict__ row_ptr,
                                                           const int32_t* __restrict__ col,
                                                           const float* __restrict__ w,
                                                           float* __restrict__ dst, int64_t ldd, int D) {
  const int r = blockIdx.x;
  const int e0 = row_ptr[r], e1 = row_ptr[r + 1];      // wave-uniform: scalar loads
  for (int c = threadIdx.x * 4; c < D; c += 256 * 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int e = e0; e < e1; ++e) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)col[e] * lds_ + c);
      const float we = w[e];
      acc[0] = __builtin_fmaf(we, v[0], acc[0]);
      acc[1] = __builtin_fmaf(we, v[1], acc[1]);
      acc[2] = __builtin_fmaf(we, v[2], acc[2]);
      acc[3] = __builtin_fmaf(we, v[3], acc[3]);
    }
    *reinterpret_cast<f32x4*>(dst + (int64_t)r * ldd + c) = acc;
  }
}

}  // namespace

extern "C" int vitmi_pos_resample(const float* src, int64_t ld_src, const int32_t* row_ptr, const int32_t* col,
                                  const float* w, float* dst, int64_t ld_dst, int64_t rows, int64_t D,
                                  void* stream_) {
  VITMI_REQUIRE(src && row_ptr && col && w && dst, VITMI_E_BADARG, "pos_resample: null argument");
  VITMI_REQUIRE(rows > 0 && rows < (1ll << 31) && D > 0 && D < (1ll << 31), VITMI_E_BADARG, "pos_resample: bad shape");
  VITMI_REQUIRE(D % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_src >= D && ld_dst >= D, VITMI_E_SHAPE,
                "pos_resample: D and the row pitches must be multiples of 4 (D=%lld)", (long long)D);
  VITMI_REQUIRE(is_aligned(src, 16) && is_aligned(dst, 16) && is_aligned(row_ptr, 4) && is_aligned(col, 4) && is_aligned(w, 4),
                VITMI_E_ALIGN, "pos_resample: pointer alignment");
  hipLaunchKernelGGL(pos_resample_kernel, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                     src, ld_src, row_ptr, col, w, dst, ld_dst, (int)D);
  return vitmi_check_launch("pos_resample_kernel");
}
