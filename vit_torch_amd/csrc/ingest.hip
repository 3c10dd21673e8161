// Device-side input pipeline: the reference runs its per-sample transforms on
// CPU DataLoader workers (/root/reference/utils_datasets.py:553-582):
//   RandomCrop(S, padding=max(2, S//12), fill=128) -> RandomHorizontalFlip -> ToTensor -> Normalize
// Here one kernel reads the uint8 NHWC batch once and writes the normalised fp32 NCHW tensor
// the patch-embedding gather consumes: dst[b,c,y,x] = (v/255 - mean[c]) / std[c] with
// v = padded_src[b, y + oy[b] - pad, xs + ox[b] - pad, c] (fill outside the image) and
// xs = flip[b] ? S-1-x : x.  The random draws are INPUTS (per-sample offsets / flags), so the
// transform is reproducible and testable; arithmetic order is torchvision's (true divisions),
// bit-exact in fp32.  HBM-bound: C bytes read + 4C bytes written per pixel.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void image_ingest_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                          const int32_t* __restrict__ oy, const int32_t* __restrict__ ox,
                                                          const uint8_t* __restrict__ flip, const float* __restrict__ mean,
                                                          const float* __restrict__ stdv, int64_t B, int H, int W, int C,
                                                          int S, int pad, int fill) {
  // one thread = 4 consecutive x of one (b, c, y) row: 16-B stores; reads are C-strided bytes
  const int xq = S / 4;
  const int64_t total = B * C * S * xq;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int x0 = (int)(idx % xq) * 4;
    int64_t t = idx / xq;
    const int y = (int)(t % S); t /= S;
    const int c = (int)(t % C);
    const int64_t b = t / C;
    const int sy = y + (oy ? oy[b] : pad) - pad;
    const int dx = (ox ? ox[b] : pad) - pad;
    const bool fl = flip && flip[b];
    const float mu = mean ? mean[c] : 0.f, sd = stdv ? stdv[c] : 1.f;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int x = x0 + e;
      const int sx = (fl ? S - 1 - x : x) + dx;
      int v = fill;
      if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = src[((b * H + sy) * (int64_t)W + sx) * C + c];
      o[e] = ((float)v / 255.f - mu) / sd;
    }
    *reinterpret_cast<f32x4*>(dst + ((b * C + c) * (int64_t)S + y) * S + x0) = o;
  }
}

}  // namespace

extern "C" int vitmi_image_ingest(const void* src_u8_nhwc, float* dst_nchw, const int32_t* off_y, const int32_t* off_x,
                                  const uint8_t* flip, const float* mean, const float* stdv, int64_t B, int64_t H,
                                  int64_t W, int64_t C, int64_t S, int64_t pad, int64_t fill, void* stream_) {
  VITMI_REQUIRE(src_u8_nhwc && dst_nchw && B > 0 && H > 0 && W > 0 && C > 0 && S > 0, VITMI_E_BADARG, "image_ingest: bad argument");
  VITMI_REQUIRE(S % 4 == 0 && is_aligned(dst_nchw, 16), VITMI_E_ALIGN, "image_ingest: output size must be a multiple of 4 and dst 16-B aligned");
  VITMI_REQUIRE(pad >= 0 && fill >= 0 && fill <= 255 && S <= H + 2 * pad && S <= W + 2 * pad, VITMI_E_SHAPE,
                "image_ingest: crop %lld does not fit the padded image (%lld+2*%lld)", (long long)S, (long long)H, (long long)pad);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  int64_t blocks = (B * C * S * (S / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(image_ingest_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t*)src_u8_nhwc, dst_nchw,
                     off_y, off_x, flip, mean, stdv, B, (int)H, (int)W, (int)C, (int)S, (int)pad, (int)fill);
  return vitmi_check_launch("image_ingest_kernel");
}
