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

// The same transform written STRAIGHT into the patch rows the patch-embedding GEMM contracts (fused with
// vitmi_patchify: no fp32 NCHW intermediate, 154 MB per 256-image step at 224^2).  One thread = 4
// consecutive k = c p^2 + i p + j of one row (4 consecutive j: 4 bytes at a stride of C in the NHWC source,
// whose lines the neighbouring c / j lanes share); stores are 8 B (bf16) / 16 B (fp32) per lane over whole
// rows.  Values are bit-identical to image_ingest followed by patchify (same expression, one rounding).
template <typename TD>
__global__ __launch_bounds__(256) void ingest_patchify_kernel(const uint8_t* __restrict__ src, TD* __restrict__ out,
                                                             int64_t out_ld, const int32_t* __restrict__ oy,
                                                             const int32_t* __restrict__ ox, const uint8_t* __restrict__ flip,
                                                             const float* __restrict__ mean, const float* __restrict__ stdv,
                                                             int64_t B, int H, int W, int C, int S, int pad, int fill, int p,
                                                             int cls_rows) {
  const int g = S / p, ntok = cls_rows + g * g, Kp = C * p * p;
  const int kq = (int)(out_ld / 4);
  const int64_t total = B * ntok * (int64_t)kq;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k4 = (int)(idx % kq);
    const int64_t row = idx / kq;
    const int t = (int)(row % ntok);
    const int64_t b = row / ntok;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t >= cls_rows && k4 * 4 < Kp) {
      const int pt = t - cls_rows, py = pt / g, px = pt % g;
      const int k = k4 * 4, c = k / (p * p), rem = k % (p * p), i = rem / p, j = rem % p;
      const int y = py * p + i, x0 = px * p + j;
      const int sy = y + (oy ? oy[b] : pad) - pad;
      const int dx = (ox ? ox[b] : pad) - pad;
      const bool fl = flip && flip[b];
      const float mu = mean ? mean[c] : 0.f, sd = stdv ? stdv[c] : 1.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int x = x0 + e;
        const int sx = (fl ? S - 1 - x : x) + dx;
        int val = fill;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) val = src[((b * H + sy) * (int64_t)W + sx) * C + c];
        v[e] = ((float)val / 255.f - mu) / sd;
      }
    }
    store4<TD>(out + row * out_ld + k4 * 4, v);
  }
}

}  // namespace

extern "C" int vitmi_ingest_patchify(const void* src_u8_nhwc, void* out, int out_dtype, int64_t out_ld, const int32_t* off_y,
                                     const int32_t* off_x, const uint8_t* flip, const float* mean, const float* stdv,
                                     int64_t B, int64_t H, int64_t W, int64_t C, int64_t S, int64_t pad, int64_t fill,
                                     int64_t p, int cls_rows, void* stream_) {
  VITMI_REQUIRE(src_u8_nhwc && out && B > 0 && H > 0 && W > 0 && C > 0 && S > 0 && p > 0, VITMI_E_BADARG, "ingest_patchify: bad argument");
  VITMI_REQUIRE(S % p == 0 && p % 4 == 0, VITMI_E_SHAPE, "ingest_patchify: S %% p and p %% 4 must be 0 (S=%lld p=%lld)", (long long)S, (long long)p);
  if (out_ld == 0) out_ld = C * p * p;
  VITMI_REQUIRE(out_ld >= C * p * p && out_ld % 4 == 0 && is_aligned(out, 16), VITMI_E_ALIGN, "ingest_patchify: out_ld / alignment");
  VITMI_REQUIRE(pad >= 0 && fill >= 0 && fill <= 255 && S <= H + 2 * pad && S <= W + 2 * pad, VITMI_E_SHAPE,
                "ingest_patchify: crop %lld does not fit the padded image (%lld+2*%lld)", (long long)S, (long long)H, (long long)pad);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int64_t g = S / p;
  int64_t blocks = (B * (cls_rows + g * g) * (out_ld / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
#define GO(TD)                                                                                                          \
  hipLaunchKernelGGL((ingest_patchify_kernel<TD>), dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t*)src_u8_nhwc, \
                     (TD*)out, out_ld, off_y, off_x, flip, mean, stdv, B, (int)H, (int)W, (int)C, (int)S, (int)pad, (int)fill, \
                     (int)p, cls_rows)
  if (out_dtype == VITMI_BF16) GO(bf16);
  else if (out_dtype == VITMI_F32) GO(float);
  else return vitmi_fail(VITMI_E_DTYPE, "ingest_patchify: bad out dtype");
#undef GO
  return vitmi_check_launch("ingest_patchify_kernel");
}

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
