// The remaining entries of the reference's optimizer table (/root/reference/utils_network.py:119-126:
// adadelta, adagrad, adabelief) as one pass each over the flat parameter / gradient / state buffers, with
// the bf16 weight shadow refreshed in the same pass (sgd / adam / adamw: elementwise.hip).  HBM-bound:
// 16 B (Adagrad: p, g, sum) to 24 B (p, g and two state arrays read + written) per parameter.
// The step count lives on the DEVICE (state[0]) and is advanced before the update, as in vitmi_adam, so a
// captured HIP graph replays the right step-dependent factors.
#include "common.h"

namespace {

constexpr int OPT_BLOCK = 256;
inline unsigned opt_grid(int64_t n) {
  int64_t b = (n / 4 + OPT_BLOCK) / OPT_BLOCK;
  if (b > 2048) b = 2048;
  return (unsigned)(b < 1 ? 1 : b);
}

__global__ void opt_tick_kernel(float* state) { state[0] += 1.f; }

// element-wise driver: F(p, g, a, b) updates one parameter and its (up to two) state values in place
template <bool TWO, typename F>
__device__ __forceinline__ void opt_walk(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ s1,
                                         float* __restrict__ s2, bf16* __restrict__ shadow, int64_t n, float gscale, F f) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = *reinterpret_cast<f32x4*>(p + i * 4);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 av = *reinterpret_cast<f32x4*>(s1 + i * 4);
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if constexpr (TWO) bv = *reinterpret_cast<f32x4*>(s2 + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pv[e], ae = av[e], be = bv[e];
      f(pe, gv[e] * gscale, ae, be);
      pv[e] = pe; av[e] = ae; bv[e] = be;
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pv;
    *reinterpret_cast<f32x4*>(s1 + i * 4) = av;
    if constexpr (TWO) *reinterpret_cast<f32x4*>(s2 + i * 4) = bv;
    if (shadow) store4<bf16>(shadow + i * 4, pv);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pi = p[i], a = s1[i], b = TWO ? s2[i] : 0.f;
    f(pi, g[i] * gscale, a, b);
    p[i] = pi; s1[i] = a;
    if constexpr (TWO) s2[i] = b;
    if (shadow) shadow[i] = (bf16)pi;
  }
}

// torch.optim.Adagrad (single tensor): g += wd p; clr = lr / (1 + (t-1) lr_decay); sum += g^2;
// p -= clr g / (sqrt(sum) + eps)
__global__ void adagrad_kernel(float* p, const float* g, float* sum, bf16* shadow, const float* state, int64_t n, float lr,
                               float lr_decay, float eps, float wd, float gscale) {
  const float clr = lr / (1.f + (state[0] - 1.f) * lr_decay);
  opt_walk<false>(p, g, sum, nullptr, shadow, n, gscale, [=](float& pi, float gi, float& s, float&) {
    gi = fmaf(wd, pi, gi);
    s = fmaf(gi, gi, s);
    pi -= clr * gi / (sqrtf(s) + eps);
  });
}

// torch.optim.Adadelta: g += wd p; sq = rho sq + (1-rho) g^2; d = sqrt(acc + eps) / sqrt(sq + eps) * g;
// acc = rho acc + (1-rho) d^2; p -= lr d
__global__ void adadelta_kernel(float* p, const float* g, float* sq, float* acc, bf16* shadow, int64_t n, float lr, float rho,
                                float eps, float wd, float gscale) {
  opt_walk<true>(p, g, sq, acc, shadow, n, gscale, [=](float& pi, float gi, float& s, float& a) {
    gi = fmaf(wd, pi, gi);
    s = rho * s + (1.f - rho) * gi * gi;
    const float d = sqrtf(a + eps) / sqrtf(s + eps) * gi;
    a = rho * a + (1.f - rho) * d * d;
    pi -= lr * d;
  });
}

// AdaBelief (Zhuang et al., NeurIPS 2020; the adabelief_pytorch package the reference imports at
// utils_network.py:17 is not in this container: restated from the published algorithm, options as the
// reference sets them at :125 — weight_decouple, rectify — plus amsgrad off, fixed_decay off,
// degenerated_to_sgd on, weight_decay 0 by default):
//   p *= 1 - lr wd (decoupled) | g += wd p;  m = b1 m + (1-b1) g;  s = b2 s + (1-b2) (g-m)^2 + eps
//   rectified (RAdam): rho_inf = 2/(1-b2) - 1, rho_t = rho_inf - 2 t b2^t / (1-b2^t);
//     rho_t >= 5: p -= lr r_t / (1-b1^t) * m / (sqrt(s) + eps),
//                 r_t = sqrt((1-b2^t) (rho_t-4)/(rho_inf-4) (rho_t-2)/rho_t rho_inf/(rho_inf-2))
//     else       : p -= lr / (1-b1^t) * m            (degenerated to SGD with momentum)
//   not rectified: p -= lr/(1-b1^t) * m / (sqrt(s)/sqrt(1-b2^t) + eps)
__global__ void adabelief_kernel(float* p, const float* g, float* m, float* s, bf16* shadow, const float* state, int64_t n,
                                 float lr, float b1, float b2, float eps, float wd, int decoupled, int rectify, float gscale) {
  const float t = state[0];
  const float b1t = powf(b1, t), b2t = powf(b2, t);
  const float bc1 = 1.f - b1t, bc2 = 1.f - b2t;
  const float rho_inf = 2.f / (1.f - b2) - 1.f;
  const float rho_t = rho_inf - 2.f * t * b2t / bc2;
  const bool adaptive = !rectify || rho_t >= 5.f;
  float step;                                        // multiplies m / denom (or m alone)
  if (!rectify) step = lr / bc1;
  else if (rho_t >= 5.f)
    step = lr * sqrtf(bc2 * (rho_t - 4.f) / (rho_inf - 4.f) * (rho_t - 2.f) / rho_t * rho_inf / (rho_inf - 2.f)) / bc1;
  else step = lr / bc1;
  const float inv_sbc2 = rectify ? 1.f : 1.f / sqrtf(bc2);
  const float keep = decoupled ? 1.f - lr * wd : 1.f;
  opt_walk<true>(p, g, m, s, shadow, n, gscale, [=](float& pi, float gi, float& mi, float& si) {
    if (decoupled) pi *= keep;
    else gi = fmaf(wd, pi, gi);
    mi = b1 * mi + (1.f - b1) * gi;
    const float r = gi - mi;
    si = b2 * si + (1.f - b2) * r * r + eps;
    if (adaptive) pi -= step * mi / (sqrtf(si) * inv_sbc2 + eps);
    else pi -= step * mi;
  });
}

int check_opt(const void* p, const void* g, const void* a, const void* b, const void* shadow, int64_t n, const char* who) {
  VITMI_REQUIRE(p && g && a && n > 0, VITMI_E_BADARG, "%s: bad argument", who);
  VITMI_REQUIRE(is_aligned(p, 16) && is_aligned(g, 16) && is_aligned(a, 16) && (!b || is_aligned(b, 16)) &&
                    (!shadow || is_aligned(shadow, 8)),
                VITMI_E_ALIGN, "%s: buffers must be 16-B aligned", who);
  return 0;
}

}  // namespace

extern "C" int vitmi_adagrad(float* p, const float* g, float* sum, void* shadow, float* state, int64_t n, float lr,
                             float lr_decay, float eps, float weight_decay, float grad_scale, void* stream_) {
  if (int rc = check_opt(p, g, sum, nullptr, shadow, n, "adagrad")) return rc;
  VITMI_REQUIRE(state, VITMI_E_BADARG, "adagrad: null state");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(opt_tick_kernel, dim3(1), dim3(1), 0, stream, state);
  if (int rc = vitmi_check_launch("opt_tick_kernel")) return rc;
  hipLaunchKernelGGL(adagrad_kernel, dim3(opt_grid(n)), dim3(OPT_BLOCK), 0, stream, p, g, sum, (bf16*)shadow, state, n, lr,
                     lr_decay, eps, weight_decay, grad_scale);
  return vitmi_check_launch("adagrad_kernel");
}

extern "C" int vitmi_adadelta(float* p, const float* g, float* square_avg, float* acc_delta, void* shadow, int64_t n,
                              float lr, float rho, float eps, float weight_decay, float grad_scale, void* stream_) {
  if (int rc = check_opt(p, g, square_avg, acc_delta, shadow, n, "adadelta")) return rc;
  VITMI_REQUIRE(acc_delta && rho >= 0.f && rho <= 1.f, VITMI_E_BADARG, "adadelta: bad argument");
  hipLaunchKernelGGL(adadelta_kernel, dim3(opt_grid(n)), dim3(OPT_BLOCK), 0, reinterpret_cast<hipStream_t>(stream_), p, g,
                     square_avg, acc_delta, (bf16*)shadow, n, lr, rho, eps, weight_decay, grad_scale);
  return vitmi_check_launch("adadelta_kernel");
}

extern "C" int vitmi_adabelief(float* p, const float* g, float* m, float* s, void* shadow, float* state, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int decoupled, int rectify,
                               float grad_scale, void* stream_) {
  if (int rc = check_opt(p, g, m, s, shadow, n, "adabelief")) return rc;
  VITMI_REQUIRE(s && state && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, VITMI_E_BADARG, "adabelief: bad argument");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(opt_tick_kernel, dim3(1), dim3(1), 0, stream, state);
  if (int rc = vitmi_check_launch("opt_tick_kernel")) return rc;
  hipLaunchKernelGGL(adabelief_kernel, dim3(opt_grid(n)), dim3(OPT_BLOCK), 0, stream, p, g, m, s, (bf16*)shadow, state, n, lr,
                     beta1, beta2, eps, weight_decay, decoupled, rectify, grad_scale);
  return vitmi_check_launch("adabelief_kernel");
}
