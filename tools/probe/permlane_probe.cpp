// what do v_permlane16_swap / v_permlane32_swap do with a register paired with itself? (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned x = threadIdx.x;
  auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  auto c = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  out[threadIdx.x] = a[0]; out[64 + threadIdx.x] = a[1]; out[128 + threadIdx.x] = c[0]; out[192 + threadIdx.x] = c[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"swap16[0]", "swap16[1]", "swap32[0]", "swap32[1]"};
  for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; i += 4) printf(" %u", h[r * 64 + i]); printf("\n"); }
  return 0;
}
