// How fast does ONE CU pull the operands of one (image, head) of the attention backward,
// as a function of the memory layout?  3072 workgroups of 448 threads, one per CU (141 KB
// of LDS), each loads five [197][64] bf16 matrices (Q, K, V from a qkv-like buffer, O and dO
// from a [rows][768] buffer) as 16-B pieces, one batch, then writes them to LDS.
//   layout 0: token-major rows, the real one (128-B pieces at a 4608-B / 1536-B stride)
//   layout 1: head-major ([image][matrix][head][197][64]: 25 KB contiguous per matrix)
// build: hipcc --offload-arch=gfx950 -O3 -o stage_probe stage_probe.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void stage(const char* qkv, const char* o, const char* dout, int layout, int N, int H,
                                             unsigned long long* times, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  u4 r[5][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * nthr, row = min(c / 8, N - 1), pc = c % 8;
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      const char* p;
      if (layout == 0) {
        if (m < 3) p = qkv + ((long)(b * N + row) * 3 * H * 64 + m * H * 64 + h * 64) * 2 + pc * 16;
        else p = (m == 3 ? o : dout) + ((long)(b * N + row) * H * 64 + h * 64) * 2 + pc * 16;
      } else {
        if (m < 3) p = qkv + (((long)(b * 3 + m) * H + h) * N + row) * 128 + pc * 16;
        else p = (m == 3 ? o : dout) + (((long)b * H + h) * N + row) * 128 + pc * 16;
      }
      r[m][i] = *reinterpret_cast<const u4*>(p);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * nthr;
#pragma unroll
    for (int m = 0; m < 5; ++m) *reinterpret_cast<u4*>(smem + m * 28672 + c * 16) = r[m][i];
  }
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (tid == 0) times[bh] = t1 - t0;
  if (tid == 1) sink[bh] = reinterpret_cast<float*>(smem)[bh & 1023];
}

int main() {
  const int B = 256, N = 197, H = 12;
  const size_t qkv_b = (size_t)B * N * 3 * H * 64 * 2, o_b = (size_t)B * N * H * 64 * 2;
  char *qkv, *o, *dout, *flush;
  unsigned long long* times;
  float* sink;
  hipMalloc(&qkv, qkv_b); hipMalloc(&o, o_b); hipMalloc(&dout, o_b); hipMalloc(&flush, 1u << 30);
  hipMalloc(&times, B * H * 8); hipMalloc(&sink, B * H * 4);
  hipMemset(qkv, 1, qkv_b); hipMemset(o, 1, o_b); hipMemset(dout, 1, o_b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(stage), hipFuncAttributeMaxDynamicSharedMemorySize, 145000);
  std::vector<unsigned long long> t(B * H);
  for (int rep = 0; rep < 2; ++rep)
    for (int layout = 0; layout < 2; ++layout) {
      hipMemset(flush, rep, 1u << 30);     // push the operands out of L2 / MALL
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(stage, dim3(B * H), dim3(448), 145000, 0, qkv, o, dout, layout, N, H, times, sink);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(t.data(), times, B * H * 8, hipMemcpyDeviceToHost);
      std::vector<unsigned long long> mid(t.begin() + 1536, t.begin() + 2048);
      std::sort(mid.begin(), mid.end());
      printf("layout %d: launch %.1f us (%.2f TB/s), mid-launch workgroup median %llu cycles (p10 %llu, p90 %llu) for 123 KB\n",
             layout, ms * 1e3, (qkv_b + 2 * o_b) / (ms * 1e-3) / 1e12, mid[mid.size() / 2], mid[mid.size() / 10],
             mid[mid.size() * 9 / 10]);
    }
  return 0;
}
