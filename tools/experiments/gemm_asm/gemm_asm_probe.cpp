// Probe of the hand-scheduled 4-wave GEMM main loop (see ../gemm_asm_plan.md): C[M][N] fp32 =
// A[M][K] * B[N][K]^T, bf16 operands, 256x256 tiles, one workgroup per CU.
// build (in this directory): python gen_gemm_asm.py && python gen_gemm_asm32.py && python gen_gemm_asm_bd.py &&
//        python -c "print('#define CLOB_V '+','.join('\"v%d\"'%i for i in range(128,256)));print('#define CLOB_V64 '+','.join('\"v%d\"'%i for i in range(64,256)));print('#define CLOB_A '+','.join('\"a%d\"'%i for i in range(256)))" > gemm_asm_clob.inc &&
//        hipcc --offload-arch=gfx950 -O3 -o gemm_asm_probe gemm_asm_probe.cpp
// run on the GPU box: ./gemm_asm_probe [M N K [variant]]   variant: 32 = 32x32x16 MFMAs, 2 = "B direct"
// the generators read switches from the environment (ORDER, STAGING, STAGGER, BURST, NO_READS, NO_DMA, NO_VM,
// NO_BAR, FILL / NFILL): every variant of ../gemm_asm_plan.md is one of them
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16;
#define LDS_ADDR(p) ((uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)(p)))

#include "gemm_asm_clob.inc"

__device__ unsigned long long g_cycles[2];
__global__ __launch_bounds__(256) void gemm_asm_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int tile = blockIdx.x;
  const int64_t m0 = (int64_t)(tile / tiles_n) * 256, n0 = (int64_t)(tile % tiles_n) * 256;
  // per-lane DMA source offset inside a 16-row x 32-k subtile (swizzled image, see gemm_fast.hip)
  const int pb = 16 * lane, lb = pb ^ (((pb >> 9) & 1) << 5);
  const int row = lb >> 6, ch = (lb & 63) >> 4;
  const uint32_t voffA = (uint32_t)((row * K + ch * 8) * 2), voffB = voffA;     // lda = ldb = K
  int pf = (lane & 15) * 64 + (lane >> 4) * 16;
  pf ^= ((pf >> 9) & 1) << 5;
  const uint32_t ring = LDS_ADDR(smem);
  const uint32_t vA01 = ring + wm * 8192 + pf, vA23 = vA01 + 65536;
  const uint32_t vB01 = ring + 16384 + wn * 8192 + pf, vB23 = vB01 + 65536;
  const uint64_t sA = (uint64_t)(uintptr_t)(A + (m0 + w * 64) * K), sB = (uint64_t)(uintptr_t)(B + (n0 + w * 64) * K);
  const uint32_t strideA = (uint32_t)(16 * K * 2), strideB = strideA;
  const uint32_t nslabs = (uint32_t)(K / 32);
  const uint32_t ldsw = ring + w * 4096;
  const uint32_t vW01 = ldsw + 16 * lane, vW23 = vW01 + 65536;   // STAGING=vgpr: per-lane LDS store bases
  const unsigned long long tc0 = __builtin_readcyclecounter();
  asm volatile(
#include "gemm_asm_loop.inc"
      :
      : "v"(vA01), "v"(vA23), "v"(vB01), "v"(vB23), "v"(voffA), "v"(voffB), "s"(sA), "s"(sB), "s"(strideA), "s"(strideB),
        "s"(nslabs), "s"(ldsw), "v"(vW01), "v"(vW23), "s"(w)
      : CLOB_V64, CLOB_A, "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s52", "s53", "s54", "s55", "s56",
        "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "m0", "scc", "vcc", "memory");
  const unsigned long long tc1 = __builtin_readcyclecounter();
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { g_cycles[0] = tc1 - tc0; }
  float acc[8][8][4];
#include "gemm_asm_readout.inc"
  // D^T blocks: lane owns row m = 16 mi + (lane & 15), columns 16 ni + 4 (lane >> 4) .. +3
#pragma unroll
  for (int ni = 0; ni < 8; ++ni)
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const int64_t m = m0 + wm * 128 + mi * 16 + (lane & 15), n = n0 + wn * 128 + ni * 16 + 4 * (lane >> 4);
      *reinterpret_cast<float4*>(C + m * N + n) = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
    }
}

// ---- 32x32x16 variant (gen_gemm_asm32.py)
__global__ __launch_bounds__(256) void gemm_asm32_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                         float* __restrict__ C, int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int tile = blockIdx.x;
  const int64_t m0 = (int64_t)(tile / tiles_n) * 256, n0 = (int64_t)(tile % tiles_n) * 256;
  // LDS-DMA writes lane-linearly: lane -> (row lane >> 2 of a 16-row subtile, chunk slot lane & 3);
  // the chunk stored in slot s of row r (r inside its 32-row block) is s ^ ((r >> 3) & 3)
  const int rl = lane >> 2, slot = lane & 3;
  const uint32_t voffe = (uint32_t)((rl * K + (slot ^ (rl >> 3)) * 8) * 2);          // even subtile: rows 0..15 of the block
  const uint32_t voffo = (uint32_t)((rl * K + (slot ^ ((rl >> 3) | 2)) * 8) * 2);    // odd subtile: rows 16..31
  const int r32 = lane & 31, cb = lane >> 5, q = (r32 >> 3) & 3;
  const uint32_t pf0 = (uint32_t)((r32 >> 4) * 1024 + (r32 & 15) * 64 + ((cb ^ q) * 16));
  const uint32_t pf1 = (uint32_t)((r32 >> 4) * 1024 + (r32 & 15) * 64 + (((2 + cb) ^ q) * 16));
  const uint32_t ring = LDS_ADDR(smem);
  const uint32_t vA0 = ring + wm * 8192 + pf0, vA1 = ring + wm * 8192 + pf1;
  const uint32_t vB0 = ring + 16384 + wn * 8192 + pf0, vB1 = ring + 16384 + wn * 8192 + pf1;
  const uint32_t vA0h = vA0 + 65536, vA1h = vA1 + 65536, vB0h = vB0 + 65536, vB1h = vB1 + 65536;
  const uint64_t sA = (uint64_t)(uintptr_t)(A + (m0 + w * 64) * K), sB = (uint64_t)(uintptr_t)(B + (n0 + w * 64) * K);
  const uint32_t strideA = (uint32_t)(16 * K * 2), strideB = strideA;
  const uint32_t nslabs = (uint32_t)(K / 32);
  const uint32_t ldsw = ring + w * 4096;
  const unsigned long long tc0 = __builtin_readcyclecounter();
  asm volatile(
#include "gemm_asm32_loop.inc"
      :
      : "v"(vA0), "v"(vA0h), "v"(vA1), "v"(vA1h), "v"(vB0), "v"(vB0h), "v"(vB1), "v"(vB1h), "v"(voffe), "v"(voffo),
        "v"(voffe), "v"(voffo), "s"(sA), "s"(sB), "s"(strideA), "s"(strideB), "s"(nslabs), "s"(ldsw)
      : CLOB_V, CLOB_A, "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "m0", "scc", "vcc", "memory");
  const unsigned long long tc1 = __builtin_readcyclecounter();
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { g_cycles[0] = tc1 - tc0; }
  float acc[4][4][4][4];
#include "gemm_asm32_readout.inc"
  // D^T 32x32 blocks: lane owns row m = 32 mi + (lane & 31), columns 32 ni + 8 q + 4 (lane >> 5) .. +3
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int64_t m = m0 + wm * 128 + mi * 32 + (lane & 31), n = n0 + wn * 128 + ni * 32 + 8 * qq + 4 * (lane >> 5);
        *reinterpret_cast<float4*>(C + m * N + n) =
            make_float4(acc[ni][mi][qq][0], acc[ni][mi][qq][1], acc[ni][mi][qq][2], acc[ni][mi][qq][3]);
      }
}

// ---- "B direct" variant (gen_gemm_asm_bd.py): A through the LDS ring, B fragments straight from global memory
__global__ __launch_bounds__(256) void gemm_asm_bd_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int tile = blockIdx.x;
  const int64_t m0 = (int64_t)(tile / tiles_n) * 256, n0 = (int64_t)(tile % tiles_n) * 256;
  const int pb = 16 * lane, lb = pb ^ (((pb >> 9) & 1) << 5);
  const int row = lb >> 6, ch = (lb & 63) >> 4;
  const uint32_t voffA = (uint32_t)((row * K + ch * 8) * 2);
  const uint32_t voffBd = (uint32_t)(((lane & 15) * K + (lane >> 4) * 8) * 2);
  int pf = (lane & 15) * 64 + (lane >> 4) * 16;
  pf ^= ((pf >> 9) & 1) << 5;
  const uint32_t ring = LDS_ADDR(smem);
  const uint32_t vA01 = ring + wm * 8192 + pf, vA23 = vA01 + 65536;
  const uint64_t sA = (uint64_t)(uintptr_t)(A + (m0 + w * 64) * K), sBw = (uint64_t)(uintptr_t)(B + (n0 + wn * 128) * K);
  const uint32_t strideA = (uint32_t)(16 * K * 2), strideB = strideA;
  const uint32_t nslabs = (uint32_t)(K / 32);
  const uint32_t ldsw = ring + w * 4096;
  const unsigned long long tc0 = __builtin_readcyclecounter();
  asm volatile(
#include "gemm_asm_bd_loop.inc"
      :
      : "v"(vA01), "v"(vA23), "v"(voffA), "v"(voffBd), "s"(sA), "s"(sBw), "s"(strideA), "s"(strideB), "s"(nslabs), "s"(ldsw)
      : CLOB_V64, CLOB_A, "s40", "s41", "s44", "s45", "s46", "s47", "s48", "s49", "s56", "s57", "s58", "s59", "s64", "s65",
        "s66", "s67", "s68", "s69", "s70", "s71", "m0", "scc", "vcc", "memory");
  const unsigned long long tc1 = __builtin_readcyclecounter();
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { g_cycles[0] = tc1 - tc0; }
  float acc[8][8][4];
#include "gemm_asm_readout.inc"
#pragma unroll
  for (int ni = 0; ni < 8; ++ni)
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const int64_t m = m0 + wm * 128 + mi * 16 + (lane & 15), n = n0 + wn * 128 + ni * 16 + 4 * (lane >> 4);
      *reinterpret_cast<float4*>(C + m * N + n) = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
    }
}

static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  const bool v32 = argc > 4 && atoi(argv[4]) == 32;            // 32 = the 32x32x16 variant
  const bool vbd = argc > 4 && atoi(argv[4]) == 2;             // 2 = the "B direct" variant
  auto kernel = v32 ? gemm_asm32_kernel : (vbd ? gemm_asm_bd_kernel : gemm_asm_kernel);
  if (M % 256 || N % 256 || K % 128) { printf("M, N %% 256 and K %% 128 required\n"); return 1; }
  std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
  srand(1);
  for (auto& v : hA) v = f2bf((rand() % 2001 - 1000) / 1000.f);
  for (auto& v : hB) v = f2bf((rand() % 2001 - 1000) / 20000.f);
  bf16 *dA, *dB;
  float* dC;
  if (hipMalloc(&dA, hA.size() * 2) != hipSuccess || hipMalloc(&dB, hB.size() * 2) != hipSuccess ||
      hipMalloc(&dC, (size_t)M * N * 4) != hipSuccess) { printf("alloc failed\n"); return 2; }
  (void)hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemset(dC, 0xff, (size_t)M * N * 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_asm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_asm32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_asm_bd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int tiles_n = N / 256, tiles = (M / 256) * tiles_n;
  hipLaunchKernelGGL(kernel, dim3(tiles), dim3(256), 131072, 0, dA, dB, dC, M, N, K, tiles_n);
  hipError_t err = hipDeviceSynchronize();
  if (err != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(err)); return 2; }
  std::vector<float> hC((size_t)M * N);
  (void)hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  int bad = 0;
  for (int t = 0; t < 4000; ++t) {
    const int m = rand() % M, n = rand() % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
    const double d = fabs(ref - hC[(size_t)m * N + n]);
    if (d > worst) worst = d;
    if (!(d <= 1e-3 * (1 + fabs(ref)))) {
      if (bad < 5) printf("  mismatch C[%d][%d] = %g, want %g\n", m, n, hC[(size_t)m * N + n], ref);
      ++bad;
    }
  }
  printf("check: worst abs diff %.3g over 4000 samples, %d bad\n", worst, bad);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(tiles), dim3(256), 131072, 0, dA, dB, dC, M, N, K, tiles_n);
  (void)hipEventRecord(e0);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kernel, dim3(tiles), dim3(256), 131072, 0, dA, dB, dC, M, N, K, tiles_n);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= it;
  printf("%d x %d x %d: %.1f us, %.1f TFLOP/s\n", M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
  unsigned long long cyc[2] = {0, 0};
  (void)hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(cyc));
  if (cyc[0]) {
    const double rounds = (double)tiles / 256.0;
    printf("  main loop of one tile: %llu cycles = %.1f per 32-deep slab; if every round took that long the clock was %.2f GHz\n",
           cyc[0], (double)cyc[0] / (K / 32), (double)cyc[0] * (rounds < 1 ? 1 : rounds) / (ms * 1e-3) / 1e9);
  }
  return bad ? 3 : 0;
}
