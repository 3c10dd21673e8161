// Round-4 probe: how fast can ONE workgroup per CU stream a GEMM operand panel HBM / Infinity Cache -> LDS with
// global_load_lds_dwordx4 (LDS-DMA), as a function of (a) the request shape, (b) the slabs kept in flight, (c) a
// consumer that takes `delay` cycles per slab (the MFMA phases), (d) three CUs of an XCD streaming the SAME panel.
// It models the A operand of gemm_fast_kernel: panels of 256 rows x KB bytes (row pitch = KB), slab j = bytes
// [SB*j, SB*(j+1)) of every row, 8 waves issuing 1-KiB pieces.
//   shape 0: piece = 16 rows x 64 B   (k-major 32-deep slab: the ring loops)            SB = 64
//   shape 1: piece = 8 rows x 128 B   (k-major 64-deep stage, full lines: PIPE 2)       SB = 128
//   shape 2: piece = 2 rows x 512 B   (k-minor operand rows)                            SB = 512 (rows = 64 per "slab")
// build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LA = slabs in flight beyond the one being consumed; PPW = pieces per wave and slab
template <int SHAPE, int LA>
__global__ __launch_bounds__(512) void stream(const char* base, long panel_bytes, int panels, int KB, int share, int delay,
                                              unsigned long long* cyc, int rot, int skew) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int SB = SHAPE == 0 ? 64 : SHAPE == 1 ? 128 : 512;
  constexpr int PPW = SHAPE == 0 ? 2 : SHAPE == 1 ? 4 : 4;           // 16 / 32 / 32 KiB per slab
  // workgroups b, b + 8, b + 16 sit on one XCD: with share = 3 they stream the same panels
  const int bid = blockIdx.x;
  const int owner = share > 1 ? (bid & 7) + 8 * ((bid >> 3) / share) : bid;
  const char* mine = base + (long)owner * panels * panel_bytes;
  unsigned voff[PPW];
  for (int i = 0; i < PPW; ++i) {
    const int p = wave * PPW + i;
    if (SHAPE == 0) voff[i] = (unsigned)((p * 16 + (lane >> 2)) * KB + (lane & 3) * 16);
    else if (SHAPE == 1) voff[i] = (unsigned)((p * 8 + (lane >> 3)) * KB + (lane & 7) * 16);
    else voff[i] = (unsigned)((p * 2 + (lane >> 5)) * KB + (lane & 31) * 16);
  }
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
  const int slabs = KB / SB;
  constexpr int RING = LA + 1;
  // rot: the `share` workgroups of a panel walk its slabs in rotated order inside groups of `share` slabs (slab
  // g*T + p -> g*T + (p + c) % g, c = position among the sharers): at any moment they stream DIFFERENT slabs, so each
  // line is missed by one of them and hit by the others.  skew: sharer c starts c * skew cycles late instead.
  const int cpos = share > 1 ? (bid >> 3) % share : 0;
  if (skew > 0 && cpos > 0) {
    const unsigned long long d0 = stamp();
    while (stamp() - d0 < (unsigned long long)skew * cpos) __builtin_amdgcn_s_sleep(8);
  }
  const unsigned long long t0 = stamp();
  long issued = 0;
  for (int pn = 0; pn < panels; ++pn) {
    const char* pb = mine + (long)pn * panel_bytes;
    for (int j = 0; j < slabs + LA; ++j) {
      if (j < slabs) {
        const unsigned dst = lds0 + (unsigned)((issued % RING) * (PPW * 8 * 1024)) + wave * PPW * 1024;
        int jp = j;
        if (rot && share > 1 && j < slabs / share * share) { const int T = j / share, p = j - T * share; jp = T * share + (p + cpos) % share; }
        for (int i = 0; i < PPW; ++i) glds16(pb + (long)jp * SB, voff[i], dst + i * 1024);
        ++issued;
      }
      if (j >= LA) {            // consume slab j - LA: it must have landed
        if (j < slabs) wait_vm<PPW * LA>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (delay > 0) {
          const unsigned long long d0 = stamp();
          while (stamp() - d0 < (unsigned long long)delay) __builtin_amdgcn_s_sleep(2);
        }
      }
    }
  }
  wait_vm<0>();
  const unsigned long long t1 = stamp();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int LA>
static void run(const char* buf, char* flush, int KB, int panels, int share, int delay, int nwg, const char* tag, int rot = 0, int skew = 0) {
  const long panel_bytes = (SHAPE == 2 ? 64L : 256L) * KB;
  unsigned long long* cyc;
  hipMalloc(&cyc, nwg * 8);
  auto k = stream<SHAPE, LA>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  float best = 1e9f;
  std::vector<unsigned long long> c(nwg);
  for (int rep = 0; rep < 3; ++rep) {
    if (flush) hipMemset(flush, rep, 1u << 30);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 160 * 1024, 0, buf, panel_bytes, panels, KB, share, delay, cyc, rot, skew);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  hipMemcpy(c.data(), cyc, nwg * 8, hipMemcpyDeviceToHost);
  std::sort(c.begin(), c.end());
  const double bytes_wg = (double)panels * panel_bytes;
  const int slabs = KB / (SHAPE == 0 ? 64 : SHAPE == 1 ? 128 : 512);
  if (rot || skew) printf("[rot %d skew %d] ", rot, skew);
  printf("%-34s shape %d LA %d KB %4d share %d delay %4d: %7.1f us  chip %5.2f TB/s  per CU %5.1f B/clk  cycles/slab %6.0f (median wg)\n", tag, SHAPE,
         LA, KB, share, delay, best * 1e3, bytes_wg * nwg / (best * 1e-3) / 1e12, bytes_wg / (double)c[nwg / 2],
         (double)c[nwg / 2] / ((double)panels * slabs));
  hipFree(cyc);
}

int main() {
  char *buf, *flush;
  const size_t total = 3ull << 30;
  hipMalloc(&buf, total);
  hipMalloc(&flush, 1u << 30);
  hipMemset(buf, 1, total);
  // ---- how much can ONE CU pull when the chip is otherwise idle (8 workgroups = one per XCD), and as the chip fills
  for (int nwg : {8, 64, 256}) {
    printf("--- %d workgroups, no consumer, distinct panels from HBM\n", nwg);
    run<0, 1>(buf, flush, 1536, 24, 1, 0, nwg, "stream");
    run<0, 3>(buf, flush, 1536, 24, 1, 0, nwg, "stream");
    run<0, 7>(buf, flush, 1536, 24, 1, 0, nwg, "stream");
    run<1, 1>(buf, flush, 1536, 24, 1, 0, nwg, "stream");
    run<1, 3>(buf, flush, 1536, 24, 1, 0, nwg, "stream");
    run<2, 3>(buf, flush, 1536, 96, 1, 0, nwg, "stream");
  }
  printf("--- L2-resident: every workgroup streams the SAME panel (share = all), no consumer\n");
  run<0, 3>(buf, nullptr, 1536, 24, 1 << 20, 0, 256, "L2 hits");
  run<1, 1>(buf, nullptr, 1536, 24, 1 << 20, 0, 256, "L2 hits");
  run<1, 3>(buf, nullptr, 1536, 24, 1 << 20, 0, 256, "L2 hits");
  printf("--- consumer overlap, ONE CU per XCD active (8 workgroups)\n");
  run<0, 3>(buf, flush, 1536, 24, 1, 1375, 8, "hbm + consumer");
  run<0, 7>(buf, flush, 1536, 24, 1, 1375, 8, "hbm + consumer");
  run<1, 1>(buf, flush, 1536, 24, 1, 2750, 8, "hbm + consumer");
  run<1, 3>(buf, flush, 1536, 24, 1, 2750, 8, "hbm + consumer");
  return 0;
}
