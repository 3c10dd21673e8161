// LDS bank-conflict probe for gfx950: cycles per wave-instruction of ds_read_b128 /
// ds_read_b64_tr_b16 for a given per-lane byte-offset pattern (one wave, back-to-back).
// build: hipcc --offload-arch=gfx950 -O3 -o lds_probe lds_probe.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <functional>

template <int KIND>   // 0 = ds_read_b128, 1 = ds_read_b64_tr_b16, 2 = ds_read_b64
__global__ void probe(const int* offs, unsigned long long* out, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 16384; i += 64) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  const unsigned a = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + offs[lane];
  unsigned long long t0, t1;
  float acc = 0.f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int r = 0; r < reps; ++r) {
    if constexpr (KIND == 0) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile(
          "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %8\n\t"
          "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8\n\tds_read_b128 %6, %8\n\tds_read_b128 %7, %8\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
          : "v"(a) : "memory");
      acc += v0[0] + v7[3];
    } else if constexpr (KIND == 1) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8\n\tds_read_b64_tr_b16 %2, %8\n\tds_read_b64_tr_b16 %3, %8\n\t"
          "ds_read_b64_tr_b16 %4, %8\n\tds_read_b64_tr_b16 %5, %8\n\tds_read_b64_tr_b16 %6, %8\n\tds_read_b64_tr_b16 %7, %8\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
          : "v"(a) : "memory");
      acc += v0[0] + v7[1];
    } else {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile(
          "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8\n\tds_read_b64 %2, %8\n\tds_read_b64 %3, %8\n\t"
          "ds_read_b64 %4, %8\n\tds_read_b64 %5, %8\n\tds_read_b64 %6, %8\n\tds_read_b64 %7, %8\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
          : "v"(a) : "memory");
      acc += v0[0] + v7[1];
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (lane == 0) { out[0] = t1 - t0; }
  if (acc == 12345.678f) out[1] = 1;
}

struct Pat { std::string name; int kind; std::function<int(int)> f; };

int main() {
  std::vector<Pat> pats;
  auto swz1 = [](int b) { return b ^ (((b >> 9) & 1) << 5); };
  // ---- ds_read_b128 patterns (lane l: row = l&15, chunk group g = l>>4) for the 16x16x32 operand
  pats.push_back({"b128 broadcast", 0, [](int l) { return 0; }});
  pats.push_back({"b128 linear 16B*lane", 0, [](int l) { return l * 16; }});
  pats.push_back({"b128 PIPE1 subtile 64B rows + bit5^=bit9", 0, [=](int l) { return swz1((l & 15) * 64 + (l >> 4) * 16); }});
  pats.push_back({"b128 PIPE1 subtile 64B rows no swizzle", 0, [](int l) { return (l & 15) * 64 + (l >> 4) * 16; }});
  pats.push_back({"b128 PIPE2 128B rows, c^((r>>1)&7)", 0, [](int l) { int r = l & 15; return r * 128 + (((l >> 4)) ^ ((r >> 1) & 7)) * 16; }});
  pats.push_back({"b128 128B rows, c^(r&7)", 0, [](int l) { int r = l & 15; return r * 128 + (((l >> 4)) ^ (r & 7)) * 16; }});
  pats.push_back({"b128 128B rows no swizzle", 0, [](int l) { int r = l & 15; return r * 128 + (l >> 4) * 16; }});
  pats.push_back({"b128 64B rows, chunk^((r>>2)&3)", 0, [](int l) { int r = l & 15; return r * 64 + (((l >> 4)) ^ ((r >> 2) & 3)) * 16; }});
  pats.push_back({"b128 64B rows, chunk^(r&3)", 0, [](int l) { int r = l & 15; return r * 64 + (((l >> 4)) ^ (r & 3)) * 16; }});
  pats.push_back({"b128 80B rows (pad)", 0, [](int l) { return (l & 15) * 80 + (l >> 4) * 16; }});
  pats.push_back({"b128 144B rows (pad), 32-row attn K: row=l&31, c=l>>5", 0, [](int l) { return (l & 31) * 144 + (l >> 5) * 16; }});
  pats.push_back({"b128 128B rows attn-style row=l&31,c=l>>5 no swz", 0, [](int l) { return (l & 31) * 128 + (l >> 5) * 16; }});
  pats.push_back({"b128 128B rows attn-style c^((r>>1)&7)", 0, [](int l) { int r = l & 31; return r * 128 + ((l >> 5) ^ ((r >> 1) & 7)) * 16; }});
  pats.push_back({"b128 128B rows attn-style c^(r&7)", 0, [](int l) { int r = l & 31; return r * 128 + ((l >> 5) ^ (r & 7)) * 16; }});
  // ---- tr16_b64 patterns: lane (g = l>>4, i = l&15): row = 8g + (i>>2), 8B piece (i&3) of a 32-B chunk
  auto key = [](int row) { return (row & 3) | (((row >> 3) & 1) << 2); };
  pats.push_back({"tr16 k-row image 512B rows, key swizzle (rb=0)", 1, [=](int l) { int g = l >> 4, i = l & 15; int row = 8 * g + (i >> 2); return row * 512 + ((0 ^ key(row)) * 32) + 8 * (i & 3); }});
  pats.push_back({"tr16 k-row image 512B rows, no swizzle", 1, [=](int l) { int g = l >> 4, i = l & 15; int row = 8 * g + (i >> 2); return row * 512 + 8 * (i & 3); }});
  pats.push_back({"tr16 attn V 192B rows: rows 4h+(i>>2), cols 16*grp", 1, [](int l) { int h5 = l >> 5, grp = (l >> 4) & 1, i = l & 15; return (4 * h5 + (i >> 2)) * 192 + (16 * grp + 4 * (i & 3)) * 2; }});
  pats.push_back({"tr16 attn K 144B rows", 1, [](int l) { int h5 = l >> 5, grp = (l >> 4) & 1, i = l & 15; return (4 * h5 + (i >> 2)) * 144 + (16 * grp + 4 * (i & 3)) * 2; }});
  pats.push_back({"b64 linear 8B*lane", 2, [](int l) { return l * 8; }});

  int* d_off; unsigned long long* d_out;
  hipMalloc(&d_off, 64 * sizeof(int));
  hipMalloc(&d_out, 2 * sizeof(unsigned long long));
  const int reps = 64;
  for (auto& p : pats) {
    int h[64];
    for (int l = 0; l < 64; ++l) h[l] = p.f(l);
    hipMemcpy(d_off, h, sizeof(h), hipMemcpyHostToDevice);
    unsigned long long best = ~0ull;
    for (int it = 0; it < 5; ++it) {
      if (p.kind == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 65536, 0, d_off, d_out, reps);
      else if (p.kind == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 65536, 0, d_off, d_out, reps);
      else hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 65536, 0, d_off, d_out, reps);
      unsigned long long o[2];
      hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
      if (o[0] < best) best = o[0];
    }
    printf("%-62s %7.2f cycles/instr\n", p.name.c_str(), (double)best / (reps * 8));
  }
  return 0;
}
