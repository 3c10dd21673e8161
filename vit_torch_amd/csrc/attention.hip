// Fused multi-head self-attention, bf16 operands / fp32 accumulation on
// v_mfma_f32_32x32x16_bf16, scores never leave the CU.
//
// Layouts: qkv [B, N, 3, H, hd] (output of the qkv Linear), out/dout
// [B, N, H, hd], lse/delta [B, H, N] fp32.
//
// Work split: one wave = 32 rows (queries; keys in dkdv); a workgroup has
// NW = ceil(min(N,256)/32) waves (7 for the 197-token ViT-B/16 sequence: 88 %
// of the padded rows are real, against 77 % with fixed 128-row blocks); the
// other side streams through LDS in chunks of up to 128 rows that are
// PREFETCHED INTO REGISTERS while the previous chunk is being computed and
// committed to LDS between two barriers (split stage, guide T14).
//
// Forward:
//   S^T[key][q] = K·Q^T        A = K rows (ds_read_b128), B = Q rows (registers)
//   online softmax per 32 keys the query is on the LANE, so max/sum/rescale are
//                              lane-local (+1 cross-half shuffle); O is rescaled
//                              only when some row's running max actually moved
//   O^T[d][q]  += V^T·P^T      A = V^T via ds_read_b64_tr_b16, B = the S^T
//                              accumulator itself re-used as operand (guide §3,
//                              "An accumulator tile as the next MFMA's operand")
// Backward = two kernels with the same building blocks and no atomics:
//   dkdv: wave = 32 keys,   S = Q·K^T, dP = dO·V^T (key on the lane),
//         dV^T += dO^T·P, dK^T += Q^T·dS
//   dq:   wave = 32 queries, S^T, dP^T (query on the lane), dQ^T += K^T·dS^T
// 7 MFMA products instead of the minimal 5, but no dQ reduction across waves.
// hd = 64: the softmax's VALU work (exp, max, sum, cvt) outweighs the MFMAs, so
// masks are applied only to ragged blocks and the scale is folded into exp2.
#include <atomic>
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int CHUNK_MAX = 128;     // rows of the streamed side resident in LDS

__device__ __forceinline__ unsigned long long attn_stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

__device__ __forceinline__ bf16x4 ds_read_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p));
}

// A-operand fragment of mfma_32x32x16 for X^T where X is an LDS image
// [image row = k][image col] (bf16, `stride` bytes per row): lane l gets
// A[row = cbase32 + (l&31)][k slot (h, j)] with k(h, j) = kbase16 + 8*(j>>2) +
// 4*h + (j&3) — the order in which a 32x32 accumulator presents its rows when
// it is re-used as the other operand.
__device__ __forceinline__ bf16x8 load_tr_frag(const char* img, int stride, int kbase16,
                                               int cbase32, int lane) {
  const int h5 = lane >> 5, grp = (lane >> 4) & 1, i = lane & 15;
  const char* p = img + (kbase16 + 4 * h5 + (i >> 2)) * stride + (cbase32 + 16 * grp + 4 * (i & 3)) * 2;
  const bf16x4 lo = ds_read_tr16(p);
  const bf16x4 hi = ds_read_tr16(p + 8 * stride);
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// split stage: global -> registers (issued early) ... registers -> LDS (after the barrier).
// `rows` rows x HD bf16 starting at row0 (token stride ts); rows >= N read as zero.
// A block of nthr threads moves rows*HD/8 16-byte pieces, at most 4 per thread.
template <int HD>
__device__ __forceinline__ void prefetch_rows(bf16x8 (&r)[4], const bf16* g, int64_t ts, int row0,
                                              int rows, int N, int tid, int nthr) {
  constexpr int CPR = HD / 8;
  // unconditional loads from clamped rows, zeroed afterwards: a branch around a load makes
  // hipcc wait for each load separately (guide §5, trap 4(c))
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * nthr;
    const int gr = min(row0 + c / CPR, N - 1);
    r[i] = *reinterpret_cast<const bf16x8*>(g + (int64_t)gr * ts + (c % CPR) * 8);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * nthr;
    if (c >= rows * CPR || row0 + c / CPR >= N) {
#pragma unroll
      for (int e = 0; e < 8; ++e) r[i][e] = (bf16)0.f;
    }
  }
}
template <int HD, int SB>
__device__ __forceinline__ void commit_rows(char* lds, const bf16x8 (&r)[4], int rows, int tid,
                                            int nthr) {
  constexpr int CPR = HD / 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * nthr;
    if (c < rows * CPR) *reinterpret_cast<bf16x8*>(lds + (c / CPR) * SB + (c % CPR) * 16) = r[i];
  }
}

template <int HD> struct AttnCfg {
  static constexpr int KS = HD * 2 + 16;              // row reads conflict-free
  static constexpr int VS = HD == 64 ? 192 : 64;      // tr reads conflict-free
  static constexpr int KSTEPS = HD / 16;
  static constexpr int DB = HD / 32;
};

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int base) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)v[base + j];
  return r;
}

// max of three in ONE instruction.  fmaxf() under the kernel's IEEE mode costs a canonicalising `v_max_f32 x, x, x` per
// operand that comes from an MFMA (hipcc cannot prove it is not a signalling NaN): 28 instructions for the 16-score row
// maximum of the forward's key block, 8 with these.
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max16(const f32x16& s) {
  const float t0 = max3f(s[0], s[1], s[2]), t1 = max3f(s[3], s[4], s[5]), t2 = max3f(s[6], s[7], s[8]);
  const float t3 = max3f(s[9], s[10], s[11]), t4 = max3f(s[12], s[13], s[14]);
  return max3f(max3f(t0, t1, t2), max3f(t3, t4, s[15]), t0);
}
// p = exp2(s * c - m) for the 16 scores of a lane, in place, and their sum.  (Round 3 tried the scale-and-shift and the sum on
// packed fp32 instructions — v_pk_fma_f32 / v_pk_add_f32, 8 + 8 instead of 16 + 16: forward 88-94 -> 92-99 us per ViT-B/16
// layer, and the same idea in the fused backward's loop 17.9 k -> 18.8 k cycles.  Packed VALU beside MFMAs is an anti-lever.)
__device__ __forceinline__ float exp2_scaled_sum16(f32x16& s, float c, float m) {
  float psum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c, -m));
    s[r] = p;
    psum += p;
  }
  return psum;
}

__device__ __forceinline__ void zero16(f32x16& v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.f;
}

template <int HD>
__device__ __forceinline__ void store_T_tile(bf16* dst_row, const f32x16 (&acc)[HD / 32], float mul,
                                             int h5) {
  // acc[db][reg]: row (d) = db*32 + 8*(reg>>2) + 4*h5 + (reg&3)
#pragma unroll
  for (int db = 0; db < HD / 32; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[db][4 * g + e] * mul);
      *reinterpret_cast<bf16x4*>(dst_row + db * 32 + 8 * g + 4 * h5) = o;
    }
}

// Column sums of a transposed accumulator over the wave's 32 rows (the lanes lr of each
// half), for the fused qkv-bias gradient.  One 32-row x 32-lane block (16 values per lane)
// at a time, to keep the register cost at 16: a transposing butterfly — at step s lane
// pairs (lr, lr ^ 2^s) split the remaining values between them and add — leaves ONE
// total per lane after 4 steps (15 shuffles instead of 4 per value); the fifth lane bit
// is folded by a plain add.  dst[d] (LDS, HD floats, this wave's) receives the sum of
// element d; rows with valid == false count 0.
__device__ __forceinline__ void colsum_T_block(float* dst32, const f32x16& acc, float mul, bool valid,
                                               int lr, int h5) {
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = valid ? acc[r] * mul : 0.f;
  int e = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int cnt = 8 >> s;
    const bool up = (lr >> s) & 1;
#pragma unroll
    for (int i = 0; i < cnt; ++i) {
      float lo = v[i], hi = v[i + cnt];
      // opaque copies: otherwise the two selects are folded into v[up ? i + cnt : i], a
      // dynamically indexed register array (a 16-way compare/select chain per access)
      asm volatile("" : "+v"(lo), "+v"(hi));
      const float keep = up ? hi : lo;
      const float send = up ? lo : hi;
      v[i] = keep + __shfl_xor(send, 1 << s, 64);
    }
    e += up ? cnt : 0;
  }
  v[0] += __shfl_xor(v[0], 16, 64);
  if (lr < 16) dst32[8 * (e >> 2) + 4 * h5 + (e & 3)] = v[0];
}
template <int HD>
__device__ __forceinline__ void colsum_T_tile(float* dst, const f32x16 (&acc)[HD / 32], float mul, bool valid,
                                              int lr, int h5) {
#pragma unroll
  for (int db = 0; db < HD / 32; ++db) colsum_T_block(dst + db * 32, acc[db], mul, valid, lr, h5);
}

// Column sums of row pieces held one per lane: lane = (row in group) * CPR + pc holds 8 consecutive
// columns (pc*8 ..) of its row(s) in v.  The same transposing butterfly over the row bits of the lane
// index: 7 shuffles, after which lane (rows-bits = e) holds the total of column pc*8 + e over the
// 8 rows (CPR = 8), or over 16 rows after one more add (CPR = 4: the lanes < 32 hold it).
template <int CPR>
__device__ __forceinline__ float piece_colsum8(float (&v)[8], int lane, int* col) {
  constexpr int LB = CPR == 8 ? 3 : 2;
  int e = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int cnt = 4 >> s;
    const bool up = (lane >> (LB + s)) & 1;
#pragma unroll
    for (int i = 0; i < cnt; ++i) {
      float lo = v[i], hi = v[i + cnt];
      asm volatile("" : "+v"(lo), "+v"(hi));     // see colsum_T_block
      const float keep = up ? hi : lo;
      const float send = up ? lo : hi;
      v[i] = keep + __shfl_xor(send, 1 << (LB + s), 64);
    }
    e += up ? cnt : 0;
  }
#pragma unroll
  for (int bit = LB + 3; bit < 6; ++bit) v[0] += __shfl_xor(v[0], 1 << bit, 64);
  *col = (lane % CPR) * 8 + e;
  return v[0];
}

// rows owned per workgroup / rows streamed per chunk for a block of nw waves
__device__ __forceinline__ int chunk_rows(int nw) { return 32 * (nw < 4 ? nw : 4); }

// ------------------------------------------------------------- forward ---
template <int HD>
__global__ __launch_bounds__(512) void attn_fwd_kernel(const bf16* __restrict__ qkv,
                                                       bf16* __restrict__ out,
                                                       float* __restrict__ lse, int N, int H,
                                                       float scale_log2e) {
  using C = AttnCfg<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kl = smem;
  char* Vl = smem + CHUNK_MAX * C::KS;
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int lr = lane & 31, h5 = lane >> 5;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int64_t ts = (int64_t)3 * H * HD;
  const bf16* qb = qkv + (int64_t)b * N * ts + h * HD;
  const bf16* kb_ = qb + H * HD;
  const bf16* vb = qb + 2 * H * HD;
  const int q0 = blockIdx.x * (32 * nw) + w * 32;
  const bool active = q0 < N;                 // wave-uniform: this wave owns >= 1 real query
  const int qrow = min(q0 + lr, N - 1);

  bf16x8 qf[C::KSTEPS];
#pragma unroll
  for (int s = 0; s < C::KSTEPS; ++s)
    qf[s] = *reinterpret_cast<const bf16x8*>(qb + (int64_t)qrow * ts + 16 * s + 8 * h5);

  f32x16 o[C::DB];
#pragma unroll
  for (int db = 0; db < C::DB; ++db) zero16(o[db]);
  float m = -INFINITY, l = 0.f;

  const int CK = chunk_rows(nw);
  const int nck = (N + CK - 1) / CK;
  bf16x8 kr[4], vr[4];
  prefetch_rows<HD>(kr, kb_, ts, 0, CK, N, tid, nthr);
  prefetch_rows<HD>(vr, vb, ts, 0, CK, N, tid, nthr);
  for (int c = 0; c < nck; ++c) {
    __syncthreads();                           // everyone is done reading the previous chunk
    commit_rows<HD, C::KS>(Kl, kr, CK, tid, nthr);
    commit_rows<HD, C::VS>(Vl, vr, CK, tid, nthr);
    __syncthreads();
    if (c + 1 < nck) {                         // in flight while this chunk is computed
      prefetch_rows<HD>(kr, kb_, ts, (c + 1) * CK, CK, N, tid, nthr);
      prefetch_rows<HD>(vr, vb, ts, (c + 1) * CK, CK, N, tid, nthr);
    }
    if (!active) continue;
    const int left = N - c * CK;
    const int nsb = ((left < CK ? left : CK) + 31) >> 5;
#pragma unroll 1
    for (int sb = 0; sb < nsb; ++sb) {
      f32x16 s;
      zero16(s);
#pragma unroll
      for (int ks = 0; ks < C::KSTEPS; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Kl + (sb * 32 + lr) * C::KS + (16 * ks + 8 * h5) * 2);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s, 0, 0, 0);
      }
      const int key0 = c * CK + sb * 32;
      if (key0 + 32 > N) {                     // ragged block only: mask keys >= N
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + mfma32_row(r, h5) >= N) s[r] = -INFINITY;
      }
      float tmax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
      for (int r = 4; r < 16; r += 2) tmax = fmaxf(tmax, fmaxf(s[r], s[r + 1]));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m, tmax * scale_log2e);
      if (!__all(m_new == m)) {                // some row's max moved: rescale once
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        l *= alpha;
#pragma unroll
        for (int db = 0; db < C::DB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        m = m_new;
      }
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m));
        s[r] = p;
        psum += p;
      }
      l += psum + __shfl_xor(psum, 32);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = pack8(s, 8 * s2);
#pragma unroll
        for (int db = 0; db < C::DB; ++db) {
          const bf16x8 a = load_tr_frag(Vl, C::VS, sb * 32 + 16 * s2, db * 32, lane);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf, o[db], 0, 0, 0);
        }
      }
    }
  }

  const int q = q0 + lr;
  if (q < N) {
    store_T_tile<HD>(out + ((int64_t)(b * (int64_t)N + q) * H + h) * HD, o, 1.f / l, h5);
    if (h5 == 0) lse[(int64_t)bh * N + q] = (m + log2f(l)) * LN2;
  }
}


// ------------------------------------ forward, whole sequence in one CU ---
// N <= 256: K and V of one (image, head) are staged ONCE (one batch of unconditional loads,
// one barrier) for all nw = ceil(N/32) waves of the workgroup; each wave then walks the key
// blocks with no further barrier.  Two such workgroups fit a CU (75 KB LDS, <= 128 VGPRs),
// so one's staging burst runs under the other's softmax.
template <int HD>
__global__ __launch_bounds__(512) void attn_fwd_whole_kernel(const bf16* __restrict__ qkv,
                                                             bf16* __restrict__ out,
                                                             float* __restrict__ lse, int N, int H,
                                                             float scale_log2e) {
  using C = AttnCfg<HD>;
  constexpr int CPR = HD / 8, IT = CPR / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int NP = nw * 32;
  char* Kl = smem;
  char* Vl = smem + NP * C::KS;
  const int lr = lane & 31, h5 = lane >> 5;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int64_t ts = (int64_t)3 * H * HD;
  const bf16* qb = qkv + (int64_t)b * N * ts + h * HD;
  const bf16* kb_ = qb + H * HD;
  const bf16* vb = qb + 2 * H * HD;
  const int q0 = w * 32;
  const int qrow = min(q0 + lr, N - 1);

  bf16x8 qf[C::KSTEPS];
#pragma unroll
  for (int s = 0; s < C::KSTEPS; ++s)
    qf[s] = *reinterpret_cast<const bf16x8*>(qb + (int64_t)qrow * ts + 16 * s + 8 * h5);
  {
    // NP*CPR 16-B pieces over nthr = NP*2 threads: CPR/2 per thread and matrix
    bf16x8 k8[IT], v8[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = tid + i * nthr, row = min(c / CPR, N - 1), pc = c % CPR;
      k8[i] = *reinterpret_cast<const bf16x8*>(kb_ + (int64_t)row * ts + pc * 8);
      v8[i] = *reinterpret_cast<const bf16x8*>(vb + (int64_t)row * ts + pc * 8);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = tid + i * nthr, row = c / CPR, pc = c % CPR;
      if (row >= N) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { k8[i][e] = (bf16)0.f; v8[i][e] = (bf16)0.f; }
      }
      *reinterpret_cast<bf16x8*>(Kl + row * C::KS + pc * 16) = k8[i];
      *reinterpret_cast<bf16x8*>(Vl + row * C::VS + pc * 16) = v8[i];
    }
  }
  __syncthreads();

  f32x16 o[C::DB];
#pragma unroll
  for (int db = 0; db < C::DB; ++db) zero16(o[db]);
  float m = -INFINITY, l = 0.f;
  const int nsb = (N + 31) >> 5;
#pragma unroll 1
  for (int sb = 0; sb < nsb; ++sb) {
    f32x16 s;
    zero16(s);
#pragma unroll
    for (int ks = 0; ks < C::KSTEPS; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Kl + (sb * 32 + lr) * C::KS + (16 * ks + 8 * h5) * 2);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s, 0, 0, 0);
    }
    const int key0 = sb * 32;
    if (key0 + 32 > N) {                       // ragged block only: mask keys >= N
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (key0 + mfma32_row(r, h5) >= N) s[r] = -INFINITY;
    }
    float tmax = max16(s);
    tmax = max3f(tmax, __shfl_xor(tmax, 32), tmax);
    const float m_new = fmaxf(m, tmax * scale_log2e);
    if (!__all(m_new == m)) {                  // some row's max moved: rescale once
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      l *= alpha;
#pragma unroll
      for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      m = m_new;
    }
    const float psum = exp2_scaled_sum16(s, scale_log2e, m);
    l += psum + __shfl_xor(psum, 32);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pf = pack8(s, 8 * s2);
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        const bf16x8 a = load_tr_frag(Vl, C::VS, sb * 32 + 16 * s2, db * 32, lane);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf, o[db], 0, 0, 0);
      }
    }
  }
  const int q = q0 + lr;
  if (q < N) {
    store_T_tile<HD>(out + ((int64_t)(b * (int64_t)N + q) * H + h) * HD, o, 1.f / l, h5);
    if (h5 == 0) lse[(int64_t)bh * N + q] = (m + log2f(l)) * LN2;
  }
}

// ------------------------------------------------------ backward: delta ---
// delta[b,h,n] = sum_d dout[b,n,h,d] * out[b,n,h,d]
template <int HD>
__global__ void attn_delta_kernel(const bf16* __restrict__ out, const bf16* __restrict__ dout,
                                  float* __restrict__ delta, int64_t rows, int N, int H) {
  constexpr int LPR = HD / 8;   // lanes per (b,n,h) row
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid / LPR;
  const int ch = (int)(gid % LPR);
  float s = 0.f;
  if (row < rows) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(out + row * HD + ch * 8);
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(dout + row * HD + ch * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)d[e];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (row < rows && ch == 0) {
    const int h = (int)(row % H);
    const int64_t bn = row / H;
    const int n = (int)(bn % N);
    const int64_t b = bn / N;
    delta[(b * H + h) * N + n] = s;
  }
}

// ------------------------------------------------- backward: dK and dV ---
template <int HD, bool DBIAS>
__global__ __launch_bounds__(512) void attn_bwd_dkdv_kernel(const bf16* __restrict__ qkv,
                                                            const bf16* __restrict__ dout,
                                                            const float* __restrict__ lse,
                                                            const float* __restrict__ delta,
                                                            bf16* __restrict__ dqkv, int N, int H,
                                                            float scale, float scale_log2e,
                                                            float* __restrict__ dbias_part) {
  using C = AttnCfg<HD>;
  constexpr int QS = C::KS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ql = smem;
  char* dOl = smem + CHUNK_MAX * QS;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * CHUNK_MAX * QS);
  float* del_s = lse_s + CHUNK_MAX;
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int lr = lane & 31, h5 = lane >> 5;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int64_t ts = (int64_t)3 * H * HD;
  const int64_t os = (int64_t)H * HD;
  const bf16* qb = qkv + (int64_t)b * N * ts + h * HD;
  const bf16* kb_ = qb + H * HD;
  const bf16* vb = qb + 2 * H * HD;
  const bf16* dob = dout + (int64_t)b * N * os + h * HD;
  const int key0 = blockIdx.x * (32 * nw) + w * 32;
  const bool active = key0 < N;
  const int krow = min(key0 + lr, N - 1);

  bf16x8 kf[C::KSTEPS], vf[C::KSTEPS];
#pragma unroll
  for (int s = 0; s < C::KSTEPS; ++s) {
    kf[s] = *reinterpret_cast<const bf16x8*>(kb_ + (int64_t)krow * ts + 16 * s + 8 * h5);
    vf[s] = *reinterpret_cast<const bf16x8*>(vb + (int64_t)krow * ts + 16 * s + 8 * h5);
  }
  f32x16 dk[C::DB], dv[C::DB];
#pragma unroll
  for (int db = 0; db < C::DB; ++db) { zero16(dk[db]); zero16(dv[db]); }

  const int CK = chunk_rows(nw);
  const int nck = (N + CK - 1) / CK;
  bf16x8 qr[4], dr[4];
  prefetch_rows<HD>(qr, qb, ts, 0, CK, N, tid, nthr);
  prefetch_rows<HD>(dr, dob, os, 0, CK, N, tid, nthr);
  for (int c = 0; c < nck; ++c) {
    __syncthreads();
    commit_rows<HD, QS>(Ql, qr, CK, tid, nthr);
    commit_rows<HD, QS>(dOl, dr, CK, tid, nthr);
    if (tid < CK) {
      const int q = c * CK + tid;
      lse_s[tid] = q < N ? lse[(int64_t)bh * N + q] * LOG2E : INFINITY;   // +inf -> p = 0 for padded queries
      del_s[tid] = q < N ? delta[(int64_t)bh * N + q] : 0.f;
    }
    __syncthreads();
    if (c + 1 < nck) {
      prefetch_rows<HD>(qr, qb, ts, (c + 1) * CK, CK, N, tid, nthr);
      prefetch_rows<HD>(dr, dob, os, (c + 1) * CK, CK, N, tid, nthr);
    }
    if (!active) continue;
    const int left = N - c * CK;
    const int nsb = ((left < CK ? left : CK) + 31) >> 5;
#pragma unroll 1
    for (int sb = 0; sb < nsb; ++sb) {
      f32x16 s, dp;
      zero16(s);
      zero16(dp);
#pragma unroll
      for (int ks = 0; ks < C::KSTEPS; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ql + (sb * 32 + lr) * QS + (16 * ks + 8 * h5) * 2);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf[ks], s, 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < C::KSTEPS; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(dOl + (sb * 32 + lr) * QS + (16 * ks + 8 * h5) * 2);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, vf[ks], dp, 0, 0, 0);
      }
      // rows = query index inside the sub-block, cols (lane) = key; the row constants
      // come in runs of 4 consecutive queries: 8*g + 4*h5 + {0..3}
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 ls = *reinterpret_cast<const f32x4*>(lse_s + sb * 32 + 8 * g + 4 * h5);
        const f32x4 de = *reinterpret_cast<const f32x4*>(del_s + sb * 32 + 8 * g + 4 * h5);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float p = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -ls[e]));
          s[r] = p;
          dp[r] = p * (dp[r] - de[e]);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = pack8(s, 8 * s2);
        const bf16x8 dsf = pack8(dp, 8 * s2);
#pragma unroll
        for (int db = 0; db < C::DB; ++db) {
          const bf16x8 a = load_tr_frag(dOl, QS, sb * 32 + 16 * s2, db * 32, lane);
          dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf, dv[db], 0, 0, 0);
          const bf16x8 a2 = load_tr_frag(Ql, QS, sb * 32 + 16 * s2, db * 32, lane);
          dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, dsf, dk[db], 0, 0, 0);
        }
      }
    }
  }
  const int key = key0 + lr;
  if (key < N) {
    bf16* row = dqkv + (int64_t)(b * (int64_t)N + key) * ts + h * HD;
    store_T_tile<HD>(row + H * HD, dk, scale, h5);
    store_T_tile<HD>(row + 2 * H * HD, dv, 1.f, h5);
  }
  if constexpr (DBIAS) {
    // per-(image, row block) column sums of dK and dV: the k and v thirds of the qkv bias
    // gradient, so that dqkv is not read back just to be summed
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();           // every wave is done with the staged Q / dO
    colsum_T_tile<HD>(red + (w * 2 + 0) * HD, dk, scale, active && key < N, lr, h5);
    colsum_T_tile<HD>(red + (w * 2 + 1) * HD, dv, 1.f, active && key < N, lr, h5);
    __syncthreads();
    for (int i = tid; i < 2 * HD; i += nthr) {
      const int which = i / HD, d = i % HD;
      float t = 0.f;
      for (int ww = 0; ww < nw; ++ww) t += red[(ww * 2 + which) * HD + d];
      dbias_part[((int64_t)b * gridDim.x + blockIdx.x) * ts + (1 + which) * H * HD + h * HD + d] = t;
    }
  }
}

// --------------------------------------------------------- backward: dQ ---
template <int HD, bool DBIAS>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const bf16* __restrict__ qkv,
                                                          const bf16* __restrict__ dout,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ delta,
                                                          bf16* __restrict__ dqkv, int N, int H,
                                                          float scale, float scale_log2e,
                                                          float* __restrict__ dbias_part) {
  using C = AttnCfg<HD>;
  constexpr int KS = C::KS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kl = smem;
  char* Vl = smem + CHUNK_MAX * KS;
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int lr = lane & 31, h5 = lane >> 5;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int64_t ts = (int64_t)3 * H * HD;
  const int64_t os = (int64_t)H * HD;
  const bf16* qb = qkv + (int64_t)b * N * ts + h * HD;
  const bf16* kb_ = qb + H * HD;
  const bf16* vb = qb + 2 * H * HD;
  const bf16* dob = dout + (int64_t)b * N * os + h * HD;
  const int q0 = blockIdx.x * (32 * nw) + w * 32;
  const bool active = q0 < N;
  const int qrow = min(q0 + lr, N - 1);

  bf16x8 qf[C::KSTEPS], dof[C::KSTEPS];
#pragma unroll
  for (int s = 0; s < C::KSTEPS; ++s) {
    qf[s] = *reinterpret_cast<const bf16x8*>(qb + (int64_t)qrow * ts + 16 * s + 8 * h5);
    dof[s] = *reinterpret_cast<const bf16x8*>(dob + (int64_t)qrow * os + 16 * s + 8 * h5);
  }
  const float lse_q = lse[(int64_t)bh * N + qrow] * LOG2E;
  const float del_q = delta[(int64_t)bh * N + qrow];
  f32x16 dq[C::DB];
#pragma unroll
  for (int db = 0; db < C::DB; ++db) zero16(dq[db]);

  const int CK = chunk_rows(nw);
  const int nck = (N + CK - 1) / CK;
  bf16x8 kr[4], vr[4];
  prefetch_rows<HD>(kr, kb_, ts, 0, CK, N, tid, nthr);
  prefetch_rows<HD>(vr, vb, ts, 0, CK, N, tid, nthr);
  for (int c = 0; c < nck; ++c) {
    __syncthreads();
    commit_rows<HD, KS>(Kl, kr, CK, tid, nthr);
    commit_rows<HD, KS>(Vl, vr, CK, tid, nthr);
    __syncthreads();
    if (c + 1 < nck) {
      prefetch_rows<HD>(kr, kb_, ts, (c + 1) * CK, CK, N, tid, nthr);
      prefetch_rows<HD>(vr, vb, ts, (c + 1) * CK, CK, N, tid, nthr);
    }
    if (!active) continue;
    const int left = N - c * CK;
    const int nsb = ((left < CK ? left : CK) + 31) >> 5;
#pragma unroll 1
    for (int sb = 0; sb < nsb; ++sb) {
      f32x16 st, dpt;
      zero16(st);
      zero16(dpt);
#pragma unroll
      for (int ks = 0; ks < C::KSTEPS; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Kl + (sb * 32 + lr) * KS + (16 * ks + 8 * h5) * 2);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], st, 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < C::KSTEPS; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Vl + (sb * 32 + lr) * KS + (16 * ks + 8 * h5) * 2);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, dof[ks], dpt, 0, 0, 0);
      }
      // zero-filled K/V rows (key >= N) give a finite p and dS that multiply a zero
      // K^T row below, so no masking is needed
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[r], scale_log2e, -lse_q));
        st[r] = p * (dpt[r] - del_q);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 dsf = pack8(st, 8 * s2);
#pragma unroll
        for (int db = 0; db < C::DB; ++db) {
          const bf16x8 a = load_tr_frag(Kl, KS, sb * 32 + 16 * s2, db * 32, lane);
          dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, dsf, dq[db], 0, 0, 0);
        }
      }
    }
  }
  const int q = q0 + lr;
  if (q < N) store_T_tile<HD>(dqkv + (int64_t)(b * (int64_t)N + q) * ts + h * HD, dq, scale, h5);
  if constexpr (DBIAS) {
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();
    colsum_T_tile<HD>(red + w * HD, dq, scale, active && q < N, lr, h5);
    __syncthreads();
    for (int i = tid; i < HD; i += nthr) {
      float t = 0.f;
      for (int ww = 0; ww < nw; ++ww) t += red[ww * HD + i];
      dbias_part[((int64_t)b * gridDim.x + blockIdx.x) * ts + h * HD + i] = t;
    }
  }
}


// ------------------------------------- backward, whole sequence in one CU ---
// For N <= 256 one workgroup holds Q, dO and K of a whole (image, head) in LDS and wave w owns
// key block w (K and V fragments in registers) AND query block w's dQ: FIVE products per
// (query block, key block) pair instead of the seven of the dkdv + dq pair, one exp per
// score instead of two, no delta kernel (rowsum(dO*O) is taken while dO is staged).
//   S = Q K^T, dP = dO V^T           key on the lane (as in dkdv)
//   dV^T += dO^T P, dK^T += Q^T dS   accumulators stay in registers (own keys)
//   dQ^T(w) += K(w')^T dS(w, w')^T   accumulators stay in registers (own queries)
// At step t wave w computes the pair (query block (w + t) mod nw, key block w) and leaves dS in its
// LDS tile [key][q]; after the step's barrier the OWNER of that query block — wave (w + t) mod nw —
// reads the tile back transposed (ds_read_b64_tr_b16) together with K^T fragments of key block w
// from the K image and adds the product to its dQ accumulators.  Seen from wave w: its source at
// step t is wave (w - t) mod nw.  The tiles are double buffered (a tile of step t is read while
// step t + 1's is written), so one barrier per step orders everything; every sum has a fixed
// order (deterministic).  (Round 1-2a kept an fp32 dQ image in LDS instead and read-modify-wrote
// it every step: 16 of the 36 KB of LDS traffic per wave and step, and 61 of 145 KB of LDS.)
template <int HD> struct FusedBwdCfg {
  static constexpr int QS = AttnCfg<HD>::KS;     // staged Q / dO / K row (bytes)
  static constexpr int TPITCH = 80;              // dS tile row pitch (bytes): 64 B of queries + pad against bank conflicts
  static constexpr int ROW_BYTES = 3 * QS + 8 + 2 * TPITCH;   // + lse, delta, two dS tile shares
};

template <int HD, bool DBIAS>
__global__ __launch_bounds__(512) void attn_bwd_fused_kernel(const bf16* __restrict__ qkv,
                                                             const bf16* __restrict__ out,
                                                             const bf16* __restrict__ dout,
                                                             const float* __restrict__ lse,
                                                             bf16* __restrict__ dqkv, int N, int H,
                                                             int npairs, float scale, float scale_log2e,
                                                             float* __restrict__ dbias_part,
                                                             unsigned long long* dbg) {
  using C = AttnCfg<HD>;
  using F = FusedBwdCfg<HD>;
  constexpr int QS = F::QS, CPR = HD / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int lr = lane & 31, h5 = lane >> 5;
  const int NP = nw * 32;
  char* Ql = smem;
  char* dOl = Ql + NP * QS;
  char* Kl = dOl + NP * QS;
  float* lse_s = reinterpret_cast<float*>(Kl + NP * QS);
  float* del_s = lse_s + NP;
  char* Tbase = reinterpret_cast<char*>(del_s + NP);              // [2][nw] dS tiles of 32 x TPITCH bytes
  const int64_t ts = (int64_t)3 * H * HD;
  const int64_t os = (int64_t)H * HD;
  const int key = w * 32 + lr;
  const int krow = min(key, N - 1);
  const bool kvalid = key < N;

  // ---- PERSISTENT walk over (image, head) pairs: a workgroup owns the whole CU (LDS), so the
  // only thing that can run beside its memory phases is its own next pair.  The global loads
  // of pair i+1 (Q, K, dO, O pieces, lse, the V fragments: one batch, registers) are issued
  // right after the loop of pair i, BEFORE its dQ / dK / dV rows are written out: the HBM
  // round trip of the staging runs under the store phase instead of after it.
  // NP*CPR pieces of 16 B over nthr = NP*2 threads = CPR/2 pieces per thread and matrix.
  constexpr int IT = CPR / 2;
  bf16x8 vf[C::KSTEPS];                            // B[k = d][n = key] of dP = dO V^T
  bf16x8 q8[IT], d8[IT], o8[IT], k8[IT];
  float lv[IT];
  auto issue_loads = [&](int bh) {
    const int b = bh / H, h = bh % H;
    const bf16* qb = qkv + (int64_t)b * N * ts + h * HD;
    const bf16* kb_ = qb + H * HD;
    const bf16* dob = dout + (int64_t)b * N * os + h * HD;
    const bf16* ob = out + (int64_t)b * N * os + h * HD;
    // the per-lane offsets are the same for every pair: hidden from the optimizer, or it keeps
    // ~40 registers of hoisted addresses alive across the main loop (spills)
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      // unconditional loads from a clamped row (a branch around a load makes hipcc wait for
      // each one separately); rows >= N are zeroed by the selects of the commit
      const int c = tid_o + i * nthr, row = min(c / CPR, N - 1), pc = c % CPR;
      q8[i] = *reinterpret_cast<const bf16x8*>(qb + (int64_t)row * ts + pc * 8);
      k8[i] = *reinterpret_cast<const bf16x8*>(kb_ + (int64_t)row * ts + pc * 8);
      d8[i] = *reinterpret_cast<const bf16x8*>(dob + (int64_t)row * os + pc * 8);
      o8[i] = *reinterpret_cast<const bf16x8*>(ob + (int64_t)row * os + pc * 8);
      lv[i] = lse[(int64_t)bh * N + row];         // with the batch: a load in the store loop would drain it each time
    }
    // hipcc sinks a load whose only use sits under `if (pc == 0)` into that branch, behind the
    // wait for the batch: a second HBM round trip.  An opaque use keeps all of them up here.
#pragma unroll
    for (int i = 0; i < IT; ++i) asm volatile("" : "+v"(lv[i]));
  };

  int bh = blockIdx.x;
  if (bh >= npairs) return;
  issue_loads(bh);

#pragma unroll 1
  for (;;) {
  const int b = bh / H, h = bh % H;
  // diagnostic timeline (armed by tools/attn_bench.py --stamps only): 64 mid-launch pairs
  const int dbg_slot = bh - npairs / 2;            // steady state, not the cold start
  const bool dbg_on = dbg != nullptr && dbg_slot >= 0 && dbg_slot < 64;
  unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (dbg_on) tl[0] = attn_stamp();
  // V fragments of the own keys straight from memory (first used in the main loop: their round
  // trip runs under the commit; staged a pair ahead they would cost 16 registers across it)
  {
    const bf16* vb = qkv + (int64_t)b * N * ts + (2 * H + h) * HD;
    int krow_o = krow;
    asm volatile("" : "+v"(krow_o));
#pragma unroll
    for (int s = 0; s < C::KSTEPS; ++s)
      vf[s] = *reinterpret_cast<const bf16x8*>(vb + (int64_t)krow_o * ts + 16 * s + 8 * h5);
  }

  // ---- commit the staged registers: Q, dO (and delta = rowsum(dO * O)), K; rows >= N are zero
  // (per-lane offsets of the memory phases derive from an opaque copy of tid: left visible, the
  // optimizer hoists them out of the pair loop and they sit in registers across the main loop)
  int tid_c = tid;
  asm volatile("" : "+v"(tid_c));
  float dosum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = tid_c + i * nthr, row = c / CPR, pc = c % CPR;
    if (row >= N) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { q8[i][e] = (bf16)0.f; d8[i][e] = (bf16)0.f; k8[i][e] = (bf16)0.f; }
    }
    if constexpr (DBIAS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) dosum[e] += (float)d8[i][e];
    }
    *reinterpret_cast<bf16x8*>(Ql + row * QS + pc * 16) = q8[i];
    *reinterpret_cast<bf16x8*>(dOl + row * QS + pc * 16) = d8[i];
    *reinterpret_cast<bf16x8*>(Kl + row * QS + pc * 16) = k8[i];
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dot = fmaf((float)d8[i][e], (float)o8[i][e], dot);
#pragma unroll
    for (int off = 1; off < CPR; off <<= 1) dot += __shfl_xor(dot, off, 64);   // CPR consecutive lanes = one row
    if (pc == 0) {
      del_s[row] = dot;
      lse_s[row] = row < N ? lv[i] * LOG2E : INFINITY;    // +inf -> p = 0 for padded queries
    }
  }
  // qkv-bias gradient of this (image, head) without touching the accumulators:
  //   v: sum_k dV[k,:] = (P 1)^T dO = 1^T dO   (softmax rows sum to one): column sums of dO, taken
  //      here from the staged pieces (one value per lane, kept across the main loop);
  //   k: sum_k dK[k,:] = sum_q (sum_k dS[q,k]) Q[q,:] = 0   (sum_k P (dP - delta) = delta - delta);
  //   q: column sums of dQ, taken while its rows are converted for the store.
  float dvb = 0.f;
  int dvb_col = 0;
  if constexpr (DBIAS) dvb = piece_colsum8<CPR>(dosum, tid_c & 63, &dvb_col);
  __syncthreads();
  if (dbg_on) tl[1] = attn_stamp();
  bf16x8 kf[C::KSTEPS];                            // B[k = d][n = key] of S = Q K^T
#pragma unroll
  for (int s = 0; s < C::KSTEPS; ++s)
    kf[s] = *reinterpret_cast<const bf16x8*>(Kl + key * QS + (16 * s + 8 * h5) * 2);

  f32x16 dk[C::DB], dv[C::DB], dq[C::DB];          // own keys; own queries (rows d, lane = key / query)
#pragma unroll
  for (int db = 0; db < C::DB; ++db) { zero16(dk[db]); zero16(dv[db]); zero16(dq[db]); }
  if (dbg_on) tl[2] = attn_stamp();

#pragma unroll 1
  for (int t = 0; t < nw; ++t) {
    int j = w + t;
    if (j >= nw) j -= nw;
    char* Tw = Tbase + ((t & 1) * nw + w) * 32 * F::TPITCH;      // this step's dS tile of this wave
    // Every LDS read of the step that does not depend on this step's dS is issued up
    // front: the compiler cannot move a read of Ql / dOl above the T-tile stores
    // (may alias), and would otherwise pay one LDS round trip per product.
    bf16x8 aq[C::KSTEPS], ado[C::KSTEPS];
#pragma unroll
    for (int ks = 0; ks < C::KSTEPS; ++ks) {
      aq[ks] = *reinterpret_cast<const bf16x8*>(Ql + (j * 32 + lr) * QS + (16 * ks + 8 * h5) * 2);
      ado[ks] = *reinterpret_cast<const bf16x8*>(dOl + (j * 32 + lr) * QS + (16 * ks + 8 * h5) * 2);
    }
    f32x4 ls[4], de[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      ls[g] = *reinterpret_cast<const f32x4*>(lse_s + j * 32 + 8 * g + 4 * h5);
      de[g] = *reinterpret_cast<const f32x4*>(del_s + j * 32 + 8 * g + 4 * h5);
    }
    bf16x8 doT[2][C::DB], qT[2][C::DB];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        doT[s2][db] = load_tr_frag(dOl, QS, j * 32 + 16 * s2, db * 32, lane);
        qT[s2][db] = load_tr_frag(Ql, QS, j * 32 + 16 * s2, db * 32, lane);
      }
    f32x16 s, dp;
    zero16(s);
    zero16(dp);
#pragma unroll
    for (int ks = 0; ks < C::KSTEPS; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks], kf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ado[ks], vf[ks], dp, 0, 0, 0);
    }
    // rows = query inside the block (runs of 4: 8*g + 4*h5 + e), lane = key
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -ls[g][e]));
      }
    }
    // zero-filled K rows (keys >= N) give p = exp(-lse) != 0: only the last wave owns such
    // keys, so the 16 selects sit behind a wave-uniform branch (loop 26 k -> 22.6 k cycles)
    if (w == nw - 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = kvalid ? s[r] : 0.f;
    }
    // (round 3: the same arithmetic on packed fp32 instructions — v_pk_fma / v_pk_add / v_pk_mul, 24 instead of 48 per
    // step — measured SLOWER: loop 17.9 k -> 18.8 k cycles, 217 -> 224 us per layer; packed VALU beside MFMAs is an anti-lever)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        dp[r] = s[r] * (dp[r] - de[g][e]);
      }
    }
    // dS -> this wave's tile T[key][q] (bf16), 4 consecutive queries per store
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = (bf16)dp[4 * g + e];
      *reinterpret_cast<bf16x4*>(Tw + lr * F::TPITCH + (8 * g + 4 * h5) * 2) = o4;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pf = pack8(s, 8 * s2);
      const bf16x8 dsf = pack8(dp, 8 * s2);
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT[s2][db], pf, dv[db], 0, 0, 0);
        dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT[s2][db], dsf, dk[db], 0, 0, 0);
      }
    }
    __syncthreads();                               // every dS tile of step t is written
    // dQ^T(w)[d][q] += K(w')^T[d][key] dS(w, w')^T[key][q],  w' = (w - t) mod nw
    int wsrc = w - t;
    if (wsrc < 0) wsrc += nw;
    const char* Ts = Tbase + ((t & 1) * nw + wsrc) * 32 * F::TPITCH;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 bT = load_tr_frag(Ts, F::TPITCH, 16 * s2, 0, lane);
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        const bf16x8 kT = load_tr_frag(Kl, QS, wsrc * 32 + 16 * s2, db * 32, lane);
        dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT, bT, dq[db], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                 // the last step's K^T / tile reads are done: all three images are dead

  if (dbg_on) tl[3] = attn_stamp();
  // dQ, dK, dV: transposed accumulators (lane = query / key, 4 consecutive d per register run) -> bf16
  // rows in the dead K / Q / dO images, so that every global store below is a 16-B piece of a
  // 128-B row segment (a lane-per-row store tail is issue-bound: guide, 'epilogue store tail').
  int tid_s = tid;
  asm volatile("" : "+v"(tid_s));
  const int lr_s = tid_s & 31, h5_s = (tid_s >> 5) & 1, key_s = w * 32 + lr_s;
  store_T_tile<HD>(reinterpret_cast<bf16*>(Kl + key_s * QS), dq, scale, h5_s);
  store_T_tile<HD>(reinterpret_cast<bf16*>(Ql + key_s * QS), dk, scale, h5_s);
  store_T_tile<HD>(reinterpret_cast<bf16*>(dOl + key_s * QS), dv, 1.f, h5_s);
  float* red = reinterpret_cast<float*>(Tbase);    // the dS tiles are dead: [2][nw][HD] partial sums
  __syncthreads();
  const int bh_next = bh + (int)gridDim.x;
  // unconditional (the last pair loads itself again, unused): under a branch the staged registers
  // of THIS pair would stay live across the main loop as the other input of the merge (spills)
  if (dbg_on) tl[5] = attn_stamp();               // rows committed to LDS + barrier
  issue_loads(min(bh_next, npairs - 1));
  if (dbg_on) tl[6] = attn_stamp();               // next pair's loads issued
  float dqsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c = tid_s; c < N * CPR; c += nthr) {
    const int row = c / CPR, pc = c % CPR;
    bf16* grow = dqkv + (int64_t)(b * (int64_t)N + row) * ts + h * HD + pc * 8;
    const bf16x8 q8s = *reinterpret_cast<const bf16x8*>(Kl + row * QS + pc * 16);
    if constexpr (DBIAS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) dqsum[e] += (float)q8s[e];
    }
    *reinterpret_cast<bf16x8*>(grow) = q8s;
    *reinterpret_cast<bf16x8*>(grow + H * HD) = *reinterpret_cast<const bf16x8*>(Ql + row * QS + pc * 16);
    *reinterpret_cast<bf16x8*>(grow + 2 * H * HD) = *reinterpret_cast<const bf16x8*>(dOl + row * QS + pc * 16);
  }
  if (dbg_on) tl[7] = attn_stamp();               // store loop issued
  if constexpr (DBIAS) {
    int dqb_col;
    const float dqb = piece_colsum8<CPR>(dqsum, tid_s & 63, &dqb_col);
    if (CPR == 8 || (tid_s & 63) < 32) {
      red[w * HD + dqb_col] = dqb;
      red[(nw + w) * HD + dvb_col] = dvb;
    }
    __syncthreads();
    for (int i = tid_s; i < 3 * HD; i += nthr) {
      const int which = i / HD, d = i % HD;
      float t = 0.f;
      if (which != 1)
        for (int ww = 0; ww < nw; ++ww) t += red[((which >> 1) * nw + ww) * HD + d];
      dbias_part[(int64_t)b * ts + which * H * HD + h * HD + d] = t;
    }
  }
  if (dbg_on) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tl[4] = attn_stamp();
    if (tid == 0)
      for (int i = 0; i < 8; ++i) dbg[dbg_slot * 8 + i] = tl[i];
  }
  if (bh_next >= npairs) break;
  bh = bh_next;
  __syncthreads();                                 // every read of the images is done: the commit may overwrite them
  }
}

inline int attn_waves(int64_t N) {
  int64_t nw = ((N < 256 ? N : 256) + 31) / 32;
  return (int)(nw < 1 ? 1 : nw);
}

}  // namespace

static int check_attn(const void* qkv, int dtype, int64_t B, int64_t N, int64_t H, int64_t hd, const char* who) {
  VITMI_REQUIRE(qkv && B > 0 && N > 0 && H > 0, VITMI_E_BADARG, "%s: null pointer or empty shape", who);
  VITMI_REQUIRE(dtype == VITMI_BF16, VITMI_E_DTYPE, "%s: this kernel is bf16 (fp32 goes to vitmi_attn_*_f32 via dtype dispatch)", who);
  VITMI_REQUIRE(hd == 32 || hd == 64, VITMI_E_SHAPE, "%s: head dim %lld not in {32, 64}", who, (long long)hd);
  VITMI_REQUIRE(B * H <= 65535, VITMI_E_SHAPE, "%s: B*H = %lld exceeds the grid limit 65535", who, (long long)(B * H));
  VITMI_REQUIRE(is_aligned(qkv, 16), VITMI_E_ALIGN, "%s: qkv must be 16-B aligned", who);
  return 0;
}

int attn_fwd_f32(const float* qkv, float* out, float* lse, int64_t B, int64_t N, int64_t H, int64_t hd, float scale, hipStream_t stream);
int attn_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int64_t B, int64_t N, int64_t H, int64_t hd, float scale, float* delta, hipStream_t stream);

static std::atomic<int> g_attn_fwd_waves{0};     // diagnostic hook: waves (32 queries each) per forward workgroup, 0 = default
extern "C" void vitmi_debug_attn_fwd_waves(int n) { g_attn_fwd_waves = n; }

extern "C" int vitmi_attn_fwd(const void* qkv, void* out, float* lse, int dtype, int64_t B,
                              int64_t N, int64_t H, int64_t hd, float scale, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VITMI_REQUIRE(out && lse, VITMI_E_BADARG, "attn_fwd: null out/lse");
  if (dtype == VITMI_F32) return attn_fwd_f32((const float*)qkv, (float*)out, lse, B, N, H, hd, scale, stream);
  int rc = check_attn(qkv, dtype, B, N, H, hd, "attn_fwd");
  if (rc) return rc;
  VITMI_REQUIRE(is_aligned(out, 8), VITMI_E_ALIGN, "attn_fwd: out must be 8-B aligned");
  if (N <= 256 && g_attn_fwd_waves <= 0) {          // whole sequence resident: one workgroup per (image, head)
    const int nwh = attn_waves(N);
    const size_t ldsw = (size_t)nwh * 32 * (hd == 64 ? (AttnCfg<64>::KS + AttnCfg<64>::VS) : (AttnCfg<32>::KS + AttnCfg<32>::VS));
    if (hd == 64) {
      auto kern = attn_fwd_whole_kernel<64>;
      if ((rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(kern), 96 * 1024, "attn_fwd"))) return rc;
      hipLaunchKernelGGL(kern, dim3((unsigned)(B * H)), dim3(64 * nwh), ldsw, stream, (const bf16*)qkv, (bf16*)out, lse, (int)N, (int)H, scale * LOG2E);
    } else {
      auto kern = attn_fwd_whole_kernel<32>;
      if ((rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(kern), 96 * 1024, "attn_fwd"))) return rc;
      hipLaunchKernelGGL(kern, dim3((unsigned)(B * H)), dim3(64 * nwh), ldsw, stream, (const bf16*)qkv, (bf16*)out, lse, (int)N, (int)H, scale * LOG2E);
    }
    return vitmi_check_launch("attn_fwd_whole_kernel");
  }
  // 4-wave workgroups (128 queries): three of them fit a CU (registers), so the K/V staging
  // of one overlaps the softmax of the others; measured 9 % faster than one 7-wave
  // workgroup per (image, head) at N = 197 although K/V are then staged twice
  int nw = attn_waves(N);
  const int fw_ = g_attn_fwd_waves; const int cap = fw_ > 0 ? fw_ : 4;
  if (cap < nw) nw = cap;
  dim3 grid((unsigned)((N + 32 * nw - 1) / (32 * nw)), (unsigned)(B * H));
  if (hd == 64) {
    const size_t lds = CHUNK_MAX * (AttnCfg<64>::KS + AttnCfg<64>::VS);
    hipLaunchKernelGGL((attn_fwd_kernel<64>), grid, dim3(64 * nw), lds, stream, (const bf16*)qkv, (bf16*)out, lse, (int)N, (int)H, scale * LOG2E);
  } else {
    const size_t lds = CHUNK_MAX * (AttnCfg<32>::KS + AttnCfg<32>::VS);
    hipLaunchKernelGGL((attn_fwd_kernel<32>), grid, dim3(64 * nw), lds, stream, (const bf16*)qkv, (bf16*)out, lse, (int)N, (int)H, scale * LOG2E);
  }
  return vitmi_check_launch("attn_fwd_kernel");
}

extern "C" size_t vitmi_attn_bwd_workspace(int64_t B, int64_t N, int64_t H) {
  return (size_t)(B * N * H) * sizeof(float);
}

static std::atomic<unsigned long long*> g_attn_dbg{nullptr};

extern "C" void vitmi_debug_attn_stamps(void* p) { g_attn_dbg = reinterpret_cast<unsigned long long*>(p); }
static std::atomic<int> g_attn_bwd_mode{-1};     // diagnostic / test hook: 0 = dkdv + dq kernels, 1 = fused where possible
extern "C" void vitmi_debug_attn_bwd(int mode) { g_attn_bwd_mode = mode; }
static bool attn_bwd_fused_ok(int64_t N, int64_t hd) {
  if (g_attn_bwd_mode == 0) return false;
  const int nw = attn_waves(N);
  const size_t lds = (size_t)nw * 32 * (hd == 64 ? FusedBwdCfg<64>::ROW_BYTES : FusedBwdCfg<32>::ROW_BYTES);
  return N <= 256 && lds <= 160 * 1024;
}

extern "C" int64_t vitmi_attn_bwd_dbias_rows(int64_t B, int64_t N) {
  const int nw = attn_waves(N);
  return B * ((N + 32 * nw - 1) / (32 * nw));
}

extern "C" int vitmi_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                              void* dqkv, int dtype, int64_t B, int64_t N, int64_t H, int64_t hd,
                              float scale, float* dbias_part, int32_t launch_flags, void* workspace,
                              size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VITMI_REQUIRE(out && dout && lse && dqkv, VITMI_E_BADARG, "attn_bwd: null argument");
  VITMI_REQUIRE(!dbias_part || dtype == VITMI_BF16, VITMI_E_DTYPE, "attn_bwd: dbias_part is produced by the bf16 kernels only");
  VITMI_REQUIRE(workspace && workspace_bytes >= vitmi_attn_bwd_workspace(B, N, H), VITMI_E_WORKSPACE, "attn_bwd: workspace too small");
  float* delta = reinterpret_cast<float*>(workspace);
  if (dtype == VITMI_F32)
    return attn_bwd_f32((const float*)qkv, (const float*)out, (const float*)dout, lse, (float*)dqkv, B, N, H, hd, scale, delta, stream);
  int rc = check_attn(qkv, dtype, B, N, H, hd, "attn_bwd");
  if (rc) return rc;
  VITMI_REQUIRE(is_aligned(out, 16) && is_aligned(dout, 16) && is_aligned(dqkv, 8), VITMI_E_ALIGN, "attn_bwd: out/dout must be 16-B, dqkv 8-B aligned");
  const int64_t rows = B * N * H;
  const int nw = attn_waves(N);
  if (attn_bwd_fused_ok(N, hd)) {
    // one workgroup per CU walking pairs bh, bh + grid, ... (see the kernel); one pair per workgroup
    // when the call carries VITMI_LAUNCH_SHARED_DEVICE (data-parallel runs: RCCL kernels hold CUs)
    const int64_t grid_f = vitmi_persist_on(launch_flags) && B * H > vitmi_cu_count() ? vitmi_cu_count() : B * H;
#define LAUNCH_FUSED(HDV, DB)                                                                            \
    do {                                                                                                 \
      auto kern = attn_bwd_fused_kernel<HDV, DB>;                                                        \
      const size_t lds = (size_t)nw * 32 * FusedBwdCfg<HDV>::ROW_BYTES;                                  \
      if (int rc_ = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, "attn_bwd")) return rc_; \
      hipLaunchKernelGGL(kern, dim3((unsigned)grid_f), dim3(64 * nw), lds, stream, (const bf16*)qkv,     \
                         (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, (int)N, (int)H,          \
                         (int)(B * H), scale, scale * LOG2E, dbias_part, g_attn_dbg.load());                    \
    } while (0)
    if (hd == 64) { if (dbias_part) LAUNCH_FUSED(64, true); else LAUNCH_FUSED(64, false); }
    else          { if (dbias_part) LAUNCH_FUSED(32, true); else LAUNCH_FUSED(32, false); }
#undef LAUNCH_FUSED
    return vitmi_check_launch("attn_bwd_fused_kernel");
  }
  dim3 grid((unsigned)((N + 32 * nw - 1) / (32 * nw)), (unsigned)(B * H));
#define LAUNCH_BWD(HDV, DB)                                                                              \
  do {                                                                                                   \
    const int64_t threads = rows * (HDV / 8);                                                            \
    hipLaunchKernelGGL((attn_delta_kernel<HDV>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,  \
                       stream, (const bf16*)out, (const bf16*)dout, delta, rows, (int)N, (int)H);        \
    rc = vitmi_check_launch("attn_delta_kernel");                                                        \
    if (rc) return rc;                                                                                   \
    const size_t lds_a = 2 * CHUNK_MAX * AttnCfg<HDV>::KS + 2 * CHUNK_MAX * sizeof(float);               \
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<HDV, DB>), grid, dim3(64 * nw), lds_a, stream,              \
                       (const bf16*)qkv, (const bf16*)dout, lse, delta, (bf16*)dqkv, (int)N, (int)H,     \
                       scale, scale * LOG2E, dbias_part);                                                \
    rc = vitmi_check_launch("attn_bwd_dkdv_kernel");                                                     \
    if (rc) return rc;                                                                                   \
    const size_t lds_b = 2 * CHUNK_MAX * AttnCfg<HDV>::KS;                                               \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HDV, DB>), grid, dim3(64 * nw), lds_b, stream,                \
                       (const bf16*)qkv, (const bf16*)dout, lse, delta, (bf16*)dqkv, (int)N, (int)H,     \
                       scale, scale * LOG2E, dbias_part);                                                \
    rc = vitmi_check_launch("attn_bwd_dq_kernel");                                                       \
  } while (0)
  if (hd == 64) { if (dbias_part) LAUNCH_BWD(64, true); else LAUNCH_BWD(64, false); }
  else          { if (dbias_part) LAUNCH_BWD(32, true); else LAUNCH_BWD(32, false); }
#undef LAUNCH_BWD
  return rc;
}

// every diagnostic switch of this file back to its default (vitmi_debug_reset, core.cpp)
void vitmi_debug_reset_attention() {
  g_attn_fwd_waves = 0;
  g_attn_dbg = nullptr;
  g_attn_bwd_mode = -1;
}
