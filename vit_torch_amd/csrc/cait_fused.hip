// Fused talking-heads attention of CaiT (/root/reference/models/cait.py:111-128), bf16 operands / fp32 accumulation:
//
//     S_h  = (q_h * scale) k_h^T              per head h                         (:114-116)
//     S'_g = sum_h Wl[g,h] S_h + bl[g]        proj_l, a Linear over the HEAD axis (:118)
//     P_g  = softmax_j(S'_g)                                                      (:120)
//     P'_g = sum_h Ww[g,h] P_h + bw[g]        proj_w                              (:122)
//     O_g  = P'_g v_g                                                             (:125)
//
// The head mixes couple all heads of a score position, so the three-call form (gemm_small q k^T -> th_softmax ->
// gemm_small P' v) keeps S, P and P' as [B,H,N,N] tensors in HBM: 73 GB of the 169 GB a cait_S24_224 step moves at
// batch 256, 21 of its 49 ms.  Here the scores never leave the CU.
//
// One workgroup = one image x RB query rows x ALL heads; 8 waves.  The score rows live in LDS as bf16 PLANES
// [row][head][key] (448 B per head, 3600 B per row: 8 planes + 16 B so that row-strided 16-B reads are conflict-free):
//   phase S   wave h = head h: S_h^T[key][q] on the matrix pipe straight from global-memory fragments (q, k rows are
//             contiguous in d: 16-B loads per lane, all issued up front), 4 consecutive keys per lane -> ds_write_b64;
//   phase R   wave w = rows: BOTH HEAD MIXES ON THE MATRIX PIPE.  ds_read_b64_tr_b16 turns the planes of one row into
//             the A operand "16 positions x (set, head)" of v_mfma_f32_16x16x32_bf16; the B operand is the 8x8 mixing
//             matrix, block-diagonal over two position sets (k-octets 0 / 1) with its bf16 rounding residual in
//             k-octets 2 / 3 (fp32-accurate weights for free).  A lane then owns ONE head and 28 positions of the
//             row: max / sum are 27 local operations + 3 cross-lane steps; P is written back into the row's planes
//             (8-B writes), mixed again (proj_w) and written back as P';
//   phase PV  wave g = head g: O_g^T[d][q] = V_g^T P'_g^T with V^T fragments from a transposed copy of V in HBM
//             (th_pack_kernel: a [key][d] operand cannot give a lane 8 consecutive KEYS) and P' fragments as
//             16-B reads of the planes; 4 consecutive d per lane -> 8-B stores.
// The round-2 fused forward (tools/experiments/cait_fused_fwd.hip.txt) did the mixes as 128 per-lane FMAs per score
// position and loaded K / V fragments piecemeal: VALU- and latency-bound, slower than the three calls.
//
// Backward (th_attn_bwd_kernel, RB = 16): recomputes S, S', P from q, k (nothing but O is kept by the forward), forms
// dP' = dO v^T on the matrix pipe, runs the transposed mixes, the softmax backward and the four parameter gradients
// (matrix pipe, planes as both operands), writes dQ, and hands dS and P' to HBM for the two remaining batched
// products (dK = scale dS^T q, dV = P'^T dO).
#include <atomic>
#include <type_traits>
#include "common.h"

namespace {

constexpr int THD = 48;               // head dimension of every CaiT variant (embed_dim / num_heads)
constexpr int TNH = 8;                // heads
constexpr int NKP = 224;              // key slots of a score row: 7 blocks of 32
constexpr int PLANE = NKP * 2;        // bytes of one head's plane of a row
constexpr int ROWB = TNH * PLANE + 16;   // 3600 B per score row
constexpr float TLOG2E = 1.4426950408889634f;

__device__ __forceinline__ unsigned long long th_stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
// phase stamps: compiled in only with -DVITMI_TH_STAMPS (tools/th_attn_bench.py's breakdown); as a run-time option every
// stamp was a branch on the buffer pointer around the phase boundaries of the production kernels
#ifdef VITMI_TH_STAMPS
#define TH_STAMP(k) do { if (dbg && threadIdx.x == 0) dbg[(int64_t)blockIdx.x * 8 + (k)] = th_stamp(); } while (0)
#else
#define TH_STAMP(k) do { } while (0)
#endif
__device__ __forceinline__ bf16x4 tr4(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p));
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 lo, bf16x4 hi) {
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16k16(bf16x4 a, bf16x4 b, f32x4 c) {     // K = 16: lane (i | j = lane & 15, kq = lane >> 4) holds k = 4 kq + e
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (bf16)0.f;
  return r;
}

// B operand of a head mix: B[k = 8 kq + e][j = n].  Column n = (position set n >> 3, output head n & 7); k-octet kq:
// position set kq & 1, input head e; octets 0 / 1 carry bf16(W), octets 2 / 3 its rounding residual.  transpose: the
// mix runs through W^T (backward).
__device__ __forceinline__ bf16x8 mix_operand(const float* __restrict__ W, bool transpose, int lane) {
  const int n = lane & 15, kq = lane >> 4;
  // every lane loads its eight weights unconditionally and selects arithmetically: with the selection as a condition hipcc
  // guarded each load with its own exec-mask branch and vmcnt(0) — 32 serialized round trips at the top of the kernel
  float w[8];
  if (!transpose) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(W + (n & 7) * TNH), b = *reinterpret_cast<const f32x4*>(W + (n & 7) * TNH + 4);
    w[0] = a[0]; w[1] = a[1]; w[2] = a[2]; w[3] = a[3]; w[4] = b[0]; w[5] = b[1]; w[6] = b[2]; w[7] = b[3];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = W[e * TNH + (n & 7)];
  }
  const bool mine = (kq & 1) == (n >> 3);
  const float s_hi = (mine && kq < 2) ? 1.f : 0.f, s_lo = (mine && kq >= 2) ? 1.f : 0.f;
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float hi = (float)(bf16)w[e];
    r[e] = (bf16)(s_hi * hi + s_lo * (w[e] - hi));
  }
  return r;
}

// One row's planes -> mixed scores: lane (n = lane & 15, pq = lane >> 4) gets d[t][r] = sum_h W[g,h] X_h[pos], head g =
// n & 7, pos = 32 t + 16 (n >> 3) + 4 pq + r.
__device__ __forceinline__ void mix_row(const char* row, bf16x8 wop, f32x4 (&d)[7], int lane) {
  const int i = lane & 15, kq = lane >> 4;
  const char* p = row + (i >> 2) * PLANE + (16 * (kq & 1) + 4 * (i & 3)) * 2;
  bf16x4 lo[7], hi[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) { lo[t] = tr4(p + 64 * t); hi[t] = tr4(p + 64 * t + 4 * PLANE); }
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    d[t] = mfma16(cat8(lo[t], hi[t]), wop, z);
  }
}
// the lane's 28 values back into the row's planes (plane g, 4 consecutive positions per tile)
__device__ __forceinline__ void store_row(char* row, const f32x4 (&d)[7], int lane) {
  const int n = lane & 15, pq = lane >> 4;
  char* p = row + (n & 7) * PLANE + (16 * (n >> 3) + 4 * pq) * 2;
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    bf16x4 v = {(bf16)d[t][0], (bf16)d[t][1], (bf16)d[t][2], (bf16)d[t][3]};
    *reinterpret_cast<bf16x4*>(p + 64 * t) = v;
  }
}
// reductions over the 8 lanes that hold one head: n ^ 8 inside the 16-lane row (DPP row_ror:8), then the four rows
// (v_permlane16_swap / v_permlane32_swap of a register with itself: rows 0|1 and 2|3, then the two halves, exchanged on
// the VALU — a ds_bpermute round trip per step made each reduction ~250 cycles of a dependent chain)
#ifdef TH_SHFL
__device__ __forceinline__ float head_max(float v) {
  v = fmaxf(v, dpp_take<0x128, 0xf>(v, v));
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}
__device__ __forceinline__ float head_sum(float v) {
  v += dpp_take<0x128, 0xf>(v, v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
#else
template <bool SWAP32> __device__ __forceinline__ void lane_swap(float v, float* a, float* b) {
  float x = v, y = v;
  // from inline asm: through the builtin, hipcc (ROCm 7.2) dropped the combine of the two results (the fmaxf / add was
  // missing from the ISA whether or not it could prove the operands equal; found by the parity test).  s_nop 1 = the wait
  // states hipcc itself puts between a VALU write and the swap.
  if constexpr (SWAP32) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  *a = x; *b = y;
}
__device__ __forceinline__ float head_max(float v) {
  float a, b;
  v = fmaxf(v, dpp_take<0x128, 0xf>(v, v));
  lane_swap<false>(v, &a, &b);
  v = fmaxf(a, b);
  lane_swap<true>(v, &a, &b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float head_sum(float v) {
  float a, b;
  v += dpp_take<0x128, 0xf>(v, v);
  lane_swap<false>(v, &a, &b);
  v = a + b;
  lane_swap<true>(v, &a, &b);
  return a + b;
}
#endif
// softmax over the row of the lane's head, in place; positions >= N are masked
__device__ __forceinline__ void softmax_row(f32x4 (&d)[7], float bias, int N, int lane) {
  const int n = lane & 15, pq = lane >> 4;
  const int p0 = 16 * (n >> 3) + 4 * pq;
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    if (32 * t + 32 <= N) {                    // wave-uniform: a whole tile of valid positions needs no per-position mask (as
#pragma unroll                                  // 28 hoisted compare results the masks cost 56 SGPRs and spilled)
      for (int r = 0; r < 4; ++r) { d[t][r] += bias; m = fmaxf(m, d[t][r]); }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = (32 * t + p0 + r < N) ? d[t][r] + bias : -INFINITY;
        d[t][r] = s;
        m = fmaxf(m, s);
      }
    }
  }
  m = head_max(m);
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __builtin_amdgcn_exp2f((d[t][r] - m) * TLOG2E);
      d[t][r] = e;
      l += e;
    }
  l = head_sum(l);
  const float inv = 1.f / l;
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) d[t][r] *= inv;
}

// ------------------------------------------------------------------------------------ fragment-major operand copies ---
// The products of this op take their k / v operands as matrix-pipe FRAGMENTS straight from global memory (the score rows
// own the LDS).  Loaded from the qkv tensor itself, a fragment is 64 lanes x 16 B out of 16-32 different token rows: 32
// cache lines per wave instruction, and the forward's two operand phases were bound by the L1 tag rate (14.5 k + 10.5 k
// of its 49 k cycles per workgroup, round 4 stamps).  th_pack_kernel rewrites the operands once per call in the order
// the lanes consume them, so that every fragment load is ONE contiguous KiB (8 full lines):
//   rows-type RF[bh][kb 0..13][ks 0..1][lane][8]  = X[token 16 kb + (lane & 15)][d = 32 ks + 8 (lane >> 4) + e]   (d >= 48, token >= N: 0)
//                                                   A operand of the score-shaped products (contraction over d, hd padded to 64)
//   T-type    TF[bh][db 0..2][ks 0..6][lane][8]   = X[token 32 ks + 8 (lane >> 4) + e][d = 16 db + (lane & 15)]   (token >= N: 0)
//                                                   A operand of the products that contract over TOKENS (P' v, dS k): a [token][d]
//                                                   operand cannot give a lane 8 consecutive tokens
//   RF16[bh][kb 0..13][ks 0..2][lane][4]          = X[token 16 kb + (lane & 15)][d = 16 ks + 4 (lane >> 4) + e]: the same operand for
//                                                   v_mfma_f32_16x16x16_bf16 (8 B per lane, hd = 48 exactly: the forward keeps
//                                                   K resident in registers, 84 of them)
constexpr int RF_FRAGS = 28, TF_FRAGS = 21, RF16_FRAGS = 42;
constexpr int RF_BYTES = RF_FRAGS * 1024, TF_BYTES = TF_FRAGS * 1024, RF16_BYTES = RF16_FRAGS * 512;
constexpr int PK_LD = 56;             // LDS row of the staged operand: 48 d + 8 zero columns
template <bool BWD>
__global__ __launch_bounds__(256) void th_pack_kernel(const bf16* __restrict__ qkv, char* __restrict__ rfK, char* __restrict__ rfV,
                                                      char* __restrict__ tf, int N, int H) {
  __shared__ __attribute__((aligned(16))) bf16 Kt[NKP][PK_LD];
  __shared__ __attribute__((aligned(16))) bf16 Vt[NKP][PK_LD];
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int64_t ts = (int64_t)3 * H * THD;
  const bf16* kp = qkv + (int64_t)b * N * ts + (H + h) * THD;
  const bf16* vp = kp + H * THD;
  for (int c = threadIdx.x; c < NKP * (PK_LD / 8); c += 256) {
    const int key = c / (PK_LD / 8), pc = c % (PK_LD / 8);
    bf16x8 k8 = zero8(), v8 = zero8();
    if (key < N && pc < THD / 8) {
      k8 = *reinterpret_cast<const bf16x8*>(kp + (int64_t)key * ts + pc * 8);
      v8 = *reinterpret_cast<const bf16x8*>(vp + (int64_t)key * ts + pc * 8);
    }
    *reinterpret_cast<bf16x8*>(&Kt[key][pc * 8]) = k8;
    *reinterpret_cast<bf16x8*>(&Vt[key][pc * 8]) = v8;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = lane & 15, kq = lane >> 4;
  for (int f = wv; f < RF16_FRAGS; f += 4) {                        // rows-type: K (and V for the backward)
    const int kbk = f / 3, ks = f % 3;
    *reinterpret_cast<bf16x4*>(rfK + ((int64_t)bh * RF16_FRAGS + f) * 512 + lane * 8) = *reinterpret_cast<const bf16x4*>(&Kt[16 * kbk + n][16 * ks + 4 * kq]);
    if (BWD) *reinterpret_cast<bf16x4*>(rfV + ((int64_t)bh * RF16_FRAGS + f) * 512 + lane * 8) = *reinterpret_cast<const bf16x4*>(&Vt[16 * kbk + n][16 * ks + 4 * kq]);
  }
  for (int f = wv; f < TF_FRAGS; f += 4) {                          // T-type: V (forward) / K (backward)
    const int db = f / 7, ks = f % 7;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = BWD ? Kt[32 * ks + 8 * kq + e][16 * db + n] : Vt[32 * ks + 8 * kq + e][16 * db + n];
    *reinterpret_cast<bf16x8*>(tf + ((int64_t)bh * TF_FRAGS + f) * 1024 + lane * 16) = v;
  }
}

// Fragment load at wave-uniform base + 32-bit lane offset.  The offset is made opaque per load: given the constant part,
// hipcc materialised a 64-bit per-lane address for every one of the ~100 fragment offsets of the backward kernel and
// hoisted them out of the block loop — 210 registers of addresses, 238 dwords of scratch.  One v_add per load instead.
template <typename T> __device__ __forceinline__ T ldg_u(const char* ubase, unsigned lane_off, unsigned const_off) {
  asm volatile("" : "+v"(lane_off));                                // the constant is added BEHIND the opaque point: not loop-invariant
  return *reinterpret_cast<const T*>(ubase + (lane_off + const_off));
}
// ... with a wave-uniform run-time offset as well (the rotated fragment order): opaque too, or each of the ~50 rotated bases
// becomes a hoisted SGPR pair
template <typename T> __device__ __forceinline__ T ldg_uu(const char* ubase, int uoff, unsigned lane_off, unsigned const_off) {
  asm volatile("" : "+s"(uoff));
  asm volatile("" : "+v"(lane_off));
  return *reinterpret_cast<const T*>(ubase + uoff + (lane_off + const_off));
}

// d-slot fragment of a 16-row operand block with hd = 48 padded to 64: k-step ks, octet kq -> d = 32 ks + 8 kq (zero beyond 48)
__device__ __forceinline__ bf16x8 load_d8(const bf16* rowp, int ks, int kq) {
  const int d0 = 32 * ks + 8 * kq;
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(rowp + min(d0, THD - 8));      // unconditional load, then select
  return d0 < THD ? v : zero8();
}

// -------------------------------------------------------------------------------------------------- forward ---
// One workgroup walks `nblk` consecutive blocks of FRB = 32 query rows of ONE image (all 7 when the batch fills the chip: one
// workgroup per CU).  Wave h keeps its head's K (84 registers) and V^T (84) fragments for the whole walk: the operands are
// read once per image, not once per block (7 x 340 KB per image through a ~25 B/clk L2 -> CU path made the two operand
// phases 31 k of the first version's 49 k cycles per block).  Per block: S (matrix pipe, registers -> planes) | barrier |
// R (rows) | barrier | PV.  The next block's S needs no barrier: a wave writes only the planes of its own head, which only
// it reads in PV.
constexpr int FRB = 32;               // query rows per block
__global__ __launch_bounds__(512) void th_attn_fwd_kernel(const bf16* __restrict__ qkv, const char* __restrict__ rfK, const char* __restrict__ tfV,
                                                          const float* __restrict__ Wl, const float* __restrict__ bl,
                                                          const float* __restrict__ Ww, const float* __restrict__ bw,
                                                          bf16* __restrict__ out, int N, float scale, int nblk,
                                                          unsigned long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // FRB rows of ROWB bytes
  TH_STAMP(0);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nqb = (N + FRB - 1) / FRB;
  const int wpi = (nqb + nblk - 1) / nblk;                         // workgroups per image
  const int b = blockIdx.x / wpi;
  const int qb0 = (blockIdx.x % wpi) * nblk, qb1 = min(qb0 + nblk, nqb);
  const int64_t ts = 3 * TNH * THD;
  const int n = lane & 15, kq = lane >> 4;
  const int bh = b * TNH + w;                                      // phases S / PV: this wave's head

  // resident operands of head w
  bf16x4 kf[14][3];
  bf16x8 vf[3][7];
  {
    const char* kfp = rfK + (int64_t)bh * RF16_BYTES + lane * 8;
#pragma unroll
    for (int kbk = 0; kbk < 14; ++kbk)
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) kf[kbk][ks] = *reinterpret_cast<const bf16x4*>(kfp + (kbk * 3 + ks) * 512);
    const char* vfp = tfV + (int64_t)bh * TF_BYTES + lane * 16;
#pragma unroll
    for (int db = 0; db < 3; ++db)
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) vf[db][ks] = *reinterpret_cast<const bf16x8*>(vfp + (db * 7 + ks) * 1024);
  }
  auto load_q = [&](int qb, bf16x4 (&qf)[2][3]) {
#pragma unroll
    for (int qbk = 0; qbk < 2; ++qbk) {
      const bf16* qp = qkv + ((int64_t)b * N + min(qb * FRB + 16 * qbk + n, N - 1)) * ts + w * THD + 4 * kq;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) qf[qbk][ks] = *reinterpret_cast<const bf16x4*>(qp + 16 * ks);
    }
  };
  bf16x4 qf[2][3];
  load_q(qb0, qf);
  const bf16x8 wl = mix_operand(Wl, false, lane), ww = mix_operand(Ww, false, lane);
  const float b_l = bl[lane & 7], b_w = bw[lane & 7];
#pragma unroll 1
  for (int qb = qb0; qb < qb1; ++qb) {
    const int q0 = qb * FRB;
    if (qb == qb0) TH_STAMP(1);
    if (qb == qb0 + 1) TH_STAMP(6);
    // ---- phase S: S_w^T[key][q] = K Q^T, scaled, into plane w of every row
    {
      char* col = smem + n * ROWB + w * PLANE + 8 * kq;            // row = query n (+ 16 qbk), keys 16 kb + 4 kq + r
#pragma unroll
      for (int kbk = 0; kbk < 14; ++kbk) {
#pragma unroll
        for (int qbk = 0; qbk < 2; ++qbk) {
          f32x4 sv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) sv = mfma16k16(kf[kbk][ks], qf[qbk][ks], sv);
          bf16x4 v = {(bf16)(sv[0] * scale), (bf16)(sv[1] * scale), (bf16)(sv[2] * scale), (bf16)(sv[3] * scale)};
          *reinterpret_cast<bf16x4*>(col + 16 * qbk * ROWB + 32 * kbk) = v;
        }
      }
    }
    if (qb + 1 < qb1) load_q(qb + 1, qf);                           // the next block's query fragments travel under phase R
    if (qb == qb0) TH_STAMP(2);
    __syncthreads();
    if (qb == qb0) TH_STAMP(3);

    // ---- phase R: rows 4 w .. 4 w + 3: proj_l -> softmax -> proj_w, in place
#pragma unroll 1
    for (int rr = 0; rr < FRB / 8; ++rr) {
      const int i = w * (FRB / 8) + rr;
      if (q0 + i >= N) break;                                       // wave-uniform
      char* row = smem + i * ROWB;
      f32x4 d[7];
      mix_row(row, wl, d, lane);
      softmax_row(d, b_l, N, lane);
      store_row(row, d, lane);                                     // P
      mix_row(row, ww, d, lane);
#pragma unroll
      for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[t][r] += b_w;
      store_row(row, d, lane);                                     // P'
    }
    if (qb == qb0) TH_STAMP(4);
    __syncthreads();
    if (qb == qb0) TH_STAMP(5);

    // ---- phase PV: O_w^T[d][q] = V_w^T P'_w^T
    {
      f32x4 acc[2][3];
#pragma unroll
      for (int qbk = 0; qbk < 2; ++qbk)
#pragma unroll
        for (int db = 0; db < 3; ++db) { acc[qbk][db][0] = 0.f; acc[qbk][db][1] = 0.f; acc[qbk][db][2] = 0.f; acc[qbk][db][3] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) {
#pragma unroll
        for (int qbk = 0; qbk < 2; ++qbk) {
          const bf16x8 pf = *reinterpret_cast<const bf16x8*>(smem + (16 * qbk + n) * ROWB + w * PLANE + (32 * ks + 8 * kq) * 2);
#pragma unroll
          for (int db = 0; db < 3; ++db) acc[qbk][db] = mfma16(vf[db][ks], pf, acc[qbk][db]);
        }
      }
#pragma unroll
      for (int qbk = 0; qbk < 2; ++qbk) {
        const int q = q0 + 16 * qbk + n;
        if (q < N) {
          bf16* o = out + ((int64_t)(b * (int64_t)N + q) * TNH + w) * THD + 4 * kq;
#pragma unroll
          for (int db = 0; db < 3; ++db) {
            bf16x4 v = {(bf16)acc[qbk][db][0], (bf16)acc[qbk][db][1], (bf16)acc[qbk][db][2], (bf16)acc[qbk][db][3]};
            *reinterpret_cast<bf16x4*>(o + 16 * db) = v;
          }
        }
      }
    }
  }
  TH_STAMP(7);
}

// ------------------------------------------------------------------------------------------------- backward ---
constexpr int BRB = 16;               // query rows per workgroup: TWO score arrays (S and dP') + a scratch row per wave
constexpr int B_SA = 0, B_DA = BRB * ROWB, B_SC = 2 * BRB * ROWB, B_LDS = 2 * BRB * ROWB + 8 * ROWB + 512;
constexpr int TH_PART = 2 * TNH * TNH + 2 * TNH;     // dWl | dbl | dWw | dbw per workgroup

__global__ __launch_bounds__(512) void th_attn_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                          const char* __restrict__ rfK, const char* __restrict__ rfV,
                                                          const float* __restrict__ Wl,
                                                          const float* __restrict__ bl, const float* __restrict__ Ww,
                                                          const float* __restrict__ bw, bf16* __restrict__ dqkv,
                                                          bf16* __restrict__ dS_out, bf16* __restrict__ Pm_out, int64_t ld,
                                                          float* __restrict__ part, int N, float scale, int nblk, int nimg,
                                                          unsigned long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TH_STAMP(0);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nqb = (N + BRB - 1) / BRB;
  const int wpi = (nqb + nblk - 1) / nblk;                         // workgroups per image
  // XCD-aware map: workgroups bid, bid + 8, ... share an XCD (and its L2); all workgroups of an image go to ONE XCD, so the
  // image's fragment-major operands (504 KB) are fetched into that L2 once and shared by its row blocks.  (One workgroup
  // walking a whole image, as in the forward, re-read them from beyond L2 for every block: 256 images in flight do not fit
  // 8 x 4 MiB — 29 k + 31 k cycles of the 82 k per block.)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int b = (slot / wpi) * 8 + xcd;
  if (b >= nimg) return;                                            // (grid padded to whole XCD rows; before any barrier)
  const int qb0 = (slot % wpi) * nblk, qb1 = min(qb0 + nblk, nqb);
  const int64_t ts = 3 * TNH * THD, tso = TNH * THD;
  const int n = lane & 15, kq = lane >> 4;
  const int bh = b * TNH + w;

  // (a resident K, as in the forward, does not fit: with the row phase's ~170 live registers hipcc spilled 158 dwords;
  // K, V and K^T fragments are re-read per block from the fragment-major copies, one contiguous 0.5 - 1 KiB per load)
  const char* kfu = rfK + (int64_t)bh * RF16_BYTES;                  // wave-uniform bases of this head's fragment-major operands
  const char* vfu = rfV + (int64_t)bh * RF16_BYTES;
  auto load_qd = [&](int qb, bf16x4 (&qf)[3], bf16x4 (&df)[3]) {
    const int qrow = min(qb * BRB + n, N - 1);
    const bf16* qp = qkv + ((int64_t)b * N + qrow) * ts + w * THD + 4 * kq;
    const bf16* dop = dout + ((int64_t)b * N + qrow) * tso + w * THD + 4 * kq;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) { qf[ks] = *reinterpret_cast<const bf16x4*>(qp + 16 * ks); df[ks] = *reinterpret_cast<const bf16x4*>(dop + 16 * ks); }
  };
  bf16x4 qf[3], df[3];
  load_qd(qb0, qf, df);
  f32x4 gWw = {0.f, 0.f, 0.f, 0.f}, gWl = {0.f, 0.f, 0.f, 0.f};   // lane (n < 8, pq < 2): [g = 4 pq + r][h = n]
  float dbw_acc = 0.f;
  const float b_l = bl[lane & 7], b_w = bw[lane & 7];
  const bf16x8 wl = mix_operand(Wl, false, lane), ww = mix_operand(Ww, false, lane);
  const bf16x8 wlT = mix_operand(Wl, true, lane), wwT = mix_operand(Ww, true, lane);
  char* sc = smem + B_SC + w * ROWB;                                // this wave's scratch row (planes)
  const int p0 = 16 * (n >> 3) + 4 * kq;                            // D-layout position base of the lane (pq = kq)
#pragma unroll 1
  for (int qb = qb0; qb < qb1; ++qb) {
    const int q0 = qb * BRB;
    if (qb == qb0) TH_STAMP(1);
    if (qb == qb0 + 1) TH_STAMP(6);
    // ---- phase S: head w: S^T[key][q] = K Q^T (scaled) and dP'^T[key][q] = V dO^T into planes w of the two arrays
    {
      char* sa = smem + B_SA + n * ROWB + w * PLANE + 8 * kq;      // row = query n, keys 16 kb + 4 kq + r
      char* da = smem + B_DA + n * ROWB + w * PLANE + 8 * kq;
      const float qvalid = (q0 + n < N) ? 1.f : 0.f;
      // the row blocks of an image run side by side on one XCD and would ask its L2 for the same fragment at the same
      // moment: block qb starts at key block qb (mod 14) — every tile is independent, the order is free
      auto rotk = [&](int k) { const int x = k + qb; return x >= 14 ? x - 14 : x; };
      auto s_half = [&](auto half_tag) {                            // 7 key blocks per batch of K / V fragment loads
        constexpr int HALF = decltype(half_tag)::value;              // compile-time: a run-time index would put kf[] into scratch
        bf16x4 kf[7][3], vf[7][3];
#pragma unroll
        for (int kk = 0; kk < 7; ++kk)
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) {
            kf[kk][ks] = ldg_uu<bf16x4>(kfu, rotk(7 * HALF + kk) * 1536, (unsigned)(lane * 8), (unsigned)(ks * 512));
            vf[kk][ks] = ldg_uu<bf16x4>(vfu, rotk(7 * HALF + kk) * 1536, (unsigned)(lane * 8), (unsigned)(ks * 512));
          }
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) {
          constexpr int dummy = 0; (void)dummy;
          const int kbk = rotk(7 * HALF + kk);
          f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) { sv = mfma16k16(kf[kk][ks], qf[ks], sv); dp = mfma16k16(vf[kk][ks], df[ks], dp); }
          bf16x4 s4 = {(bf16)(sv[0] * scale), (bf16)(sv[1] * scale), (bf16)(sv[2] * scale), (bf16)(sv[3] * scale)};
          bf16x4 d4 = {(bf16)dp[0], (bf16)dp[1], (bf16)dp[2], (bf16)dp[3]};
          dbw_acc += qvalid * ((dp[0] + dp[1]) + (dp[2] + dp[3]));   // d bw[w] = sum of dP'_w over valid (q, key): V's padded keys are zero rows
          *reinterpret_cast<bf16x4*>(sa + 32 * kbk) = s4;
          *reinterpret_cast<bf16x4*>(da + 32 * kbk) = d4;
        }
      };
      s_half(std::integral_constant<int, 0>{});
      asm volatile("" ::: "memory");                                // keep the second batch of loads behind the first batch's products:
      __builtin_amdgcn_sched_barrier(0);                            // hoisted together, 168 fragment registers spilled (157 dwords of scratch)
      s_half(std::integral_constant<int, 1>{});
    }
    if (qb == qb0) TH_STAMP(2);                                 // (stamps 1..7: the SECOND block of the walk, steady state)
    __syncthreads();
    if (qb == qb0) TH_STAMP(3);

    // ---- phase R: rows 2 w, 2 w + 1
#pragma unroll 1
    for (int rr = 0; rr < BRB / 8; ++rr) {
      const int i = w * (BRB / 8) + rr;
      if (q0 + i >= N) break;
      char* srow = smem + B_SA + i * ROWB;
      char* drow = smem + B_DA + i * ROWB;
      f32x4 P[7], dP[7];
      mix_row(srow, wl, P, lane);
      softmax_row(P, b_l, N, lane);                                 // P (fp32, lane's head)
      store_row(sc, P, lane);                                       // P planes -> scratch
      mix_row(drow, wwT, dP, lane);                                 // dP_h = sum_g Ww[g,h] dP'_g
      float dl = 0.f;
#pragma unroll
      for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dl += dP[t][r] * P[t][r];
      dl = head_sum(dl);
      // dS' = P (dP - delta), kept in dP's registers
#pragma unroll
      for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dP[t][r] = P[t][r] * (dP[t][r] - dl);
      // parameter gradients, part 1: dWw[g,h] += sum_pos dP'_g P_h  (dbw comes from phase S)
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(drow + (n & 7) * PLANE + (32 * t + 8 * kq) * 2);
        const bf16x8 bq = *reinterpret_cast<const bf16x8*>(sc + (n & 7) * PLANE + (32 * t + 8 * kq) * 2);
        gWw = mfma16(a, bq, gWw);
      }
      // P' = proj_w(P) -> HBM for the dV product
      {
        f32x4 pm[7];
        mix_row(sc, ww, pm, lane);
        bf16* po = Pm_out + (((int64_t)b * TNH + (n & 7)) * N + q0 + i) * ld + p0;
#pragma unroll
        for (int t = 0; t < 7; ++t) {                               // all 224 key slots: ld >= NKP, no per-tile branch around the store
          bf16x4 v = {(bf16)(pm[t][0] + b_w), (bf16)(pm[t][1] + b_w), (bf16)(pm[t][2] + b_w), (bf16)(pm[t][3] + b_w)};
          *reinterpret_cast<bf16x4*>(po + 32 * t) = v;
        }
      }
      store_row(sc, dP, lane);                                      // dS' planes -> scratch
      // parameter gradients, part 2: dWl[g,h] += sum_pos dS'_g S_h
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(sc + (n & 7) * PLANE + (32 * t + 8 * kq) * 2);
        const bf16x8 bq = *reinterpret_cast<const bf16x8*>(srow + (n & 7) * PLANE + (32 * t + 8 * kq) * 2);
        gWl = mfma16(a, bq, gWl);
      }
      // dS_h = sum_g Wl[g,h] dS'_g -> HBM (th_attn_prod_kernel: dQ, dK)
      {
        f32x4 ds[7];
        mix_row(sc, wlT, ds, lane);
        bf16* so = dS_out + (((int64_t)b * TNH + (n & 7)) * N + q0 + i) * ld + p0;
#pragma unroll
        for (int t = 0; t < 7; ++t) {
          bf16x4 v = {(bf16)ds[t][0], (bf16)ds[t][1], (bf16)ds[t][2], (bf16)ds[t][3]};
          *reinterpret_cast<bf16x4*>(so + 32 * t) = v;
        }
      }
    }
    if (qb == qb0) TH_STAMP(4);
    __syncthreads();
    if (qb == qb0) TH_STAMP(5);

  }
  // ---- per-workgroup partial sums of the parameter gradients: lanes (n < 8 | n == 8, pq < 2) hold [g = 4 pq + r][n]
  __syncthreads();                    // every wave is done with the arrays: the S array's first bytes become 8 slots of 160 floats
  {
    // one slot per wave, summed in wave order below: deterministic (LDS float atomics would add in arrival order)
    float* slot = reinterpret_cast<float*>(smem + B_SA) + w * 160;
    if (kq < 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int g = 4 * kq + r;
        if (n < 8) { slot[g * TNH + n] = gWl[r]; slot[TNH * TNH + TNH + g * TNH + n] = gWw[r]; }
      }
    }
    const float dbw_w = wave_sum(dbw_acc);                          // this wave's head
    if (lane < 16) slot[(lane < 8 ? TNH * TNH : 2 * TNH * TNH + TNH) + (lane & 7)] = (lane >= 8 && (lane & 7) == w) ? dbw_w : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < TH_PART) {
    const float* slots = reinterpret_cast<const float*>(smem + B_SA);
    float sum = 0.f;
#pragma unroll
    for (int ww_ = 0; ww_ < 8; ++ww_) sum += slots[ww_ * 160 + threadIdx.x];
    part[((int64_t)b * wpi + slot % wpi) * TH_PART + threadIdx.x] = sum;
  }
  TH_STAMP(7);
}

// ------------------------------------------------------------------------- backward: the three operand gradients ---
// dQ = scale dS K, dK = scale dS^T Q, dV = P'^T dO for one (image, head) per workgroup, from the dS / P' rows the row kernel
// left in HBM (each read ONCE; the two batched gemm_small calls read them at 2.3 TB/s and took 150 us per layer, and dQ
// inside the row kernel cost it a third operand phase).  Half of the [query][key] image at a time sits in LDS (464-B rows,
// 64.5 KB with the q / dO rows: two workgroups per CU, one's staging under the other's products); the products
// that contract over QUERIES take both operands through ds_read_b64_tr_b16 from row-major [query][.] images (the
// fragment of v_mfma_f32_16x16x16_bf16 is exactly what one transposing read delivers); dQ contracts over keys: K^T
// fragments from the fragment-major copy, dS rows as plain 16-B reads.
constexpr int PH = NKP / 2;                           // query rows staged at a time: the contraction over queries runs in two halves
constexpr int PS_PITCH = NKP * 2 + 16, PQ_PITCH = THD * 2 + 16;
constexpr int P_IMG_S = 0, P_IMG_Q = PH * PS_PITCH, P_LDS = PH * PS_PITCH + PH * PQ_PITCH;     // 64.5 KB: two workgroups per CU
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void th_attn_prod_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                           const bf16* __restrict__ dS, const bf16* __restrict__ Pm, int64_t ld,
                                                           const char* __restrict__ tfK, bf16* __restrict__ dqkv, int N, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x, b = bh / TNH, h = bh % TNH;
  const int64_t ts = 3 * TNH * THD, tso = TNH * THD;
  const int n = lane & 15, kq = lane >> 4;
  char* imgS = smem + P_IMG_S;
  char* imgQ = smem + P_IMG_Q;
  // rows [r0, r0 + PH) of a [N][cols] bf16 matrix (row stride rs elements) -> LDS image with `pitch`-byte rows, rows >= N zeroed
  // (all loads of a batch are issued before the first store: as a load -> store loop every 16-B piece paid its own round trip)
  auto stage = [&](char* img, int pitch, const bf16* src, int64_t rs, int r0, auto cols8_tag) {
    constexpr int COLS8 = decltype(cols8_tag)::value;
    constexpr int IT = (PH * COLS8 + 511) / 512;
#pragma unroll
    for (int i0 = 0; i0 < IT; i0 += 4) {
      bf16x8 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i0 + i < IT) {
          const int c = min(tid + (i0 + i) * 512, PH * COLS8 - 1);
          v[i] = *reinterpret_cast<const bf16x8*>(src + (int64_t)min(r0 + c / COLS8, N - 1) * rs + (c % COLS8) * 8);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = tid + (i0 + i) * 512;
        if (i0 + i < IT && c < PH * COLS8) {
          const int row = c / COLS8;
          *reinterpret_cast<bf16x8*>(img + row * pitch + (c % COLS8) * 16) = r0 + row < N ? v[i] : zero8();
        }
      }
    }
  };
  // this wave's column blocks of the [d][key] products: cb = w and w + 8 (14 blocks: waves 6, 7 have one); all three d-blocks
  // of a column block share its B fragments
  f32x4 acc[2][3];
  auto zero_acc = [&]() {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int db = 0; db < 3; ++db) { acc[u][db][0] = 0.f; acc[u][db][1] = 0.f; acc[u][db][2] = 0.f; acc[u][db][3] = 0.f; }
  };
  // acc[u][db] += sum over the PH staged query rows of X[q][16 db + .] Y[q][16 cb_u + .]   (X in imgQ, Y in imgS)
  auto contract_q = [&]() {
    const int i = lane & 15;
    const char* pa = imgQ + (4 * kq + (i >> 2)) * PQ_PITCH + (4 * (i & 3)) * 2;
    const char* pb = imgS + (4 * kq + (i >> 2)) * PS_PITCH + (4 * (i & 3)) * 2;
#pragma unroll 1
    for (int qb = 0; qb < PH / 16; ++qb) {
      bf16x4 a[3];
#pragma unroll
      for (int db = 0; db < 3; ++db) a[db] = tr4(pa + qb * 16 * PQ_PITCH + 32 * db);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int cb = w + 8 * u;
        if (cb < 14) {                                               // wave-uniform
          const bf16x4 bfr = tr4(pb + qb * 16 * PS_PITCH + 32 * cb);
#pragma unroll
          for (int db = 0; db < 3; ++db) acc[u][db] = mfma16k16(a[db], bfr, acc[u][db]);
        }
      }
    }
  };
  auto store_cols = [&](int slot_off, float mul) {                  // acc[u][db] = D^T[d = 16 db + 4 kq + r][key = 16 cb_u + n]
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int key = 16 * (w + 8 * u) + n;
      if (w + 8 * u < 14 && key < N) {
#pragma unroll
        for (int db = 0; db < 3; ++db) {
          bf16x4 v = {(bf16)(acc[u][db][0] * mul), (bf16)(acc[u][db][1] * mul), (bf16)(acc[u][db][2] * mul), (bf16)(acc[u][db][3] * mul)};
          *reinterpret_cast<bf16x4*>(dqkv + ((int64_t)b * N + key) * ts + slot_off + h * THD + 16 * db + 4 * kq) = v;
        }
      }
    }
  };
  // dQ: wave w owns d-block w % 3 (its 7 K^T fragments stay in 28 registers) and every (waves on that d-block)-th row block
  const int qdb = w % 3, qfirst = w / 3, qstep = qdb < 2 ? 3 : 2;
  bf16x8 ktf[7];
  {
    const char* ktu = tfK + (int64_t)bh * TF_BYTES + lane * 16;
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) ktf[ks] = *reinterpret_cast<const bf16x8*>(ktu + (qdb * 7 + ks) * 1024);
  }
  // ---- dK^T[d][key] = scale sum_q Q[q][d] dS[q][key]  and  dQ^T[d][q] = scale sum_key K[key][d] dS[q][key]
  zero_acc();
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();
    stage(imgS, PS_PITCH, dS + (int64_t)bh * N * ld, ld, half * PH, std::integral_constant<int, NKP / 8>{});
    stage(imgQ, PQ_PITCH, qkv + (int64_t)b * N * ts + h * THD, ts, half * PH, std::integral_constant<int, THD / 8>{});
    __syncthreads();
    contract_q();
#pragma unroll 1
    for (int qb = qfirst; qb < PH / 16; qb += qstep) {             // dQ of the staged rows
      f32x4 aq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) {
        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(imgS + (16 * qb + n) * PS_PITCH + (32 * ks + 8 * kq) * 2);
        aq = mfma16(ktf[ks], pf, aq);
      }
      const int q = half * PH + 16 * qb + n;
      if (q < N) {
        bf16x4 v = {(bf16)(aq[0] * scale), (bf16)(aq[1] * scale), (bf16)(aq[2] * scale), (bf16)(aq[3] * scale)};
        *reinterpret_cast<bf16x4*>(dqkv + ((int64_t)b * N + q) * ts + h * THD + 16 * qdb + 4 * kq) = v;
      }
    }
  }
  store_cols((int)tso, scale);
  // ---- dV^T[d][key] = sum_q dO[q][d] P'[q][key]
  zero_acc();
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    stage(imgS, PS_PITCH, Pm + (int64_t)bh * N * ld, ld, half * PH, std::integral_constant<int, NKP / 8>{});
    stage(imgQ, PQ_PITCH, dout + (int64_t)b * N * tso + h * THD, tso, half * PH, std::integral_constant<int, THD / 8>{});
    __syncthreads();
    contract_q();
  }
  store_cols((int)(2 * tso), 1.f);
}

}  // namespace

static std::atomic<unsigned long long*> g_th_dbg{nullptr};
// diagnostic hook (tools/th_attn_bench.py): phase stamps (s_memtime) of every workgroup, 8 slots each; nullptr = off
extern "C" void vitmi_debug_th_attn_stamps(unsigned long long* buf) { g_th_dbg = buf; }
void vitmi_debug_reset_cait_fused() { g_th_dbg = nullptr; }

extern "C" int vitmi_th_attn_supported(int dtype, int64_t H, int64_t N, int64_t hd) {
  return dtype == VITMI_BF16 && H == TNH && hd == THD && N >= 8 && N <= NKP && N % 4 == 0;
}
// fragment-major operand copies (forward: RF(K) | TF(V); backward: RF(K) | RF(V) | TF(K)) + the backward's per-workgroup
// parameter-gradient partials
extern "C" size_t vitmi_th_attn_workspace(int64_t B, int64_t H, int64_t N, int64_t hd) {
  (void)hd;
  const size_t packs = (size_t)B * H * (2 * RF16_BYTES + TF_BYTES);
  const size_t parts = (size_t)B * ((N + BRB - 1) / BRB) * TH_PART * sizeof(float);
  return packs + ((parts + 255) / 256) * 256;
}

extern "C" int vitmi_th_attn_fwd(const void* qkv, const float* Wl, const float* bl, const float* Ww, const float* bw,
                                 void* out, int dtype, int64_t B, int64_t H, int64_t N, int64_t hd, float scale,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
  VITMI_REQUIRE(qkv && Wl && bl && Ww && bw && out && B > 0, VITMI_E_BADARG, "th_attn_fwd: null argument");
  VITMI_REQUIRE(vitmi_th_attn_supported(dtype, H, N, hd), VITMI_E_SHAPE, "th_attn_fwd: bf16, H = 8, hd = 48, N <= 224 and N %% 4 == 0 (got dtype %d, H %lld, hd %lld, N %lld)", dtype, (long long)H, (long long)hd, (long long)N);
  VITMI_REQUIRE(is_aligned(qkv, 16) && is_aligned(out, 8), VITMI_E_ALIGN, "th_attn_fwd: qkv must be 16-B, out 8-B aligned");
  VITMI_REQUIRE(workspace && is_aligned(workspace, 256) && workspace_bytes >= vitmi_th_attn_workspace(B, H, N, hd), VITMI_E_WORKSPACE, "th_attn_fwd: workspace too small or misaligned");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  char* rfK = reinterpret_cast<char*>(workspace);
  char* tfV = rfK + (size_t)B * H * RF16_BYTES;
  const bf16* q = reinterpret_cast<const bf16*>(qkv);
  hipLaunchKernelGGL(th_pack_kernel<false>, dim3((unsigned)(B * H)), dim3(256), 0, stream, q, rfK, (char*)nullptr, tfV, (int)N, (int)H);
  if (int rc = vitmi_check_launch("th_pack_kernel")) return rc;
  const int lds = FRB * ROWB;
  if (int rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(th_attn_fwd_kernel), lds, "th_attn_fwd")) return rc;
  const int64_t nqb = (N + FRB - 1) / FRB;
  // blocks per workgroup: a whole image when the batch fills the chip (operands read once per image), fewer on small batches
  int nblk = (int)(B * nqb / vitmi_cu_count());
  if (nblk < 1) nblk = 1;
  if (nblk > nqb) nblk = (int)nqb;
  const int64_t wpi = (nqb + nblk - 1) / nblk;
  hipLaunchKernelGGL(th_attn_fwd_kernel, dim3((unsigned)(B * wpi)), dim3(512), lds, stream, q, rfK, tfV, Wl, bl, Ww, bw, reinterpret_cast<bf16*>(out), (int)N, scale, nblk, g_th_dbg.load());
  return vitmi_check_launch("th_attn_fwd_kernel");
}

extern "C" int vitmi_th_attn_bwd(const void* qkv, const void* dout, const float* Wl, const float* bl, const float* Ww,
                                 const float* bw, void* dqkv, void* dS, void* Pm, int64_t ld, float* dWl, float* dbl,
                                 float* dWw, float* dbw, int dtype, int64_t B, int64_t H, int64_t N, int64_t hd, float scale,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
  VITMI_REQUIRE(qkv && dout && Wl && bl && Ww && bw && dqkv && dS && Pm && dWl && dbl && dWw && dbw && B > 0, VITMI_E_BADARG, "th_attn_bwd: null argument");
  VITMI_REQUIRE(vitmi_th_attn_supported(dtype, H, N, hd), VITMI_E_SHAPE, "th_attn_bwd: bf16, H = 8, hd = 48, N <= 224 and N %% 4 == 0 required");
  VITMI_REQUIRE(ld >= NKP && ld % 8 == 0 && is_aligned(dS, 16) && is_aligned(Pm, 16) && is_aligned(qkv, 16) && is_aligned(dout, 16) && is_aligned(dqkv, 8),
                VITMI_E_ALIGN, "th_attn_bwd: ld >= 224 (every key slot of a row is written), ld %% 8 == 0 (the products kernel reads 16-B row pieces), qkv / dout / dS / Pm 16-byte and dqkv 8-byte aligned");
  VITMI_REQUIRE(workspace && is_aligned(workspace, 256) && workspace_bytes >= vitmi_th_attn_workspace(B, H, N, hd), VITMI_E_WORKSPACE, "th_attn_bwd: workspace too small or misaligned");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  char* rfK = reinterpret_cast<char*>(workspace);
  char* rfV = rfK + (size_t)B * H * RF16_BYTES;
  char* tfK = rfV + (size_t)B * H * RF16_BYTES;
  float* part = reinterpret_cast<float*>(tfK + (size_t)B * H * TF_BYTES);
  const bf16* q = reinterpret_cast<const bf16*>(qkv);
  hipLaunchKernelGGL(th_pack_kernel<true>, dim3((unsigned)(B * H)), dim3(256), 0, stream, q, rfK, rfV, tfK, (int)N, (int)H);
  if (int rc = vitmi_check_launch("th_pack_kernel")) return rc;
  if (int rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(th_attn_bwd_kernel), B_LDS, "th_attn_bwd")) return rc;
  const int64_t nqb = (N + BRB - 1) / BRB;
  const int nblk = 1;                                                // one block per workgroup: an image's blocks share its operands in L2
  const int64_t wpi = (nqb + nblk - 1) / nblk;
  const int64_t img_rows = (B + 7) / 8;                              // images per XCD
  hipLaunchKernelGGL(th_attn_bwd_kernel, dim3((unsigned)(8 * img_rows * wpi)), dim3(512), B_LDS, stream, q, reinterpret_cast<const bf16*>(dout), rfK, rfV, Wl, bl, Ww, bw,
                     reinterpret_cast<bf16*>(dqkv), reinterpret_cast<bf16*>(dS), reinterpret_cast<bf16*>(Pm), ld, part, (int)N, scale, nblk, (int)B, g_th_dbg.load());
  if (int rc = vitmi_check_launch("th_attn_bwd_kernel")) return rc;
  if (int rc = vitmi_raise_dynamic_lds(reinterpret_cast<const void*>(th_attn_prod_kernel), P_LDS, "th_attn_bwd(products)")) return rc;
  hipLaunchKernelGGL(th_attn_prod_kernel, dim3((unsigned)(B * H)), dim3(512), P_LDS, stream, q, reinterpret_cast<const bf16*>(dout),
                     reinterpret_cast<const bf16*>(dS), reinterpret_cast<const bf16*>(Pm), ld, tfK, reinterpret_cast<bf16*>(dqkv), (int)N, scale);
  if (int rc = vitmi_check_launch("th_attn_prod_kernel")) return rc;
  float* const outs[4] = {dWl, dbl, dWw, dbw};
  const int widths[4] = {TNH * TNH, TNH, TNH * TNH, TNH};
  return vitmi_reduce_rows_segs(part, (int)(B * wpi), TH_PART, outs, widths, stream);
}
