/*
 * vitmi.h — C ABI of libvitmi.so, the MI355X (gfx950) kernels behind the ViT
 * forward/backward training path.
 *
 * The reference (khuongnd6/ViT_torch) has no FFI: its hot path is the chain of
 * stock PyTorch ops executed by `self.model(x)` / `loss.backward()` /
 * `optimizer.step()` in utils_network.py:418-442.  Each entry point below
 * replaces one group of those ops; the reference line each one stands in for
 * is cited next to it.  Nothing here takes a torch type: plain device
 * pointers, sizes, strides (in ELEMENTS) and a hipStream_t passed as void*.
 *
 * Conventions
 *  - every call only ENQUEUES work on `stream`: no allocation, no sync (safe under
 *    hipGraph capture).  Calls are re-entrant across host threads and streams.  The only
 *    process-wide state is (a) a per-device table, filled under a lock, of the
 *    CU count and of which kernels have had their dynamic-LDS limit raised, and (b) the
 *    `vitmi_debug_*` switches (not declared here; test / profiling hooks that select kernel
 *    variants process-wide and must not be flipped while another thread is launching);
 *  - return 0 on success, <0 for a rejected argument (VITMI_E_*), >0 = the
 *    hipError_t of a failed launch; vitmi_last_error_string() describes the
 *    last failure on the calling thread;
 *  - dtype codes: VITMI_F32 / VITMI_BF16.  "T" below = activation dtype,
 *    "R" = residual-stream dtype; statistics, biases, LayerNorm/LayerScale
 *    parameters, gradients of parameters and the loss are always fp32.
 */
#ifndef VITMI_H
#define VITMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITMI_VERSION 109

enum { VITMI_F32 = 0, VITMI_BF16 = 1 };

enum {
  VITMI_E_BADARG = -1,    /* null pointer / non-positive size                */
  VITMI_E_ALIGN = -2,     /* pointer or stride not aligned as required       */
  VITMI_E_DTYPE = -3,     /* dtype combination not built                     */
  VITMI_E_SHAPE = -4,     /* shape outside what the kernel supports          */
  VITMI_E_WORKSPACE = -5  /* workspace too small                             */
};

int vitmi_version(void);
const char* vitmi_last_error_string(void);
/* launch_flags (vitmi_gemm_desc.launch_flags, vitmi_attn_bwd): per-call launch form.
 * By default the GEMM and attention-backward kernels launch one workgroup per CU that walks a fixed
 * list of tiles / (image, head) pairs (persistent grids).  A caller that runs other kernels BESIDE them
 * on the same device (RCCL collectives overlapping the backward pass: vit_torch_amd.ddp.GradReducer)
 * passes VITMI_LAUNCH_SHARED_DEVICE for those calls: one tile / pair per workgroup, placed by the
 * hardware dispatcher on whatever CUs are free.  Results are bit-identical either way.  The choice is
 * an argument of the call (it was a process-wide switch in ABI 104): concurrent engines, threads and
 * streams cannot race on it.
 * VITMI_LAUNCH_ROWS_PADDED (ABI 108, vitmi_gemm only): the caller vouches that C, C2, R and AUX are ALLOCATED up to the
 * next multiple of 256 rows (the rows M .. ceil(M / 256) * 256 - 1 are padding the kernel may read and overwrite).  A
 * bf16 product with a k-major A whose M is not a multiple of 256 (N % 256 == 0, K % 64 == 0) then runs on the 256 x 256
 * tile kernel: the last row tile stages A's last row in place of the missing ones and stores whole tiles.  Without the
 * flag such shapes take the 256 x 128 ragged kernel (no padding needed, 7-30 % slower).  Results for rows < M are the same. */
enum { VITMI_LAUNCH_SHARED_DEVICE = 1, VITMI_LAUNCH_ROWS_PADDED = 2 };

/* ---------------------------------------------------------------- GEMM ---
 * C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ), fp32 accumulation.
 *   A(m,k) = a_kmajor ? A[m*lda + k] : A[k*lda + m]
 *   B(n,k) = b_kmajor ? B[n*ldb + k] : B[k*ldb + n]
 * so  nn.Linear forward  y = x W^T      : A=x  (kmajor), B=W  (kmajor)
 *     input gradient     dx = dy W      : A=dy (kmajor), B=W  (k-minor)
 *     weight gradient    dW = dy^T x    : A=dy (k-minor), B=x (k-minor)
 * Replaces nn.Linear / nn.Conv2d(k=s=p) forward and their autograd backward:
 * models/swin.py:19,21,109,111,304,434; models/cait.py:29-33,101-105; the
 * classifier head models/vision_all.py:310-319.
 */
enum {
  VITMI_EPI_STORE = 0,     /* C = alpha*acc (+bias[n])                        */
  VITMI_EPI_BIAS_GELU = 1, /* C2 = acc+bias ; C = gelu_erf(C2)  (Mlp.fc1+act,
                              models/swin.py:25-26)                           */
  VITMI_EPI_RESIDUAL = 2,  /* C = R + gamma[n]*(acc+bias[n])    (x + g*f(x),
                              models/cait.py:148-149, models/swin.py:267-268;
                              gamma==NULL -> 1); C2 (optional, dtype of A/B)
                              = acc+bias, the branch output kept for d gamma  */
  VITMI_EPI_DGELU = 3,     /* C = acc * gelu_erf'(AUX)          (backward of
                              EPI_BIAS_GELU's activation)                     */
  VITMI_EPI_PATCH_POS = 4  /* t = m % n_tok:  t==0 ? cls[n]+pos[0,n]
                              : acc+bias[n]+pos[t,n]   (cat(cls,x)+pos_embed,
                              models/cait.py:231-234 / DINO prepare_tokens)   */
};

enum { VITMI_GEMM_AUTO = 0, VITMI_GEMM_GENERIC = 1, VITMI_GEMM_FAST = 2 };

typedef struct vitmi_gemm_desc {
  int64_t struct_size;         /* = sizeof(vitmi_gemm_desc): a caller built against an
                                  older / shorter layout is rejected (VITMI_E_BADARG)
                                  instead of having the library read past its struct  */
  int64_t M, N, K;
  const void* A; int64_t lda; int32_t a_kmajor;
  const void* B; int64_t ldb; int32_t b_kmajor;
  int32_t in_dtype;            /* dtype of A, B and AUX                       */
  int32_t epilogue;
  void* C; int64_t ldc; int32_t c_dtype;
  void* C2; int64_t ldc2;      /* second output or NULL (dtype: see epilogues) */
  const float* bias;           /* [N] or NULL                                 */
  const void* R; int64_t ldr; int32_t r_dtype;   /* residual input           */
  const float* gamma;          /* [N] LayerScale or NULL                      */
  const void* AUX; int64_t ldaux;                /* pre-activation (DGELU)   */
  const float* pos; int64_t n_tok; const float* cls; /* PATCH_POS            */
  float alpha;                 /* scale on acc for EPI_STORE (0 -> 1)        */
  int32_t accumulate;          /* EPI_STORE fp32 only: C += ...              */
  int32_t impl;                /* VITMI_GEMM_*                               */
  void* workspace;             /* optional scratch, >= vitmi_gemm_workspace(d) */
  size_t workspace_bytes;      /* (without it split-K is not used: slower)    */
  /* batched form (generic kernel, EPI_STORE only): problem z = zo*batch_inner + zi,
   * zo < batch/batch_inner, uses A + zo*a_bs[0] + zi*a_bs[1] (elements), same for B, C.
   * Lets per-(image, head) products read q/k/v in place from the [B,N,3,H,hd] qkv
   * tensor (models/cait.py:113).  batch <= 1 = plain GEMM. */
  int64_t batch, batch_inner;
  int64_t a_bs[2], b_bs[2], c_bs[2];
  /* EPI_RESIDUAL only: per-row-group scale of the branch, C = R + rowscale[m/rows_per_group]
   * * gamma[n]*(acc+bias) — DropPath's per-sample keep/keep_prob (timm DropPath as used at
   * models/swin.py:203,267-268) with rows_per_group = tokens per image.  NULL -> 1. */
  const float* rowscale; int64_t rows_per_group;
  /* EPI_DGELU on the bf16 tile paths only: optional fp32 [ceil(M/128)][N] (ld = N);
   * row r receives the column sums of the fp32 epilogue results of rows [128r, 128r+128).
   * Their sum over r (vitmi_colsum) is the bias gradient of the Linear whose
   * pre-activation is AUX (torch.nn.Linear backward, db = sum_rows dH) — fused here so dH
   * is not read back from HBM just to be summed.  Ask vitmi_gemm_uses_fast() first: any
   * other path rejects a non-NULL colsum_part. */
  float* colsum_part;
  /* EPI_BIAS_GELU / EPI_DGELU pair: 0 -> C2 / AUX hold the pre-activation (as above);
   * 1 -> EPI_BIAS_GELU stores gelu_erf'(acc+bias) in C2 (it shares the forward's exp) and
   * EPI_DGELU computes C = acc * AUX: the backward epilogue is a multiply instead of a second
   * erf/exp evaluation.  Same function of the same pre-activation either way
   * (torch.nn.GELU backward); a C2 written with one value must be read with the same. */
  int32_t aux_is_derivative;
  int32_t launch_flags;        /* VITMI_LAUNCH_* (0 = default)                 */
} vitmi_gemm_desc;

int vitmi_gemm(const vitmi_gemm_desc* d, void* stream);
/* Two products in ONE launch: C0 = op(A0) op(B0)^T and C1 = op(A1) op(B1)^T.  Meant for the two weight gradients of an
 * attention block (dW_proj = dY^T x: 768 x 768, and dW_qkv: 2304 x 768; autograd of models/swin.py:119-144): both contract
 * over all tokens, and the small one alone has too few output tiles to fill the chip without cutting K into 28 slices.
 * Sharing a grid, the 9 + 27 tiles take 7 slices each (a quarter of the partial-tile traffic, one pipeline fill per 113
 * k-steps instead of 29).  Pairable: bf16 operands, both k-minor (a_kmajor = b_kmajor = 0), fp32 C, EPI_STORE without bias /
 * accumulate, the same K, whole 256 x 256 x 64 tiles, `workspace` >= vitmi_gemm_pair_workspace (the descriptors' own
 * workspace fields are used only by the fallback).  Anything else runs as two vitmi_gemm calls: same results either way
 * (the split-K reduction sums the slices in a fixed order). */
size_t vitmi_gemm_pair_workspace(const vitmi_gemm_desc* d0, const vitmi_gemm_desc* d1);
int vitmi_gemm_pair(const vitmi_gemm_desc* d0, const vitmi_gemm_desc* d1, void* workspace, size_t workspace_bytes,
                    void* stream);
/* bytes of scratch that let the fast kernel split the contraction over more
 * workgroups (weight gradients: few output tiles, K = all tokens); 0 if none */
size_t vitmi_gemm_workspace(const vitmi_gemm_desc* d);
/* 1 if the aligned-shape MFMA/LDS-DMA kernel would be used for d, 0 if the
 * generic strided kernel would (tests assert the hot shapes take the fast one) */
int vitmi_gemm_uses_fast(const vitmi_gemm_desc* d);

/* ------------------------------------------------------------ LayerNorm --
 * nn.LayerNorm over the last dim: models/swin.py:198,204,305,436,551
 * (eps 1e-5), models/cait.py:64,68,137,141,203 (eps 1e-6).  One wavefront
 * per row, fp32 statistics, rows may be strided (x_stride/y_stride in
 * elements) so the CLS-only final norm reads x[:,0] in place.
 * Requires D % 4 == 0, D <= 2048.
 */
int vitmi_layernorm_fwd(const void* x, int x_dtype, int64_t x_stride,
                        const float* gamma, const float* beta,
                        void* y, int y_dtype, int64_t y_stride,
                        float* mean, float* rstd,
                        int64_t M, int64_t D, float eps, void* stream);

size_t vitmi_layernorm_bwd_workspace(int64_t M, int64_t D);
/* g_out = (g_in ? g_in : 0) + dLN/dx ; gb_out (optional) = cast(g_out);
 * dgamma/dbeta overwritten (fp32 [D]); gsum (optional, fp32 [D]) = column sum of g_out
 * over the M rows = the bias gradient of the Linear whose output feeds this residual
 * position (fused here to save a pass over g_out).  gb_scale (optional, fp32 [D]):
 * LayerScale of the branch that consumes gb_out — gb_out and gsum are then taken of
 * g_out * gb_scale (models/cait.py:148-149); gb_rowscale (optional, fp32, one per
 * rows_per_group rows): DropPath factor of that branch, applied the same way. */
int vitmi_layernorm_bwd(const void* dy, int dy_dtype, int64_t dy_stride,
                        const void* x, int x_dtype, int64_t x_stride,
                        const float* mean, const float* rstd, const float* gamma,
                        const void* g_in, void* g_out, int g_dtype, int64_t g_stride,
                        void* gb_out, int gb_dtype, int64_t gb_stride,
                        float* dgamma, float* dbeta, float* gsum, const float* gb_scale,
                        const float* gb_rowscale, int64_t rows_per_group,
                        int64_t M, int64_t D,
                        void* workspace, size_t workspace_bytes, void* stream);

/* The same pass with the fold of its partial sums left to the caller (ABI 106): the per-workgroup partial rows of
 * dgamma | dbeta | gsum stay in `workspace` — which the caller must then keep untouched until the fold has run — and
 * *fold describes them for vitmi_fold_many.  A training step queues these (and the bias partials of the GEMM / attention
 * epilogues) and folds a whole backward pass, or one gradient-bucket section, in one launch.  vitmi_layernorm_bwd is
 * this call followed by vitmi_fold_many(fold, 1). */
typedef struct vitmi_fold_desc {
  int64_t struct_size;   /* = sizeof(vitmi_fold_desc) */
  const float* part;     /* fp32 [S][ld] partial sums */
  int32_t S;             /* partial rows */
  int32_t nseg;          /* 1..3 segments of N consecutive columns in a row */
  int64_t N, ld;
  float* out[3];         /* fp32 [N] per segment, overwritten with the column sums over the S rows */
} vitmi_fold_desc;
int vitmi_layernorm_bwd_deferred(const void* dy, int dy_dtype, int64_t dy_stride,
                                 const void* x, int x_dtype, int64_t x_stride,
                                 const float* mean, const float* rstd, const float* gamma,
                                 const void* g_in, void* g_out, int g_dtype, int64_t g_stride,
                                 void* gb_out, int gb_dtype, int64_t gb_stride,
                                 float* dgamma, float* dbeta, float* gsum, const float* gb_scale,
                                 const float* gb_rowscale, int64_t rows_per_group,
                                 int64_t M, int64_t D,
                                 void* workspace, size_t workspace_bytes, vitmi_fold_desc* fold, void* stream);
/* n folds (host array) in ceil(n / 32) launches; each output column is summed over its S rows in the same fixed order as
 * by the single-fold calls (bit-identical, deterministic). */
int vitmi_fold_many(const vitmi_fold_desc* descs, int n, void* stream);

/* ------------------------------------------------------------ Attention --
 * Multi-head self-attention core: softmax(scale * q k^T) v, fused (scores
 * never reach HBM).  qkv is the output of the qkv Linear viewed as
 * [B, N, 3, H, hd] (models/swin.py:120, models/cait.py:113); out is
 * [B, N, H*hd] ready for the proj Linear (models/swin.py:142).  lse[B,H,N] is
 * the natural-log-sum-exp of the scaled scores, kept for backward.
 * hd in {32, 64}.
 */
int vitmi_attn_fwd(const void* qkv, void* out, float* lse, int dtype,
                   int64_t B, int64_t N, int64_t H, int64_t hd, float scale,
                   void* stream);
size_t vitmi_attn_bwd_workspace(int64_t B, int64_t N, int64_t H);
/* dbias_part (optional, bf16 kernels only): fp32 [vitmi_attn_bwd_dbias_rows(B, N)][3*H*hd];
 * each row holds the column sums of the dqkv rows one workgroup produced, so their sum
 * (vitmi_colsum) is the qkv Linear's bias gradient without reading dqkv back. */
int64_t vitmi_attn_bwd_dbias_rows(int64_t B, int64_t N);
int vitmi_attn_bwd(const void* qkv, const void* out, const void* dout,
                   const float* lse, void* dqkv, int dtype,
                   int64_t B, int64_t N, int64_t H, int64_t hd, float scale,
                   float* dbias_part, int32_t launch_flags,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------- CaiT ops --
 * Talking-heads softmax (models/cait.py:118-122) on score tensors [B,H,N,ld] (row length
 * Nk <= 256 valid columns, H <= 8):  S' = Wl S + bl over the head axis, P = softmax_j(S'),
 * Pm = Ww P + bw.  P is kept for backward.  The q k^T, P v products around it are batched
 * vitmi_gemm calls that read q/k/v in place from the qkv tensor. */
int vitmi_th_softmax_fwd(const void* S, const float* Wl, const float* bl, const float* Ww,
                         const float* bw, void* P, void* Pm, int dtype,
                         int64_t B, int64_t H, int64_t N, int64_t Nk, int64_t ld, void* stream);
size_t vitmi_th_softmax_bwd_workspace(int64_t B, int64_t H, int64_t N);
/* dS from dPm; dWl/dbl/dWw/dbw (fp32, overwritten) = gradients of proj_l / proj_w */
int vitmi_th_softmax_bwd(const void* S, const void* P, const void* dPm, const float* Wl,
                         const float* Ww, void* dS, float* dWl, float* dbl, float* dWw, float* dbw,
                         int dtype, int64_t B, int64_t H, int64_t N, int64_t Nk, int64_t ld,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Talking-heads attention as ONE op (ABI 108; models/cait.py:111-128): O = proj_w(softmax(proj_l(scale q k^T))) v with the
 * score rows resident in LDS and both head mixes on the matrix pipe; the forward keeps nothing but O.  bf16, H = 8,
 * hd = 48, N <= 224, N % 4 == 0 (vitmi_th_attn_supported); other shapes / fp32 take the three-call form above.
 * qkv [B,N,3,H,hd], out / dout [B,N,H,hd].  The backward recomputes the scores and writes dqkv (all of it) and the four
 * proj_l / proj_w gradients (fp32, overwritten); dS and Pm ([B,H,N,ld] bf16, ld >= 224, ld % 8 == 0, 16-B aligned) are caller-owned SCRATCH through
 * which dS and P' travel once between its two kernels (the row kernel and the dQ / dK / dV products).  workspace: vitmi_th_attn_workspace bytes, 256-B aligned. */
int vitmi_th_attn_supported(int dtype, int64_t H, int64_t N, int64_t hd);
size_t vitmi_th_attn_workspace(int64_t B, int64_t H, int64_t N, int64_t hd);
int vitmi_th_attn_fwd(const void* qkv, const float* Wl, const float* bl, const float* Ww, const float* bw,
                      void* out, int dtype, int64_t B, int64_t H, int64_t N, int64_t hd, float scale,
                      void* workspace, size_t workspace_bytes, void* stream);
int vitmi_th_attn_bwd(const void* qkv, const void* dout, const float* Wl, const float* bl, const float* Ww,
                      const float* bw, void* dqkv, void* dS, void* Pm, int64_t ld, float* dWl, float* dbl,
                      float* dWw, float* dbw, int dtype, int64_t B, int64_t H, int64_t N, int64_t hd, float scale,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Class attention core (models/cait.py:44-52): one query per image (the projected CLS
 * token, q [B, H*hd]) against k/v rows [B, N, .] with token stride kv_token_stride
 * (elements); out [B, H*hd]; p_save [B,H,N] fp32 (softmax, kept for backward). */
int vitmi_class_attn_fwd(const void* q, const void* k, const void* v, int64_t kv_token_stride,
                         void* out, float* p_save, int dtype,
                         int64_t B, int64_t H, int64_t N, int64_t hd, float scale, void* stream);
int vitmi_class_attn_bwd(const void* q, const void* k, const void* v, int64_t kv_token_stride,
                         const void* dout, const float* p_save, void* dq, void* dk, void* dv,
                         int64_t dkv_token_stride, int dtype,
                         int64_t B, int64_t H, int64_t N, int64_t hd, float scale, void* stream);

/* out[n] = sum_m x[m*ldx+n] * y[m*ldy+n]: LayerScale gradient d gamma = sum_rows dY * f(x)
 * (models/cait.py:148-149) */
size_t vitmi_colsum_mul_workspace(int64_t M, int64_t N);
int vitmi_colsum_mul(const void* x, int x_dtype, int64_t ldx, const void* y, int y_dtype, int64_t ldy,
                     int64_t M, int64_t N, float* out, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ------------------------------------------------------------- Swin ops --
 * (Shifted-)window attention, models/swin.py:113-144 inside SwinTransformerBlock.forward
 * :241-261.  qkv [B*L, 3*H*hd] and out/dout [B*L, H*hd] stay in TOKEN order: the cyclic
 * shift, window_partition and window_reverse are folded into the kernels' row addressing.
 * Bw = B * (Himg/ws) * (Wimg/ws) windows of N = ws*ws <= 64 tokens; bias [H,N,N] fp32 (the
 * gathered relative position bias), mask [nW,N,N] fp32 or NULL (shift mask, -100/0);
 * lse [Bw,H,N]. */
int vitmi_win_attn_fwd(const void* qkv, void* out, float* lse, const float* bias, const float* mask,
                       int dtype, int64_t Bw, int64_t H, int64_t N, int64_t hd,
                       int64_t Himg, int64_t Wimg, int64_t ws, int64_t shift, float scale, void* stream);
size_t vitmi_win_attn_bwd_workspace(int64_t Bw, int64_t H, int64_t N);
/* dqkv (token order) and dbias [H,N,N] = sum over windows of d(score) (deterministic).
 * dqkv_bias (optional, fp32 [3*H*hd], overwritten): column sums of dqkv = the qkv Linear's bias
 * gradient, without reading dqkv back; only where vitmi_win_attn_bwd_fuses_qkv_bias() is 1. */
int vitmi_win_attn_bwd_fuses_qkv_bias(int dtype, int64_t hd);
int vitmi_win_attn_bwd(const void* qkv, const void* dout, const float* lse, const float* bias,
                       const float* mask, void* dqkv, float* dbias, float* dqkv_bias, int dtype,
                       int64_t Bw, int64_t H, int64_t N, int64_t hd,
                       int64_t Himg, int64_t Wimg, int64_t ws, int64_t shift, float scale,
                       void* workspace, size_t workspace_bytes, void* stream);
/* relative_position_bias_table[T,H] <-> bias[H,N,N] through relative_position_index[N*N]
 * (models/swin.py:126-129): gather when (table,bias) given, gradient scatter when
 * (dbias,dtable) given (dtable overwritten). */
int vitmi_relpos_bias(const float* table, const int64_t* index, float* bias, const float* dbias,
                      float* dtable, int64_t T, int64_t H, int64_t N, void* stream);
/* PatchMerging's 2x2 gather (models/swin.py:317-323): [B,Hh*Ww,C] -> [B,Hh/2*Ww/2,4C]
 * (inverse=1: the scatter back, i.e. its backward) */
int vitmi_patch_merge(const void* src, void* dst, int dtype, int64_t B, int64_t Hh, int64_t Ww,
                      int64_t C, int inverse, void* stream);
/* AdaptiveAvgPool1d(1) over tokens (models/swin.py:584): out[B,C] fp32 = mean_l x[B,L,C]
 * when (x,out) given; dx = dout/L when (dout,dx) given */
int vitmi_token_mean(const void* x, float* out, const float* dout, void* dx, int dtype,
                     int64_t B, int64_t L, int64_t C, void* stream);

/* ---------------------------------------------------------- Elementwise --*/
/* "bf16x3" parity mode (ABI 109): fp32 arithmetic of the reference (/root/reference/main.py:244,
 * utils_network.py:120 — nn.Linear / Conv2d on fp32 tensors) on the bf16 matrix pipe.  x = hi + lo with
 * hi = bf16(x), lo = bf16(x - hi); an fp32 [rows, cols] matrix is written as the bf16 image
 *   A pattern (b_pattern = 0): hi | lo | hi        B pattern (b_pattern = 1): hi | hi | lo
 * so that ONE vitmi_gemm over K' = 3K of an A-pattern and a B-pattern operand accumulates
 * a_hi b_hi + a_lo b_hi + a_hi b_lo in fp32 (error O(2^-16) per product instead of bf16's 2^-9).
 * stacked = 0: the three parts side by side along the row (out is [rows, 3 cols], ldo >= 3 cols): a k-major operand;
 * stacked = 1: three row blocks (out is [3 rows, cols], ldo >= cols): a k-minor operand (k = the row index).
 * cols, ldx, ldo multiples of 4; x 16-byte, out 8-byte aligned. */
int vitmi_split3(const float* x, int64_t ldx, int64_t rows, int64_t cols, void* out_bf16, int64_t ldo,
                 int b_pattern, int stacked, void* stream);
/* fp32 halves of the two GELU epilogues for that mode (the tile kernel builds them for bf16 outputs only):
 * out = gelu(pre) (erf form, nn.GELU() default: models/swin.py:14-30);  out = dh * gelu'(pre). */
int vitmi_gelu_fwd(const float* pre, int64_t ldp, float* out, int64_t ldo, int64_t M, int64_t N, void* stream);
int vitmi_gelu_bwd(const float* dh, int64_t ldd, const float* pre, int64_t ldp, float* out, int64_t ldo,
                   int64_t M, int64_t N, void* stream);

/* fp32 -> bf16 shadow copy of the flat parameter buffer */
int vitmi_cast(const void* src, int src_dtype, void* dst, int dst_dtype,
               int64_t n, void* stream);

/* y[i] += a * x[i], fp32, 16-byte aligned (ABI 107): gradient accumulation — a second backward() before
 * zero_grad() must ADD to .grad (torch.autograd's contract, /root/reference/utils_network.py:440-442 relies on
 * zero_grad -> backward -> step); the engines overwrite their flat gradient buffer, so the wrapper saves the
 * accumulated part and adds it back with this call */
int vitmi_axpy(const float* x, float* y, float a, int64_t n, void* stream);

/* out[m*ldo+n] = cast(x[m*ldx+n] * scale[n] * rowscale[m/rows_per_group]) (NULL -> 1):
 * strided row copy with optional per-column LayerScale / per-sample DropPath factor; also
 * the plain strided copy/cast of row blocks */
int vitmi_scale_cast(const void* x, int x_dtype, int64_t ldx, const float* scale,
                     const float* rowscale, int64_t rows_per_group,
                     void* out, int out_dtype, int64_t ldo, int64_t M, int64_t N, void* stream);

/* im2col for Conv2d(C, D, kernel=p, stride=p) (models/swin.py:434,445):
 * x[B,C,H,W] fp32 with element strides (sb,sc,sh,sw) — NCHW or channels_last —
 * -> rows [B*(cls_rows + (H/p)*(W/p)), C*p*p], k = c*p*p + i*p + j.  With
 * cls_rows=1 row 0 of every image is zero (placeholder for the CLS token).
 * out_ld (elements, % 4 == 0, 0 -> C*p*p): row pitch of `out`; columns [C*p*p, out_ld) are
 * written as zeros, so a contraction that is not a multiple of the GEMM's 32-deep step (Swin's
 * 4x4x3 = 48) can be padded to one that is (64) and take the tile kernel. */
int vitmi_patchify(const float* x, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                   void* out, int out_dtype, int64_t out_ld,
                   int64_t B, int64_t C, int64_t H, int64_t W, int64_t p,
                   int cls_rows, void* stream);

/* Bicubic resize of pos_embed to the input's patch grid (upstream DINO interpolate_pos_encoding, the
 * module loaded at models/vision_all.py:156; oracle/vit_ref.py:110-127) as a row-sparse fp32 product:
 *   dst[r, 0:D) = sum_{e in [row_ptr[r], row_ptr[r+1])} w[e] * src[col[e], 0:D),  r < rows.
 * The tables depend only on the two grids and are built by the host (vit_torch_amd/posembed.py: 16
 * taps per output position, aten's bicubic arithmetic); the transposed table gives the backward
 * pass (fixed entry order per row: deterministic).  D, ld_src, ld_dst multiples of 4. */
int vitmi_pos_resample(const float* src, int64_t ld_src, const int32_t* row_ptr, const int32_t* col,
                       const float* w, float* dst, int64_t ld_dst, int64_t rows, int64_t D, void* stream);

/* out[n] = sum_m x[m*ld + n]  (bias gradients, pos_embed/cls gradients) */
size_t vitmi_colsum_workspace(int64_t M, int64_t N);
int vitmi_colsum(const void* x, int dtype, int64_t M, int64_t N, int64_t ld,
                 float* out, void* workspace, size_t workspace_bytes, void* stream);

/* nn.CrossEntropyLoss() (main.py:244, utils_network.py:430), mean reduction:
 * loss[0] = mean_b( -log_softmax(logits[b])[label[b]] ), loss[1+b] = the
 * per-sample terms; dlogits = (softmax - onehot) / B  (gradient for dloss = 1).
 * correct[0] = count of argmax==label (utils_network.py:85-95), correct[1+b]
 * the per-sample 0/1.  `loss` must hold 1+B floats and `correct` 1+B int32 —
 * the per-sample slots double as the scratch of the two-stage, deterministic
 * reduction, so the call needs no workspace. */
int vitmi_softmax_xent(const float* logits, const int64_t* labels,
                       float* loss, float* dlogits, int32_t* correct,
                       int64_t B, int64_t K, void* stream);

/* optim.SGD(momentum) step (utils_network.py:120,442) over a flat buffer:
 * buf = momentum*buf + grad_scale*g ; p -= lr*buf ; optional bf16 shadow
 * refresh of p in the same pass. */
int vitmi_sgd_momentum(float* p, const float* g, float* buf, void* p_shadow_bf16,
                       int64_t n, float lr, float momentum, float grad_scale,
                       void* stream);

/* Device-side input transform (utils_datasets.py:553-582: RandomCrop(S, padding, fill=128),
 * RandomHorizontalFlip, ToTensor, Normalize): uint8 NHWC [B,H,W,C] -> fp32 NCHW [B,C,S,S].
 * off_y/off_x [B] = top-left of the crop inside the padded image (NULL -> pad: centred, the
 * test-time transform), flip [B] (NULL -> none), mean/std [C] (NULL -> 0 / 1).  The random
 * draws are the caller's; torchvision's arithmetic order, bit-exact in fp32. */
int vitmi_image_ingest(const void* src_u8_nhwc, float* dst_nchw, const int32_t* off_y, const int32_t* off_x,
                       const uint8_t* flip, const float* mean, const float* std, int64_t B, int64_t H,
                       int64_t W, int64_t C, int64_t S, int64_t pad, int64_t fill, void* stream);

/* vitmi_image_ingest fused with vitmi_patchify: the same per-sample transform of the uint8 NHWC batch written
 * straight into the patch rows [B*(cls_rows + (S/p)^2), out_ld] (dtype out_dtype, k = c*p*p + i*p + j, columns
 * beyond C*p*p zero, CLS placeholder rows zero) that the patch-embedding GEMM contracts — no fp32 NCHW
 * intermediate (utils_datasets.py:553-582 + models/swin.py:440-448 / the DINO patch conv).  Bit-identical to the
 * two calls in sequence.  p % 4 == 0, S % p == 0. */
int vitmi_ingest_patchify(const void* src_u8_nhwc, void* out, int out_dtype, int64_t out_ld, const int32_t* off_y,
                          const int32_t* off_x, const uint8_t* flip, const float* mean, const float* std,
                          int64_t B, int64_t H, int64_t W, int64_t C, int64_t S, int64_t pad, int64_t fill,
                          int64_t p, int cls_rows, void* stream);

/* optim.Adam / optim.AdamW step (utils_network.py:121,124; torch defaults betas (0.9, 0.999),
 * eps 1e-8, AdamW weight_decay 1e-2) over a flat buffer.  state[0] (device, fp32) is the step
 * count: it is advanced by this call BEFORE the update, so a captured HIP graph replays the
 * right bias corrections.  decoupled != 0: AdamW (p *= 1 - lr*wd); else L2 (g += wd*p). */
int vitmi_adam(float* p, const float* g, float* m, float* v, void* p_shadow_bf16, float* state,
               int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
               int decoupled, float grad_scale, void* stream);

/* The remaining entries of the reference's optimizer table (utils_network.py:119-126), same conventions
 * as vitmi_adam (flat fp32 buffers, optional bf16 shadow refreshed in the pass, grad_scale on g first,
 * device-side step count state[0] advanced BEFORE the update where the rule depends on the step).
 * vitmi_adagrad: torch.optim.Adagrad (:123; lr_decay 0, eps 1e-10, initial accumulator 0 by default).
 * vitmi_adadelta: torch.optim.Adadelta (:122; rho 0.9, eps 1e-6).
 * vitmi_adabelief: adabelief_pytorch.AdaBelief as configured at :125 (eps 1e-16, betas (0.9, 0.999),
 *   weight_decouple, rectify; amsgrad off, fixed_decay off, degenerated_to_sgd on).  That package is
 *   not in this container: the rule is restated from the published algorithm (Zhuang et al. 2020) and
 *   checked against a torch restatement of it (oracle/optim_ref.py) — parity unpinned. */
int vitmi_adagrad(float* p, const float* g, float* sum, void* p_shadow_bf16, float* state, int64_t n, float lr,
                  float lr_decay, float eps, float weight_decay, float grad_scale, void* stream);
int vitmi_adadelta(float* p, const float* g, float* square_avg, float* acc_delta, void* p_shadow_bf16, int64_t n,
                   float lr, float rho, float eps, float weight_decay, float grad_scale, void* stream);
int vitmi_adabelief(float* p, const float* g, float* m, float* s, void* p_shadow_bf16, float* state, int64_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled, int rectify,
                    float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITMI_H */
