/*
 * mtt_hip.h — C ABI of libmtt_hip.so: the MI355X (gfx950) kernels behind the TaskPrompter / InvPT
 * hot path.  This is the drop-in boundary BELOW the Python nn.Module mirror of the reference
 * (SURVEY.md §8b).  Plain pointers and sizes only — no torch types.
 *
 * Conventions (all entry points):
 *   - return 0 on success, <0 = MTT_E_* argument error, >0 = hipError_t from the launch
 *   - asynchronous on the caller's stream; no allocation, no synchronisation (safe under hipGraph capture); every buffer,
 *     including kernel workspaces, is owned by the caller.  No floating-point atomics: every cross-workgroup reduction writes
 *     per-workgroup partials to a caller-owned workspace and sums them in a fixed order, so results are run-to-run reproducible.  No environment variables and no process-global switches: which
 *     kernel runs is a pure function of the descriptor.  The only process state is a per-device "already configured" bit per
 *     kernel for the one-time hipFuncSetAttribute(MaxDynamicSharedMemorySize) opt-in (idempotent, thread-safe)
 *   - reduction-contiguous GEMM operands (MTT_OP_K) are read in 8-element chunks: when K is not a multiple
 *     of 8 the elements K..pad8(K)-1 of every row must exist and be zero (ld >= pad8(K))
 *   - activations are token-major / NHWC: a feature map is a row-major [rows = B*H*W, channels]
 *     matrix with a leading dimension (ld, in elements) that is a multiple of 8; 16-byte aligned
 *   - dtype codes: MTT_F32 = 0, MTT_BF16 = 1, MTT_SPLIT = 2 (hi / lo bf16 planes, see below)
 *   - prec: MTT_PREC_BF16 = 0 (bf16 MFMA, fp32 accumulate — the throughput path; fp32 operands are rounded while staged)
 *           MTT_PREC_X3   = 1 (operands split hi+lo bf16, 3 MFMAs — fp32-class accuracy, the
 *                              1e-3 parity gate; operands MTT_F32 (split while staged) or both MTT_SPLIT (pre-split planes
 *                              streamed by LDS-DMA))
 *
 * Each entry names the reference code it replaces (paths relative to /root/reference).
 */
#ifndef MTT_HIP_H
#define MTT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTT_ABI_VERSION 13

/* MTT_SPLIT: an fp32-class value stored as TWO bf16 planes of identical layout, x = hi + lo with hi = bf16(x), lo = bf16(x - hi)
 * (~16 mantissa bits).  The main pointer of an operand addresses the hi plane, its `*_lo` companion the lo plane.  The hi plane alone is
 * an ordinary bf16 tensor — which is what lets a training step run its forward in fp32-class arithmetic (3 MFMAs per product on the
 * LDS-DMA kernel, MTT_PREC_X3) and its backward in bf16 on the very same buffers. */
enum { MTT_F32 = 0, MTT_BF16 = 1, MTT_SPLIT = 2 };
enum { MTT_PREC_BF16 = 0, MTT_PREC_X3 = 1 };
enum { MTT_E_BADARG = -1, MTT_E_ALIGN = -2, MTT_E_UNSUPPORTED = -3 };

/* operand layouts of mtt_gemm */
enum {
  MTT_OP_K = 0,     /* element(r,k) at base + r*ld + k   (reduction index contiguous)            */
  MTT_OP_R = 1,     /* element(r,k) at base + k*ld + r   (row index contiguous; transposed view) */
  MTT_OP_CONV_K = 2,/* A only: implicit im2col, r = output pixel, k = (tap, ci)                   */
  MTT_OP_CONV_R = 3 /* B only: implicit im2col^T, k = pixel (reduction), r = (tap, ci)            */
};
enum { MTT_ACT_NONE = 0, MTT_ACT_GELU = 1, MTT_ACT_RELU = 2, MTT_ACT_GELU_BWD = 3, MTT_ACT_RELU_BWD = 4,
       /* ABI 10 — the GELU of a bf16-arithmetic backward with its derivative taken in the FORWARD: the fc1 epilogue has z in registers and
        * stores GELU'(z) (instead of z) next to GELU(z); the fc2 input-gradient epilogue is then one multiply per element instead of an
        * erf + exp evaluation (that epilogue ran ~25 VALU instructions per element against a 16-step K loop).  Same bytes, same bf16
        * storage rounding class: GELU' in [-0.13, 1.13] rounded to bf16 vs GELU'(bf16(z)). */
       MTT_ACT_GELU_DAUX = 5 /* D = GELU(z), aux_out = GELU'(z)   (aux_out required) */,
       MTT_ACT_MUL_AUX = 6   /* D = epi(acc) * aux_in              (aux_in required) */ };
enum { MTT_STORE_ROWS = 0, MTT_STORE_PIXSHUF2 = 1 };
/* kernel selection of mtt_gemm / mtt_attn_fwd: AUTO = the library's policy (a pure function of the descriptor); the other values
 * force one kernel where it is applicable (tests of a kernel on small shapes, A/B measurements).  There is no process-global switch and
 * no environment variable: the library keeps no mutable state that affects results. */
enum { MTT_GEMM_AUTO = 0,
       MTT_GEMM_GENERAL = 1           /* the register-staged 128 x 128 kernel (any layout / dtype / precision) */,
       MTT_GEMM_DMA256 = 3            /* the 256 x 256 LDS-DMA kernel (or, for two MTT_OP_R operands, the token-major weight-gradient kernel) on any shape it supports */,
       MTT_GEMM_DMA128 = 4            /* the 128 x 128 LDS-DMA kernel (K-contiguous bf16 operands; narrow or mid-size outputs) */,
       MTT_GEMM_GENERAL_EPILOGUE = 11 /* policy kernel, but always the general (run-time configured) epilogue: A/B of the specialised one */ };
enum { MTT_ATTN_AUTO = 0, MTT_ATTN_PLAIN = 1,
       MTT_ATTN_FAST_V0 = 2 /* A/B: the first flash kernels (run-time LDS stage, predicated register staging) */,
       MTT_ATTN_FAST_V1 = 3 /* A/B: register-staged tiles, unrolled stages (AUTO = LDS-DMA staging + transpose reads) */ };

/* 3x3 (dilated) "same" convolution geometry for MTT_OP_CONV_* operands; stride 1, pad = dil. */
typedef struct {
  int32_t H, W;     /* feature-map height/width (rows = B*H*W)                       */
  int32_t C;        /* valid input channels                                           */
  int32_t Cp;       /* channel pitch of the K ordering k = tap*Cp + ci (multiple of 8)*/
  int32_t dil;      /* dilation (1, or 2 for InvPT UpEmbed, invpt.py:33)              */
  int32_t flip;     /* 1: taps visited mirrored (dgrad)                               */
} mtt_conv_geom;

/*
 * D[z][m,n] = epilogue( alpha * sum_k A[z][m,k] * B[z][n,k] )      m<M, n<N, k<K, z<batch
 *
 * Replaces every nn.Linear / 1x1 nn.Conv2d / 3x3 nn.Conv2d / nn.ConvTranspose2d(k=s=2) matmul of
 * the hot path and their dgrad/wgrad (taskprompter.py:175-187,201,212,219,250 Linear; :362-366
 * fea_fuse / fea_decode convs; :692-709 heads; timm Mlp fc1/fc2; invpt.py:108-113; vit.py:179-181;
 * transformer_decoder.py:56-67,110).  Rows of A, D, aux and resid are addressed in groups:
 *   row m -> base + (m / mb) * bs + (m % mb) * ld          (mb = 0 means one group)
 * so a GEMM can read/write a token subset of a [B, N, C] buffer in place.
 * Batch index z = zo*batch_inner + zi adds zo*?_zo + zi*?_zi elements to each base.
 *
 * epilogue, in order:  v = alpha*acc;  v = v*colscale[n] + colshift[n];
 *   act: GELU(erf) / ReLU / GELU_BWD: v *= gelu'(aux_in[m,n]) / RELU_BWD: v *= (aux_in[m,n] > 0)
 *   aux_out[m,n] = pre-activation value (optional, training)
 *   v *= rowscale[(m / d_mb)*2 + ((m % d_mb) >= n_prompt)]   (DropPath per-sample scale, optional)
 *   v += resid[m,n] (fp32, own row mapping; may alias D)
 *   store D (dtype d_dtype; MTT_SPLIT: hi plane at D, lo plane at D_lo); columns N <= n < n_store are written as zeros (channel padding).
 * Operand dtypes: MTT_PREC_BF16 takes bf16 or fp32 operands in any mix (fp32 is rounded to bf16 while staged); MTT_PREC_X3 takes
 * fp32 operands (general kernel) or A and B both MTT_SPLIT with a_op = b_op = MTT_OP_K, K % 64 == 0 (LDS-DMA kernel).
 */
typedef struct {
  const void* A; const void* B; void* D;
  int32_t M, N, K;
  int32_t a_op, b_op;            /* MTT_OP_* */
  int32_t a_dtype, b_dtype, d_dtype;
  int32_t prec;
  int64_t lda, ldb, ldd;
  int32_t a_mb; int64_t a_bs;
  int32_t d_mb; int64_t d_bs;
  int32_t batch, batch_inner;    /* batch >= 1; batch_inner >= 1 */
  int64_t a_zo, a_zi, b_zo, b_zi, d_zo, d_zi;
  mtt_conv_geom conv;            /* used when a_op/b_op is MTT_OP_CONV_* */
  float alpha;
  const float* colscale; const float* colshift; int64_t col_zo, col_zi;   /* [N] per batch */
  int32_t act;
  const void* aux_in; void* aux_out; int32_t aux_dtype; int64_t ldaux; int64_t aux_zo, aux_zi; /* rows mapped like D */
  const float* rowscale; int32_t n_prompt;
  const float* resid; int64_t ldr; int32_t r_mb; int64_t r_bs; int64_t r_zo, r_zi;
  int32_t n_store;               /* >= N, <= ldd; 0 means N */
  int32_t store_mode;            /* MTT_STORE_PIXSHUF2: n = (dy*2+dx)*Co + co, D is [B,2H,2W,ldd] (ConvTranspose2d k=s=2, taskprompter.py:705) */
  int32_t ps_H, ps_W, ps_Co;
  int32_t variant;               /* MTT_GEMM_* (0 = AUTO) */
  const void* A_lo; const void* B_lo; void* D_lo;   /* lo planes of MTT_SPLIT operands / output (same layout, strides and batch offsets as the hi plane) */
  float* colsum_out; float* colsum_ws;   /* optional: colsum_out[n] = sum_m D[m,n] of the values AS STORED (bf16 D: the rounded values), n < N
                                            — the bias gradient when D is the gradient of a Linear's output (the fc2 input gradient with its
                                            GELU' epilogue IS fc1's output gradient), taken in the epilogue instead of re-reading D.  Per
                                            row-block partials go to colsum_ws (>= mtt_gemm_colsum_ws_floats(d) floats) and are summed in
                                            block order (deterministic).  batch == 1, MTT_STORE_ROWS only. */
  /* ABI 10 — A-operand PROLOGUE of the exact-fp32 tall GEMM (the kernel behind variant 11: fp32 operands, N <= 32 outputs, batch == 1):
   * the A element of channel k enters the product as act_a(A[m, k] * a_scale[k] + a_shift[k]) — BatchNorm (batch statistics folded into a
   * per-channel scale / shift, K floats each; padding channels: scale = shift = 0) + activation applied while the operand is loaded, so the
   * normalised + activated map of a conv head (8.7 GB in fp32 at the benchmark's batch) is never written in fp32 nor re-read: ConvHead's
   * BatchNorm + GELU + 1x1 prediction (taskprompter.py:692-694) is ONE pass over the conv output.  a_aux16 (optional): the transformed
   * operand rounded to bf16, [M, ld_a16] — what the bf16 backward of that layer reads (weight gradient of the predictions).
   * a_scale == NULL: no prologue.  Other kernels return MTT_E_UNSUPPORTED when it is set. */
  const float* a_scale; const float* a_shift; int32_t a_act; void* a_aux16; int64_t ld_a16;
} mtt_gemm_desc;
size_t mtt_gemm_colsum_ws_floats(const mtt_gemm_desc* d);

int mtt_abi_version(void);
/* sizeof(descriptor): 0 gemm, 1 attn, 2 softmax, 3 ln, 4 chanlogit, 5 modulate, 6 ctr, 7 resize, 8 bn, 9 conv_geom, (15 adam, 16 loss, 17 upconv, 18 gather, 19 winattn, 20 chanattn, 21 conv3s2)
 * 10 dwconv, 11 pool, 12 lnmt, 13 attnmsg, 14 convt, 17 upconv */
size_t mtt_desc_size(int which);
int mtt_gemm(const mtt_gemm_desc* d, void* stream);
/* which kernel mtt_gemm dispatches this descriptor to: 0 register-staged 128x128 (general), 3 phased LDS-DMA 256x256
 * (gemm_dma_kernel), 6 token-major weight-gradient kernel (gemm_tn_kernel: both operands MTT_OP_R, LDS-DMA + LDS transpose reads),
 * 8 gemm_dma_kernel on MTT_SPLIT operands (K-concatenated x3 product) */
int mtt_gemm_variant(const mtt_gemm_desc* d);

/*
 * Fused global attention over [T prompts || hw patches] with the prompt-row logit side channel.
 * Replaces Attention.forward's spatial part, taskprompter.py:201-210 (and vit.py:184-191 with T=0):
 *   qkv   [B, N, 3, nH, 64]  (dtype act)   ->   out [B, N, nH*64] = softmax(q k^T / 8) v
 *   rawlog[B, nH, T, N] fp32 = UNSCALED q.k of the first T query rows (what cal_task_feature reads,
 *   taskprompter.py:436-438,482); never materialises the N x N matrix.  head_dim is fixed at 64.
 */
typedef struct {
  const void* qkv; void* out; float* rawlog; float* lse;   /* lse [B,nH,N] fp32 (optional, for backward) */
  int32_t B, N, nH, T;
  int32_t dtype, prec;
  float scale;
  int32_t variant;   /* MTT_ATTN_* (0 = AUTO: the swapped-product flash kernel for bf16 storage, the plain kernel otherwise) */
  const void* qkv_lo; void* out_lo;   /* dtype == MTT_SPLIT (forward, MTT_PREC_X3): lo planes of qkv / out, same layout as the hi planes */
} mtt_attn_desc;
int mtt_attn_fwd(const mtt_attn_desc* d, void* stream);
/* Flash backward of mtt_attn_fwd (autograd of taskprompter.py:201-210 / vit.py:184-191), bf16 storage + MTT_PREC_BF16 only
 * (the x3 parity mode uses the materialised batched-GEMM backward).  Inputs: d->qkv, d->out (forward output), d->lse (forward
 * log-sum-exp), dout [B*N, nH*64] bf16, drawlog fp32 [B,nH,T,N] = gradient of the rawlog side channel (NULL if T == 0).
 * Outputs: dqkv [B*N, 3*nH*64] bf16 (every row written); stat fp32 [B,nH,2,pad4(N)] scratch (row 0: rowsum(dO*O), row 1:
 * lse*log2(e); pads zero).  No N x N tensor, no atomics. */
int mtt_attn_bwd(const mtt_attn_desc* d, const void* dout, const float* drawlog, void* dqkv, float* stat, void* stream);

/* Row softmax (+ its backward) on a materialised score matrix: used by the InvPT decoder attention
 * (invpt.py:232) and by the round-1 attention backward.  rows x cols, ld in elements.
 *   fwd : P = softmax(scale * S)                         S fp32/bf16 -> P (dtype)
 *   bwd : dS = scale * P * (dP - rowsum(dP*P)) (+ extra[r,c] for r < extra_rows, unscaled)  */
typedef struct {
  const void* S; void* P; const void* dP; void* dS; const float* extra;
  int64_t rows, cols, ld; int32_t s_dtype, p_dtype; float scale;
  int64_t rows_per_mat; int32_t extra_rows; int64_t extra_ld;
} mtt_softmax_desc;
int mtt_softmax_fwd(const mtt_softmax_desc* d, void* stream);
int mtt_softmax_bwd(const mtt_softmax_desc* d, void* stream);

/* LayerNorm over the last dim C (eps 1e-6 ViT: taskprompter.py:310; 1e-5 InvPT: invpt.py:256).
 * x fp32 [rows, C] (ldx) -> y (y_dtype) [rows, C] (ldy); mean/rstd [rows] fp32 saved for backward.
 * bwd: dx (fp32) = dx_in + dLN/dx   (dx_in: same layout as dx, e.g. the residual-stream gradient the LayerNorm branch joins;
 *      NULL = accumulate in place, dx += ...); dgamma / dbeta = the column sums of dy * xhat / dy (overwritten), reduced
 *      deterministically through the caller-owned workspace ws (>= mtt_layernorm_bwd_ws_floats(rows, C) floats; required with dgamma). */
typedef struct {
  const float* x; void* y; const float* gamma; const float* beta; float* mean; float* rstd;
  const void* dy; float* dx; float* dgamma; float* dbeta;
  int64_t rows; int32_t C; int64_t ldx, ldy; int32_t y_dtype; float eps;
  const float* dx_in;
  float* ws;
  void* y_lo;                    /* fwd, y_dtype == MTT_SPLIT: lo plane of y (same pitch ldy) */
  float* y32; int64_t ldy32;     /* fwd, optional: an additional fp32 copy of y (consumers that want the exact rows next to the split planes) */
} mtt_ln_desc;
size_t mtt_layernorm_bwd_ws_floats(int64_t rows, int32_t C);
int mtt_layernorm_fwd(const mtt_ln_desc* d, void* stream);
int mtt_layernorm_bwd(const mtt_ln_desc* d, void* stream);

/* Patchify for the k=s=16 patch-embed conv (timm PatchEmbed used at taskprompter.py:313,393):
 * img fp32 NCHW [B,3,H,W] -> cols [B*h*w, 768] (k = (c, py, px)), dtype out_dtype. */
int mtt_patchify16(const float* img, void* cols, int B, int H, int W, int out_dtype, void* stream);

/* Channel-attention logits, taskprompter.py:217-246:  rawchan[b,t,win,c] = sum_{p in win} q[b,t,p] * xn[b,p,c]
 * q [B,T,hw] (dtype), xn = norm1-ed tokens [B, N, C] (dtype; patch rows start at row T), rawchan fp32 [B,T,nwin,C]. */
typedef struct {
  const void* q; const void* xn; float* rawchan;
  int32_t B, T, N, C, h, w, nh, nw; int32_t dtype; int64_t ldq;
  float* ws;                     /* forward: workspace of mtt_chan_logits_ws_floats(d) floats (may be NULL when that is 0) */
  const void* xn_lo;             /* ABI 9, forward only: dtype == MTT_SPLIT reads the normalised tokens as hi (xn) + lo (xn_lo) bf16 planes — the
                                    LayerNorm output the split-plane GEMMs stream, so no fp32 copy of it is written for this kernel; q is then fp32 */
} mtt_chanlogit_desc;
/* rawchan is WRITTEN.  Reductions over the pixels of a window that are split across workgroups go through per-split partials in the
 * caller-owned workspace and are summed in split order (deterministic; no atomics — as every reduction of this library since ABI 6). */
size_t mtt_chan_logits_ws_floats(const mtt_chanlogit_desc* d);
int mtt_chan_logits(const mtt_chanlogit_desc* d, void* stream);
/* backward: dq[b,t,p] = sum_c drawchan*xn (written, dq_dtype, pitch ldq); dxn[b,T+p,c] += sum_t drawchan*q (fp32 [B,N,C]). */
int mtt_chan_logits_bwd(const mtt_chanlogit_desc* d, const float* drawchan, void* dq, int dq_dtype, float* dxn, void* stream);

/* Task-feature modulation, taskprompter.py:436-467: from x fp32 [B, hw, C] (row pitch/batch stride given)
 *   out[2t  ][b,p,c] = x[b,p,c] * (1 + rawlog[b, c/hg, t, T+p])          (rawlog fp32 [B, C/hg, T, N])
 *   out[2t+1][b,p,c] = x[b,p,c] * (1 + rawchan[b, t, win(p), c])            out dtype act (or MTT_SPLIT: two bf16 planes), [2T, B*hw, C] */
typedef struct {
  const float* x; int64_t x_ld, x_bs; const float* rawlog; const float* rawchan; void* out;
  int32_t B, T, N, C, h, w, nh, nw; int32_t out_dtype;
  int32_t hg;                    /* channels per attention head (a multiple of 8; 0 = 64, the ViT variants; 32 in the last Swin stage) */
  void* out_lo;                  /* ABI 8: out_dtype == MTT_SPLIT writes hi = bf16(v) at out and lo = bf16(v - hi) here (same layout): the
                                    fea_decode GEMMs of the fp32-class forward then stream pre-split planes (gemm_ring3_kernel) */
} mtt_modulate_desc;
int mtt_modulate(const mtt_modulate_desc* d, void* stream);
/* backward: dout [2T, B*hw, C] (d->out_dtype) -> dx = (fp32, same addressing as x; WRITTEN), drawlog[b,head,t,T+p] = (fp32 [B,nH,T,N],
 * caller zeroes the first T columns), drawchan = (fp32 [B,T,nwin,C], WRITTEN: per-split partials in ws, mtt_modulate_bwd_ws_floats(d)
 * floats, summed in split order). */
size_t mtt_modulate_bwd_ws_floats(const mtt_modulate_desc* d);
int mtt_modulate_bwd(const mtt_modulate_desc* d, const void* dout, float* dx, float* drawlog, float* drawchan, float* ws, void* stream);

/* Cross-task reweighting, taskprompter.py:478-485: w[b,t,s] from the per-head MLP on the prompt<->prompt
 * logits, then out[t][b,p,:] = sum_s w[b,t,s] * fea[s][b,p,:] (+= into acc when accumulate=1). */
typedef struct {
  const void* fea; void* out; const float* wmix;   /* fea [T, rows, ld]; wmix [B,T,T] fp32; out [T, rows, ld] (fp32 unless out_dtype says bf16) */
  int32_t T, B; int64_t rows_per_b, ld; int32_t C; int32_t fea_dtype; int32_t accumulate;
  int32_t out_dtype;             /* MTT_F32 (0, default) or MTT_BF16 (mtt_ctr_mix without accumulate only: the backward's gradient of bf16 features) */
} mtt_ctr_desc;
int mtt_ctr_mix(const mtt_ctr_desc* d, void* stream);
/* dwmix[b,t,s] = sum_{rows of b, c} dout[t][row,c] * fea[s][row,c]   (dout fp32 [T, rows, ld]; dwmix WRITTEN; ws: mtt_ctr_dw_ws_floats(d)
 * floats of per-workgroup partials, summed in workgroup order) */
size_t mtt_ctr_dw_ws_floats(const mtt_ctr_desc* d);
int mtt_ctr_dw(const mtt_ctr_desc* d, const float* dout, float* dwmix, float* ws, void* stream);

/* Cross-task reweighting WEIGHTS (taskprompter.py:482-484; modules defined at :364 `ctr_attn_conv`): for every image b and task pair
 * (t, s) the nH prompt<->prompt raw logits z[h] = rawlog[b, h, t, s] (the first T columns of the prompt rows' logits) go through task t's
 * 1x1-conv MLP over the head dimension:   wmix[b, t, s] = b2[t] + sum_j w2[t, j] * gelu(b0[t, j] + sum_h w0[t, j, h] * z[h])   (exact-erf GELU).
 * rawlog fp32 [B, nH, T, N] (row pitch N >= T); w0 [T, nH, nH], b0 [T, nH], w2 [T, nH], b2 [T] fp32 contiguous (ops.stack_vec packs);
 * wmix fp32 [B, T, T] written.  nH <= 32.
 * bwd: given dwmix [B, T, T]: drawlog[b, h, t, s] for s < T is WRITTEN (the other columns of drawlog are not touched: the caller zeroes
 * them), dw0 / db0 / dw2 / db2 are WRITTEN; the sums over (b, s) run inside one workgroup per (t, j) in a fixed order (deterministic,
 * no atomics, no workspace). */
typedef struct {
  const float* rawlog; const float* w0; const float* b0; const float* w2; const float* b2; float* wmix;
  int32_t B, T, nH; int64_t N;
} mtt_ctrw_desc;
int mtt_ctr_weights(const mtt_ctrw_desc* d, void* stream);
int mtt_ctr_weights_bwd(const mtt_ctrw_desc* d, const float* dwmix, float* drawlog, float* dw0, float* db0, float* dw2, float* db2, void* stream);

/* Losses of the 3-D detection branch (TaskPrompter/detection_toolbox/det_losses.py): element-wise loss on [N, C] with the
 * reference's weighting (weight_reduce_loss, :28-54) and a deterministic sum.
 *   kind 0  sigmoid focal loss (:226-345; mmcv-full 1.6.2 `sigmoid_focal_loss` on the device, py_sigmoid_focal_loss on the host, :183-224):
 *           target = int64 labels [N] in [0, C], C = background (no positive class);  p = sigmoid(x);
 *           loss[n, c] = -alpha (1 - p)^gamma log p            if c == target[n]
 *                        -(1 - alpha) p^gamma log(1 - p)        otherwise
 *   kind 1  smooth L1 (:102-123): target fp32 [N, C];  d = |x - y|;  loss = 0.5 d^2 / beta if d < beta else d - 0.5 beta
 * weight: NULL, [N] per sample (wmode 1) or [N, C] per element (wmode 2).  fwd: out[n, c] = w * loss (skipped when NULL), sum[0] = the sum of
 * out over all elements (skipped when NULL; ws: mtt_detloss_ws_floats(d) floats of per-workgroup partials summed in workgroup order).
 * bwd: dpred[n, c] = w * d loss / d x * (gelem ? gelem[n, c] : gscale[0]) * scale   (gelem: the upstream gradient of reduction 'none';
 * gscale: the upstream scalar of 'mean' / 'sum', read on the device; scale = loss_weight / avg_factor etc., a host constant). */
typedef struct {
  const float* pred; const void* target; const float* weight; float* out; float* sum; float* ws;
  int64_t N; int32_t C; int32_t kind; int32_t wmode; float gamma, alpha, beta;
} mtt_detloss_desc;
size_t mtt_detloss_ws_floats(const mtt_detloss_desc* d);
int mtt_detloss_fwd(const mtt_detloss_desc* d, void* stream);
int mtt_detloss_bwd(const mtt_detloss_desc* d, const float* gscale, const float* gelem, float scale, float* dpred, void* stream);

/* Bilinear resize, align_corners=False (F.interpolate at taskprompter.py:420, taskprompter_wrapper.py:36,
 * invpt.py:221,303,537, transformer_net.py:35-36).  NHWC in -> NHWC out or NCHW fp32 out.
 * bwd: din (fp32) += scatter of dout (atomics). */
typedef struct {
  const void* in; void* out; int32_t B, C, Hin, Win, Hout, Wout; int64_t ld_in, ld_out;
  int32_t in_dtype, out_dtype; int32_t out_nchw; int32_t accumulate;
} mtt_resize_desc;
int mtt_bilinear_fwd(const mtt_resize_desc* d, void* stream);
int mtt_bilinear_bwd(const mtt_resize_desc* d, void* stream);   /* in = dout, out = din (fp32) */

/* BatchNorm2d (+GELU/ReLU) on Z stacked NHWC maps [Z][rows, ld] (one BatchNorm per map; Z <= 1 = a single map), training mode
 * (batch statistics; nn.BatchNorm2d at taskprompter.py:362,692,705; SyncBatchNorm invpt.py:14).  Map z starts x_zs elements after
 * map z-1; its per-channel vectors (mean, rstd, gamma, beta, outputs) start p_zs floats after the previous map's.
 *   stats     : mean_out[z][c] = batch mean, m2_out[z][c] = sum_r (x - mean)^2 (centred; biased variance = m2 / rows).  Two-level
 *               deterministic reduction through the caller's workspace `ws` (mtt_bn_reduce_ws_floats(rows, C, Z) floats); no atomics,
 *               nothing to zero.  (mean, m2, rows) triplets of several ranks merge exactly (SyncBatchNorm) with Chan's update.
 *   apply     : y = act((x-mean)*rstd*gamma+beta), channels C..ld-1 written as zeros.
 *   bwd_reduce: dsum[z][c] = sum_r du, dsumxh[z][c] = sum_r du*xhat with du = dy*act'(u)   (same workspace, written not accumulated)
 *   bwd_apply : dx = gamma*rstd*(du - dsum/rows - xhat*dsumxh/rows)   (multi-rank: pass the all-reduced sums scaled by rows/rows_total) */
typedef struct {
  const void* x; void* y; const void* dy; void* dx;
  float* mean_out; float* m2_out; const float* mean; const float* rstd; const float* gamma; const float* beta;
  float* dsum; float* dsumxh;
  int64_t rows; int32_t C; int64_t ld; int32_t dtype; int32_t act;
  int32_t Z; int64_t x_zs; int64_t p_zs;
  int32_t g_dtype;   /* ABI 9, bwd_reduce / bwd_apply: storage of dy and dx = MTT_* dtype code + 1 (same pitch and map stride in ELEMENTS as
                      * x); 0 = the dtype of x.  Lets a bf16-arithmetic backward keep its gradient maps in bf16 next to an fp32-stored x */
} mtt_bn_desc;
size_t mtt_bn_reduce_ws_floats(int64_t rows, int32_t C, int32_t Z);
int mtt_bn_stats(const mtt_bn_desc* d, float* ws, void* stream);
int mtt_bn_apply(const mtt_bn_desc* d, void* stream);
int mtt_bn_bwd_reduce(const mtt_bn_desc* d, float* ws, void* stream);
int mtt_bn_bwd_apply(const mtt_bn_desc* d, void* stream);

/* fp32 [rows, cols] (pitch lds) -> MTT_SPLIT planes hi / lo [rows, ldd] (bf16; columns cols..ldd-1 zero): hi = bf16(x), lo = bf16(x - hi) */
int mtt_split_cast(const float* src, void* hi, void* lo, int64_t rows, int64_t cols, int64_t lds, int64_t ldd, void* stream);

/* Multi-segment strided copy / cast: ONE launch re-packs every weight of a model (the reference re-reads nn.Parameter tensors directly;
 * the packs are this implementation's operand layouts: [N, pad8(K)] casts, tap-major conv matrices, padded concatenations, hi/lo planes)
 * and scatters packed weight gradients back to parameter layout.  Segment g copies the logical box [n0][n1][n2] (n2 innermost):
 *     dst[i0*d0 + i1*d1 + i2*d2] = cast(src[i0*s0 + i1*s1 + i2*s2])
 * `table` is a DEVICE array of n_seg rows of MTT_SEG_WORDS int64 words:
 *     0 src address, 1 dst address, 2 dst_lo address (MTT_SPLIT destinations: hi -> dst, lo -> dst_lo; else 0), 3 total = n0*n1*n2,
 *     4 n1, 5 n2, 6..8 s0 s1 s2, 9..11 d0 d1 d2 (elements), 12 src dtype, 13 dst dtype, 14 vec (1: n2, s0, s1, d0, d1 multiples of 4,
 *     s2 = d2 = 1 and both addresses 16-byte aligned -> four elements per lane), 15 transpose (1: s1 = 1 and d2 = 1 — the source is
 *     contiguous along i1, the destination along i2; copied in 64 x 64 tiles of (i1, i2) through LDS; word 3 is then
 *     n0 * ceil(n1/64) * ceil(n2/64) * 4096 and chunk offsets count tile slots of 4096).
 * src_base / dst_base (bytes) are added to every src / dst(+dst_lo) address: 0 for tables of absolute addresses (persistent packs),
 * the buffers' addresses for tables of offsets (gradient scatter out of a fresh buffer).  Work is cut into chunks of
 * mtt_segcopy_chunk() logical elements: chunk c covers [chunk_off[c], chunk_off[c] + chunk) of segment chunk_seg[c].
 * Destination elements outside the boxes (padding) are not touched: the caller zeroes the buffers once. */
#define MTT_SEG_WORDS 16
typedef struct {
  const int64_t* table; const int32_t* chunk_seg; const int64_t* chunk_off; int32_t n_chunks;
  int64_t src_base, dst_base;
} mtt_segcopy_desc;
int mtt_segcopy_chunk(void);
int mtt_segcopy(const mtt_segcopy_desc* d, void* stream);

/* Small utilities: dtype cast / strided 2-D copy, column sums (bias gradients), axpy-style accumulate. */
int mtt_cast2d(const void* src, void* dst, int64_t rows, int64_t cols, int64_t lds, int64_t ldd,
               int src_dtype, int dst_dtype, int zero_pad_cols, void* stream);
/* ABI 12.  Pixel shuffle of nn.ConvTranspose2d(k = 2, s = 2) (taskprompter.py:705) computed as a PLAIN GEMM z = x W4^T, W4 row (dy*2+dx)*Co + co:
 *   out[((b*2H + 2y+dy)*2W + 2x+dx), co] = z[(b*H + y)*W + x, (dy*2+dx)*Co + co]   (co < Co; channels Co .. ldo-1 written as zeros)
 * z [B*H*W, ldz] (z_dtype), out [B*2H*2W, ldo] (out_dtype); ldo % 8 == 0.  (mtt_gemm's MTT_STORE_PIXSHUF2 does the same inside the general
 * kernel's epilogue; this pass lets the product run on the split-plane LDS-DMA kernel in the x3f mode.) */
int mtt_pixshuf2(const void* z, void* out, int32_t B, int32_t H, int32_t W, int32_t Co, int64_t ldz, int64_t ldo, int z_dtype, int out_dtype,
                 void* stream);
/* dst[r,:] = rowscale[(r/mb)*2 + ((r%mb) >= n_prompt)] * src[r,:] with dtype cast (DropPath scale of a branch gradient) */
int mtt_rowscale_cast(const void* src, void* dst, int64_t rows, int32_t cols, int64_t lds, int64_t ldd, int src_dtype, int dst_dtype,
                      const float* rowscale, int32_t mb, int32_t n_prompt, void* stream);
/* the same + colsum_out[c] = sum_r dst[r, c] of the values as stored (c < cols): when src is the gradient of a Linear's output (the
 * residual-stream gradient entering proj / fc2), the cast that prepares it for the bf16 GEMMs also yields that layer's bias gradient.
 * cols, lds, ldd multiples of 8; ws: mtt_rowscale_cast_colsum_ws_floats(rows, cols) floats of per-row-block partials (fixed-order sum). */
size_t mtt_rowscale_cast_colsum_ws_floats(int64_t rows, int32_t cols);
int mtt_rowscale_cast_colsum(const void* src, void* dst, int64_t rows, int32_t cols, int64_t lds, int64_t ldd, int src_dtype, int dst_dtype,
                             const float* rowscale, int32_t mb, int32_t n_prompt, float* colsum_out, float* ws, void* stream);
/* dst[c] = sum_r src[r, c] for c < cols (bias gradients; ld >= pad8(cols), columns up to pad8(cols) are read).  Deterministic two-stage
 * reduction through the caller-owned workspace ws (>= mtt_colsum_ws_floats(rows, cols) floats, contents unspecified afterwards). */
size_t mtt_colsum_ws_floats(int64_t rows, int32_t cols);
int mtt_colsum(const void* src, float* dst, int64_t rows, int32_t cols, int64_t ld, int src_dtype, float* ws, void* stream);
/* Z maps in one launch pair: map z starts src_zs elements after map z - 1, its sums dst_zs floats after those of z - 1;
 * ws >= Z * mtt_colsum_ws_floats(rows, cols) floats. */
int mtt_colsum_batched(const void* src, float* dst, int64_t rows, int32_t cols, int64_t ld, int src_dtype, int32_t Z, int64_t src_zs,
                       int64_t dst_zs, float* ws, void* stream);
int mtt_add_rows(const void* src, float* dst, int64_t rows, int32_t cols, int64_t lds, int64_t ldd, int src_dtype,
                 float alpha, void* stream);


/* ---------------------------------------------------------------------------------------------------------
 * InvPT decoder kernels (InvPT/models/transformers/invpt.py, transformer_decoder.py)
 * --------------------------------------------------------------------------------------------------------- */
/* Depthwise 3x3 stride-2 pad-1 conv, the query projection of invpt.py:124-138 (bias-free), Z tasks at once.
 * x [Z, B*H*W, ld] -> y [Z, B*Ho*Wo, ld], Ho = (H-1)/2+1; w fp32 [Z, 9, ld] (tap-major); optional per-channel
 * scale/shift [Z, ld] = eval BatchNorm folded in. */
typedef struct {
  const void* x; const float* w; void* y; const float* scale; const float* shift;
  int32_t Z, B, H, W; int64_t ld; int32_t dtype;
} mtt_dwconv_desc;
int mtt_dwconv3x3s2(const mtt_dwconv_desc* d, void* stream);
/* backward (scale/shift ignored: apply BN backward first): dx [Z, B*H*W, ld] (same dtype, optional), dw fp32 [Z, 9, ld] (optional) */
int mtt_dwconv3x3s2_bwd(const mtt_dwconv_desc* d, const void* dy, void* dx, float* dw, void* stream);

/* nn.AvgPool2d(kernel=stride=k, padding=0, ceil_mode=True) on NHWC (keys/values, invpt.py:139-149). */
typedef struct { const void* x; void* y; int32_t B, H, W, k; int64_t ld; int32_t dtype; } mtt_pool_desc;
int mtt_avgpool_ceil(const mtt_pool_desc* d, void* stream);
int mtt_avgpool_ceil_bwd(const mtt_pool_desc* d, const void* dy, void* dx, void* stream);

/* LayerNorm over the concatenated channels of all T tasks (norm_mts, invpt.py:482,526): x fp32 [T, rows, ldx],
 * gamma/beta fp32 [T*D] -> y [T, rows, ldy] (channels >= D written as zeros). */
typedef struct {
  const float* x; void* y; const float* gamma; const float* beta;
  int64_t rows; int32_t T, D; int64_t ldx, ldy; int32_t y_dtype; float eps;
} mtt_lnmt_desc;
int mtt_layernorm_mt(const mtt_lnmt_desc* d, void* stream);

/* Cross-stage attention message fusion (invpt.py:208-229): previous-stage scores upsampled x2 per task and mixed with
 * the current scores by the 1x1 conv fuse_attn [heads, 2*heads].  cur/out fp32 [B, heads, T*qh*qw, ldk],
 * prev fp32 [B, heads, T*(qh/2)*(qw/2), ldkp]; K valid key columns. */
typedef struct {
  const float* cur; const float* prev; float* out; const float* w; const float* bias;
  int32_t B, heads, T, qh, qw, K; int64_t ldk, ldkp;
} mtt_attnmsg_desc;
int mtt_attn_msg(const mtt_attnmsg_desc* d, void* stream);
/* backward of mtt_attn_msg (d->out unused): dcur, dup fp32 [B, heads, T*qh*qw, ldk] = gradients w.r.t. `cur` and w.r.t. the x2-upsampled
 * `prev` (the gradient of `prev` itself is mtt_bilinear_bwd of dup); dw [heads, 2*heads] and dbias [heads] are WRITTEN (per-workgroup
 * partials in ws, mtt_attn_msg_bwd_ws_floats(d) floats, summed in workgroup order). */
size_t mtt_attn_msg_bwd_ws_floats(const mtt_attnmsg_desc* d);
int mtt_attn_msg_bwd(const mtt_attnmsg_desc* d, const float* dout, float* dcur, float* dup, float* dw, float* dbias, float* ws, void* stream);

/* Gather half of nn.ConvTranspose2d(k=3, s=2, p=1, output_padding=1) (scale_embed[0], transformer_decoder.py:64):
 * yall [B*H*W, 9*Cop] = x @ Wall^T computed by mtt_gemm (column = tap*Cop + co) -> out [B*2H*2W, Cop] (+ bias[Cop]). */
typedef struct {
  const void* yall; void* out; const float* bias; int32_t B, H, W, Cop; int32_t dtype, out_dtype;
} mtt_convt_desc;
int mtt_convt3x3s2_gather(const mtt_convt_desc* d, void* stream);
/* backward: dyall [B*H*W, 9*Cop] (d->dtype) gathered from dout [B*2H*2W, Cop] (d->out_dtype) */
int mtt_convt3x3s2_gather_bwd(const mtt_convt_desc* d, const void* dout, void* dyall, void* stream);

/* ConvHead's first stage on the backbone's low-resolution task features: F.interpolate(x, scale_factor=4, 'bilinear') followed by
 * Conv2d(C, C, 3, padding=1)  (taskprompter.py:420 -> :692), evaluated "taps first".  Both operators are linear, so
 *   conv3x3(up4(x))(Y, X) = bias + sum over taps (ky, kx) of [ up4(W[ky,kx] x) ](Y + ky - 1, X + kx - 1)   (zero outside the 4h x 4w map):
 * the channel mixing runs as ONE GEMM on the h x w map (mtt_gemm, N = 9 * Cp: 16x fewer MACs than the conv on the upsampled map), and
 * these two entry points do what is left — an HBM-bound 36-point stencil — without materialising the upsampled features:
 *   mtt_upconv4_expand : z [Z][B*h*w][9*Cp] (column (ky*3+kx)*Cp + c holds W[ky,kx] x, channel padding zero)
 *                        -> y [Z][B*4h*4w][Cp] = act((sum of the shifted x4 bilinear expansions) * colscale + bias)
 *   mtt_upconv4_gather : the adjoint, dz [Z][B*h*w][9*Cp] from dy [Z][B*4h*4w][Cp] (the caller applies act' / BatchNorm backward first).
 * Bilinear semantics are PyTorch's align_corners=False (source index clamped at the borders); the x4 phase weights are the constants
 * 1/8, 3/8, 5/8, 7/8.  bias / colscale: fp32 [Z][C] or NULL.  z_dtype / y_dtype: MTT_F32 or MTT_BF16 (both the same). */
typedef struct {
  void* z; void* y; const float* bias; const float* colscale;
  int32_t Z, B, h, w, C, Cp; int32_t z_dtype, y_dtype, act;
} mtt_upconv_desc;
int mtt_upconv4_expand(const mtt_upconv_desc* d, void* stream);
int mtt_upconv4_gather(const mtt_upconv_desc* d, void* stream);

/* Fused multi-tensor clip_grad_norm_ + Adam (TaskPrompter/utils/train_utils.py:47-51, torch.optim.Adam semantics: L2 weight decay
 * added to the gradient, bias-corrected moments).  All n tensors are fp32; grads/params/exp_avg/exp_avg_sq/numel are DEVICE arrays
 * of n pointers / element counts; work is cut into chunks of mtt_adam_chunk() elements: chunk c covers elements
 * [chunk_off[c], chunk_off[c] + chunk) of tensor chunk_tensor[c] (device arrays of n_chunks entries, built once by the host).
 *   mtt_grad_sqnorm: *out_sq += sum of squares of all gradients (out_sq zeroed by the caller)
 *   mtt_adam_step  : g = grad * min(1, max_norm / (sqrt(*total_sq) + 1e-6))  (no clipping if max_norm <= 0 or total_sq NULL);
 *                    g += weight_decay * p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *                    p -= step_size * m / (sqrt(v) * inv_sqrt_bc2 + eps)   with step_size = lr / (1 - b1^t), inv_sqrt_bc2 = 1/sqrt(1 - b2^t) */
typedef struct {
  const void* const* grads; float* const* params; float* const* exp_avg; float* const* exp_avg_sq; const int64_t* numel;
  const int32_t* chunk_tensor; const int64_t* chunk_off; int32_t n_chunks;
  float max_norm, step_size, beta1, beta2, eps, weight_decay, inv_sqrt_bc2;
  const float* hyper;  /* optional DEVICE pair {step_size, inv_sqrt_bc2}: when non-NULL it replaces the two by-value fields, so that a step
                          captured in a hipGraph can be replayed with the next step count / learning rate (graphs.py) */
  float* ws;           /* mtt_grad_sqnorm: n_chunks floats of per-chunk partial sums (summed in chunk order: a deterministic clip norm) */
} mtt_adam_desc;
int mtt_adam_chunk(void);
int mtt_grad_sqnorm(const mtt_adam_desc* d, float* out_sq, void* stream);
int mtt_adam_step(const mtt_adam_desc* d, const float* total_sq, void* stream);

/* Per-task training losses from the full-resolution logits (TaskPrompter/losses/loss_functions.py:15-177; MultiTaskLoss,
 * loss_schemes.py:9-39, sums them with the task weights on the host side).  kind: 0 CrossEntropy(ignore) | 1 CrossEntropy with
 * 2-class label-frequency weights (`balanced`) | 2 balanced BCE-with-logits (pos_weight) | 3 L1(ignore) | 4 L1 on L2-normalised pred.
 * pred fp32 [B, C, HW] (NCHW), label fp32 [B, Cl, HW]; a pixel is valid when all its Cl label channels != ignore.
 *   mtt_loss_label_stats: stats[0] += number of valid pixels, stats[1] += sum of valid labels (channel 0)   (stats zeroed by caller)
 *   mtt_loss_fwd        : loss[0] += task loss (mean over valid pixels; normalisation read from `stats` on the device)
 *   mtt_loss_bwd        : dpred [B, C, HW] = gout[0] * d loss / d pred */
typedef struct {
  const float* pred; const float* label; float* dpred; float* loss; const float* stats;
  int64_t B, HW; int32_t C, Cl, kind; float ignore, pos_weight;
  float* ws;           /* label_stats / fwd: mtt_loss_ws_floats(d) floats of per-workgroup partials (summed in workgroup order) */
} mtt_loss_desc;
size_t mtt_loss_ws_floats(const mtt_loss_desc* d);
int mtt_loss_label_stats(const mtt_loss_desc* d, float* stats, void* stream);
int mtt_loss_fwd(const mtt_loss_desc* d, void* stream);
int mtt_loss_bwd(const mtt_loss_desc* d, const float* gout, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------------
 * TaskPrompter-Swin forward (TaskPrompter/models/transformers/taskprompter_swin.py).
 * --------------------------------------------------------------------------------------------------------------------------- */

/* Patchify for a k = s = P patch-embed conv, P in {2, 4, 8} (timm PatchEmbed, taskprompter_swin.py:605-607):
 * img fp32 NCHW [B,3,H,W] -> cols [B*(H/P)*(W/P), ldc] (k = (c*P + dy)*P + dx, columns >= 3*P*P zero), dtype out_dtype. */
int mtt_patchify(const float* img, void* cols, int B, int H, int W, int P, int64_t ldc, int out_dtype, void* stream);

/* Bilinear resize (align_corners = False, as F.interpolate) of `planes` fp32 maps [Hin, Win] -> [Hout, Wout]: the img_ds_ratio
 * input resize (taskprompter_swin.py:666-667) on the NCHW image. */
int mtt_resize_nchw(const float* in, float* out, int planes, int Hin, int Win, int Hout, int Wout, void* stream);

/* Row gather with zero fill: for b < B, r < rows, c < C:
 *   dst[b*dst_bs + r*ld_dst + c] = idx[b*idx_bs + r] >= 0 ? src[b*src_bs + idx[...]*ld_src + c] : 0      (dtype conversion on the way)
 * One kernel for the window partition of the (cyclically shifted, zero padded) token map with the prompts joined to every window
 * (taskprompter_swin.py:340-354,175-177), its inverse (:366-384), and the 2x2 patch-merging gather (:450-456: 4 calls, one per column
 * block of dst).  C % 8 == 0; pitches / strides in elements. */
typedef struct {
  const void* src; void* dst; const int32_t* idx;
  int64_t rows; int32_t C; int64_t ld_src, ld_dst; int32_t src_dtype, dst_dtype;
  int32_t B; int64_t src_bs, dst_bs, idx_bs;
  int32_t skip_neg;              /* 1: rows with idx < 0 are left untouched instead of zero-filled (scatter-style adjoints) */
} mtt_gather_desc;
int mtt_gather_rows(const mtt_gather_desc* d, void* stream);

/* Window attention with the task prompts as the first T tokens of every window (taskprompter_swin.py:168-210), head_dim 32:
 *   qkv (dtype) [nwin, N = T + ws2, 3*nH*32] -> out (dtype) [nwin, N, nH*32],
 *   S = scale * q k^T;  S[i >= T, j >= T] += bias[h, i-T, j-T] (+ mask[w % nW, i-T, j-T] when mask != NULL);  P = softmax(S);  out = P v
 *   rawmap fp32: the UNSCALED q.k of the prompt rows against the window's pixels, written straight into image layout
 *                rawmap[(b*nH + h)*T + t][map_off + pix[w % nW][j]]  (row pitch map_ld; pix < 0 = padding pixel, dropped; b = w / nW)
 *                — the re-assembly, reverse shift and un-padding of :376-388 folded into the store.
 * bias fp32 [nH, ws2, ws2] (the relative-position table gathered by the host once per parameter version), mask fp32 [nW, ws2, ws2]
 * (0 / -100), pix int32 [nW, ws2].  N <= 160. */
typedef struct {
  const void* qkv; void* out; float* rawmap; const float* bias; const float* mask; const int32_t* pix;
  int32_t nwin, nW, nH, T, ws2, dtype; float scale; int64_t map_ld, map_off;
  int32_t mfma;                  /* ABI 11, fp32 storage only: 1 = matrix-core arithmetic — forward: every product as 3 bf16 MFMAs on hi / lo split
                                    operands (fp32-class, the x3 / x3f modes); backward: bf16 MFMAs (the bf16 backward of the x3f mode).
                                    0 = the exact fp32 VALU kernels.  bf16 storage always runs on the matrix cores. */
  const float* biasT;            /* ABI 13, optional (mtt_winattn_bwd, matrix-core kernels): bias with its last two axes transposed, fp32 [nH, ws2, ws2] —
                                    the key-owner pass then reads bias / mask of 4 consecutive queries with one 16-byte load each (the shift mask
                                    must be symmetric, as Swin's region mask is).  NULL = strided scalar loads. */
} mtt_winattn_desc;
int mtt_winattn_fwd(const mtt_winattn_desc* d, void* stream);
/* backward (d as in the forward, d->out = the forward's output): dout (dtype) [nwin, N, nH*32]; drawmap fp32 = gradient of rawmap (same
 * layout; NULL = none) -> dqkv (dtype) [nwin, N, 3*nH*32] (written), dS_out fp32 [nwin, nH, ws2, ws2] = dS of the window x window
 * part (written when not NULL; the relative-position-bias gradient is its sum over the windows). */
int mtt_winattn_bwd(const mtt_winattn_desc* d, const void* dout, const float* drawmap, void* dqkv, float* dS_out, void* stream);

/* Channel attention of the prompts over the feature channels (taskprompter_swin.py:391-409), windows over the sqrt(ce) x sqrt(ce) grid
 * of the channel-embedding dimension:
 *   q fp32 [B, T, ce];  kvT (dtype) [B, 2*ce, ldk]: rows 0..ce-1 = k^T, ce..2ce-1 = v^T, columns = feature channels c < C
 *   rawchan[b, t, win, c] = sum_{e in win} q[b,t,e] * kT[b,e,c]        (fp32 [B, T, nwin, C], unscaled)
 *   cx[b, t, e in win]    = sum_c softmax_c(scale * rawchan[b,t,win,:]) * vT[b,e,c]      (fp32 [B, T, ce]) */
typedef struct {
  const float* q; const void* kvT; float* rawchan; float* cx;
  int32_t B, T, C, ce, nh, nw, kv_dtype; int64_t ldk; float scale;
  const float* kvbias;           /* fp32 [2*ce] added to the rows of kvT (the bias of the Linear that produced them), or NULL */
} mtt_chanattn_desc;
int mtt_chanattn_fwd(const mtt_chanattn_desc* d, void* stream);
/* backward (d as in the forward with d->rawchan = the forward's logits, d->kvbias NULL): drawchan fp32 [B, T, nwin, C] (NULL = none) and
 * dcx fp32 [B, T, ce] -> dq fp32 [B, T, ce] and dkvT fp32 [B, 2*ce, ldg] (rows: d kT, then d vT), both WRITTEN; ws: mtt_chanattn_bwd_ws_floats(d)
 * floats.  Fixed summation orders, no atomics. */
size_t mtt_chanattn_bwd_ws_floats(const mtt_chanattn_desc* d);
int mtt_chanattn_bwd(const mtt_chanattn_desc* d, const float* drawchan, const float* dcx, float* dq, float* dkvT, int64_t ldg, float* ws, void* stream);

/* Small-channel 3x3 stride-2 pad-1 convolution on fp32 maps (PatchMerging.spa_attn_ds, taskprompter_swin.py:437,463):
 *   x[b][ci][y*W + x] at x + b*x_bs + ci*x_cs + x_off;  w fp32 [Co, Ci, 3, 3];  y[b][co][...] at y + b*y_bs + co*y_cs + y_off  (H, W even) */
typedef struct {
  const float* x; const float* w; const float* bias; float* y;
  int32_t B, Ci, Co, H, W; int64_t x_bs, x_cs, x_off, y_bs, y_cs, y_off;
} mtt_conv3s2_desc;
int mtt_conv3s2_nchw(const mtt_conv3s2_desc* d, void* stream);
/* backward: dy laid out as y -> dx laid out as x (written for every (b, ci, pixel); may be NULL), dw fp32 [Co, Ci, 3, 3] and db fp32 [Co]
 * (written; dw NULL skips both; db may be NULL) */
int mtt_conv3s2_nchw_bwd(const mtt_conv3s2_desc* d, const float* dy, float* dx, float* dw, float* db, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------------
 * Bird's-eye-view rotated boxes [x1, y1, x2, y2, ry] of the 3-D detection branch (TaskPrompter/detection_toolbox/iou3d: iou3d_kernel.cu
 * :241-275 pairwise kernels, :277-377 NMS mask kernels, iou3d.cpp:103-203 greedy reduction).
 *   mtt_boxes_overlap_bev: out fp32 [na, nb] = overlap area (iou = 0) or IoU (iou = 1) of every pair
 *   mtt_nms_bev          : boxes [n, 5] ALREADY sorted by descending score; rotated = 1 (nms_gpu) / 0 (nms_normal_gpu: axis-aligned IoU);
 *                          keep int64 [n] <- kept indices in score order, *num_out <- their count, both on the DEVICE (no host round trip:
 *                          the reference copies the suppression masks to the host and reduces there); ws >= mtt_nms_ws_bytes(n) bytes. */
int mtt_boxes_overlap_bev(const float* boxes_a, int na, const float* boxes_b, int nb, float* out, int iou, void* stream);
size_t mtt_nms_ws_bytes(int n);
int mtt_nms_bev(const float* boxes, int n, float thresh, int rotated, long long* keep, int* num_out, void* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MTT_HIP_H */
