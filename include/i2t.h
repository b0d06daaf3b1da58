/*
 * i2t.h -- C ABI of libi2t_hip.so: the gfx950 (MI355X) kernels under the image-captioning hot path.
 *
 * The reference (iitmdinesh/image2text) is pure Python over torch ops; it has no FFI of its own.  Each entry
 * point below replaces the torch dispatch site cited next to it (paths relative to the reference repo), and is
 * what a reference-side ctypes binding would call (INTEGRATION.md shows that binding).
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the caller (the PyTorch caching allocator);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); no hidden allocation, no host sync,
 *     so every call is legal inside hipGraph stream capture;
 *   - return 0 on success, negative I2T_E* otherwise; never throws.  i2t_last_error() gives the message;
 *   - bf16 tensors are raw uint16 bit patterns; "f32" = IEEE float; row-major with an explicit leading dimension
 *     (elements).  bf16 leading dimensions and base pointers must be multiples of 8 elements / 16 bytes;
 *   - all arithmetic accumulates in fp32.
 */
#ifndef I2T_H
#define I2T_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I2T_OK 0
#define I2T_EINVAL (-1)   /* bad argument (shape/alignment/unsupported size) */
#define I2T_EHIP (-2)     /* a HIP runtime call failed */

#define I2T_ABI_VERSION 2

int i2t_abi_version(void);
/* copies the last error message of the calling thread into buf (NUL-terminated); returns its length */
int i2t_last_error(char* buf, size_t n);

/* ---------------------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = epilogue(alpha * op(A)[M,K] . op(B)[K,N])           bf16 MFMA 16x16x32, fp32 accumulate
 *   replaces: every nn.Linear on the path -- layers.py:437-439,452,469 (c_attn/c_proj), :476-485 (MLP),
 *   :537-542 (cross-attn in/out proj), encoder.py:147-149 (projector), vision_encoder_decoder.py:34-37 (bridge),
 *   decoder.py:189,256 (tied lm_head) -- and their autograd backward (dX = dY.W, dW = dY^T.X).
 *   a_kmajor = 0: A stored [M][K] (K contiguous)      a_kmajor = 1: A stored [K][M] (M contiguous)
 *   b_kmajor = 0: B stored [N][K] (K contiguous)      b_kmajor = 1: B stored [K][N] (N contiguous)
 *   epilogue order: v = alpha*acc; v += bias[n]; if aux_out: aux_out[m][n] = bf16(v)  (pre-activation)
 *                   act 1: v = gelu_tanh(v);  act 2: v *= gelu_tanh'(aux_in[m][n])
 *                   dropout (training): drop_mode 1: v = keep(key, m*N+n) ? v*scale : 0  (resid / MLP dropout,
 *                   layers.py:469,485); drop_mode 2: v *= keep(key + n/(N/3), m) ? scale : 0  (the per-token q/k/v
 *                   multipliers of layers.py:454-461 on the fused c_attn output); keep(key, i) = u8(key, i) >= thr, where u8 is
 *                   byte (i & 3) of lowbias32((i >> 2) ^ key) and thr = round(p * 256); scale = 256 / (256 - thr)
 *                   v += residual[m][n] (f32);  accumulate: v += C[m][n] (f32 C only);  store C as f32 or bf16.
 *   K must be a multiple of 8.  Columns [N, ldc) of C are never written.
 * --------------------------------------------------------------------------------------------------------- */
#define I2T_ACT_NONE 0
#define I2T_ACT_GELU 1
#define I2T_ACT_DGELU 2
#define I2T_ACT_GELU_ERF 3  /* exact GELU x Phi(x) (torchvision ViT MLP); generic epilogue class only */
#define I2T_ACT_DGELU_ERF 4 /* v *= gelu_erf'(aux_in[m][n]) */
#define I2T_ACT_GELU_DOUT 5 /* C = gelu_tanh(v) and aux_out = gelu_tanh'(v) (bf16) instead of the pre-activation: the layer's backward is then */
#define I2T_ACT_MUL_AUX 6   /* v *= aux_in[m][n] -- one multiply per element where I2T_ACT_DGELU re-evaluates exp + rcp (the GELU' epilogue */
                            /* of a K = 512 GEMM cost ~8 us of VALU per 256 x 256 tile against 11.6 us of MFMA work) */
int i2t_gemm_bf16(void* stream,
                  const void* A, int lda, int a_kmajor,
                  const void* B, int ldb, int b_kmajor,
                  void* C, int ldc, int c_is_f32,
                  int M, int N, int K, float alpha,
                  const float* bias, int act,
                  const void* aux_in, int ld_aux_in,
                  void* aux_out, int ld_aux_out,
                  const float* residual, int ldr,
                  int accumulate,
                  int drop_mode, unsigned drop_key, unsigned drop_thr, float drop_scale);
/* The same call with a gradient normaliser folded in: alpha_sumsq (device scalar, or NULL) = sum(g^2) of the tensor the A operand was
 * cut from; alpha is then multiplied by 1 / (sqrt(*alpha_sumsq) + 1e-6) on the device (reference models/functions.py:19-24 divides the
 * block-output gradient by its norm + 1e-6: the backward GEMMs of the block's last linear read the UN-normalised bf16 gradient and
 * apply the factor in their epilogues, so no pass over the gradient exists just to rescale it). */
int i2t_gemm_bf16_ex(void* stream,
                  const void* A, int lda, int a_kmajor,
                  const void* B, int ldb, int b_kmajor,
                  void* C, int ldc, int c_is_f32,
                  int M, int N, int K, float alpha,
                  const float* bias, int act,
                  const void* aux_in, int ld_aux_in,
                  void* aux_out, int ld_aux_out,
                  const float* residual, int ldr,
                  int accumulate,
                  int drop_mode, unsigned drop_key, unsigned drop_thr, float drop_scale,
                     const float* alpha_sumsq);

/* Fused cross-attention forward (reference models/layers.py:537-542,600-605: nn.MultiheadAttention over the encoder output):
 *   kv[b][key][0:d | d:2d] = mem[b][key][:] . [W_k ; W_v]^T + bias_kv          (bf16, written once: the backward pass reads it)
 *   o[q][64 h ..] = softmax(Q_h K_h^T / 8) V_h   per image b and head h,  lse as i2t_attention_fwd
 * in ONE launch: the K/V projection is the persistent 256 x 256 MFMA GEMM, tiled so that a wave ends its K loop holding K_h(b)^T and
 * V_h(b) of one (image, head) in its accumulators, and that wave runs the image's attention straight out of those registers
 * (csrc/gemm.hip::xattn_epilogue) -- K and V are never read back from HBM in the forward.
 * mem bf16 [B*S][ld_mem] (S must be 64), w_kv = in_proj_weight rows d..3d (bf16 [2d][ld_w], d = 64 H, H even), bias_kv f32 [2d];
 * q / o bf16: [B][Tq] views (batch / row strides) or, with cu_q (device int[B+1]), packed rows (batch strides unused, Tq = max);
 * dropout on the probabilities: same index space and rule as i2t_attention_fwd (drop_thr 0 = off), so i2t_attention_bwd on the
 * stored q / kv / o / lse is its backward. */
int i2t_xattn_kv_fused(void* stream, const void* mem, int ld_mem, const void* w_kv, int ld_w, const float* bias_kv,
                       const void* q, long q_bs, int q_rs, const int* cu_q, int total_q, void* kv, int ld_kv, void* o,
                       long o_bs, int o_rs, float* lse, int B, int S, int H, int Tq, unsigned drop_key, unsigned drop_thr,
                       float drop_scale);

/* column sums: out[n] (+)= sum_m X[m][n]  (bias gradients; X bf16 [M][ld]) */
/* i2t_gemm_bf16 for row-major operands with few output tiles and a long K (a decode step at a mid-sized caption batch): K is cut
 * into slices, each (tile, slice) workgroup writes its raw accumulators to its own [M][N] fp32 plane of `workspace` (ws_floats
 * floats), a second launch adds the planes in slice order and applies the epilogue -- bias, act NONE | GELU, fp32 residual, bf16 or
 * fp32 C.  Deterministic (no atomics).  Falls through to i2t_gemm_bf16 when splitting does not pay, M <= 64 (the weight-streaming
 * kernel), M > 2048 (the planes' traffic outweighs the idle CUs) or the workspace is too small. */
int i2t_gemm_bf16_ws(void* stream, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_is_f32, int M, int N, int K,
                     const float* bias, int act, const float* residual, int ldr, float* workspace, long ws_floats);

/* The large-tile GEMM kernels are persistent: one 8-wave workgroup per CU holds the CU's whole LDS and register file and
 * walks a static share of the output tiles.  A kernel from another stream that needs whole CUs for a long time (an RCCL
 * collective overlapped with the backward pass) would leave that many GEMM workgroups waiting for a second round, i.e. double
 * the GEMM's duration.  i2t_gemm_reserve_cus(n) makes later launches leave n CUs free (0 = use every CU again); the
 * data-parallel gradient exchange brackets its overlap window with it (training/dp.py).  Process-wide, takes effect at the
 * next launch. */
int i2t_gemm_reserve_cus(int n_reserved);
/* the reservation currently in force (>= 0); never fails */
int i2t_gemm_reserved_cus(void);

int i2t_colsum_bf16(void* stream, const void* X, int ld, int M, int N, float* out, int accumulate);
/* out[n] (+)= (sum_m X[m][n]) / (sqrt(*alpha_sumsq) + 1e-6)  (alpha_sumsq NULL: plain sums): the bias gradient of a linear layer whose
 * output gradient X is kept un-normalised (see i2t_gemm_bf16_ex) */
int i2t_colsum_bf16_ex(void* stream, const void* X, int ld, int M, int N, float* out, int accumulate, const float* alpha_sumsq);

/* ---------------------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (layers.py:349-358, F.layer_norm eps 1e-5, optional bias)
 *   fwd: x f32 [M][d] -> y (bf16 or f32) [M][d]; saves mean/rstd f32 [M] when non-null
 *   bwd: dx[M][d] (f32) (+)= LN'(dy); dgamma/dbeta f32 [d] are ACCUMULATED (atomics); dy bf16 or f32;
 *        dx_bf16 (nullable): bf16 copy of the final dx rows, i.e. the A operand of the next backward GEMMs
 * --------------------------------------------------------------------------------------------------------- */
int i2t_layernorm_fwd(void* stream, const float* x, const float* gamma, const float* beta,
                      void* y, int y_is_f32, float* mean, float* rstd, int M, int d);
/* the same forward with an explicit eps (torchvision's ViT blocks: 1e-6); the backward only needs the saved rstd */
int i2t_layernorm_fwd_eps(void* stream, const float* x, const float* gamma, const float* beta,
                          void* y, int y_is_f32, float* mean, float* rstd, int M, int d, float eps);
int i2t_layernorm_bwd(void* stream, const void* dy, int dy_is_f32, const float* x, const float* gamma,
                      const float* mean, const float* rstd,
                      float* dx, int dx_accumulate, void* dx_bf16, float* dgamma, float* dbeta, int M, int d,
                      unsigned drop_key, unsigned drop_thr, float drop_scale, float* sumsq_out, const float* dx_pre_sumsq);
/* dx_mask_* (nullable by thr = 0): an elementwise dropout mask (index row * d + column, i2t_dropout_apply mode 1) applied to the f32 dx
 * this call stores -- the embedding dropout's backward (decoder.py:243, encoder.py:170) folded into the lowest block's last LayerNorm
 * backward; sumsq_out and the bf16 copy see the unmasked value */
int i2t_layernorm_bwd_ex(void* stream, const void* dy, int dy_is_f32, const float* x, const float* gamma,
                         const float* mean, const float* rstd,
                         float* dx, int dx_accumulate, void* dx_bf16, float* dgamma, float* dbeta, int M, int d,
                         unsigned drop_key, unsigned drop_thr, float drop_scale, float* sumsq_out, const float* dx_pre_sumsq,
                         unsigned dx_mask_key, unsigned dx_mask_thr, float dx_mask_scale, int acc_period, int acc_rows);
/* acc_period / acc_rows (0 = every row): with dx_accumulate, only rows r with r % acc_period < acc_rows are added onto -- the others
 * are written.  The encoder's CLS-only last block (encoder.py:172-173): dx holds the CLS rows' residual gradient and nothing else, so
 * the zero fill of the patch rows and its read-back are not needed. */
/* dx_pre_sumsq (bwd, nullable, needs dx_accumulate): the dx accumulated onto is still un-normalised -- its old value is
 * multiplied by 1 / (sqrt(*dx_pre_sumsq) + 1e-6) while adding (the gradient normaliser of i2t_grad_normalize flag 2).
 * sumsq_out (bwd, nullable): += sum of squares of the f32 dx written by this call (after accumulation) -- lets the
 * gradient normaliser that consumes dx next skip its own reduction pass (i2t_grad_normalize presummed).
 * drop_* (bwd): optional elementwise dropout (rule of i2t_gemm_bf16, idx = row*d + col, drop_thr 0 = off) applied to the
 * bf16 copy dx_bf16 only -- the copy feeds the backward of a dropped-out branch, the f32 dx is the residual gradient. */

/* ---------------------------------------------------------------------------------------------------------
 * LayerNormND (layers.py:361-370 via encoder.py:150,166,170): one normalisation per image over the joint
 *   (rows x d) slab, affine (rows, d).   y[b] = LN(x[b] + add) * gamma + beta   (add = wpe or NULL)
 *   x,add,gamma,beta f32; y f32 with its own batch stride (lets the 2nd call write behind the CLS rows);
 *   stats = workspace f32 [B][I2T_LNND_STATS_STRIDE]: (mean, rstd) then per-split partials; written by fwd,
 *   read (and its partial slots reused) by bwd.  Each LayerNormND application needs its own stats buffer.
 *   bwd: dx[b] (f32, own batch stride, overwritten) ; dgamma/dbeta (+ dadd when non-null) accumulated.
 * --------------------------------------------------------------------------------------------------------- */
#define I2T_LNND_STATS_STRIDE 34
int i2t_layernorm_nd_fwd(void* stream, const float* x, const float* add, const float* gamma, const float* beta,
                         float* y, long y_batch_stride, float* stats, int B, int rows, int d);
/* the same with the elementwise dropout of the tensor that y is a slab of (the embedding dropout of encoder.py:170 applied by the producer of
 * the patch rows): element (b, i) of y is dropped by the decision of index b * y_batch_stride + drop_base + i (i2t_dropout_apply mode 1's rule) */
int i2t_layernorm_nd_fwd_drop(void* stream, const float* x, const float* add, const float* gamma, const float* beta,
                              float* y, long y_batch_stride, float* stats, int B, int rows, int d,
                              unsigned drop_key, unsigned drop_thr, float drop_scale, long drop_base);
int i2t_layernorm_nd_bwd(void* stream, const float* dy, long dy_batch_stride, const float* x, const float* add,
                         const float* gamma, const float* stats, float* dx, float* dgamma, float* dbeta,
                         float* dadd, int B, int rows, int d);

/* ---------------------------------------------------------------------------------------------------------
 * Attention, head_dim 64 (F.scaled_dot_product_attention at layers.py:465 and inside nn.MultiheadAttention
 *   layers.py:537-542).  softmax(q k^T / 8 [+ causal]) v per (batch, head).
 *   q/k/v/o are bf16 with element strides: batch stride, row (token) stride; head h starts at column 64*h.
 *   Packed c_attn output: q=base, k=base+d, v=base+2d, row stride 3d.  lse f32 [B][H][Tq] (natural log).
 *   causal: key j visible to query i iff j <= i + (Tk - Tq)   (Tk == Tq in training; Tk > Tq with a KV cache)
 *   bwd: dq/dk/dv written (not accumulated), same strides convention as their forward operands.
 *   packed variable-length batches (training skips the caption rows past the last label: they are dead under the causal
 *   mask and carry zero loss weight): cu_q / cu_k (device int[B+1], nullable) give sequence b the rows [cu[b], cu[b+1]) of
 *   a packed [total, width] tensor (batch strides then unused; Tq/Tk = the maxima); with cu_q, lse/delta are [H][total_q].
 *   dropout on the attention probabilities (SDPA dropout_p, layers.py:465; nn.MultiheadAttention dropout): drop_thr = round(p*256)
 *   (0 = off), probability (b,h,q,key) kept iff u8(drop_key, ((b*H+h)*Tq+q)*Tk+key) >= drop_thr, scaled by drop_scale.
 * --------------------------------------------------------------------------------------------------------- */
int i2t_attention_fwd(void* stream, const void* q, long q_bs, int q_rs, const void* k, long k_bs, int k_rs,
                      const void* v, long v_bs, int v_rs, void* o, long o_bs, int o_rs, float* lse,
                      int B, int H, int Tq, int Tk, int causal, unsigned drop_key, unsigned drop_thr, float drop_scale,
                      const int* cu_q, const int* cu_k, int total_q);
int i2t_attention_bwd(void* stream, const void* q, long q_bs, int q_rs, const void* k, long k_bs, int k_rs,
                      const void* v, long v_bs, int v_rs, const void* o, long o_bs, int o_rs,
                      const void* d_o, long do_bs, int do_rs, const float* lse, float* delta_ws,
                      void* dq, long dq_bs, int dq_rs, void* dk, long dk_bs, int dk_rs,
                      void* dv, long dv_bs, int dv_rs, int B, int H, int Tq, int Tk, int causal,
                      unsigned drop_key, unsigned drop_thr, float drop_scale,
                      const int* cu_q, const int* cu_k, int total_q,
                      unsigned out_drop_key, unsigned out_drop_thr, float out_drop_scale);
/* out_drop_* (bwd): the per-token q/k/v multipliers of the fused c_attn output (i2t_gemm_bf16 drop_mode 2) applied to
 * dq / dk / dv on the way out: row r of dq is scaled by keep(key, r), of dk by keep(key + 1, r), of dv by keep(key + 2, r);
 * r = token index (b*T + t, or the packed row).  out_drop_thr 0 = off. */
/* the same with out_drop_q_seq != 0: the Tq query rows are the FIRST rows of sequences of out_drop_q_seq rows (the encoder's last
 * block computes its CLS rows only, layers.py:465 over encoder.py:172-173's slice), so row r of dq is token b*out_drop_q_seq + t in the
 * index space of the multipliers; 0 = Tq.  Offered by the dense resident-operand kernel only. */
int i2t_attention_bwd_ex(void* stream, const void* q, long q_bs, int q_rs, const void* k, long k_bs, int k_rs,
                         const void* v, long v_bs, int v_rs, const void* o, long o_bs, int o_rs,
                         const void* d_o, long do_bs, int do_rs, const float* lse, float* delta_ws,
                         void* dq, long dq_bs, int dq_rs, void* dk, long dk_bs, int dk_rs,
                         void* dv, long dv_bs, int dv_rs, int B, int H, int Tq, int Tk, int causal,
                         unsigned drop_key, unsigned drop_thr, float drop_scale,
                         const int* cu_q, const int* cu_k, int total_q,
                         unsigned out_drop_key, unsigned out_drop_thr, float out_drop_scale, int out_drop_q_seq);

/* ---------------------------------------------------------------------------------------------------------
 * Token + position embedding (decoder.py:231-243): x[b][t] = wte[ids[b][t]] + wpe[t + pos_offset]  (f32)
 *   bwd: dwte[ids] += dx (atomics), dwpe[t+pos_offset] += sum_b dx[b][t]
 *   pos (device int[B*T], nullable): explicit position of every row for packed variable-length batches (then B*T is just
 *   the row count and the wpe gradient is accumulated with atomics).
 * --------------------------------------------------------------------------------------------------------- */
int i2t_embed_fwd(void* stream, const int64_t* ids, const float* wte, const float* wpe, float* x,
                  int B, int T, int d, int pos_offset, int vocab, const int* pos);
int i2t_embed_bwd(void* stream, const int64_t* ids, const float* dx, float* dwte, float* dwpe,
                  int B, int T, int d, int pos_offset, int vocab, const int* pos);

/* ---------------------------------------------------------------------------------------------------------
 * Weighted cross-entropy over bf16 logits (wrapper.py:147-151: F.cross_entropy(logits/T, labels, ignore_index,
 *   'none') * weights, summed).  logits [M][ld] bf16, V valid columns; labels int64 [M] (ignore_index rows skip);
 *   fwd: lse[m] = logsumexp(logits[m]/T); loss += sum_m w[m]*(lse[m] - logits[m][label]/T)   (atomic into *loss)
 *   bwd: logits[m][:] <- bf16( gscale * w[m]/T * (softmax(logits[m]/T) - onehot) ), in place;  gscale read
 *        from device memory (*gscale_ptr) so the upstream gradient never needs a host sync.
 * --------------------------------------------------------------------------------------------------------- */
int i2t_ce_fwd(void* stream, const void* logits, int ld, const int64_t* labels, const float* w, float inv_temp,
               int64_t ignore_index, float* lse, float* loss, int M, int V);
int i2t_ce_bwd(void* stream, void* logits, int ld, const int64_t* labels, const float* w, float inv_temp,
               int64_t ignore_index, const float* lse, const float* gscale_ptr, int M, int V);
/* One-pass form for training steps (V <= 65536): reads every live row ONCE, leaves lse[m] and the loss term as i2t_ce_fwd does and
 * overwrites the row with the UN-scaled gradient bf16(w[m]/T (softmax(z/T) - onehot)) (dead rows: zeros).  The upstream gradient is
 * applied afterwards with i2t_scale_bf16 (x[0..n) *= *scale_ptr in place; returns without touching x when the scale is exactly 1). */
int i2t_ce_fwd_bwd(void* stream, void* logits, int ld, const int64_t* labels, const float* w, float inv_temp,
                   int64_t ignore_index, float* lse, float* loss, int M, int V);
int i2t_scale_bf16(void* stream, void* x, long n, const float* scale_ptr);
/* Momentum-distillation loss (reference training/wrapper.py:134-144): targets alpha * softmax(teacher/T) + (1 - alpha) * onehot.
 * teacher = the momentum twin's logits of the same rows (bf16 [M][ld_t], constant); lse_t [M] keeps its row lse for backward.
 *   fwd: loss += sum_m w[m] (lse[m] - (1 - alpha) z[m][label]/T - alpha/T sum_v softmax(teacher[m]/T)[v] z[m][v])
 *   bwd: z[m][:] <- bf16(gscale w[m]/T (softmax(z[m]/T) - (1 - alpha) onehot - alpha softmax(teacher[m]/T))), in place. */
int i2t_ce_distill_fwd(void* stream, const void* logits, int ld, const void* teacher, int ld_t, float alpha, const int64_t* labels,
                       const float* w, float inv_temp, int64_t ignore_index, float* lse, float* lse_t, float* loss, int M, int V);
int i2t_ce_distill_bwd(void* stream, void* logits, int ld, const void* teacher, int ld_t, float alpha, const int64_t* labels,
                       const float* w, float inv_temp, int64_t ignore_index, const float* lse, const float* lse_t,
                       const float* gscale_ptr, int M, int V);
/* EMA update of the momentum twin's flat arena (wrapper.py:52-59): pm <- pm momentum + p (1 - momentum); pm_bf16 (nullable) = its shadow */
int i2t_ema_update(void* stream, float* pm, const float* p, void* pm_bf16, long n, float momentum);
/* Decoder inputs of a training step from its labels (wrapper.py:154-196): ids[b][0] = bos, ids[b][t] = label t-1 (ignored -> eos), with
 * the optional MLM corruption of labelled tokens: with probability mask_fraction -> mask_id, or (random_fraction of those) a random
 * id in [0, vocab); draws are counter hashes of (seed, element) -- image2text_amd/rng.py::mlm_draws is the host replica. */
int i2t_lm_inputs(void* stream, const int64_t* labels, int64_t* ids, int B, int L, int64_t bos, int64_t eos, int64_t mask_id, int vocab,
                  int64_t ignore_index, float mask_fraction, float random_fraction, unsigned seed_lo, unsigned seed_hi);


/* ---------------------------------------------------------------------------------------------------------
 * Gradient normaliser (functions.py:19-24): g <- g / (||g||_2 + 1e-6) over the whole f32 tensor, in place.
 *   ws = 1 float of zero-initialised-by-the-call scratch; g_bf16 (nullable) receives a bf16 copy of the result.
 * --------------------------------------------------------------------------------------------------------- */
int i2t_grad_normalize(void* stream, float* g, long n, float* ws, void* g_bf16,
                       unsigned drop_key, unsigned drop_thr, float drop_scale,      /* dropout on the bf16 copy, as i2t_layernorm_bwd */
                       int presummed, float* clear_after);
/* ws[0] (+)= sum(g^2): the first half of i2t_grad_normalize on its own.  Two gradient tensors that the reference normalises as ONE
 * (the prompt rows and the text rows of the decoder when a loss differentiates both) sum their parts and normalise with
 * i2t_grad_normalize(..., presummed = 1). */
int i2t_sumsq(void* stream, const float* g, long n, float* ws, int accumulate);
/* presummed bit 0: *ws already holds sum(g^2) (accumulated by the producer through i2t_layernorm_bwd's sumsq_out) and the
 * reduction pass is skipped; bit 1 (2): g itself is left as it is and only the normalised bf16 copy is written -- the first
 * i2t_layernorm_bwd that accumulates onto g applies the factor (its dx_pre_sumsq = ws), 6 instead of 10 bytes per element;
 * clear_after (nullable): a float zeroed after the call, i.e. the accumulator of the next producer. */

/* ---------------------------------------------------------------------------------------------------------
 * ConvMLP feature extractor (layers.py:258-282): Conv2d(k x k, padding='same', k even => pad (k-1)/2 before,
 *   k/2 after), GELU(tanh) applied to the INPUT when in_gelu (the previous layer stores pre-activations).
 *   x: in_is_f32 ? f32 : bf16 [B][Cin][H][W];  w f32 [Cout][Cin][k][k];  y bf16 [B][Cout][H][W] (pre-activation)
 *   bwd_data:  dx[B][Cin][H][W] bf16 = convT(dy) * (in_gelu ? gelu'(x) : 1)     (skipped for the first layer)
 *   bwd_weight: dw f32 [Cout][Cin][k][k], db f32 [Cout] accumulated (atomics)
 *   w_ws: f32 scratch of Cout*Cin*k*k elements (weights repacked tap-major for scalar loads)
 * --------------------------------------------------------------------------------------------------------- */
int i2t_conv_fwd(void* stream, const void* x, int in_is_f32, int in_gelu, const float* w, const float* bias,
                 void* y, float* w_ws, int B, int Cin, int Cout, int H, int W, int k);
int i2t_conv_bwd_data(void* stream, const void* dy, const float* w, const void* x, int in_gelu, void* dx,
                      float* w_ws, int B, int Cin, int Cout, int H, int W, int k);
int i2t_conv_bwd_weight(void* stream, const void* dy, const void* x, int in_is_f32, int in_gelu,
                        float* dw, float* db, int B, int Cin, int Cout, int H, int W, int k);

/* MFMA implicit-GEMM form of the same 6x6 convolutions (the path the engine uses when every intermediate channel
 * count is 8, 16 or 32).  Layout codes: 0 = NCHW f32 (the image), 1 = NCHW bf16, 2 = NHWC bf16.  Intermediate
 * pre-activations are NHWC bf16; the last layer writes NCHW bf16 (y_nchw) because its output is the flat-patch operand
 * of the projector GEMM (encoder.py:166), whose gradient therefore arrives NCHW (dy_layout 1).
 *   w_ws:    >= 32*36*32 bf16 scratch (weights repacked [co][tap][ci], flipped + transposed for bwd_data)
 *   scratch: >= 32*36*16 f32 (partial dW in [co][tap][ci] order, folded into dw[Cout][Cin][6][6] by the call)
 *   bwd_data: dx (NHWC bf16, Cin channels) = convT(dy) * gelu'(x_pre);  bwd_weight: dw, db accumulated. */
int i2t_conv6_fwd(void* stream, const void* x, int x_layout, int in_gelu, const float* w, const float* bias,
                  void* y, int y_nchw, void* w_ws, int B, int Cin, int Cout, int H, int W);
int i2t_conv6_bwd_data(void* stream, const void* dy, int dy_layout, const float* w, const void* x_pre, void* dx,
                       void* w_ws, int B, int Cin, int Cout, int H, int W);
int i2t_conv6_bwd_weight(void* stream, const void* dy, int dy_layout, const void* x, int x_layout, int in_gelu,
                         float* dw, float* db, float* scratch, int B, int Cin, int Cout, int H, int W);
/* dst[b][y][x][c] = src[b][c][y][x] (bf16, C <= 32): the flat-patch gradient of the last conv layer -> channels-last */
int i2t_nchw_to_nhwc_bf16(void* stream, const void* src, void* dst, int B, int C, int H, int W);

/* ---------------------------------------------------------------------------------------------------------
 * Elementwise / arena utilities
 *   cast:  dst bf16[n] = src f32[n]                     (bf16 weight shadows after an optimizer step)
 *   adamw: torch.optim.AdamW semantics (decoupled decay, bias correction from `step`), one launch over a flat
 *          f32 arena; also refreshes the bf16 shadow of every parameter.  lr/wd are per-element-range tables:
 *          seg_end[i] = exclusive end of segment i, seg_lr[i], seg_wd[i] (nseg small; device arrays).  A segment with
 *          seg_lr < 0 is FROZEN (requires_grad off, or in no parameter group): skipped entirely -- no state, no update, no traffic.
 *   add_rows / cls concat helpers for the encoder token buffer.
 * --------------------------------------------------------------------------------------------------------- */
int i2t_cast_f32_bf16(void* stream, const float* src, void* dst, long n);
/* hi[i] = bf16(f(src[i])), lo[i] = bf16(f(src[i]) - hi[i]) (lo may be null); f by act: I2T_ACT_NONE, I2T_ACT_GELU, I2T_ACT_GELU_ERF.
 * The operand producer of the opt-in parity mode I2T_PRECISE=1 (image2text_amd/ops.py: every GEMM as hi.hi + lo.hi + hi.lo through the
 * accumulate class; inference only, never part of a measured step).  The reference computes these operands in fp32
 * (models/layers.py: nn.Linear on fp32 activations under precision '32'). */
int i2t_split_f32_bf16(void* stream, const float* src, void* hi, void* lo, long n, int act);
/* in-place dropout of x[rows][cols] (f32 or bf16) with the same counter-based keep rule as the fused epilogues:
 * mode 1: element (r, c) kept iff u8(key, r*cols + c) >= thr; mode 2: all of (r, third t of cols) kept iff
 * u8(key + t, r) >= thr (u8 and thr as for i2t_gemm_bf16).  Used for the embedding dropouts and to re-apply a forward mask to a gradient. */
int i2t_dropout_apply(void* stream, void* x, int is_f32, long rows, int cols, int mode, unsigned key, unsigned thr,
                      float scale);
int i2t_adamw_step(void* stream, float* p, const float* g, float* m, float* v, void* p_bf16, long n,
                   const long* seg_end, const float* seg_lr, const float* seg_wd, int nseg,
                   float beta1, float beta2, float eps, int step, float grad_scale);
/* SNRAdam (reference models/optimizer.py:56-113; trainer.py:169 `use_snr_optim`): Adam with the gradient's running VARIANCE
 * in the denominator.  Same contract as i2t_adamw_step (frozen segments: lr < 0); a segment at lr == 0 (warm-up from 0) still advances its moments, as the reference does. */
int i2t_snradam_step(void* stream, float* p, const float* g, float* m, float* v, void* p_bf16, long n,
                     const long* seg_end, const float* seg_lr, const float* seg_wd, int nseg,
                     float beta1, float beta2, float eps, int step, float grad_scale);
/* y[b][r][:] = src[r][:] for r < rows (broadcast a (rows,d) f32 block into a strided batch buffer) */
int i2t_bcast_rows(void* stream, const float* src, float* y, long y_batch_stride, int B, int rows, int d);
/* ... with the same elementwise dropout on the rows it writes (index b * y_batch_stride + i: they head their slab) */
int i2t_bcast_rows_drop(void* stream, const float* src, float* y, long y_batch_stride, int B, int rows, int d,
                        unsigned drop_key, unsigned drop_thr, float drop_scale);
/* dst[r][:] (+)= sum_b x[b][r][:] */
int i2t_sum_over_batch(void* stream, const float* x, long x_batch_stride, float* dst, int B, int rows, int d,
                       int accumulate);
/* strided f32 row copy with optional bf16 output: y[b][r][:] = x[b][r][:] for r < rows */
int i2t_copy_rows(void* stream, const float* x, long x_bs, void* y, long y_bs, int y_is_bf16, int B, int rows,
                  int d);
int i2t_add_f32(void* stream, float* dst, const float* src, long n);

/* ---------------------------------------------------------------------------------------------------------
 * Greedy decode step pieces (vision_encoder_decoder.py:136-182 with top_k=1, temperature=1):
 *   decode_attention: one new query token per caption against the self K/V cache (keys 0..*pos); head h of caption b starts at
 *     kcache + b * cache_bs + h * cache_hs and its keys are cache_rs apart: cache_hs = 64, cache_rs = d is the token-major layout
 *     [B][Tmax][d] (what a projection GEMM writes: the cross-attention K/V, fixed n_keys, pos_ptr NULL), cache_hs = Tmax * 64,
 *     cache_rs = 64 the head-major layout [B][H][Tmax][64] of the self-attention cache (a head's keys are one contiguous run:
 *     the step is HBM-bound on exactly these reads);
 *     append_dm = d (0 = off): q is a packed [q|k|v] row and the new k/v are written to the cache at *pos by the same
 *     launch (fused kv_append);
 *   ngram_ban_argmax: HF NoRepeatNGramLogitsProcessor for every size in ngram_sizes (prompt included), then argmax
 *     (first index on ties); appends the token at ids[b][*len] and (block 0) bumps *len / *pos after the grid.
 *   Positions live in device memory so that a captured hipGraph replays the same launch for every step.
 * --------------------------------------------------------------------------------------------------------- */
int i2t_decode_attention(void* stream, const void* q, int q_rs, void* kcache, void* vcache,
                         long cache_bs, int cache_rs, long cache_hs, void* o, int o_rs, const int* pos_ptr, int n_keys_fixed,
                         int append_dm, int B, int H);
int i2t_kv_append(void* stream, const void* qkv, int qkv_rs, void* kcache, void* vcache, long cache_bs,
                  int cache_rs, const int* pos_ptr, int B, int d);
int i2t_ngram_ban_argmax(void* stream, const void* logits, int ld, int logits_is_f32, int64_t* ids, int ids_ld,
                         int* len_ptr, const int* ngram_sizes, int n_sizes, int B, int V, float* margin_out);
/* The greedy step's lm_head without its logits (vision_encoder_decoder.py:143-180 with top_k = 1): i2t_gemm_bf16_top2 runs
 * logits = A [M][lda] . B^T (B = the head's rows [N][ldb], bf16, K % 128 == 0) on the persistent GEMM kernel and leaves, for every
 * 64-column segment of a row, its two largest values and their columns -- top2[M][nseg][4] = {v1, column1 (int bits), v2, column2},
 * value descending, lower column first on ties, nseg = ceil(N / 64); i2t_top2_ngram_argmax then applies the n-gram ban and takes the
 * argmax over the segments (a segment whose two leaders are both banned is re-evaluated from `hidden` and `w_head`), writing the token
 * at ids[b][*len_ptr] as i2t_ngram_ban_argmax does.  No margin output: callers that want margins use the logits form. */
int i2t_gemm_bf16_top2(void* stream, const void* A, int lda, const void* B, int ldb, int M, int N, int K, float* top2, int nseg);
int i2t_top2_ngram_argmax(void* stream, const float* top2, int nseg, const void* hidden, int ld_hidden, const void* w_head, int ld_w, int d,
                          int64_t* ids, int ids_ld, int* len_ptr, const int* ngram_sizes, int n_sizes, int B, int V);
int i2t_embed_step(void* stream, const int64_t* ids, int ids_ld, const int* len_ptr, const float* wte,
                   const float* wpe, float* x, int B, int d, int pos_offset, int vocab);
/* One sampling step of generate() for B captions (reference models/vision_encoder_decoder.py:150-180, the non-greedy modes; the
 * call shape of trainer.py:41-56 eval_model is temperature 0.7 / nucleus 0.6): logits f32 [B][ld] of the last position ->
 * / temperature -> no-repeat-n-gram ban over ids[b][0 .. len) -> top-k crop (top_k <= 0: none; ties at the k-th value stay) ->
 * softmax -> nucleus cut (nucleus_p < 0: none; keeps the sorted prefix whose running sum <= max(nucleus_p, largest probability))
 * -> renormalise -> draw -> ids[b][len] = token.  *len_ptr (device) is the current length; seed = 2 device words.  The draw is
 * the inverse CDF in vocabulary order at u = uniform(seed, len, b) (image2text_amd/rng.py::sample_uniform is the host replica):
 * reproducible per (seed, step, row).  dist_out (optional, f32 [B][dist_ld]) receives the kept, renormalised distribution. */
int i2t_sample_token(void* stream, const float* logits, int ld, int64_t* ids, int ids_ld, const int* len_ptr,
                     const int* ngram_sizes, int n_sizes, int B, int V, float temperature, int top_k, float nucleus_p,
                     const unsigned* seed, float* dist_out, int dist_ld);
int i2t_advance(void* stream, int* counters, int n, int delta);   /* counters[0..n) += delta */

/* -----------------------------------------------------------------------------------------------------------
 * The nano-mini block family (reference training_configs/gpu/nano-mini.yaml).
 *
 * Grouped-query attention, head_dim 16 / 32 / 64 / 128 (reference models/layers.py:391-430 MultiQueryAttention = H query heads
 * on ONE shared key/value head, Hkv = 1; Hkv = H is multi-head attention at widths i2t_attention_* does not cover, i.e.
 * nn.MultiheadAttention with 128-wide heads, layers.py:537-542).  Same conventions as i2t_attention_fwd / _bwd (strides, packed
 * rows, lse layout, causal rule, dropout index space, out_drop multipliers); query head h starts at column hd*h of q / o / dq,
 * key/value head h / (H / Hkv) at column hd*(h / (H / Hkv)) of k / v / dk / dv; dk / dv hold the SUM over the query heads that
 * share a key/value head. */
int i2t_gq_attention_fwd(void* stream, const void* q, long q_bs, int q_rs, const void* k, long k_bs, int k_rs,
                         const void* v, long v_bs, int v_rs, void* o, long o_bs, int o_rs, float* lse,
                         int B, int H, int Hkv, int hd, int Tq, int Tk, int causal,
                         unsigned drop_key, unsigned drop_thr, float drop_scale,
                         const int* cu_q, const int* cu_k, int total_q, int split);
/* split (forward only, dense non-causal self-attention; 0 = off): query rows >= split do not see keys < split -- the mask of a
 * NON-causal decoder over [soft prompt | text] (reference models/vision_encoder_decoder.py:93-99,106-113: prompt rows see every
 * column, text rows never see the prompt columns). */
int i2t_gq_attention_bwd(void* stream, const void* q, long q_bs, int q_rs, const void* k, long k_bs, int k_rs,
                         const void* v, long v_bs, int v_rs, const void* o, long o_bs, int o_rs,
                         const void* d_o, long do_bs, int do_rs, const float* lse, float* delta_ws,
                         void* dq, long dq_bs, int dq_rs, void* dk, long dk_bs, int dk_rs,
                         void* dv, long dv_bs, int dv_rs, int B, int H, int Hkv, int hd, int Tq, int Tk, int causal,
                         unsigned drop_key, unsigned drop_thr, float drop_scale,
                         const int* cu_q, const int* cu_k, int total_q,
                         unsigned out_drop_key, unsigned out_drop_thr, float out_drop_scale);
/* x[m][n] *= keep(key0 + n / section, m) ? scale : 0 on a bf16 [rows][ld] matrix: the per-token q / k / v multipliers of
 * layers.py:412-420 on the separate q_proj (section = d, key0 = key) and kv_proj (section = hd, key0 = key + 1) outputs. */
int i2t_row_sections_dropout(void* stream, void* x, int ld, long rows, int cols, int section, unsigned key0, unsigned thr, float scale);

/* Sparse token subsets (layers.py:570-577, 609-614): out[i][:] = src[idx[i]][:] (fp32 and / or bf16 copy), dst[idx[i]][:] = src[i][:]. */
int i2t_gather_rows(void* stream, const float* src, const int* idx, float* out_f32, void* out_bf16, long n, int d);
int i2t_scatter_rows(void* stream, const float* src, const int* idx, float* dst, long n, int d);

/* MoELinear routing (layers.py:330-346).  U f32 [M][ldu] = x [l1_0; ..; l1_{E-1}; gate layer 0]^T + bias: columns [0, E P) the
 * experts' pre-activations, then G gate-hidden pre-activations (G > 0: gate = Linear-GELU-Linear, wg2 f32 [E][G], bg2 f32 [E] or
 * null) or the E gate logits themselves (G = 0).  gates = softmax(logits * inv_sqrt_in); the top_k largest are the routing
 * weights w (not renormalised), all others 0.  A bf16 [M][Kp] = [w_e gelu(U[e P + j]) | w_0 .. w_{E-1} | 0 pad] is the left
 * operand of y = A W2aug^T (i2t_moe_pack_w2).  gates / wsel f32 [M][E] are kept for the backward pass.
 * bwd: dA bf16 [M][Kp] = dy W2aug -> D1 bf16 [M][ldd] = dL/dU (pad columns zeroed), dwg2 / dbg2 += the gate's second layer
 * gradients (fixed summation order; part_ws f32 [i2t_moe_gate_bwd_blocks(M)][E G + E]). */
int i2t_moe_gate_fwd(void* stream, const float* U, int ldu, const float* wg2, const float* bg2, void* A, int Kp, float* gates,
                     float* wsel, int M, int E, int P, int G, int top_k, float inv_sqrt_in);
int i2t_moe_gate_bwd_blocks(int M);
int i2t_moe_gate_bwd(void* stream, const void* dA, int Kp, const float* U, int ldu, const float* gates, const float* wsel,
                     const float* wg2, void* D1, int ldd, float* dwg2, float* dbg2, float* part_ws, int M, int E, int P, int G,
                     int top_k, float inv_sqrt_in);
/* W2aug bf16 [out][Kp] = [l2_0.weight | .. | l2_{E-1}.weight | l2_0.bias .. l2_{E-1}.bias | 0] from the stacked parameters
 * l2w bf16 [E][out][P], l2b f32 [E][out]; unpack adds a gradient dW f32 [out][Kp] back onto gw f32 [E][out][P], gb f32 [E][out]. */
int i2t_moe_pack_w2(void* stream, const void* l2w, const float* l2b, void* W, int out, int E, int P, int Kp);
int i2t_moe_unpack_dw2(void* stream, const float* dW, float* gw, float* gb, int out, int E, int P, int Kp);

/* Decode step of the family (static KV cache under hipGraph, see i2t_decode_attention): the new query row q [B][H hd] against the
 * cached keys 0 .. *pos_ptr of its key/value head (Hkv heads of width hd in the cache rows); k_new / v_new [B][>= Hkv hd] are this
 * token's key / value, written to cache slot *pos_ptr and attended to; null k_new / v_new: a fixed memory of n_keys_fixed keys
 * (cross-attention).  max_keys bounds *pos_ptr + 1 (<= 1024). */
int i2t_gq_decode_attention(void* stream, const void* q, int q_rs, const void* k_new, const void* v_new, int kv_rs, void* kcache,
                            void* vcache, long cache_bs, int cache_rs, void* out, int out_rs, const int* pos_ptr,
                            int n_keys_fixed, int max_keys, int B, int H, int Hkv, int hd);
/* Sparse blocks in the decode step: a layer's cache holds only its kept positions, so the token at text position *pos_ptr uses
 * slot rank[l][pos] (= kept positions before it) and runs the block only when member[l][pos]; both tables int [L][tmax] on the
 * device.  setup writes lpos[l] / lmem[l] for the current position; i2t_select_rows picks the block's or the null connector's
 * output (out = *flag ? a : b, fp32 [n]). */
int i2t_sparse_step_setup(void* stream, const int* pos_ptr, const int* rank, const int* member, int* lpos, int* lmem, int L, int tmax);
int i2t_select_rows(void* stream, const int* flag, const float* a, const float* b, float* out, long n);

/* -----------------------------------------------------------------------------------------------------------
 * Llama-2 / Qwen2 decoder blocks (reference models/decoder.py:404-440: Llama2HuggingfaceDecoder / Qwen2HuggingfaceDecoder wrap
 * transformers' LlamaForCausalLM / Qwen2ForCausalLM, a go-through dependency -- transformers 5.x models/llama/modeling_llama.py:
 * LlamaRMSNorm, apply_rotary_pos_emb with rotate_half, LlamaMLP).  Projections are i2t_gemm_bf16, attention
 * is i2t_gq_attention_* (H query heads on Hkv key/value heads).
 *
 * RMSNorm: y (bf16) / y_f32 [M][d] = w * x * rstd (either output may be null), rstd[m] = rsqrt(mean(x[m]^2) + eps) (fp32 x; rstd
 * may be null).  Backward:
 * dx (+)= rstd (dy w - xh mean(dy w xh)), xh = x rstd; dw += sum_m dy xh (null: skipped); dx_bf16 (nullable) = bf16 copy of the
 * dx written (the next GEMM's operand). */
int i2t_rmsnorm_fwd(void* stream, const float* x, const float* w, void* y, float* y_f32, float* rstd, int M, int d, float eps);
int i2t_rmsnorm_bwd(void* stream, const void* dy, int dy_is_f32, const float* x, const float* w, const float* rstd,
                    float* dx, int dx_accumulate, void* dx_bf16, float* dw, int M, int d);
/* Rotary position embedding in place on n_heads heads of width hd starting at column col0 of the bf16 rows x [M][rs]:
 * [x1 | x2] -> [x1 cos - x2 sin | x2 cos + x1 sin] (halves of the head; transformers' rotate_half convention).  cos_sin fp32
 * [n_positions][hd] = [cos(p f_i), i < hd/2 | sin(p f_i)].  Position of row m: pos[m] (packed rows) | *pos_ptr + pos_offset (decode
 * step under hipGraph) | pos_offset + m % T.  inverse != 0 rotates by the negative angle (= the backward of the forward). */
int i2t_rope(void* stream, void* x, int rs, int col0, int n_heads, int hd, const float* cos_sin, int n_positions,
             const int* pos, const int* pos_ptr, int pos_offset, int T, int M, int inverse);
/* SwiGLU on the fused projection gate_up bf16 [M][ld] = [gate (ff) | up (ff)]: h [M][ff] = silu(gate) * up; backward writes
 * d_gate_up [M][ld] from dh [M][ff]. */
int i2t_swiglu_fwd(void* stream, const void* gate_up, int ld, void* h, int M, int ff);
int i2t_swiglu_bwd(void* stream, const void* dh, const void* gate_up, int ld, void* d_gate_up, int M, int ff);

/* LoRA adapters (reference models/utils.py:46-65 -> peft: y = base(x) + lora_B(lora_A(dropout(x))) * alpha / r).  The rank-r products
 * are i2t_gemm_bf16 calls (rank padded to 64 by zero rows), the input dropout is i2t_dropout_apply; this is the GELU derivative behind
 * an adapted mlp.c_proj, whose input gradient is the fp32 sum of the base and adapter paths: out (bf16) = dh (fp32) * gelu_tanh'(pre). */
int i2t_dgelu_mul(void* stream, const float* dh, const void* pre, void* out, long n);
/* ... and behind an exact (erf) GELU (transformers' Falcon MLP, nn.GELU()): out = dh * gelu_erf'(pre). */
int i2t_dgelu_erf_mul(void* stream, const float* dh, const void* pre, void* out, long n);
/* out (bf16) = gelu(pre (bf16)), tanh form or (erf != 0) exact: the activation behind a projection whose GEMM cannot apply it while
 * also writing the pre-activation (the fp8 classes of i2t_gemm_fp8). */
int i2t_gelu_fwd(void* stream, const void* pre, void* out, long n, int erf);
/* One pass over an adapted layer's input x bf16 [M][K]: xcat[m][0..K) = x[m] (xcat bf16 [M][ldc] is the K-concatenated operand
 * [x | u] of the layer's GEMM) and, when xd is not null, xd [M][K] = dropout(x) with the elementwise mask (key, thr, scale; index
 * m * K + k, the index space of i2t_dropout_apply) -- the adapter's input (peft's lora_dropout). */
int i2t_lora_stage(void* stream, const void* x, void* xcat, int ldc, void* xd, long M, int K, unsigned drop_key, unsigned drop_thr,
                   float drop_scale);

/* Grouped small GEMMs for AdvancedPositionalBiasMLP (reference models/layers.py:617-638, decoder.py:231-232: every position owns a
 * private MLP, so one layer of the module is one GEMM per position).  Group g = position; its rows are rows [seg[g], seg[g+1]) of
 * the row-major operands (position-major order; seg = device int[n_groups + 1]; max_rows = the largest group), its weights
 * W_g = B + (g + group0) * b_group_stride (bf16 [N][K]), bias b_g = bias + (g + group0) * bias_group_stride (f32 [N]).
 *   mode 0: C[rows][N] = act(A[rows][K] . W_g^T + b_g) (+ residual)   act = I2T_ACT_GELU keeps the pre-activation in aux_out (bf16)
 *   mode 1: C[rows][K] = (A[rows][N] . W_g) * gelu'(aux_in) (+ residual)          (act = I2T_ACT_DGELU, or none)
 *   mode 2: C_g[N][K] (+)= A_g[rows][N]^T . B_g[rows][K]   with C_g = C + (g + group0) * c_group_stride (f32; B = the layer input rows)
 * group_ptr (device int, modes 0 / 1 with n_groups = 1, seg null): every one of the max_rows rows uses group *group_ptr + group0
 * (the decode step under hipGraph replay).  N, K multiples of 32.
 * i2t_grouped_colsum: out[(g + group0) * out_group_stride + n] += sum over the rows of group g of X[row][n]  (bias gradients). */
int i2t_grouped_gemm(void* stream, int mode, const void* A, int lda, const void* B, int ldb, long b_group_stride, void* C, int ldc,
                     long c_group_stride, int c_is_f32, const float* bias, long bias_group_stride, int act, const void* aux_in,
                     void* aux_out, int ld_aux, const float* residual, int ldr, int accumulate, const int* seg, int n_groups,
                     int max_rows, const int* group_ptr, int group0, int N, int K);
int i2t_grouped_colsum(void* stream, const void* X, int ld, const int* seg, int n_groups, float* out, long out_group_stride,
                       int group0, int N);

/* hipGraph capture around any sequence of the calls above (replaces the Python loop of
 * vision_encoder_decoder.py:143-180 with one replayed launch per token) */
int i2t_graph_capture_begin(void* stream);
int i2t_graph_capture_end(void* stream, void** graph_exec_out);
int i2t_graph_launch(void* graph_exec, void* stream);
int i2t_graph_destroy(void* graph_exec);

/* ---------------------------------------------------------------------------------------------------------
 * PretrainedViT (reference models/encoder.py:56-127): torchvision ViT-B/16 backbone + three heads.  The backbone's blocks run on
 * i2t_gemm_bf16 (I2T_ACT_GELU_ERF), i2t_attention_* and i2t_layernorm_fwd_eps; these are the pieces around them (csrc/vit.hip).
 *   i2t_patchify: conv_proj (p x p, stride p; encoder.py:60 -> torchvision VisionTransformer._process_input) as a GEMM: images f32
 *     [B][C][H][W] -> out bf16 [B (H/p) (W/p)][C p p], column order (c, ky, kx) = the flattened conv weight's.
 *   i2t_vit_tokens: x f32 [B][T][d] = [class_token | proj rows of the image] + pos_embedding (T = P^2 + 1).
 *   i2t_l2norm_fwd / _bwd: F.normalize(p = 2, eps 1e-12) over the last dim of M rows (encoder.py:118-119); y f32 and / or bf16,
 *     inv_norm f32 [M] saved; bwd: dx (+)= (dy - y <y, dy>) inv_norm.
 *   i2t_transpose_last2: dst[b][c][r] = src[b][r][c] (the (e, s) <-> (s, e) exchange around the peer_proj_wt GEMM, encoder.py:116).
 *   i2t_peer_lookup_fwd / _bwd: PeerLookup.forward after its linear maps (models/layers.py:78-109), one workgroup per row:
 *     scores f32 [M nhead][2 nq] = [query_left | query_right] scores, inp_proj bf16 [M][nhead din], residual f32 [M][dout],
 *     emb_in bf16 [units][din], emb_out bf16 [units][dout] -> out f32 [M][dout]; saved per (row, head, j < topk): expert unit,
 *     (left, right) query-unit indices, softmax score, pre-GELU dot.  bwd: dscores f32 (PRE-ZEROED by the caller), dinp_proj
 *     bf16, expert-table gradients accumulated with fp32 atomics (null = frozen).
 *   i2t_gemm_f32: z[M][N] = x[M][K] . P[K][N], fp32 FMA (the LSH projections: bucket decisions must not see bf16 rounding).
 *   i2t_lsh_embed_fwd / _bwd: CompositeCosineVectorEmbedding per slot (models/layers.py:112-143,190-219): z f32 [B][n_cls nK n_proj]
 *     -> bucket = #(grid points < z) -> mean over projections of table rows, summed over the nK resolutions; table k of slot s =
 *     tables + s slot_stride + tab_off[k] (f32 [(nbins[k] + 1) n_proj][dout]); grid k = grids + grid_off[k]; rows int [B][n_cls][nK][n_proj]
 *     saved; bwd scatters dy / n_proj into the tables' gradients (atomics).
 * --------------------------------------------------------------------------------------------------------- */
int i2t_patchify(void* stream, const float* images, void* out, int B, int C, int H, int W, int p);
int i2t_vit_tokens(void* stream, const float* proj, const float* cls, const float* pos, float* x, int B, int T, int d);
int i2t_l2norm_fwd(void* stream, const float* x, float* y, void* y_bf16, float* inv_norm, int M, int d);
int i2t_l2norm_bwd(void* stream, const float* dy, const float* x, const float* inv_norm, float* dx, int accumulate, int M, int d);
int i2t_transpose_last2(void* stream, const float* src, float* dst, void* dst_bf16, long B, int R, int C);
int i2t_peer_lookup_fwd(void* stream, const float* scores, const void* inp_proj, const float* residual, const void* emb_in,
                        const void* emb_out, float* out, int* sv_unit, int* sv_lr, float* sv_score, float* sv_dot, int M, int nhead,
                        int nq, int topk, int din, int dout);
int i2t_peer_lookup_bwd(void* stream, const float* dout, const void* inp_proj, const void* emb_in, const void* emb_out,
                        const int* sv_unit, const int* sv_lr, const float* sv_score, const float* sv_dot, float* dscores,
                        void* dinp_proj, float* g_emb_in, float* g_emb_out, int M, int nhead, int nq, int topk, int din, int dout_w);
int i2t_gemm_f32(void* stream, const float* x, const float* P, float* z, int M, int N, int K);
int i2t_lsh_embed_fwd(void* stream, const float* z, const float* tables, long slot_stride, const long* tab_off, const int* nbins,
                      const float* grids, const int* grid_off, float* out, int* rows, int B, int n_cls, int nK, int n_proj, int dout);
int i2t_lsh_embed_bwd(void* stream, const float* dy, const int* rows, float* g_tables, long slot_stride, const long* tab_off, int B,
                      int n_cls, int nK, int n_proj, int dout);

/* ---------------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange (replaces accelerate's DDP wrap, reference trainer.py:108-114,173-174):
 * one RCCL communicator per process (one process per GPU), all-reduce of the flat fp32 gradient arena in place over xGMI.
 * RCCL is bound at run time (the copy torch already loaded, else librccl.so.1); i2t_comm_available() = 0 means none was found.
 *   i2t_comm_unique_id: rank 0 fills a 128-byte id, the caller ships it to the other ranks (any out-of-band channel)
 *   i2t_comm_init: collective over all ranks (current HIP device of the calling thread); *comm_out = opaque handle
 *   i2t_comm_allreduce: buf[count] fp32 <- sum (mean != 0: mean) over ranks, asynchronous on `stream`; bf16_staging non-null
 *     (count bf16 elements, count % 4 == 0): the values travel as bf16 (half the bytes, one extra rounding per element)
 *   i2t_workspace_bytes: *bytes_out = scratch bytes the named entry point needs for a problem (M, N, K); 0 = none
 * --------------------------------------------------------------------------------------------------------- */
int i2t_comm_available(void);
int i2t_comm_unique_id(void* id_out, int bytes);
int i2t_comm_init(const void* id, int world, int rank, void** comm_out);
int i2t_comm_allreduce(void* comm, void* stream, float* buf, long count, int mean, void* bf16_staging);
int i2t_comm_destroy(void* comm);
int i2t_workspace_bytes(const char* entry, long M, long N, long K, long* bytes_out);

/* Deterministic mode: every reduction of the gradient path that combines workgroup partials with fp32 atomics (dW split-K
 * slices, column sums, LayerNorm / RMSNorm gain gradients, the gradient normaliser's sum, embedding scatter-adds, convolution
 * weight gradients) runs in one fixed order instead (one K slice, one workgroup, or one launch per workgroup).  Slow; two backward
 * passes of the same step are then bit-equal.  Default: the environment variable I2T_DETERMINISTIC (unset / 0 = off). */
int i2t_set_deterministic(int on);
int i2t_deterministic(void);

/* ---------------------------------------------------------------------------------------------------------
 * fp8 (OCP e4m3) operand path for frozen weights (BASELINE.json configs[4]; csrc/fp8.hip): C = (A8 . B8^T) * sa[m] * sb[n]
 * (+ bias[n]) (act: I2T_ACT_NONE | I2T_ACT_GELU | I2T_ACT_GELU_ERF, forward-only: no pre-activation output) (+ residual f32) on
 * v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales and per-row fp32 scales, fp32
 * accumulate, C bf16 or f32.  Rows of both operands are K contiguous bytes, zero-padded to the leading dimension (% 16).
 *   i2t_quant_rows_fp8: x (bf16 or f32) [M][ld] -> out e4m3 [M][ld_out], scale[m] = amax(row m) / 448 (1 for a zero row)
 *   i2t_quant_cols_fp8: W bf16 [N][ld] -> out e4m3 [K][ld_out] = W^T with a scale per k: the operand of dx = dy . W
 * --------------------------------------------------------------------------------------------------------- */
int i2t_quant_rows_fp8(void* stream, const void* x, int x_is_f32, int ld, void* out, int ld_out, float* scale, int M, int K);
int i2t_quant_cols_fp8(void* stream, const void* w, int ld, void* out, int ld_out, float* scale, int N, int K);
/* Producers that emit the e4m3 operand of the next i2t_gemm_fp8 themselves (no bf16 copy, no quantisation pass): RMSNorm forward
 * (d <= 8192; rstd for i2t_rmsnorm_bwd may be null), SwiGLU forward (h [M][ff]) and backward ([d gate | d up] [M][2 ff]); ff <= 12288.
 * Rows zero-padded to ld8 (% 16); scale[m] = amax(row m) / 448 as i2t_quant_rows_fp8, taken from the fp32 values.  The last argument: an
 * optional contiguous bf16 copy of the same row ([M][d] / [M][ff] / [M][2 ff]) for consumers that are not fp8 GEMMs (LoRA's rank products). */
int i2t_rmsnorm_fwd_fp8(void* stream, const float* x, const float* w, void* y8, int ld8, float* scale, float* rstd, int M, int d, float eps,
                        void* y_bf16);
int i2t_swiglu_fwd_fp8(void* stream, const void* gate_up, int ld, void* h8, int ld8, float* scale, int M, int ff, void* h_bf16);
int i2t_swiglu_bwd_fp8(void* stream, const void* dh, const void* gate_up, int ld, void* dgu8, int ld8, float* scale, int M, int ff,
                       void* dgu_bf16);
int i2t_gemm_fp8(void* stream, const void* A8, int lda, const float* sa, const void* B8, int ldb, const float* sb, void* C, int ldc,
                 int c_is_f32, int M, int N, int K, const float* bias, int act, const float* residual, int ldr);

#ifdef __cplusplus
}
#endif
#endif /* I2T_H */
