/*
 * capdec.h -- C ABI of libcapdec.so: the MI355X (gfx950) caption-decoder hot path.
 *
 * The reference (JeremyNathanJusuf/image-captioning) is pure Python on torch.nn and has no FFI; every
 * entry point below replaces the *implicit* ATen/cuBLAS/cuDNN kernel sequence behind the cited
 * reference lines.  Citations are file:line under the reference checkout.
 *
 * Conventions
 *  - all pointers are DEVICE pointers to contiguous fp32 / int32 / int64 data unless a parameter
 *    says "host"; the caller owns every buffer (the library never allocates or frees device memory);
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *  - return value: 0 on success, a negative CAPDEC_E* code on bad arguments (nothing was launched),
 *    or a positive hipError_t from the launch;
 *  - no entry point synchronises the host; no global mutable state: calls on different streams
 *    are independent.  (Two exceptions, both documented at their entry points: capdec_vocab_argmax_static owns a
 *    fixed device array, and the kernel-selection overrides of include/capdec_tuning.h exist only in the separate
 *    tuning build libcapdec_tune.so, never in this library.)
 */
#ifndef CAPDEC_H
#define CAPDEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAPDEC_OK 0
#define CAPDEC_EINVAL (-22)   /* bad shape / null pointer / unsupported size */
#define CAPDEC_ENOMEM (-12)   /* caller-provided workspace too small        */
#define CAPDEC_EALIGN (-14)   /* pointer not 16-byte aligned where required */
#define CAPDEC_E2BIG (-7)     /* problem larger than the fixed capacity of an allocation-free entry point */

#define CAPDEC_ABI_VERSION 2
int capdec_abi_version(void);
/* static string for a return code of this library (never NULL) */
const char* capdec_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * Dense contraction  C[i,j] = epi( sum_k P[i,k] * Q[j,k] )   (fp32 MFMA, v_mfma_f32_32x32x2_f32)
 *   replaces nn.Linear: transformer.py:38-40,43,54,137; cnnattn_model.py:27; vitattn_model.py:43;
 *   vitcnnattn_model.py:47-48; yolocnnattn_model.py:75-76; cnnrnn_model.py:23,33
 * Up to 4 independent problems ("groups") that share K run in one launch (Q/K/V projections).
 * epi(y) = relu?( (y + bias_i[i] + bias_j[j]) * scale_j[j] + shift_j[j] ) + R[i,j]
 * Element (i,j) of C / R lives at C[i*ldci + j*ldcj] / R[i*ldri + j*ldrj].
 * nn.Linear y = x W^T + b is  P = x [M,K], Q = W [N,K], bias_j = b, ldci = N, ldcj = 1.
 * ------------------------------------------------------------------------------------------ */
typedef struct capdec_gemm_group {
    const float* P;        /* [I, K], row stride ldp */
    const float* Q;        /* [J, K], row stride ldq */
    float* C;
    const float* bias_i;   /* nullable */
    const float* bias_j;   /* nullable */
    const float* scale_j;  /* nullable; BatchNorm1d eval folded as per-column scale/shift */
    const float* shift_j;  /* nullable */
    const float* R;        /* nullable residual */
    int64_t ldp, ldq, ldci, ldcj, ldri, ldrj;
    int32_t I, J;
    const int32_t* c_rows; /* nullable, device [I]: output row i is stored at row c_rows[i] of C (scatter) */
} capdec_gemm_group;

#define CAPDEC_GEMM_RELU 1
#define CAPDEC_GEMM_TP 2     /* P is stored [K, I] (element (i,k) at P[k*ldp + i]); needs TQ as well   */
#define CAPDEC_GEMM_TQ 4     /* Q is stored [K, J] (element (j,k) at Q[k*ldq + j])                      */
#define CAPDEC_GEMM_ATOMIC 8 /* C += result with float atomics; the contraction may then be split over     */
                             /* workgroups (small I x J, long K: weight gradients, wide feature heads).    */
                             /* C must hold the value to add to (zero it for a plain product); only bias   */
                             /* and scale/shift epilogues are allowed; sums differ in the last bits from    */
                             /* run to run (arrival order).                                                 */
#define CAPDEC_GEMM_SPLITK 16 /* one group, long K, few output tiles (the 86436-wide YOLO projection,          */
                             /* yolocnnattn_model.py:75, yoloattn_model.py:70): the contraction is split over     */
                             /* workgroups as with ATOMIC, but every k-range stores its partial sums to its own   */
                             /* slab of extra->splitk_ws and a second kernel adds the slabs in index order and    */
                             /* applies the epilogue: the result is the same bit for bit from run to run.         */
                             /* Needs capdec_gemm_ex with splitk_ws of capdec_gemm_splitk_workspace_elems floats. */
int capdec_gemm(const capdec_gemm_group* groups /*host*/, int ngroups, int K, int flags, void* stream);
size_t capdec_gemm_splitk_workspace_elems(int I, int J, int K);

/* Backward forms of nn.Linear y = x W^T + b (autograd of the same reference lines) use the layout flags:
 *   dX[m,k] = sum_n dY[m,n] W[n,k] :  P = dY [M,N], Q = W stored [N(k-axis), K(j-axis)]  -> TQ
 *   dW[n,k] = sum_m dY[m,n] X[m,k] :  P = dY stored [M(k-axis), N], Q = X stored [M, K]   -> TP | TQ
 * and two extra epilogues (applied after bias/scale/relu, before the residual R, which doubles as
 * "accumulate into C" when R == C):
 *   mask  : v = 0 where mask[i*ldm + j] <= 0        (ReLU backward with the saved activation)
 *   lse   : v = (exp(v - lse[i]) - (tgt[i] == j)) * gscale[0], or 0 where tgt[i] == ignore_index
 *           (gradient of CrossEntropyLoss(ignore_index) w.r.t. the logits, recomputed from the projection;
 *            gscale is a device scalar, normally 1 / number of counted targets; train.py:216,254-256) */
typedef struct capdec_gemm_extra {
    const float* mask; int64_t ldm;
    const float* lse; const int64_t* tgt; const float* gscale; int64_t ignore_index;
    const int32_t* rows_dev;   /* nullable device scalar: only rows i < min(I, rows_dev[0]) are computed */
    const int32_t* k_dev;      /* nullable device scalar: the contraction runs over k < min(K, k_dev[0]) only; whole
                                  32-deep k-tiles are read, so both operands must hold zeros (or finite values against
                                  zeros) from k_dev[0] up to the next multiple of 32 */
    float* splitk_ws;          /* CAPDEC_GEMM_SPLITK: device scratch for the partial sums */
    int64_t splitk_ws_elems;   /* its size in floats */
} capdec_gemm_extra;
int capdec_gemm_ex(const capdec_gemm_group* groups /*host*/, int ngroups, int K, int flags,
                   const capdec_gemm_extra* extra /*host, nullable*/, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-sample contractions over a ragged batch (fp32 MFMA, the k-loop of capdec_gemm).
 *   used by the cross-attention of a training step over a long memory (S rows per sample, T << S query rows): with
 *   the key / value projections folded onto the query side,
 *       scores = (q_h W_k,h) mem^T / sqrt(dk),   out_h = (softmax(scores) mem) W_v,h^T + b_v,h,
 *   which is scaled_dot_product_attention(W_q x, W_k mem, W_v mem) of transformer.py:21-27,45-60 re-associated: the
 *   memory is contracted against H*T rows per sample instead of being projected row by row (S*E*2E multiply-adds per
 *   sample and layer become about 2*H*T*S*E); b_k shifts every score of a row by the same amount and drops out of the
 *   softmax, b_v passes through because the probabilities of a row sum to one.
 * Sample b owns the rows [seg_off[b]*mult, seg_off[b+1]*mult) of the ragged operands (seg_off == NULL: the rows
 * [b*seg*mult, (b+1)*seg*mult)), n_b of them (at most seg*mult), and the slab dense + b*dense_stride of the dense one.
 *   CAPDEC_BGEMM_NT: C_b[n_b,J] = scale * P_b[n_b,K] Q_b[J,K]^T          P, C ragged, Q dense;  K % 4 == 0
 *   CAPDEC_BGEMM_NN: C_b[n_b,J] = scale * P_b[n_b,K] Q_b[K,J]            P, C ragged, Q dense;  J % 4 == 0; the rows of P
 *                                                                       hold zeros from column K to the next multiple of 4
 *   CAPDEC_BGEMM_TN: C_b[I,J] (+)= scale * (P_b[n_b,I]^T Q_b[n_b,J] + P2_b[n_b,I]^T Q2_b[n_b,J])
 *                                                                       P, Q (P2, Q2: nullable pair, same strides) ragged,
 *                                                                       C dense; J % 4 == 0; the rows of P, P2 hold zeros
 *                                                                       from column I to the next multiple of 4
 * All pointers 16-byte aligned, ldp, ldq, dense_stride multiples of 4.  accumulate (TN only): add onto C.
 * ------------------------------------------------------------------------------------------ */
typedef struct capdec_bgemm_desc {
    const float* P; const float* Q; float* C;
    const float* P2; const float* Q2;
    int64_t ldp, ldq, ldc, dense_stride;
    const int32_t* seg_off;    /* device [B+1], nullable */
    int32_t mult, seg;         /* rows per segment unit (heads); segment length when seg_off == NULL, else its maximum */
    int32_t B, I, J, K;        /* I: TN only; K: NT / NN only */
    float scale;
    int32_t accumulate;
} capdec_bgemm_desc;
#define CAPDEC_BGEMM_NT 0
#define CAPDEC_BGEMM_NN 1
#define CAPDEC_BGEMM_TN 2
int capdec_bgemm(const capdec_bgemm_desc* d /*host*/, int form, void* stream);

/* Rows of scores -> probabilities, in place (softmax over the first `cols` columns of every row; columns
 * [cols, pad_to) are set to zero), and its backward  dp <- p * (dp - sum_j p_j dp_j) * scale  (same zero padding).
 *   transformer.py:23-24 (F.softmax(scores, dim=-1)) for score rows held in memory, as capdec_bgemm produces them.
 * live (nullable device scalar array + index): only the first live[live_idx]*mult rows (at most `rows`) are touched. */
int capdec_softmax_rows(float* s, int64_t ld, int rows, int cols, int pad_to, const int32_t* live, int live_idx, int mult,
                        void* stream);
int capdec_softmax_rows_bwd(const float* p, float* dp, int64_t ld, int rows, int cols, int pad_to, float scale,
                            const int32_t* live, int live_idx, int mult, void* stream);

/* ------------------------------------------------------------------------------------------
 * Cross-attention over exactly two memory rows, training form (the ViT+CNN / YOLO+CNN heads: vitcnnattn_model.py:47-53,
 * yolocnnattn_model.py:75-81 feeding transformer.py:86).  A softmax over two keys is a logistic function of the score
 * difference: with dK = k0 - k1, dV = v0 - v1 (the projected memory rows of sample b) the block's output for a row x1 is
 *     a2 = c1[b] + sum_h p_h D[b,h],   p_h = sigmoid((x1 . G[b,h] + sb[b,h]) * scale),
 *     G[b,h] = W_q,h^T dK_h,  sb[b,h] = b_q,h . dK_h,  D[b,h] = W_o[:,h] dV_h,  c1[b] = W_o v1 + b_o,  scale = 1/sqrt(dk)
 * so the W_q / W_o contractions over all rows and the attention launch become H dot products per row; everything with a
 * [b] is B rows of work done with capdec_gemm.  G, D: [B][H][E]; c1: [B][E]; sb: [B][H]; p: [M][H].
 *   capdec_pair_diff      k2, v2 [B*2][E] -> dK, dV [B][E], sb
 *   capdec_pair_gate_fwd  one row per wavefront; sample (nullable): row -> sample, else row / T; rows_dev as elsewhere
 *   capdec_pair_gate_bwd  one workgroup per sample (rows [seg_off[b], seg_off[b+1]) or [b*T, (b+1)*T)):
 *                         d_x1 += sum_h dz_h G_h, dG, dD, dc1, dsb  (dz_h = (d_a2 . D_h) p_h (1 - p_h) scale)
 *   capdec_pair_diff_bwd  d(dK) (+ the sb term), d(dV), d(v1) -> dk2, dv2 [B*2][E]; qb [B][E] whose column sum is d b_q
 * E % (4*H) == 0 (diff), E % 4 == 0 and E <= 2048 (gate_fwd) / E <= 1024 (gate_bwd); pointers 16-byte aligned; CAPDEC_EINVAL /
 * CAPDEC_EALIGN otherwise.  An empty batch (B == 0, M == 0) is a no-op.
 * ------------------------------------------------------------------------------------------ */
int capdec_pair_diff(const float* k2, const float* v2, const float* bq, int B, int E, int H, float* dk, float* dv, float* sb,
                     void* stream);
int capdec_pair_diff_bwd(const float* ddk, const float* ddv, const float* dsb, const float* dv1, const float* bq,
                         const float* dk, int B, int E, int H, float* dk2, float* dv2, float* qb, void* stream);
int capdec_pair_gate_fwd(const float* x1, const float* G, const float* D, const float* c1, const float* sb,
                         const int32_t* sample, int T, int M, int E, int H, float scale, float* a2, float* p,
                         const int32_t* rows_dev, void* stream);
int capdec_pair_gate_bwd(const float* d_a2, const float* x1, const float* p, const float* G, const float* D,
                         const int32_t* seg_off, int B, int T, int E, int H, float scale, float* d_x1, float* dG, float* dD,
                         float* dc1, float* dsb, void* stream);

/* Vocabulary projection fused with row arg-max (logits are never written):
 *   replaces fc2 + argmax, transformer.py:137,155 and cnnrnn_model.py:84-90 (log_softmax is
 *   monotone, so the arg-max of the logits is the arg-max of the log-probabilities).
 * x [M,K] (row stride ldx), W [V,K], bias [V].  Writes per-row partial winners for each vocabulary
 * tile into part_val/part_idx: capdec_vocab_partial_elems(M, V) elements each (blocks of 64 rows,
 * [row block][tile][64], ntiles = capdec_vocab_tiles(V) tiles; opaque to the caller, read back by the
 * finalize / advance calls).  Ties resolve to the lowest index, as torch.argmax does.  Finish with
 * capdec_argmax_finalize / capdec_greedy_advance. */
int capdec_vocab_tiles(int V);
size_t capdec_vocab_partial_elems(int M, int V);
int capdec_vocab_argmax_partial(const float* x, int64_t ldx, const float* W, const float* bias, int M, int V,
                                int K, float* part_val, int32_t* part_idx, void* stream);
int capdec_argmax_finalize(const float* part_val, const int32_t* part_idx, int M, int ntiles,
                           int32_t* out_idx, float* out_val /*nullable*/, void* stream);

/* Allocation-free form of capdec_vocab_argmax_partial + capdec_argmax_finalize: out_idx[M] (and out_val[M], nullable)
 * = arg-max / max over the vocabulary of x W^T + bias, ties to the lowest index, bit-identical to the two-call form.
 * It has no counterpart in the reference.  It exists for callers that cannot allocate device memory or hand over a
 * workspace: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  The per-tile winners
 * go to a fixed device array inside the library (CAPDEC_VOCAB_STATIC_MAX_ELEMS values + ids = 8 MiB, held by every GPU
 * the library runs on until the process ends); the call performs no hipMalloc / hipFree and calls no allocating
 * routine.  Limit: capdec_vocab_partial_elems(M, V) <= CAPDEC_VOCAB_STATIC_MAX_ELEMS (e.g. M = 4096 rows at
 * V = 16384, or M = 6656 at V = 10000), otherwise CAPDEC_E2BIG and nothing is launched.  All calls on one GPU share
 * that GPU's array: calls that overlap in time on one GPU (different streams or threads) race and must be serialised
 * by the caller; back-to-back calls on one stream are safe, and nothing needs clearing between calls.  The first call
 * on a device resolves the array's address there (hipGetSymbolAddress): make it outside graph capture. */
#define CAPDEC_VOCAB_STATIC_MAX_ELEMS (1 << 20)
int capdec_vocab_argmax_static(const float* x, int64_t ldx, const float* W, const float* bias, int M, int V, int K,
                               int32_t* out_idx, float* out_val, void* stream);

/* Vocabulary projection fused with log-sum-exp + target pick (CrossEntropyLoss(ignore_index)):
 *   replaces fc2 + criterion, transformer.py:137 + train.py:216,254-256.
 * Writes per-row lse[M] and picked target logit tgt_logit[M]; rows whose target == ignore_index
 * get nll contribution 0.  loss_sum[0] += sum nll, loss_sum[1] += number of valid rows (both must be
 * zeroed by the caller).  part_* are scratch of capdec_vocab_partial_elems(M, V) floats each.  A counted target outside
 * [0, V) is a caller error (torch raises there): its loss term, and with it the loss, comes out NaN; nothing outside
 * W is read.  The same holds for capdec_softmax_nll_grad* below. */
int capdec_vocab_lse_partial(const float* x, int64_t ldx, const float* W, const float* bias, int M, int V, int K,
                             float* part_max, float* part_sum, void* stream);
int capdec_nll_finalize(const float* part_max, const float* part_sum, int M, int V, const float* x,
                        int64_t ldx, const float* W, const float* bias, int K, const int64_t* targets,
                        int64_t ignore_index, float* lse, float* nll, float* loss_sum, void* stream);

/* ------------------------------------------------------------------------------------------
 * Embedding gather + positional encoding:  x[b*T+i,:] = emb[ids[b*ldid+i],:] + pe[pos0+i,:]
 *   replaces transformer.py:126 (nn.Embedding + PositionalEncoding.forward :70); pe == NULL gives
 *   the plain gather of cnnrnn_model.py:27,79,106.   ids are int64 (ids32 == 0) or int32.
 * ------------------------------------------------------------------------------------------ */
int capdec_embed_pe(const void* ids, int ids32, int64_t ldid, const float* emb, const float* pe, int pos0, int B,
                    int T, int E, int V, float* x, void* stream);

/* y[r,:] = LayerNorm(x[r,:] + add[(r / add_div),:]) * gamma + beta   (eps inside the sqrt, biased var)
 *   replaces the residual add + nn.LayerNorm of transformer.py:85,87,89.  add may be NULL.
 *   mean[r] / rstd[r] are optional outputs for the backward pass.  In-place (y == x) is allowed. */
int capdec_add_layernorm(const float* x, const float* add, int add_div, const float* gamma, const float* beta,
                         float eps, int M, int E, float* y, float* mean, float* rstd, void* stream);

/* Fused attention  out = softmax(mask(q k^T / sqrt(dk))) v  for B*H*Tq query rows, one wavefront each:
 *   replaces scaled_dot_product_attention + split/combine_heads, transformer.py:21-35, with the mask of
 *   generate_mask (transformer.py:117-123) recomputed from the token ids instead of materialised.
 * q row (b,i) at q[(b*Tq+i)*ldq + h*dk]; key (b,j) at k[b*sk + j*ldk + h*dk]; same for v; out row
 * stride ldo.  causal != 0: query i sees keys j <= i + causal_off.  ids (nullable, int32, [B, >=Tq] with row
 * stride ldid): a query row whose own token id is 0 has every score masked to -1e9, i.e. its
 * softmax is uniform over ALL Tk keys (the reference masks the PAD *query row*).
 * lse (nullable) [B,H,Tq] receives max + log(sum) per row for the backward pass. */
int capdec_attention(const float* q, int64_t ldq, const float* k, int64_t ldk, int64_t sk, const float* v,
                     int64_t ldv, int64_t sv, float* out, int64_t ldo, int B, int H, int Tq, int Tk, int dk,
                     int causal, int causal_off, const int32_t* ids, int64_t ldid, float* lse, void* stream);
/* Causal self-attention over a ragged batch (see capdec_compact_prefix): sample b is the rows [seq_off[b], seq_off[b+1])
 * of q, k, v, out (row strides ld*) and of ids (nullable; <PAD> query rows as in capdec_attention).  T = the longest
 * possible sample (<= 64); lse [B,H,T] nullable.  The rows from seq_off[B] up to the next multiple of 32 (capped at
 * B*T) are zeroed, later rows are not written. */
int capdec_attention_ragged(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            float* out, int64_t ldo, int B, int H, int T, int dk, const int32_t* ids,
                            const int32_t* seq_off, float* lse, void* stream);
/* Cross-attention of a ragged batch over dense memories: q, out by slot as above; k, v hold S rows per sample
 * ([B*S, .], row strides ldk / ldv); no mask (transformer.py:84: cross_attn(x, enc, enc, None)).  T <= 64; S <= 64, or
 * longer memories (up to 2048 rows) when T >= 4. */
int capdec_cross_attention_ragged(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                  float* out, int64_t ldo, int B, int H, int T, int S, int dk, const int32_t* seq_off,
                                  float* lse, void* stream);

/* BatchNorm1d over [B,C] (cnnattn_model.py:13,29; vitattn_model.py:29,45; vitcnnattn_model.py:33,51;
 * yolocnnattn_model.py:61,79; cnnrnn_model.py:14,24).
 * eval: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale, to be applied as the
 * scale_j/shift_j epilogue of the feature-head contraction.
 * train: y = (x - mean_b) / sqrt(var_b + eps) * gamma + beta with biased batch variance; running buffers are
 * updated in place with momentum and the unbiased variance (pass NULL for both to skip); save_mean/save_rstd
 * ([C], nullable) feed the backward pass.  B >= 2. */
int capdec_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                   float eps, int C, float* scale, float* shift, void* stream);
int capdec_bn_train_forward(const float* x, const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, int B, int C, float* y, float* save_mean,
                            float* save_rstd, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-side kernels: autograd of the same reference lines (train.py:252-262 drives them through
 * loss.backward(); dropout is nn.Dropout at transformer.py:85,87,89,126 and in the wrappers).
 * Dropout masks come from a counter-based generator keyed by (seed, element index): the backward pass
 * regenerates the mask from the same seed, nothing is stored.  Column-reduced gradients (dgamma, dbeta,
 * bias and embedding gradients) are ACCUMULATED into their outputs with float atomics: zero them first.
 * ------------------------------------------------------------------------------------------ */
/* seed_dev (nullable, here and below): a device-resident 64-bit value that the kernel ADDS to `seed` when it starts.  A
 * training step captured in a HIP graph keeps its by-value arguments for ever; with the per-step part of the seed in
 * device memory (capdec_train_state.seed, advanced by a kernel inside the graph) every replay draws new masks, and
 * `seed` only tells the dropout sites of one step apart. */
int capdec_dropout(const float* x, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, float* y, void* stream);
/* s = add + dropout_p(x) ; y = LayerNorm(s) ; s_out/mean/rstd (nullable) are saved for the backward pass.
 * Row r of the sum reads row r / x_div of x (x_div = T broadcasts one row per sample over its T positions, the
 * single-memory-row cross-attention output); the dropout mask is still drawn per element of the [M,E] sum. */
int capdec_add_layernorm_train(const float* x, int x_div, const float* add, const float* gamma, const float* beta,
                               float eps, int M, int E, float p, uint64_t seed, float* y, float* s_out, float* mean,
                               float* rstd, void* stream);
/* The same with the broadcast row taken from x_map (nullable: x row of output row r is x_map[r] instead of r / x_div)
 * and only the first rows_dev[0] rows processed (rows_dev: nullable device scalar). */
int capdec_add_layernorm_train_rows(const float* x, int x_div, const int32_t* x_map, const float* add, const float* gamma,
                                    const float* beta, float eps, int M, int E, float p, uint64_t seed,
                                    const uint64_t* seed_dev, float* y, float* s_out, float* mean, float* rstd,
                                    const int32_t* rows_dev, void* stream);
/* d_add = dLN/ds ; d_x = d_add * dropout mask (same p, seed as the forward) ; dgamma/dbeta accumulated;
 * dxsum (nullable, [E], accumulated): column sums of d_x = the bias gradient of the linear layer that produced x */
int capdec_layernorm_bwd(const float* dy, const float* s, const float* mean, const float* rstd, const float* gamma,
                         int M, int E, float p, uint64_t seed, float* d_add, float* d_x, float* dgamma, float* dbeta,
                         float* dxsum, void* stream);
/* The same over the first min(M, rows_dev[0]) rows (rows_dev: nullable device scalar). */
int capdec_layernorm_bwd_rows(const float* dy, const float* s, const float* mean, const float* rstd, const float* gamma,
                              int M, int E, float p, uint64_t seed, const uint64_t* seed_dev, float* d_add, float* d_x,
                              float* dgamma, float* dbeta, float* dxsum, const int32_t* rows_dev, void* stream);
/* The same with the three column sums taken in two stages (autograd of nn.LayerNorm, transformer.py:85,87,89): every
 * workgroup stores its sums as one row of `workspace` ([workgroups][3*E] floats, capdec_layernorm_bwd_workspace_floats(M, E)
 * of them; nullable: then one float atomic per column and workgroup as above) and a second launch adds the rows to
 * dgamma / dbeta / dxsum in a fixed order.  CAPDEC_ENOMEM if the workspace is too small. */
size_t capdec_layernorm_bwd_workspace_floats(int M, int E);
int capdec_layernorm_bwd_staged(const float* dy, const float* s, const float* mean, const float* rstd, const float* gamma,
                                int M, int E, float p, uint64_t seed, const uint64_t* seed_dev, float* d_add, float* d_x,
                                float* dgamma, float* dbeta, float* dxsum, const int32_t* rows_dev, float* workspace,
                                size_t workspace_floats, void* stream);
/* out[j] += sum_i x[i*ldx + j] */
int capdec_colsum(const float* x, int64_t ldx, int M, int N, float* out, void* stream);
/* The same over the first min(M, rows_dev[0]) rows (rows_dev: nullable device scalar). */
int capdec_colsum_rows(const float* x, int64_t ldx, int M, int N, float* out, const int32_t* rows_dev, void* stream);
/* out[g,:] = sum over t < T of x[g*T + t, :]  (gradient of a row that was broadcast over T positions) */
int capdec_group_sum(const float* x, int G, int T, int E, float* out, void* stream);
/* y[r,:] = (accumulate ? y[r,:] : 0) + scale * x[r / x_div, :] for r < M */
int capdec_axpy_rows(const float* x, int x_div, float scale, int M, int E, float* y, int accumulate, void* stream);
/* out[g,:] = sum of the rows [seg_off[g], seg_off[g+1]) of x (seg_off: device, G+1 entries) */
int capdec_group_sum_segments(const float* x, int G, const int32_t* seg_off, int E, float* out, void* stream);
/* demb[ids[r],:] += dx[r,:]  (nn.Embedding backward, transformer.py:126 / cnnrnn_model.py:27) */
int capdec_embed_bwd(const int64_t* ids, const float* dx, int rows, int E, int V, float* demb, void* stream);
/* The same over the first min(rows, rows_dev[0]) rows (rows_dev: nullable device scalar). */
int capdec_embed_bwd_rows(const int64_t* ids, const float* dx, int rows, int E, int V, float* demb,
                          const int32_t* rows_dev, void* stream);
/* Embedding + positional encoding + dropout of a training step in one launch (transformer.py:126 with self.dropout in
 * training mode): x[s,:] = dropout(emb[id,:] + pe[i,:]) for the M slots s, where slot s is dense row r = order[s]
 * (order nullable: r = s) = position i = r % T of sample r / T, and its id is ids[(r / T)*ldid + i] (ldid >= T: the
 * ids may be a view of a wider matrix, e.g. captions[:, :-1]).  The mask is the one capdec_dropout draws for element
 * s*E + c of x (same seed, seed_dev).  ids_out (nullable) [M] receives the slots' ids.  pe nullable. */
int capdec_embed_train_fwd(const int64_t* ids, int64_t ldid, const int64_t* order, const float* emb, const float* pe, int M,
                           int T, int E, int V, float p, uint64_t seed, const uint64_t* seed_dev, float* x, int64_t* ids_out,
                           void* stream);
/* Its backward: capdec_embed_bwd_rows on dropout(dx) with that same mask, without the launch in between. */
int capdec_embed_train_bwd(const int64_t* ids, const float* dx, int rows, int E, int V, float p, uint64_t seed,
                           const uint64_t* seed_dev, float* demb, const int32_t* rows_dev, void* stream);
int capdec_bn_train_backward(const float* dy, const float* x, const float* gamma, const float* save_mean,
                             const float* save_rstd, int B, int C, float* dx, float* dgamma, float* dbeta,
                             void* stream);
/* Backward of capdec_attention (same layout and mask arguments): lse from the forward; dsum is scratch
 * [B,H,Tq]; dq/dk/dv are written (not accumulated).  A PAD query row has a constant uniform softmax: it
 * contributes to dv only.  out (nullable, row stride ldout) is the forward output: with it an unmasked
 * attention over a long memory (Tq <= 64) runs as one pass that streams the keys through LDS. */
int capdec_attention_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, int64_t sk, const float* v,
                         int64_t ldv, int64_t sv, const float* dout, int64_t ldo, const float* out, int64_t ldout,
                         const float* lse, float* dsum,
                         float* dq, int64_t lddq, float* dk, int64_t lddk, int64_t sdk, float* dv, int64_t lddv,
                         int64_t sdv, int B, int H, int Tq, int Tk, int dkh, int causal, int causal_off,
                         const int32_t* ids, int64_t ldid, void* stream);
/* Backward of capdec_attention_ragged (same layout; the dq, dk, dv rows from seq_off[B] up to the next multiple of 32
 * are zeroed, later rows are not written). */
int capdec_attention_bwd_ragged(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                const float* dout, int64_t ldo, const float* lse, float* dq, int64_t lddq, float* dk,
                                int64_t lddk, float* dv, int64_t lddv, int B, int H, int T, int dkh, const int32_t* ids,
                                const int32_t* seq_off, void* stream);
/* Backward of capdec_cross_attention_ragged: dq by slot (tail rows zeroed as above), dk / dv dense [B*S, .].
 * out (the forward output, by slot) is required for S > 64 and ignored otherwise. */
int capdec_cross_attention_bwd_ragged(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                                      int64_t ldv, const float* dout, int64_t ldo, const float* out, int64_t ldout,
                                      const float* lse, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv,
                                      int64_t lddv, int B, int H, int T, int S, int dkh, const int32_t* seq_off,
                                      void* stream);
/* Materialised form of the same loss (train.py:216,254-256) for training, where the backward pass needs the
 * softmax anyway: logits [M,V] (row stride ld) are overwritten in place with d(mean NLL)/d(logits) =
 * (softmax - onehot(target)) / n_counted (zero rows for ignored targets).  sums must be zeroed by the caller:
 * sums[0] += sum of the rows' NLL, sums[1] = n_counted; gscale[0] = 1 / n_counted; lse_out [M] nullable. */
int capdec_softmax_nll_grad(float* logits, int64_t ld, const int64_t* targets, int M, int V, int64_t ignore_index,
                            float* sums, float* gscale, float* lse_out, void* stream);
/* The same when only the first rows_dev[0] rows hold logits (rows_dev: nullable device scalar): those rows are
 * processed, the rows up to the next multiple of 32 are zeroed, the rest is left untouched. */
int capdec_softmax_nll_grad_rows(float* logits, int64_t ld, const int64_t* targets, int M, int V, int64_t ignore_index,
                                 float* sums, float* gscale, float* lse_out, const int32_t* rows_dev, void* stream);
/* Rows with a counted target first (the loss of train.py:216,254-256 ignores <PAD> targets, so their logits, and
 * the three vocabulary-sized contractions over them, need not be computed; the reference computes and discards them).
 * Stable.  order [M]: row of slot s (slots >= count name row 0); slot [M]: slot of row r, or M for an ignored row;
 * targets_c [M]: targets[order[s]] (ignore_index from slot count on); count [1]: number of counted targets. */
int capdec_compact_targets(const int64_t* targets, int M, int64_t ignore_index, int64_t* order, int64_t* slot,
                           int64_t* targets_c, int32_t* count, void* stream);
/* Live prefixes of a padded caption batch, as a ragged batch.  In the teacher-forced decoder (transformer.py:125-132,
 * causal self-attention) the rows of sample b from its last counted target on feed nothing but ignored loss terms
 * (train.py:216,254-256), so the whole stack may leave them out.  len[b] = 1 + last position whose target is not
 * `pad`, or T if an input id inside that prefix equals `pad` (such a query row attends uniformly over all T keys in the
 * reference, transformer.py:18-22).  ids, targets [B,T].  order [M=B*T]: dense row of slot s; slot [M]: slot of dense
 * row r, or M for a dead row; seq_off [B+1]: first slot of each sample; sample [M]: sample of slot s; ids_c [M],
 * targets_c [M]: ids / targets by slot; count [2]: (live rows rounded up to a multiple of 32, live rows).  The slots
 * between count[1] and count[0] are copies of row 0 with an ignored target: they are computed like live rows, so that
 * contractions over the rows may read whole k-tiles, and contribute exactly zero to every gradient. */
int capdec_compact_prefix(const int64_t* ids, const int64_t* targets, int B, int T, int64_t pad, int64_t* order,
                          int64_t* slot, int32_t* seq_off, int32_t* sample, int32_t* ids_c, int64_t* targets_c,
                          int32_t* count, void* stream);
/* The same with element (b, p) of ids and targets at b*ld + p (ld >= T): the inputs captions[:, :-1] and the targets
 * captions[:, 1:] of train.py:253-256 as two views (ld = T + 1) of the caption matrix, without a copy of either. */
int capdec_compact_prefix_strided(const int64_t* ids, const int64_t* targets, int64_t ld, int B, int T, int64_t pad,
                                  int64_t* order, int64_t* slot, int32_t* seq_off, int32_t* sample, int32_t* ids_c,
                                  int64_t* targets_c, int32_t* count, void* stream);
/* sums = (sum of nll, number of counted targets) from capdec_nll_finalize:
 * loss[0] = sums[0] / sums[1] (nn.CrossEntropyLoss mean over non-ignored targets), gscale[0] = 1 / sums[1] */
int capdec_loss_finish(const float* sums, float* loss, float* gscale, void* stream);
/* torch.optim.Adam defaults (train.py:217): p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps) over one flat
 * buffer; g is multiplied by grad_scale first (1/world for a summed all-reduce). */
int capdec_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, int step, float grad_scale, void* stream);

/* Per-step scalars of the training loop (train.py:217-218,244-262,366) held in DEVICE memory, so that a whole step --
 * forward, loss, backward, Adam -- can be captured once in a HIP graph and replayed with no per-step kernel arguments:
 * the dropout seed of the step, Adam's step count with its two bias corrections, and the learning rate (written by the
 * host when StepLR changes it).  capdec_train_state_advance starts the next step: step += 1, seed = a new 62-bit value
 * (splitmix64 of the old one), bias1 = 1 - beta1^step, bias2_sqrt = sqrt(1 - beta2^step) (computed in double).
 * capdec_adam_step_dev is capdec_adam_step with lr and the corrections read from the state. */
typedef struct capdec_train_state {
    uint64_t seed;
    int32_t step;
    float lr;
    float bias1;
    float bias2_sqrt;
    int32_t reserved[2];
} capdec_train_state;   /* 32 bytes */
int capdec_train_state_advance(capdec_train_state* st /*device*/, float beta1, float beta2, void* stream);
int capdec_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                         const capdec_train_state* st /*device*/, float grad_scale, void* stream);

/* One LSTM step, gate order i,f,g,o (nn.LSTM, cnnrnn_model.py:17,29-30,83):
 *   gates = x W_ih^T + b_ih + h W_hh^T + b_hh ; c' = s(f) c + s(i) tanh(g) ; h' = s(o) tanh(c')
 * x [B,E] (row stride ldx), h0 [B,Hd] (row stride ldh0), c [B,Hd]; h0/c0 NULL means zero state.  The gate GEMM and the cell
 * update run in one kernel; the gates are not written unless gates_out != NULL ([B,4Hd], for backward). */
int capdec_lstm_step(const float* x, int64_t ldx, const float* h0, int64_t ldh0, const float* c0, const float* w_ih,
                     const float* w_hh, const float* b_ih, const float* b_hh, int B, int E, int Hd, float* h1,
                     int64_t ldh1, float* c1, float* gates_out, void* stream);

/* Backward of one capdec_lstm_step (autograd of nn.LSTM inside DecoderRNN.forward, cnnrnn_model.py:29-30):
 * gates [B,4Hd] = the pre-activations saved by the forward step, c0 (NULL = zero) / c1 the cell state before /
 * after it.  The gradient reaching h' is dh_out (row stride lddh; NULL = none, e.g. the image step whose output
 * is discarded) + dh_rec (from the following step; NULL = none); dc_in is the gradient reaching c' from the
 * following step (NULL = none).  Writes dgates [B,4Hd] (gradient of the pre-activations) and dc0 [B,Hd]
 * (element-wise: dc0 may alias dc_in). */
int capdec_lstm_step_bwd(const float* gates, const float* c0, const float* c1, const float* dh_out, int64_t lddh,
                         const float* dh_rec, const float* dc_in, int B, int Hd, float* dgates, float* dc0,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole-loop greedy decoders.  One call enqueues every kernel of every step; the host never
 * synchronises inside.  They replace DecoderTransformer.greedy_inference (transformer.py:141-171)
 * and the loop of CNNtoRNN.caption_images (cnnrnn_model.py:75-106), up to detokenisation.
 * ------------------------------------------------------------------------------------------ */
typedef struct capdec_attn_w {
    const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;   /* [E,E] / [E] */
} capdec_attn_w;

typedef struct capdec_tf_layer {
    capdec_attn_w self_attn, cross_attn;
    const float *ff_w1, *ff_b1, *ff_w2, *ff_b2;            /* d_ff == E */
    const float *n1_g, *n1_b, *n2_g, *n2_b, *n3_g, *n3_b;
} capdec_tf_layer;

typedef struct capdec_tf_model {
    int32_t E, H, L, V, max_pos;
    const float* emb;      /* decoder_embedding.weight [V,E] */
    const float* pe;       /* positional_encoding.pe [max_pos,E] */
    const float* fc2_w;    /* [V,E] */
    const float* fc2_b;    /* [V] */
    const capdec_tf_layer* layers;   /* host array of L entries */
} capdec_tf_model;

/* bytes of caller-provided device workspace for the calls below */
size_t capdec_tf_decode_workspace_bytes(const capdec_tf_model* m, int B, int S, int max_len, int extra_cap);

/* KV-cached greedy decode.  enc [B,S,E].  out_ids [B,max_len] int64: every predicted id before the
 * row's first <eos>, then -1.  out_len [B].
 * Generated <PAD> tokens: the reference masks a PAD *query row*, which then attends uniformly over the CURRENT
 * length, so every position from a sample's first PAD on changes at each later step.  Those positions are re-run
 * inside the step (through the same kernels, k/v rewritten in the cache) as extra rows after the B new ones; at
 * most extra_cap of them per step (0 = default max(64, B/4)).  out_dirty [B] is set to 1 for a sample whose
 * re-run rows did not fit: only such samples need capdec_tf_greedy_decode_exact afterwards.
 * All max_len steps are always enqueued without a host sync; the reference's all-rows-done early exit
 * (transformer.py:166-168) happens on the device: the kernel that picks the tokens counts the captions that
 * finish at each step, and once all B have finished the later steps run on a device-side row count of zero (every
 * kernel of a step returns at once; about a sixth of a full step's time).  The outputs are the same either way.
 * With S == 1 the softmax over the single memory row is exactly 1,
 * so the cross-attention block reduces to one vector per sample and layer that is computed once. */
/* Optional timing probe: HIP events recorded on `stream` right before / after the vocabulary-projection
 * kernel of decode step t, for every t < n whose two array entries are both non-null (a record costs about 3 us of
 * stream time, so a caller may time every k-th step only).  Events are created with capdec_event_create; read them with
 * capdec_event_elapsed_ms after the stream has been synchronised.  Measurement plumbing only: it has no
 * counterpart in the reference. */
typedef struct capdec_probe {
    void** begin;   /* host array of n hipEvent_t */
    void** end;     /* host array of n hipEvent_t */
    int32_t n;
} capdec_probe;
int capdec_event_create(void** ev);
int capdec_event_destroy(void* ev);
int capdec_event_record(void* ev, void* stream);
int capdec_event_elapsed_ms(void* begin, void* end, float* ms);

#define CAPDEC_DECODE_FOLD_MEMORY 2      /* S >= 3: attend to the raw memory with the key / value projections folded into
                                            the query and output projections (the form taken by itself from
                                            CAPDEC_DECODE_FOLD_MIN_ROWS memory rows on; E in {64,128,256,512}) */
#define CAPDEC_DECODE_FOLD_MIN_ROWS 48
#define CAPDEC_DECODE_NO_S1_SHORTCUT 1   /* run the general cross-attention kernels even when S == 1 or S == 2 (the
                                            closed forms: one key -> the block is a per-sample constant; two keys ->
                                            a per-sample constant plus one logistic gate per head) */
#define CAPDEC_DECODE_NO_FUSED_LAYERS 4   /* S <= 2, E == 512, H == 4, L <= 2, max_len <= 64: a step's decoder layers run as
                                            ONE launch (teams of 8 workgroups walk blocks of 16 rows through the whole
                                            chain, csrc/decode_fused.hip); this flag keeps the launch-per-operation form.
                                            A workgroup of that launch whose bounded wait for its team runs out sets
                                            out_dirty to 2 for every sample: redo them with capdec_tf_greedy_decode_exact */
int capdec_tf_greedy_decode(const capdec_tf_model* m, const float* enc, int B, int S, int max_len, int sos, int eos,
                            int flags, int extra_cap, void* workspace, size_t workspace_bytes, int64_t* out_ids,
                            int32_t* out_len, int32_t* out_dirty, const capdec_probe* probe /*nullable*/, void* stream);

/* Exact full-prefix greedy decode (re-runs the whole prefix every step, as the reference does).
 * Same outputs; used for rows flagged dirty above and as an in-library cross-check. */
size_t capdec_tf_exact_workspace_bytes(const capdec_tf_model* m, int B, int S, int max_len);
int capdec_tf_greedy_decode_exact(const capdec_tf_model* m, const float* enc, int B, int S, int max_len, int sos,
                                  int eos, void* workspace, size_t workspace_bytes, int64_t* out_ids,
                                  int32_t* out_len, void* stream);

/* Teacher-forced decoder_forward (transformer.py:125-138): logits [B,T,V] (or [B,V] with take_last).
 * ids int64 [B,T]. */
size_t capdec_tf_forward_workspace_bytes(const capdec_tf_model* m, int B, int S, int T);
int capdec_tf_forward_logits(const capdec_tf_model* m, const float* enc, const int64_t* ids, int B, int S, int T,
                             int take_last, void* workspace, size_t workspace_bytes, float* logits, void* stream);

typedef struct capdec_rnn_model {
    int32_t E, Hd, V;
    const float* emb;                          /* embed.weight [V,E] */
    const float *w_ih, *w_hh, *b_ih, *b_hh;    /* lstm.*_l0 */
    const float* fc2_w;                        /* [V,Hd] */
    const float* fc2_b;
} capdec_rnn_model;

size_t capdec_rnn_decode_workspace_bytes(const capdec_rnn_model* m, int B, int max_len);
/* x0 [B,E] is the BatchNorm'ed image projection (time-step 0 of the LSTM, zero initial state).
 * Greedy decode (cnnrnn_model.py:57-110): all max_len steps are enqueued without a host sync; the reference's exit
 * once every caption holds an <EOS> (:101-103) is taken on the device, as in capdec_tf_greedy_decode. */
int capdec_rnn_greedy_decode(const capdec_rnn_model* m, const float* x0, int B, int max_len, int sos, int eos,
                             void* workspace, size_t workspace_bytes, int64_t* out_ids, int32_t* out_len,
                             void* stream);

/* ------------------------------------------------------------------------------------------
 * Beam search.  Replaces DecoderTransformer.beam_search_inference (transformer.py:177-244) and the loop of
 * CNNtoRNN.caption_images_beam_search (cnnrnn_model.py:131-197).  B samples, W hypotheses each (W <= 8); all B*W
 * hypotheses are extended every step: per hypothesis the W most probable next tokens (log-softmax); the first step
 * expands hypothesis 0 only; the best W of the W*W candidates survive, in the order torch.topk returns them on the CPU
 * (equal scores included, see capdec_topk_rows).  <EOS> follows the reference's code, not its comments: its
 * `mask[:, :, 0] = False` (transformer.py:199-201, cnnrnn_model.py:161-163) writes through a stride-0 view and clears
 * the `done` tensor at every step, so a hypothesis is frozen only for the one step after its <EOS> (its W proposals
 * all carry its unchanged score and all stay eligible), accumulates log-probabilities again afterwards, and the loop
 * ends early only when all B*W hypotheses produce <EOS> in the same step.  The Transformer form recomputes the full
 * prefix each step, as the reference does, so generated <PAD> tokens behave exactly as there.
 * One call runs steps [t_begin, t_end) (t_begin == 0 initialises the state held in the workspace) and then writes
 * the current best hypothesis of every sample (without <SOS>, cut at its first <EOS>) to out_ids [B,max_len] (padded
 * with -1) / out_len [B], the W scores to out_scores [B,W] (nullable), and to n_unfinished[0] (device) the number of
 * samples that received a token other than <EOS> at step t_end - 1.  0 there is the reference's
 * `if done.all(): break`: the state is frozen on the device from that step on, so steps already enqueued beyond it do
 * not change the result and the caller may stop issuing more.
 * ------------------------------------------------------------------------------------------ */
size_t capdec_tf_beam_workspace_bytes(const capdec_tf_model* m, int B, int W, int S, int max_len);
int capdec_tf_beam_search(const capdec_tf_model* m, const float* enc /*[B,S,E]*/, int B, int W, int S, int max_len,
                          int sos, int eos, int t_begin, int t_end, void* workspace, size_t workspace_bytes,
                          int64_t* out_ids, int32_t* out_len, float* out_scores, int32_t* n_unfinished, void* stream);
/* KV-cached form of capdec_tf_beam_search: one new position per hypothesis and step, the key/value caches re-gathered
 * after every selection.  The reference lets a <PAD> query row attend uniformly over the CURRENT length
 * (transformer.py:117-123), which a plain cache does not reproduce.  pad_seen [B] (device) is set for the samples the
 * caller must redo with capdec_tf_beam_search: with the one-launch decoder layers (the shapes listed at
 * CAPDEC_DECODE_NO_FUSED_LAYERS, over B*W rows) the <PAD> positions of a hypothesis follow along and only a block of 16
 * hypotheses with more than 16 of them, or a wait that ran out, reports its samples; otherwise every sample in which
 * some hypothesis received a <PAD> token. */
size_t capdec_tf_beam_cached_workspace_bytes(const capdec_tf_model* m, int B, int W, int S, int max_len);
int capdec_tf_beam_search_cached(const capdec_tf_model* m, const float* enc, int B, int W, int S, int max_len, int sos,
                                 int eos, int t_begin, int t_end, void* workspace, size_t workspace_bytes,
                                 int64_t* out_ids, int32_t* out_len, float* out_scores, int32_t* n_unfinished,
                                 int32_t* pad_seen, void* stream);
size_t capdec_rnn_beam_workspace_bytes(const capdec_rnn_model* m, int B, int W, int max_len);
/* x0 [B,E]: the BatchNorm'ed image projection (time-step 0 of the LSTM) */
int capdec_rnn_beam_search(const capdec_rnn_model* m, const float* x0, int B, int W, int max_len, int sos, int eos,
                           int t_begin, int t_end, void* workspace, size_t workspace_bytes, int64_t* out_ids,
                           int32_t* out_len, float* out_scores, int32_t* n_unfinished, void* stream);

/* torch.topk(x, k, dim=-1) (largest, sorted) over `rows` short rows x [rows, n], n <= 64, k <= 8, k * 64 > n, with the
 * order among EQUAL values that ATen's CPU kernel produces (std::nth_element + std::sort over (value, index) pairs,
 * GNU libstdc++): the selection step of the reference's beam search (transformer.py:211, cnnrnn_model.py:172), where
 * exact ties are routine (a hypothesis that has just produced <EOS> proposes W candidates with one score).
 * out_val, out_idx [rows, k].  The beam-search entry points above use the same device routine internally. */
int capdec_topk_rows(const float* x, int rows, int n, int k, float* out_val, int32_t* out_idx, void* stream);

/* ------------------------------------------------------------------------------------------
 * Soft-attention LSTM decoder (additive attention over the memory rows feeding an LSTM cell).
 * No counterpart in the reference (its "Attn" models are Transformer decoders); this is the Show-Attend-Tell
 * style decoder named by the project's goals, so its results are checked against a PyTorch composition of the same
 * formulas only.
 *   e[b,s] = sum_a v[a] * tanh(u[b,a] + pf[b,s,a]) ; alpha = softmax_s(e) ; ctx[b,:] = sum_s alpha[b,s] * enc[b,s,:]
 * u [B,A] (row stride ldu) = att_h(h) of the current state, pf [B,S,A] = att_f(enc) computed once per caption,
 * enc [B,S,E]; ctx rows have stride ldctx (so they can land in the right half of the LSTM's [emb | ctx] input rows);
 * alpha [B,S] is optional.
 * ------------------------------------------------------------------------------------------ */
int capdec_additive_attention(const float* u, int64_t ldu, const float* pf, const float* v, const float* enc, int B, int S,
                              int A, int E, float* ctx, int64_t ldctx, float* alpha, void* stream);
/* Backward of capdec_additive_attention for one step.  alpha [B,S] as the forward wrote it, dctx [B,E] (row stride
 * lddctx).  du [B,A] (row stride lddu) is written; dpf [B,S,A], denc [B,S,E] and dvb [B,A] are ACCUMULATED (they collect
 * the steps of a caption; zero them before the last step's call); the gradient of v is the column sum of dvb. */
int capdec_additive_attention_bwd(const float* u, int64_t ldu, const float* pf, const float* v, const float* enc,
                                  const float* alpha, const float* dctx, int64_t lddctx, int B, int S, int A, int E,
                                  float* du, int64_t lddu, float* dpf, float* denc, float* dvb, void* stream);
typedef struct capdec_satt_model {
    int32_t E, Hd, A, V;
    const float* emb;                          /* [V,E] */
    const float *w_ih, *w_hh, *b_ih, *b_hh;    /* LSTM over the 2E-wide input [emb | ctx]: w_ih [4Hd,2E] */
    const float *att_h_w, *att_h_b, *att_v;    /* [A,Hd], [A], [A] */
    const float *fc2_w, *fc2_b;                /* [V,Hd], [V] */
} capdec_satt_model;
size_t capdec_satt_decode_workspace_bytes(const capdec_satt_model* m, int B, int max_len);
/* Greedy decode with the bookkeeping of the reference's greedy loops (transformer.py:155-171): h0/c0 [B,Hd] initial
 * state, per step: u = att_h(h), context, LSTM step on [emb | ctx], vocabulary arg-max, next embedding. */
int capdec_satt_greedy_decode(const capdec_satt_model* m, const float* enc, const float* pf, const float* h0,
                              const float* c0, int B, int S, int max_len, int sos, int eos, void* workspace,
                              size_t workspace_bytes, int64_t* out_ids, int32_t* out_len, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CAPDEC_H */
