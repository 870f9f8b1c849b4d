/*
 * uniir_hip.h -- C ABI of libuniir_hip.so, the MI355X (gfx950) hot path of the UniIR
 * contrastive-train / embed / brute-force-retrieve pipeline.
 *
 * Boundary contract: plain pointers and sizes only, no torch types.  Every
 * entry point is asynchronous on the caller's hipStream_t (passed as void*), never allocates device
 * memory, never synchronises the host (the two top-k helpers that return host-visible counts say so),
 * returns 0 (UNIIR_OK) or a negative UNIIR_E* code and never throws.  All pointers are DEVICE pointers
 * unless the parameter name ends in _host.  bf16 / fp16 tensors are passed as const void*.
 *
 * Each group names the reference code it replaces (paths relative to the UniIR tree):
 *   [ENC]   openai/CLIP VisionTransformer / Transformer forward+backward as called from
 *           src/models/uniir_clip/clip_scorefusion/clip_sf.py:43-47 (encode_text / encode_image)
 *   [FUSE]  clip_sf.py:53-63 (mask * emb fuse), :88-97 (row select + F.normalize)
 *   [NCE]   clip_sf.py:133-144 (similarity * exp(logit_scale), CrossEntropy, argmax accuracy)
 *   [OPT]   clip_scorefusion/train.py:52-61,195-199 (two-group AdamW) + uniir_clip/engine.py:41-46
 *   [TOPK]  src/common/mbeir_retriever.py:76,85-103 (normalize_L2 + IDMap,Flat IP index),
 *           :188-232 (search_index: normalize queries, exact top-k by inner product)
 */
#ifndef UNIIR_HIP_H
#define UNIIR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNIIR_OK 0
#define UNIIR_EINVAL (-1)       /* null pointer / negative size / bad enum */
#define UNIIR_ESHAPE (-2)       /* shape not supported by the kernels (see each function) */
#define UNIIR_EALIGN (-3)       /* pointer or leading dimension not 16-byte aligned */
#define UNIIR_ELAUNCH (-4)      /* HIP launch failed */
#define UNIIR_EUNSUPPORTED (-5)

const char* uniir_strerror(int code);
/* ABI version; bumped on any signature change. */
int uniir_abi_version(void);
/* Reproducible reductions (round 6).  Bias, LayerNorm-weight and token-embedding gradients are sums over every row of a batch taken by
 * many workgroups; added with fp32 atomics they depend on the arrival order, and two runs of one training step differ in their last
 * bits.  Register a caller-owned device buffer for a stream (256-byte aligned; 64 MiB covers ViT-L/14 at 1024 items; buf NULL removes
 * the entry) and every such reduction launched on that stream -- uniir_gemm's colsum / a_rowsum, uniir_layernorm_bwd*, uniir_colsum_bf16,
 * uniir_text_embed_bwd* and the tower entry points that use them -- stores per-workgroup partials there and adds them in a fixed
 * order (one extra small launch each): same inputs, same bits.  Kernels on one stream run one after the other, so one buffer per
 * stream is enough.  A reduction that needs more than the buffer holds keeps its atomics.  Host-side table (16 streams), not
 * thread-safe: one training thread per process, as in the reference (train.py). */
int uniir_reduce_scratch(void* buf, int64_t bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [ENC] building block 1: 16-bit MFMA GEMM with fused epilogues.
 *   C[M,N] = alpha * op(A)[M,K] . op(B)[K,N]  (+ epilogue)
 * a_tmaj = 0: A stored [M][lda] (K contiguous);  1: A stored [K][lda] (M contiguous)
 * b_tmaj = 0: B stored [N][ldb] (K contiguous; torch Linear weight layout);  1: B stored [K][ldb]
 * Requirements: K-contiguous operands need K % 8 == 0, M/N-contiguous ones need that extent % 8 == 0,
 * N % 8 == 0, all base pointers and leading dimensions multiples of 8 elements (16 B).
 * ---------------------------------------------------------------------------------------------- */
enum {
    UNIIR_EPI_BF16 = 0,          /* C(bf16) = v + bias?                                          */
    UNIIR_EPI_BIAS_ACT = 1,      /* f = bf16(v + bias) -> C ; act(f) -> C2 (both bf16)           */
    UNIIR_EPI_RESID_F32 = 2,     /* C(f32) = v + bias? + resid(f32, ldc) ; C2(bf16 copy) optional */
    UNIIR_EPI_DACT = 3,          /* C(bf16) = v * act'(aux[m][n]) (aux bf16, ldaux); C2(bf16, ldaux) = act(aux) opt. */
    UNIIR_EPI_F32 = 4,           /* C(f32) = v  (beta = 0)                                        */
    UNIIR_EPI_ATOMIC_F32 = 5,    /* C(f32) += v via atomics; enables split-K (wgrad accumulate)  */
    UNIIR_EPI_ACT_ONLY = 6       /* C(bf16) = act(bf16(v + bias)): BIAS_ACT without the pre-activation output (forward-only
                                    passes: embedding extraction); same rounding as BIAS_ACT's second output */
};
enum { UNIIR_ACT_QUICKGELU = 0, UNIIR_ACT_GELU_ERF = 1, UNIIR_ACT_RELU = 2 };
enum { UNIIR_DT_BF16 = 0, UNIIR_DT_F16 = 1 };

typedef struct {
    const void* A;
    const void* B;
    void* C;
    void* C2;            /* optional second output */
    const float* bias;   /* [N] or NULL */
    const float* resid;  /* [M][ldc] f32 or NULL */
    const void* aux;     /* [M][ldaux] bf16 or NULL */
    int32_t M, N, K;
    int64_t lda, ldb, ldc, ldaux;
    int32_t a_tmaj, b_tmaj;
    int32_t epilogue, act, dtype;
    int32_t k_splits;    /* >= 1; > 1 only with UNIIR_EPI_ATOMIC_F32 */
    float alpha;
    void* splitk_ws;          /* optional scratch for split-K: when it holds k_splits*M*N floats the splits write */
    int64_t splitk_ws_bytes;  /* plain fp32 slabs that one reduce kernel adds into C (no atomics)               */
    float* colsum;            /* optional [N] fp32: += column sums of the result before rounding (bias gradient);
                                 honoured by the 256x256 kernel with EPI_RESID_F32 / EPI_DACT / EPI_F32 only     */
    const float* row_scale;   /* optional [M] fp32, EPI_RESID_F32 only: C = (v + bias) * row_scale[m] + resid -- DropPath /
                                 stochastic depth of a residual branch (BLIP ViT, backbone/vit.py:79-80) without a separate pass */
    float* a_rowsum;          /* optional [M] fp32, a_tmaj only: += sum over K of op(A)'s rows -- the bias gradient of a
                                 weight-gradient GEMM dW = dy^T x, taken from the dy fragments inside the 256x256 transposed
                                 kernel (bf16, b_tmaj) and by a separate pass over A otherwise                              */
} uniir_gemm_desc;

int uniir_gemm(const uniir_gemm_desc* d, void* stream);
/* measurement hook (bench.py): stride > 0 brackets every stride-th uniir_gemm call -- from any caller, the tower entry
 * points included -- with HIP events on its launch stream; 0 switches it off.  uniir_gemm_timing_read (after a stream
 * synchronise) returns the sums over the sampled launches: 2 M N K, elapsed ms, count.  Single measuring thread.
 * uniir_gemm_timing_on: the same, counting and sampling only the calls launched on `stream`.  An event pair measures a kernel's own
 * duration only while no other stream shares the device: with the towers on two streams the measurement follows one of them, the
 * uniir_gemm calls on other streams are bracketed as "device shared" windows, and the read functions leave out the samples that
 * intersect a window (uniir_gemm_timing_read_ex also returns their count in *shared; shared may be NULL).  Windows closer than two
 * mean sample durations (at most 3 ms) count as one; when fewer than 8 samples (or than all, if fewer were taken) would be left the
 * sums cover ALL samples and *fallback (may be NULL) is 1 -- a read after at least one sampled launch never returns zero launches.
 * uniir_gemm_timing_filter is that rule alone on caller-supplied times, no device involved: windows = nwin x (begin, end) ms,
 * samples = n x (begin, duration) ms on the same axis, merge_ms < 0 = the automatic gap; writes keep[n] (1 = counts) and returns the
 * number kept (negative UNIIR_E* on bad arguments). */
int uniir_gemm_timing(int32_t stride);
int uniir_gemm_timing_on(int32_t stride, void* stream);
int uniir_gemm_timing_read(double* flop, double* ms, int32_t* launches);
int uniir_gemm_timing_read_ex(double* flop, double* ms, int32_t* launches, int32_t* shared, int32_t* fallback);
int uniir_gemm_timing_filter(const float* windows, int32_t nwin, const float* samples, int32_t n, float merge_ms, uint8_t* keep,
                             int32_t* fallback);

/* ------------------------------------------------------------------------------------------------
 * [ENC] building block 2: LayerNorm over the last dim (fp32 statistics, CLIP eps 1e-5).
 * x is the fp32 residual stream [rows][x_stride]; y is bf16 [rows][width] (GEMM operand) and/or f32.
 * bwd: dx_f32 = (dres ? dres : 0) + LN'(dy); also writes a bf16 copy when dx_bf16 != NULL;
 * dgamma/dbeta are ACCUMULATED (+=) in fp32 (zero them once per optimizer step); dx_colsum (optional, [width]) +=
 * the column sums of dx_f32 as written (the bias gradient of the linear layer that produced x).
 * dy may be bf16 (dy_is_f32 = 0) or fp32.
 * ---------------------------------------------------------------------------------------------- */
int uniir_layernorm_fwd(const float* x, int64_t x_stride, const float* gamma, const float* beta,
                        void* y_bf16, float* y_f32, int32_t rows, int32_t width, float eps, void* stream);
int uniir_layernorm_bwd(const float* x, int64_t x_stride, const float* gamma, const void* dy,
                        int32_t dy_is_f32, const float* dres, float* dx_f32, int64_t dx_stride,
                        void* dx_bf16, float* dgamma, float* dbeta, float* dx_colsum, int32_t rows,
                        int32_t width, float eps, void* stream);
/* the same with a per-row factor (fp32 [rows]) for what leaves towards the next residual branch: dx_bf16 = bf16(branch_scale[r] *
 * dx) and dx_colsum += branch_scale[r] * dx; dx_f32 stays the unscaled residual-stream gradient (DropPath in backward) */
int uniir_layernorm_bwd_ex(const float* x, int64_t x_stride, const float* gamma, const void* dy,
                           int32_t dy_is_f32, const float* dres, float* dx_f32, int64_t dx_stride,
                           void* dx_bf16, float* dgamma, float* dbeta, float* dx_colsum, const float* branch_scale,
                           int32_t rows, int32_t width, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [ENC] building block 3: fused multi-head attention, head_dim 64, seq <= 512.
 * qkv bf16 [batch*seq][3*heads*64] as produced by nn.MultiheadAttention.in_proj ([q | k | v] per row);
 * out bf16 [batch*seq][heads*64]; lse f32 [batch][heads][seq] (natural log-sum-exp of scaled scores).
 * causal = 1 applies CLIP's build_attention_mask (key <= query).
 * ---------------------------------------------------------------------------------------------- */
int uniir_attention_fwd(const void* qkv, void* out, float* lse, int32_t batch, int32_t seq,
                        int32_t heads, int32_t causal, void* stream);
int uniir_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                        void* dqkv, int32_t batch, int32_t seq, int32_t heads, int32_t causal,
                        void* stream);
/* Packed rows: item m owns rows row_off[m] .. row_off[m + 1] - 1 (its own length <= max_seq) of qkv / out / dqkv; lse stays
 * [batch][heads][max_seq].  Every live row's result is bitwise that of the dense call on zero-padded-behind items under causal = 1. */
int uniir_attention_fwd_packed(const void* qkv, void* out, float* lse, const int32_t* row_off, int32_t batch,
                               int32_t max_seq, int32_t heads, int32_t causal, void* stream);
int uniir_attention_bwd_packed(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                               const int32_t* row_off, int32_t batch, int32_t max_seq, int32_t heads, int32_t causal, void* stream);
/* General form: separate Q [batch*tq][q_ld] and K/V [batch*tk][kv_ld] tensors (head h at column h*64) for the BLIP
 * MED cross-attention (uniir_blip/backbone/med.py:160-232 with encoder_hidden_states), and an optional per-item key
 * length (keys >= key_len[m] masked: the BERT padding mask, med.py:687-688 "(1 - mask) * -10000").
 * lse is [batch][heads][tq].  bwd writes dq [batch*tq][dq_ld], dk / dv [batch*tk][dkv_ld]. */
int uniir_attention_fwd_ex(const void* q, int64_t q_ld, const void* k, const void* v, int64_t kv_ld, void* out,
                           int64_t out_ld, float* lse, const int32_t* key_len, int32_t batch, int32_t tq,
                           int32_t tk, int32_t heads, int32_t causal, float drop_p, uint32_t drop_seed, void* stream);
int uniir_attention_bwd_ex(const void* q, int64_t q_ld, const void* k, const void* v, int64_t kv_ld,
                           const void* out, const void* dout, int64_t out_ld, const float* lse,
                           const int32_t* key_len, void* dq, int64_t dq_ld, void* dk, void* dv, int64_t dkv_ld,
                           int32_t batch, int32_t tq, int32_t tk, int32_t heads, int32_t causal, float drop_p,
                           uint32_t drop_seed, void* stream);
/* drop_p > 0 (BERT attention_probs_dropout_prob in train mode): P V uses P * mask / (1 - drop_p), mask regenerated from
 * (drop_seed, ((item * heads + head) * tq + query) * tk + key); backward must get the same pair. */
/* The general form on PACKED QUERY ROWS (BLIP's MED BERT on the tokens up to each caption's valid length only; replaces the padded
 * rows of backbone/med.py:160-232, whose keys are masked by (1 - m) * -10000, :687-688): item m owns the rows q_row_off[m] ..
 * q_row_off[m + 1] - 1 of q / out / dout / dq (length <= tq).  kv_packed != 0: self-attention -- k / v / dk / dv are rows of the same
 * packed numbering, the item's key count is its own length (what key_len is in the padded call; key_len must be NULL, tq == tk);
 * kv_packed == 0: cross-attention -- k / v / dk / dv stay dense [batch][tk] rows, key_len optional.  lse keeps the dense
 * [batch][heads][tq] layout; the dropout mask is drawn at the dense coordinates above, so every live row equals the padded call's
 * bit for bit, train mode included. */
int uniir_attention_fwd_rows(const void* q, int64_t q_ld, const void* k, const void* v, int64_t kv_ld, void* out, int64_t out_ld,
                             float* lse, const int32_t* q_row_off, int32_t kv_packed, const int32_t* key_len, int32_t batch,
                             int32_t tq, int32_t tk, int32_t heads, float drop_p, uint32_t drop_seed, void* stream);
int uniir_attention_bwd_rows(const void* q, int64_t q_ld, const void* k, const void* v, int64_t kv_ld, const void* out,
                             const void* dout, int64_t out_ld, const float* lse, const int32_t* q_row_off, int32_t kv_packed,
                             const int32_t* key_len, void* dq, int64_t dq_ld, void* dk, void* dv, int64_t dkv_ld, int32_t batch,
                             int32_t tq, int32_t tk, int32_t heads, float drop_p, uint32_t drop_seed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [ENC] small fused pieces of the towers.
 * ---------------------------------------------------------------------------------------------- */
/* images f32 NCHW [n][3][res][res] -> patches bf16 [n*grid*grid][kpad], k = c*P*P + py*P + px (the
 * Conv2d weight's own flattening), zero-padded to kpad (multiple of 64). */
int uniir_patchify(const float* images, void* patches, int32_t n, int32_t res, int32_t patch,
                   int32_t kpad, void* stream);
/* x[n][1+g][w] (f32) = concat(class_emb, patch_out[n][g][w] (bf16)) + pos_emb[1+g][w] */
int uniir_vit_assemble(const void* patch_out, const float* class_emb, const float* pos_emb, float* x,
                       int32_t n, int32_t tokens, int32_t width, void* stream);
/* backward of the above: dpatch bf16 [n*g][w] = dx[:,1:,:]; dclass += sum_n dx[:,0,:]; dpos += sum_n dx */
int uniir_vit_assemble_bwd(const float* dx, void* dpatch_out, float* dclass, float* dpos, int32_t n,
                           int32_t tokens, int32_t width, void* stream);
/* x[n][ctx][w] (f32) = token_emb[text[n][t]] + pos_emb[t]; also eot[n] = argmax_t text[n][t] (first max) */
int uniir_text_embed(const int32_t* text, const float* token_emb, const float* pos_emb, float* x,
                     int32_t* eot, int32_t n, int32_t ctx, int32_t width, int32_t vocab, void* stream);
int uniir_text_embed_bwd(const int32_t* text, const float* dx, float* dtoken_emb, float* dpos,
                         int32_t n, int32_t ctx, int32_t width, int32_t vocab, void* stream);
/* the same on packed rows: x row row_off[n] + t for t < row_off[n + 1] - row_off[n]; last_row[n] (optional) = the item's last row */
int uniir_text_embed_packed(const int32_t* text, const float* token_emb, const float* pos_emb, const int32_t* row_off,
                            float* x, int32_t* last_row, int32_t n, int32_t ctx, int32_t width, int32_t vocab, void* stream);
int uniir_text_embed_bwd_packed(const int32_t* text, const float* dx, const int32_t* row_off, float* dtoken_emb,
                                float* dpos, int32_t n, int32_t ctx, int32_t width, int32_t vocab, void* stream);
/* out[i][:] = x[(i*seq + idx[i])][:] (idx NULL -> 0, the class token): rows for ln_post / ln_final */
int uniir_gather_rows(const float* x, const int32_t* idx, float* out, int32_t n, int32_t seq,
                      int32_t width, void* stream);
/* dx[(i*seq + idx[i])][:] += dout[i][:] ; all other rows of dx must have been zeroed by the caller */
int uniir_scatter_rows(const float* dout, const int32_t* idx, float* dx, int32_t n, int32_t seq,
                       int32_t width, void* stream);
/* act / colsum / casts */
int uniir_act_fwd(const void* f_bf16, void* g_bf16, int64_t count, int32_t act, void* stream);
/* out[n] += sum_m x[m][n]  (x bf16 [rows][ld]) */
int uniir_colsum_bf16(const void* x, int64_t ld, float* out, int32_t rows, int32_t cols, void* stream);
int uniir_cast_f32_to_bf16(const float* src, void* dst, int64_t count, void* stream);
int uniir_cast_f32_to_f16(const float* src, void* dst, int64_t count, void* stream);     /* the fp16 forward's weight copies */
int uniir_cast_bf16_to_f32(const void* src, float* dst, int64_t count, void* stream);
/* dst bf16 [rows][ld_dst] (zero padded) = src f32 [rows][cols] */
int uniir_cast_pad_rows(const float* src, void* dst, int32_t rows, int32_t cols, int32_t ld_dst,
                        void* stream);
int uniir_cast_pad_rows_f16(const float* src, void* dst, int32_t rows, int32_t cols, int32_t ld_dst,
                            void* stream);      /* the same into fp16 (the fp16 forward's patch-embedding weight) */
/* dst f32 [rows][cols] += src f32 [rows][ld_src][:cols]  (un-pad a wgrad result) */
int uniir_unpad_add(const float* src, float* dst, int32_t rows, int32_t cols, int32_t ld_src,
                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * [FUSE] emb = txt_emb * txt_mask[:,None] + img_emb * img_mask[:,None]  (clip_sf.py:61-63), then
 * q = normalize(emb[idx_q]), p = normalize(emb[idx_p])  (clip_sf.py:88-97; F.normalize eps 1e-12).
 * Masks are int64 like the collator's (mbeir_dataset.py:427-434).  All fp32.
 * ---------------------------------------------------------------------------------------------- */
int uniir_fuse_embeddings(const float* txt_emb, const float* img_emb, const int64_t* txt_mask,
                          const int64_t* img_mask, float* emb, int32_t n, int32_t dim, void* stream);
int uniir_select_normalize(const float* emb, const int32_t* idx, float* out, float* inv_norm,
                           int32_t rows, int32_t dim, void* stream);
/* demb[idx[i]] += (dout[i] - out[i] * <out[i], dout[i]>) * inv_norm[i]  (demb zeroed by the caller) */
int uniir_select_normalize_bwd(const float* out, const float* inv_norm, const float* dout,
                               const int32_t* idx, float* demb, int32_t rows, int32_t dim, void* stream);
/* dtxt = demb * txt_mask, dimg = demb * img_mask */
int uniir_fuse_embeddings_bwd(const float* demb, const int64_t* txt_mask, const int64_t* img_mask,
                              float* dtxt, float* dimg, int32_t n, int32_t dim, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [NCE] in-batch InfoNCE, fp32 end to end (f32 MFMA 16x16x4: exact fp32 fma chains).
 *   score[i][j] = <q[i], all_p[j]> * scale            (scale = exp(logit_scale), device scalar)
 *   loss = mean_i( logsumexp_j score[i][:] - score[i][t_i] ),  t_i = target_offset + i
 *   acc  = mean_i( argmax_j score[i][:] == t_i )      (first max on ties, torch.max semantics)
 * score may be NULL only in uniir_infonce_bwd (it is required by fwd as the stash for bwd).
 * bwd: with g = dloss (device scalar), G = (softmax(score) - onehot) * g / b,
 *   dq = scale * G . all_p,  d_all_p = scale * G^T . q,  dscale = sum(G * score) / scale.
 * ---------------------------------------------------------------------------------------------- */
int uniir_infonce_fwd(const float* q, const float* all_p, const float* scale, int32_t b, int32_t B,
                      int32_t dim, int32_t target_offset, float* score, float* row_lse, float* loss,
                      float* acc, void* stream);
int uniir_infonce_bwd(const float* q, const float* all_p, const float* scale, const float* score,
                      const float* row_lse, const float* dloss, int32_t b, int32_t B, int32_t dim,
                      int32_t target_offset, float* gbuf, float* dq, float* d_all_p, float* dscale,
                      void* stream);
/* Hard-negative branch of the same loss (clip_sf.py:105-131).  q, p [b][dim], n [b][N][dim] are L2-normalised fp32 rows;
 * the logit row of query i is [<q_i,p_i>, <q_i,n_i,0..N-1>, I more copies of <q_i,p_i>] * scale, I = min(b - 1,
 * in_batch_neg_num): the reference's expand/mask expression for the "in-batch negatives" yields the query's own
 * positive I times, which is reproduced as is (golden G3).  fwd: logits [b][1+N+I], row_lse, row_loss (= -log_softmax(row)[0]) and
 * row_hit (first arg-max == 0) per query; loss = mean(row_loss), accuracy = mean(row_hit).
 * bwd (dloss: device scalar): dq, dn written; dp and dscale ACCUMULATED (zero them first). */
int uniir_hardneg_fwd(const float* q, const float* p, const float* n, const float* scale, int32_t b, int32_t N,
                      int32_t I, int32_t dim, float* logits, float* row_lse, float* row_loss, float* row_hit,
                      void* stream);
int uniir_hardneg_bwd(const float* q, const float* p, const float* n, const float* scale, const float* logits,
                      const float* row_lse, const float* dloss, int32_t b, int32_t N, int32_t I, int32_t dim,
                      float* dq, float* dp, float* dn, float* dscale, void* stream);
/* generic strided fp32 MFMA GEMM used by the two above (exported for tests):
 * C[m][n] (ldc) = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]  */
int uniir_sgemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, float alpha, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [TOWER] a whole CLIP tower per call: what the reference's FFI for this path would bind for
 * `clip_model.encode_image(image)` / `clip_model.encode_text(text)` (clip_sf.py:44,47; openai/CLIP
 * VisionTransformer.forward / CLIP.encode_text) and their autograd backward.  The per-layer launch sequence lives in the
 * library (csrc/tower.hip); the op-level entry points above stay available.
 *   weights / gradients : raw device pointers in the POD structs below.  "16" fields are the bf16 shadow of the fp32
 *                         master weights (kept fresh by uniir_adamw_step or uniir_cast_f32_to_bf16); g_* are fp32
 *                         gradients, ACCUMULATED (+=), may be NULL for a forward-only description.
 *   activations         : ONE caller-owned workspace of uniir_clip_tower_workspace_bytes() bytes (256-B aligned); with
 *                         save_for_backward it is the activation stash and must be passed unchanged to the bwd calls.
 *   forward             : input = images f32 NCHW [batch][3][res][res] (vision) or token ids int32 [batch][tokens] (text)
 *                         -> emb_out f32 [batch][embed_dim].
 *   backward            : demb f32 [batch][embed_dim].  uniir_clip_tower_bwd = head + all blocks + stem; the three stages
 *                         are exported so that a caller can hand finished layers to its gradient all-reduce between
 *                         uniir_clip_tower_bwd_blocks calls (descending ranges [lo, hi) covering all layers).
 * Asynchronous on `stream`, no allocation, no host synchronisation; 0 or a negative UNIIR_E* code.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;                 /* [W] */
    const void *wqkv16, *wo16, *wfc16, *wproj16;                /* bf16 [3W][W], [W][W], [4W][W], [W][4W] */
    const float *bqkv, *bo, *bfc, *bproj;                       /* [3W], [W], [4W], [W] */
    float *g_ln1_w, *g_ln1_b, *g_ln2_w, *g_ln2_b;
    float *g_wqkv, *g_bqkv, *g_wo, *g_bo, *g_wfc, *g_bfc, *g_wproj, *g_bproj;
} uniir_clip_block;

typedef struct {
    int32_t is_text;                          /* 0: vision tower, 1: text tower (causal attention, EOT pooling) */
    int32_t layers, width, heads, tokens;     /* width = 64 * heads; tokens = 1 + (resolution/patch)^2 or context_length */
    int32_t embed_dim;
    int32_t resolution, patch, kpad;          /* vision: kpad = 3*patch*patch rounded up to 64 */
    int32_t vocab;                            /* text */
    const uniir_clip_block* blocks;           /* HOST array [layers] */
    const void* conv16;                       /* vision: bf16 [W][kpad] (conv1.weight flattened, zero padded) */
    const float *class_emb, *pos_emb;         /* vision [W], [tokens][W]; text: pos_emb [tokens][W] */
    const float *ln_pre_w, *ln_pre_b;         /* vision */
    const float* token_emb;                   /* text [vocab][W] */
    const float *ln_post_w, *ln_post_b;       /* visual.ln_post / ln_final */
    const void* proj16;                       /* bf16 [W][embed_dim]: visual.proj / text_projection */
    float *g_conv;                            /* fp32 [W][3*patch*patch] (conv1.weight.grad) */
    float *g_class, *g_pos, *g_ln_pre_w, *g_ln_pre_b, *g_token, *g_ln_post_w, *g_ln_post_b, *g_proj;
    void* splitk_ws;                          /* optional scratch for the weight-gradient GEMMs' split-K slabs */
    int64_t splitk_ws_bytes;
    int32_t stash_act;                        /* with save_for_backward: 0 = the MLP's act(f) lives in one scratch buffer and the
                                                 backward re-materialises it from the stashed pre-activation f inside the c_proj dgrad
                                                 epilogue; 1 = act(f) is stashed per layer next to f (round 4: +2 bytes x rows x 4 width
                                                 per layer of workspace; that epilogue loses its second output: 3.31 -> 2.99 ms at ViT-L/14 x
                                                 1024 items).  Same forward bit for bit; the c_proj weight gradient reads the forward's
                                                 act(f) instead of the backward's recomputation (equal up to rare one-ulp differences).
                                                 Part of the workspace layout: every call on one workspace must see the same value */
    int32_t dtype16;                          /* 0 = bf16 (training and the default forward); 1 = fp16, FORWARD ONLY: wqkv16 .. proj16 /
                                                 conv16 then point at fp16 copies of the weights (uniir_cast_f32_to_f16) and every
                                                 16-bit activation is fp16 -- the reference embedder's autocast(fp16),
                                                 mbeir_embedder.py:52-56; save_for_backward with it is UNIIR_EUNSUPPORTED */
    int32_t pool_last_block;                  /* 1 = the LAST residual block runs its Q projection, attention, out_proj, ln_2 and MLP on
                                                 the one pooled row of every item only (class token / EOT row; ln_1 and the K | V
                                                 projection still see every row): the reference computes the other rows' outputs of
                                                 that block and discards them (VisionTransformer.forward: ln_post(x[:, 0, :]);
                                                 CLIP.encode_text: x[arange, text.argmax(-1)]).  Same embedding bit for bit; the same
                                                 gradient sums without their exact-zero terms (the one-query attention backward is the
                                                 general kernel: at 193..288 tokens, where the full block runs the pair-tile kernel,
                                                 upstream gradients agree to bf16 rounding, 4e-3); 10 of the block's 12 WxW GEMM units
                                                 leave the step.  Part of the
                                                 workspace layout like stash_act.  0 = every row through every sublayer */
} uniir_clip_tower;

int64_t uniir_clip_tower_workspace_bytes(const uniir_clip_tower* t, int32_t batch, int32_t save_for_backward);
int uniir_clip_tower_fwd(const uniir_clip_tower* t, const void* input, int32_t batch, float* emb_out, void* workspace,
                         int64_t workspace_bytes, int32_t save_for_backward, void* stream);
int uniir_clip_tower_bwd(const uniir_clip_tower* t, const void* input, const float* demb, int32_t batch, void* workspace,
                         int64_t workspace_bytes, void* stream);
int uniir_clip_tower_bwd_head(const uniir_clip_tower* t, const float* demb, int32_t batch, void* workspace,
                              int64_t workspace_bytes, void* stream);
int uniir_clip_tower_bwd_blocks(const uniir_clip_tower* t, int32_t batch, int32_t layer_lo, int32_t layer_hi, void* workspace,
                                int64_t workspace_bytes, void* stream);
int uniir_clip_tower_bwd_stem(const uniir_clip_tower* t, const void* input, int32_t batch, void* workspace,
                              int64_t workspace_bytes, void* stream);
/* The TEXT tower on packed rows (exact): only the tokens up to and including each caption's EOT are rows of the residual stream.
 * Under CLIP's causal mask nothing behind the EOT reaches the pooled feature (clip_sf.py:43-44 -> CLIP.encode_text pools at
 * argmax(tokens)), so embeddings and activation gradients are bitwise those of the dense calls above and the weight gradients the same
 * sums without their zero terms (equal up to the order of fp32 additions).  row_off: device int32 [batch + 1], prefix sums of the
 * live lengths (argmax + 1); live_rows = row_off[batch] as known to the host.  Same workspace discipline as the dense calls. */
int64_t uniir_clip_tower_workspace_bytes_packed(const uniir_clip_tower* t, int32_t batch, int32_t live_rows,
                                                int32_t save_for_backward);
int uniir_clip_tower_fwd_packed(const uniir_clip_tower* t, const void* tokens, int32_t batch, const int32_t* row_off,
                                int32_t live_rows, float* emb_out, void* workspace, int64_t workspace_bytes,
                                int32_t save_for_backward, void* stream);
int uniir_clip_tower_bwd_head_packed(const uniir_clip_tower* t, const float* demb, int32_t batch, const int32_t* row_off,
                                     int32_t live_rows, void* workspace, int64_t workspace_bytes, void* stream);
int uniir_clip_tower_bwd_blocks_packed(const uniir_clip_tower* t, int32_t batch, const int32_t* row_off, int32_t live_rows,
                                       int32_t layer_lo, int32_t layer_hi, void* workspace, int64_t workspace_bytes, void* stream);
int uniir_clip_tower_bwd_stem_packed(const uniir_clip_tower* t, const void* tokens, int32_t batch, const int32_t* row_off,
                                     int32_t live_rows, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [FP32] forward path of the encoders in fp32 ("model.float()" of the reference: clip_sf.py:25-26 keeps fp32 weights,
 * the embedder only autocasts when use_fp16 is set).  Linear layers = uniir_sgemm (exact fp32 MFMA) followed by
 * uniir_bias_act_f32: y = (resid ? resid : 0) + act(y + bias) in place (act < 0: none, else UNIIR_ACT_*);
 * uniir_vit_assemble_f32 = uniir_vit_assemble on an fp32 patch projection; uniir_attention_f32_fwd = softmax(scale q k^T
 * [causal / key_len masks]) v with separate Q / K / V views like uniir_attention_fwd_ex, fp32 in and out, any seq.
 * Forward only; 157 TFLOP/s peak -- reference precision, not throughput.
 * ---------------------------------------------------------------------------------------------- */
int uniir_bias_act_f32(float* y, const float* bias, const float* resid, int64_t rows, int32_t cols, int32_t act,
                       void* stream);
int uniir_vit_assemble_f32(const float* patch_out, const float* class_emb, const float* pos_emb, float* x, int32_t n,
                           int32_t tokens, int32_t width, void* stream);
int uniir_attention_f32_fwd(const float* q, int64_t q_ld, const float* k, const float* v, int64_t kv_ld, float* out,
                            int64_t out_ld, const int32_t* key_len, int32_t batch, int32_t tq, int32_t tk,
                            int32_t heads, int32_t causal, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [OPT] fused AdamW over one flat fp32 parameter group (torch.optim.AdamW semantics, no amsgrad):
 *   p *= 1 - lr*wd ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ;
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * grad_scale multiplies g first (1/accumulation_steps or 1/world); also refreshes the bf16 shadow.
 * ---------------------------------------------------------------------------------------------- */
int uniir_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                     void* param_bf16, int64_t count, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int32_t step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [BLIP] extra pieces of the BLIP_FF path (src/models/uniir_blip): tanh pooler (backbone/med.py:499-511), momentum
 * EMA p_m = m p_m + (1-m) p (blip_featurefusion/blip_ff.py:288-292) over a flat buffer (+ bf16 shadow), and the
 * soft-target contrastive loss of blip_ff.py:219-231 on one matrix of dot products sim [b][n] (n = b + queue) with
 * logits = sim / *temp (temp NULL -> 1):
 *   target = alpha * softmax(sim_m / temp) + (1-alpha) * pos / sum(pos), pos_j = (ids_all[j] == ids_row[i]);
 *   row_loss[i] = -sum_j log_softmax(logits[i])_j target_j;
 *   with g = (softmax(logits) - target) * gscale * (dloss ? *dloss : 1):
 *     dsim = g / temp (optional), row_dtemp[i] = -sum_j g_j logits_j / temp (optional);
 *   row_hit[i] = pos[argmax_j sim[i][j]]  (accuracy of blip_ff.py:250-252, first max).
 * ---------------------------------------------------------------------------------------------- */
int uniir_tanh_fwd(const float* x, float* y, int64_t count, void* stream);
int uniir_tanh_bwd(const float* y, const float* dy, float* dx, int64_t count, void* stream);
int uniir_ema_update(float* param_m, const float* param, void* param_m_bf16, int64_t count, float momentum,
                     void* stream);
int uniir_softce(const float* sim, const float* sim_m, const float* temp, const int64_t* ids_row,
                 const int64_t* ids_all, int32_t b, int32_t n, float alpha, float gscale, const float* dloss,
                 float* row_loss, float* row_hit, float* dsim, float* row_dtemp, void* stream);
/* uniir_sgemm with C += instead of C = (dq = dsim[:, :b] p_m + dsim[:, b:] queue^T without a concatenated copy) */
int uniir_sgemm_acc(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                    float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, float alpha, void* stream);
/* The same product for a LONG reduction with few output tiles (BLIP: d feat = dsim[b][K] x queue[K][E], K = 57 344): deterministic
 * split-K -- slices of whole K steps into slabs of the caller's workspace, added in slice order.  Reproducible run to run; differs
 * from uniir_sgemm's k-ordered chain in the last bits (so the logits never take it).  accumulate != 0: C += alpha * A B.
 * Falls back to the plain kernel when there is nothing to split. */
int64_t uniir_sgemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t K);
int uniir_sgemm_splitk(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C,
                       int64_t ldc, int32_t M, int32_t N, int32_t K, float alpha, int32_t accumulate, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [DROPOUT] train-mode dropout / DropPath of the BLIP and T5 stacks.  Masks are counter based: element idx of a call
 * with seed s is kept iff fmix32(idx * 0x9E3779B1 ^ s) >= p * 2^32 (kept values are scaled by 1 / (1 - p)); backward
 * regenerates the mask from the same (seed, idx).  idx = row * cols + col of the logical [rows][cols] tensor.
 *   dropout_f32 : y = (resid ? resid : 0) + x * mask [* rowscale[row / rows_per_scale]]  -> fp32 and / or bf16
 *   dropout_bf16: y = x * mask [* rowscale[...]]  (bf16 rows of pitch ld; in place allowed)
 *   dropout_mask: out[idx] = mask value (0 or 1 / (1 - p)); for tests
 * rowscale (optional) is the per-item DropPath factor (0 or 1 / keep) of blip's ViT-large.
 * ---------------------------------------------------------------------------------------------- */
int uniir_dropout_f32(const float* x, const float* resid, float* y_f32, void* y_bf16, int64_t rows, int32_t cols, float p,
                      uint32_t seed, const float* rowscale, int32_t rows_per_scale, void* stream);
int uniir_dropout_bf16(const void* x, void* y, int64_t rows, int32_t cols, int64_t ld, float p, uint32_t seed,
                       const float* rowscale, int32_t rows_per_scale, void* stream);
int uniir_dropout_mask(float* out, int64_t count, float p, uint32_t seed, void* stream);
/* the same on PACKED rows (no rowscale): row r of x / resid / y is row row_map[r] of the logical tensor, idx = row_map[r] * cols + col
 * -- a BERT that runs on the live rows of its captions only draws the mask elements those rows have in the padded batch */
int uniir_dropout_f32_rows(const float* x, const float* resid, float* y_f32, void* y_bf16, int64_t rows, int32_t cols, float p,
                           uint32_t seed, const int32_t* row_map, void* stream);
int uniir_dropout_bf16_rows(const void* x, void* y, int64_t rows, int32_t cols, int64_t ld, float p, uint32_t seed,
                            const int32_t* row_map, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [CLIP_FF] pieces of the feature-fusion stack of src/models/uniir_clip/clip_featurefusion/clip_ff.py:80-96,161-192
 * (transformers T5Stack, encoder mode): RMS norm (T5LayerNorm: y = gamma * x * rsqrt(mean(x^2) + eps), no bias; same
 * argument meaning as uniir_layernorm_*), self-attention with logits = scale * q.k + rel_emb[rel_bucket[key - query +
 * seq - 1]][head] on packed qkv (T5Attention: scale 1, bucketed relative position bias; rel_bucket is the host-computed
 * bucket of every offset, [2 seq - 1] int32; bwd ACCUMULATES d rel_emb into drel [buckets][heads]), and the mean over the
 * tokens of an item.
 * ---------------------------------------------------------------------------------------------- */
int uniir_rmsnorm_fwd(const float* x, int64_t x_stride, const float* gamma, void* y_bf16, float* y_f32, int32_t rows,
                      int32_t width, float eps, void* stream);
int uniir_rmsnorm_bwd(const float* x, int64_t x_stride, const float* gamma, const void* dy, int32_t dy_is_f32,
                      const float* dres, float* dx_f32, int64_t dx_stride, void* dx_bf16, float* dgamma, int32_t rows,
                      int32_t width, float eps, void* stream);
int uniir_attention_rel_fwd(const void* qkv, void* out, float* lse, const float* rel_emb, const int32_t* rel_bucket,
                            int32_t nbuckets, float scale, int32_t batch, int32_t seq, int32_t heads, float drop_p,
                            uint32_t drop_seed, void* stream);
int uniir_attention_rel_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                            const float* rel_emb, const int32_t* rel_bucket, int32_t nbuckets, float scale, float* drel,
                            int32_t batch, int32_t seq, int32_t heads, float drop_p, uint32_t drop_seed, void* stream);
int uniir_meanpool_fwd(const float* x, float* out, int32_t n, int32_t tokens, int32_t width, void* stream);
int uniir_meanpool_bwd(const float* dout, float* dx, int32_t n, int32_t tokens, int32_t width, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [TOPK] exact brute-force inner-product top-k over an fp16 pool (FAISS "IDMap,Flat" + normalize_L2).
 *   uniir_pool_inv_norms: inv[i] = 1/sqrt(sum_j x[i][j]^2) in the oracle's summation order
 *                         (sequential fp32, no fma), 0 for an all-zero row (FAISS leaves it untouched).
 *   uniir_topk_coarse:    MFMA fp16 scan; per query a shortlist of uniir_topk_ncand(nq, kc) pool rows that is
 *                         guaranteed to contain the kc best by approximate score; kc >= k, kc <= 64.
 *   uniir_topk_rescore:   exact fp32 re-score of a shortlist in the oracle's summation order, sort by
 *                         (score desc, id asc), keep k.  exact_ws: nq*ncand floats of scratch.
 * Layouts: pool fp16 [n][dim] (dim % 64 == 0), queries fp16 [nq][dim] as stored by the embedder
 * (un-normalised), ids int64 [n].  Outputs: cand_idx int32 [nq][ncand] pool-row indices (-1 = empty),
 * out_scores f32 [nq][k] descending, out_ids int64 [nq][k] (-1 padded like FAISS).
 * workspace: uniir_topk_workspace_bytes(nq, kc) bytes.
 * ---------------------------------------------------------------------------------------------- */
int uniir_pool_inv_norms(const void* x_f16, int64_t n, int32_t dim, float* inv_norm, void* stream);
int64_t uniir_topk_workspace_bytes(int32_t nq, int32_t kc, int64_t rows);
/* candidates per query that uniir_topk_coarse writes into cand_idx (group path: 2*kc groups of 16 rows) */
int32_t uniir_topk_ncand(int32_t nq, int32_t kc);
int uniir_topk_coarse(const void* pool_f16, const float* pool_inv_norm, int64_t rows, int32_t dim,
                      const void* queries_f16, int32_t nq, int32_t kc, int32_t* cand_idx,
                      float* cand_score, void* workspace, int64_t workspace_bytes, void* stream);
int uniir_topk_rescore(const void* pool_f16, const float* pool_inv_norm, const int64_t* pool_ids,
                       int64_t rows, int32_t dim, const void* queries_f16, const float* query_inv_norm,
                       int32_t nq, const int32_t* cand_idx, int32_t ncand, int32_t k, float* exact_ws,
                       float* out_scores, int64_t* out_ids, void* stream);
/* The whole search of one pool shard in ONE call -- what the reference's FFI for this path would bind
 * (mbeir_retriever.py:188-232 search_index: faiss.normalize_L2(queries); index.search(queries, k)):
 * per chunk of <= 1024 queries one sweep of the shard (MFMA group-max scan: queries in registers and the pool streamed once for
 * <= 256 queries, GEMM-shaped above), then a fused group selection + query norm + exact re-score launch and the final sort.
 * k <= 56.  Same results as coarse + rescore, bit for bit.
 * rows * dim * 2 must stay below 2 GiB (UNIIR_ESHAPE otherwise: the shard is addressed through 31-bit buffer offsets); a larger
 * resident shard -- the whole 5.6 M x 768 M-BEIR pool on one GPU -- is searched as equal row ranges and merged with
 * uniir_topk_merge (uniir_amd/retrieval.py subshard_bounds), which is the search of the whole shard.
 * workspace: uniir_topk_ip_workspace_bytes_ex(nq, k, rows, dim) bytes (the exact requirement; the dim-agnostic form is an upper
 * bound over the sweep widths), 256-B aligned. */
int64_t uniir_topk_ip_workspace_bytes(int32_t nq, int32_t k, int64_t rows);
int64_t uniir_topk_ip_workspace_bytes_ex(int32_t nq, int32_t k, int64_t rows, int32_t dim);
int uniir_topk_ip(const void* pool_f16, const float* pool_inv_norm, const int64_t* pool_ids, int64_t rows,
                  int32_t dim, const void* queries_f16, int32_t nq, int32_t k, float* out_scores,
                  int64_t* out_ids, void* workspace, int64_t workspace_bytes, void* stream);
/* The same for ONE resident shard of any size (round 5) -- the whole 5.6 M x 768 M-BEIR pool on one GPU: equal logical sub-shards of
 * uniir_topk_subshard_rows(rows, dim) rows (a multiple of 32, below the 2-GiB buffer bound and within the fused tail's 786 432 rows;
 * = rows when the shard needs no cut), per sweep one scan launch per sub-shard, then ONE fused tail, ONE sort and ONE merge launch
 * for all of them; identical to the merge of per-sub-shard uniir_topk_ip searches, bit for bit.  pool_ids must be given (unique). */
int64_t uniir_topk_subshard_rows(int64_t rows, int32_t dim);
int64_t uniir_topk_ip_multi_workspace_bytes(int32_t nq, int32_t k, int64_t rows, int32_t dim);
int uniir_topk_ip_multi(const void* pool_f16, const float* pool_inv_norm, const int64_t* pool_ids, int64_t rows,
                        int32_t dim, const void* queries_f16, int32_t nq, int32_t k, float* out_scores,
                        int64_t* out_ids, void* workspace, int64_t workspace_bytes, void* stream);
/* queries per sweep inside uniir_topk_ip: 0 (default) = automatic -- 256 where the streaming scan applies (dim 768, >= 32768 rows,
 * shard < 2 GiB), else 1024 = the maximum.  Results never depend on it: the hook exists so that tests can drive the sweep loop with
 * small inputs, and for tuning.  Process-wide host setting.  uniir_topk_ip_sweep_queries: the value a search of this shape uses. */
int uniir_topk_set_chunk(int32_t queries_per_sweep);
int32_t uniir_topk_ip_sweep_queries(int32_t dim, int64_t rows);
/* k-way merge of per-shard results (score desc, id asc): in [nshard][nq][k] -> out [nq][k] */
int uniir_topk_merge(const float* scores, const int64_t* ids, int32_t nshard, int32_t nq, int32_t k,
                     float* out_scores, int64_t* out_ids, void* stream);
/* the same for lists of k_in entries per shard merged into the k_out best (k_out may exceed k_in) */
int uniir_topk_merge_ex(const float* scores, const int64_t* ids, int32_t nshard, int32_t nq, int32_t k_in,
                        int32_t k_out, float* out_scores, int64_t* out_ids, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [IMAGE] device-side image transform (input pipeline).  Replaces, for a decoded RGB uint8
 * image [h][w][3] already in HBM, the CPU chain of upstream clip._transform (preprocess returned by clip.load, used at
 * src/models/uniir_clip/clip_scorefusion/clip_sf.py:25-26 and applied in src/data/mbeir_dataset.py:92-100) and of BLIP's
 * eval transform (src/models/uniir_blip/backbone/transform/blip_transform.py:41-48):
 *   PIL resize to (oh, ow) with BICUBIC -> crop rows [top, top+n) x columns [left, left+n) -> x / 255 ->
 *   (v - mean[c]) / std[c] -> out fp32 [3][n][n].
 * The resize is bit-exact with Pillow's 8-bit resample; mean3 / std3 are HOST pointers to 3 floats (read at call time).
 * (oh, ow, top, left) are the caller's geometry (torchvision Resize + CenterCrop: short side n, long side
 * int(n * long / short), offsets round-half-even((size - n) / 2); BLIP: oh = ow = n, no crop).
 * ---------------------------------------------------------------------------------------------- */
int64_t uniir_image_workspace_bytes(int32_t h, int32_t w, int32_t oh, int32_t ow, int32_t n);
int uniir_image_preprocess(const void* rgb_u8, int32_t h, int32_t w, int32_t oh, int32_t ow, int32_t top, int32_t left,
                           int32_t n, const float* mean3, const float* std3, float* out, void* workspace,
                           int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNIIR_HIP_H */
