// Tower-level entry points of the C ABI (include/uniir_hip.h [TOWER]): one call runs a whole CLIP tower forward or
// backward -- the per-layer launch sequence lives HERE, not in the host language, so a binding of libuniir_hip.so gets
// `encode_image` / `encode_text` (openai/CLIP VisionTransformer.forward / CLIP.encode_text as called from UniIR
// clip_sf.py:44,47) and their backward without re-writing ~90 launches per direction.  Pure host code: it sequences the
// op-level entry points of this library on the caller's stream, never allocates, never synchronises.  Weights and
// gradients are raw device pointers in a POD description; activations live in ONE caller-provided workspace whose layout
// (plan()) is a pure function of the description and the batch size, so forward and backward agree on it.
#include "common.h"
#include "../../include/uniir_hip.h"
#include <string.h>

// the forward's ops with the 16-bit storage type as a parameter (defined next to their extern "C" bf16 entry points)
int patchify_impl(const float* images, void* patches, int32_t n, int32_t res, int32_t patch, int32_t kpad, int f16, void* stream);
int vit_assemble_impl(const void* patch_out, const float* class_emb, const float* pos_emb, float* x, int32_t n, int32_t tokens,
                      int32_t width, int f16, void* stream);
int layernorm_fwd_impl(const float* x, int64_t x_stride, const float* gamma, const float* beta, void* y_16, float* y_f32,
                       int32_t rows, int32_t width, float eps, int f16, void* stream);
int attention_fwd_impl(const void* qkv, void* out, float* lse, const int32_t* row_off, int32_t batch, int32_t seq, int32_t heads,
                       int32_t causal, int f16, void* stream);
// one query row per item (attention.hip) and raw row gather / scatter-overwrite (elementwise.hip): the pooled last block below
int attention_pooled_fwd_impl(const void* q, const void* k, const void* v, int64_t kv_ld, void* out, float* lse, const int32_t* klen,
                              int32_t klen_add, const int32_t* kv_row_off, int32_t batch, int32_t tk, int32_t heads, int f16,
                              void* stream);
int attention_pooled_bwd_impl(const void* q, const void* k, const void* v, int64_t kv_ld, const void* out, const void* dout,
                              const float* lse, const int32_t* klen, int32_t klen_add, const int32_t* kv_row_off, void* dq,
                              int64_t dq_ld, void* dk, void* dv, int64_t dkv_ld, int32_t batch, int32_t tk, int32_t heads, void* stream);
int rows_copy_impl(const void* src, const int32_t* idx, void* dst, int32_t n, int32_t seq, int32_t row_bytes, int64_t big_pitch,
                   int64_t small_pitch, int32_t scatter, void* stream);

namespace {

inline int64_t al(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct Plan {
    // sizes
    int M, T, W, H, L, E, R, G, kpad;
    bool save;
    // head / stem
    int64_t eot, rows, pooled, patches, po, x0;
    // per layer (stride lay_stride when save, 0 otherwise)
    int64_t lay0, lay_stride, o_x, o_qkv, o_ao, o_lse, o_x2, o_f, o_g, o_h1, o_h2;
    bool stash_act;                 // uniir_clip_tower.stash_act with save: act(f) of every layer is kept in o_g
    int64_t x_last;                 // the residual stream after the last block
    // transients shared by forward and backward (union)
    int64_t tmp;
    int64_t g, df, dh, dx, dx2, dxb, dqkv, demb16, dpooled, drows, dx0, dpo, dconv;
    int64_t total;
    // packed text rows (uniir_clip_tower_*_packed): R = the live rows, item m = rows row_off[m] .. row_off[m + 1] - 1
    const int32_t* row_off;
    // pool_last_block: the last block's sublayers behind the K / V projection run on the ONE pooled row of every item ([M]-row
    // buffers): forward (kept for the backward with save) ...
    bool pool_last;
    int64_t pl_q, pl_h1, pl_x, pl_ao, pl_lse, pl_x2, pl_h2, pl_f, pl_g;
    // ... and backward transients
    int64_t pl_dxb, pl_df, pl_dh, pl_dx2, pl_dqkv, pl_dhq;
};

// rows < 0: the dense layout (batch x tokens rows)
Plan plan(const uniir_clip_tower* t, int batch, bool save, int rows = -1, const int32_t* row_off = nullptr) {
    Plan p;
    memset(&p, 0, sizeof(p));
    p.M = batch; p.T = t->tokens; p.W = t->width; p.H = t->heads; p.L = t->layers; p.E = t->embed_dim;
    p.R = rows >= 0 ? rows : batch * t->tokens; p.G = t->tokens - 1; p.kpad = t->kpad; p.save = save;
    p.row_off = row_off;
    const int64_t R = p.R, W = p.W, M = p.M;
    int64_t cur = 0;
    auto take = [&](int64_t bytes) { const int64_t o = cur; cur += al(bytes); return o; };
    p.eot = take(M * 4);
    p.rows = take(M * W * 4);
    p.pooled = take(M * W * 2);
    if (!t->is_text) {
        p.patches = take((int64_t)M * p.G * p.kpad * 2);
        p.po = take((int64_t)M * p.G * W * 2);
        p.x0 = take(R * W * 4);
    }
    // one layer's stash: x, qkv, ao, lse, x2, f, h1, h2
    int64_t lc = 0;
    auto ltake = [&](int64_t bytes) { const int64_t o = lc; lc += al(bytes); return o; };
    p.o_x = ltake(R * W * 4);
    p.o_qkv = ltake(R * 3 * W * 2);
    p.o_ao = ltake(R * W * 2);
    p.o_lse = ltake((int64_t)M * p.H * p.T * 4);
    p.o_x2 = ltake(R * W * 4);
    p.o_f = save ? ltake(R * 4 * W * 2) : 0;      // the pre-activation is stashed for the backward only
    p.stash_act = save && t->stash_act != 0;       // ... and (stash_act) act(f) next to it instead of re-materialising it there
    p.o_g = p.stash_act ? ltake(R * 4 * W * 2) : 0;
    p.o_h1 = ltake(R * W * 2);
    p.o_h2 = ltake(R * W * 2);
    p.lay0 = cur;
    p.lay_stride = save ? lc : 0;
    cur += save ? lc * p.L : lc;
    p.x_last = take(R * W * 4);          // with !save the stream ping-pongs between o_x of the single layer set and this
    p.pool_last = t->pool_last_block != 0;
    if (p.pool_last) {
        p.pl_q = take(M * W * 2); p.pl_h1 = take(M * W * 2); p.pl_x = take(M * W * 4); p.pl_ao = take(M * W * 2);
        p.pl_lse = take(M * (int64_t)p.H * 4); p.pl_x2 = take(M * W * 4); p.pl_h2 = take(M * W * 2);
        p.pl_f = take(M * 4 * W * 2); p.pl_g = take(M * 4 * W * 2);
    }
    // transients: forward needs g only; backward the rest
    p.tmp = cur;
    p.g = take(R * 4 * W * 2);
    if (!save) {            // a forward-only pass (embedding extraction) needs none of the backward transients below
        p.total = cur;
        return p;
    }
    p.df = take(R * 4 * W * 2);
    p.dh = take(R * W * 2);
    p.dx = take(R * W * 4);
    p.dx2 = take(R * W * 4);
    p.dxb = take(R * W * 2);
    p.dqkv = take(R * 3 * W * 2);
    p.demb16 = take(M * (int64_t)p.E * 2);
    p.dpooled = take(M * W * 2);
    p.drows = take(M * W * 4);
    if (p.pool_last) {
        p.pl_dxb = take(M * W * 2); p.pl_df = take(M * 4 * W * 2); p.pl_dh = take(M * W * 2); p.pl_dx2 = take(M * W * 4);
        p.pl_dqkv = take(M * 3 * W * 2); p.pl_dhq = take(M * W * 2);
    }
    if (!t->is_text) {
        p.dx0 = take(R * W * 4);
        p.dpo = take((int64_t)M * p.G * W * 2);
        p.dconv = take((int64_t)W * p.kpad * 4);
    }
    p.total = cur;
    return p;
}

int check_tower(const uniir_clip_tower* t, int batch) {
    if (!t || !t->blocks || batch < 0) return UNIIR_EINVAL;
    if (t->layers <= 0 || t->width <= 0 || t->heads <= 0 || t->tokens < 2 || t->embed_dim <= 0) return UNIIR_EINVAL;
    if (t->width != t->heads * 64 || t->width % 64 || t->embed_dim % 8 || t->tokens > 512) return UNIIR_ESHAPE;
    if (!t->pos_emb || !t->ln_post_w || !t->ln_post_b || !t->proj16) return UNIIR_EINVAL;
    if (t->is_text) {
        if (!t->token_emb || t->vocab <= 0) return UNIIR_EINVAL;
    } else {
        if (!t->conv16 || !t->class_emb || !t->ln_pre_w || !t->ln_pre_b) return UNIIR_EINVAL;
        if (t->patch <= 0 || t->resolution % t->patch || t->kpad % 64 || t->kpad < 3 * t->patch * t->patch) return UNIIR_ESHAPE;
        const int g = t->resolution / t->patch;
        if (g * g + 1 != t->tokens) return UNIIR_ESHAPE;
    }
    return UNIIR_OK;
}

#define TRY(call)                 \
    do {                          \
        const int rc__ = (call);  \
        if (rc__) return rc__;    \
    } while (0)

void base_desc(uniir_gemm_desc& d) {
    memset(&d, 0, sizeof(d));
    d.k_splits = 1;
    d.alpha = 1.0f;
    d.dtype = UNIIR_DT_BF16;
    d.act = UNIIR_ACT_QUICKGELU;
}

// y[M,N] = x[M,K] @ w[N,K]^T (+ epilogue)
int linear_fwd(const void* x, const void* w, void* out, int M, int N, int K, int epi, const float* bias, const float* resid,
               void* C2, void* st, int f16 = 0) {
    uniir_gemm_desc d;
    base_desc(d);
    if (f16) d.dtype = UNIIR_DT_F16;
    d.A = x; d.B = w; d.C = out; d.C2 = C2; d.bias = bias; d.resid = resid;
    d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.ldc = N; d.epilogue = epi;
    return uniir_gemm(&d, st);
}

// dx[M,K] = dy[M,N] @ w[N,K]  (optionally * act'(aux), act(aux) -> act_out, column sums of dx += colsum).  uniir_gemm produces
// act_out / colsum itself whichever kernel shape the problem runs (fused in the 256-tile epilogue, separate passes otherwise).
int linear_dgrad(const void* dy, const void* w, void* out, int M, int N, int K, const void* aux, void* act_out, float* colsum,
                 void* st) {
    uniir_gemm_desc d;
    base_desc(d);
    d.A = dy; d.B = w; d.C = out; d.C2 = aux ? act_out : nullptr; d.aux = aux; d.ldaux = K;
    d.M = M; d.N = K; d.K = N; d.lda = N; d.ldb = K; d.ldc = K; d.b_tmaj = 1;
    d.epilogue = aux ? UNIIR_EPI_DACT : UNIIR_EPI_BF16;
    d.colsum = aux ? colsum : nullptr;
    TRY(uniir_gemm(&d, st));
    if (colsum && !aux) TRY(uniir_colsum_bf16(out, K, colsum, M, K, st));
    return UNIIR_OK;
}

int wgrad_splits(int rows, int tiles) {     // ops.wgrad_splits: fill the 256 CUs, >= 16 K steps per split
    const int ncu = 256;
    int max_s = rows / (64 * 16);
    if (max_s < 1) max_s = 1;
    if (max_s > 64) max_s = 64;
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= max_s; ++s) {
        const int blocks = tiles * s;
        const int rounds = (blocks + ncu - 1) / ncu;
        const double eff = (double)blocks / ((double)rounds * ncu);
        if (blocks >= 0.9 * ncu && eff >= 0.9) return s;
        if (eff > best_eff + 1e-9) { best = s; best_eff = eff; }
    }
    return best;
}

// dw[N,K] (fp32) += dy[M,N]^T @ x[M,K]
// dbias (optional): += column sums of dy, from the same pass over dy
int linear_wgrad(const uniir_clip_tower* t, const void* dy, const void* x, float* dw, int M, int N, int K, void* st,
                 float* dbias = nullptr) {
    uniir_gemm_desc d;
    base_desc(d);
    d.A = dy; d.B = x; d.C = dw; d.a_rowsum = dbias;
    d.M = N; d.N = K; d.K = M; d.lda = N; d.ldb = K; d.ldc = K; d.a_tmaj = 1; d.b_tmaj = 1;
    d.epilogue = UNIIR_EPI_ATOMIC_F32;
    d.k_splits = wgrad_splits(M, ((N + 255) / 256) * ((K + 255) / 256));
    if (d.k_splits > 1 && t->splitk_ws && t->splitk_ws_bytes >= (int64_t)4 * d.k_splits * N * K) {
        d.splitk_ws = t->splitk_ws;
        d.splitk_ws_bytes = t->splitk_ws_bytes;
    }
    return uniir_gemm(&d, st);
}

struct Lay {
    float *x, *x2, *lse;
    void *qkv, *ao, *f, *g, *h1, *h2;
};
Lay layer_bufs(const Plan& p, char* ws, int i) {
    char* b = ws + p.lay0 + p.lay_stride * i;
    Lay l;
    l.x = (float*)(b + p.o_x); l.qkv = b + p.o_qkv; l.ao = b + p.o_ao; l.lse = (float*)(b + p.o_lse);
    l.x2 = (float*)(b + p.o_x2); l.f = b + p.o_f; l.g = b + p.o_g; l.h1 = b + p.o_h1; l.h2 = b + p.o_h2;
    return l;
}
// where the residual stream that ENTERS block i lives (block i's stash slot; the tower output after the last block)
float* stream_in(const Plan& p, char* ws, int i) {
    if (i == p.L) return (float*)(ws + p.x_last);
    if (p.save) return (float*)(ws + p.lay0 + p.lay_stride * i + p.o_x);
    return (i & 1) ? (float*)(ws + p.x_last) : (float*)(ws + p.lay0 + p.o_x);      // ping-pong without a stash
}

int blocks_fwd(const uniir_clip_tower* t, const Plan& p, char* ws, void* st) {
    const int R = p.R, W = p.W;
    const int f16 = t->dtype16 != 0;            // forward-only (tower_fwd refuses save_for_backward with it)
    for (int i = 0; i < p.L - (p.pool_last ? 1 : 0); ++i) {      // (pool_last: the last block is last_block_fwd_pooled)
        const uniir_clip_block& b = t->blocks[i];
        Lay l = layer_bufs(p, ws, i);
        float* x = stream_in(p, ws, i);
        float* xn = (!p.save && i + 1 == p.L) ? ((i & 1) ? (float*)(ws + p.lay0 + p.o_x) : (float*)(ws + p.x_last))
                                               : stream_in(p, ws, i + 1);
        TRY(layernorm_fwd_impl(x, W, b.ln1_w, b.ln1_b, l.h1, nullptr, R, W, 1e-5f, f16, st));
        TRY(linear_fwd(l.h1, b.wqkv16, l.qkv, R, 3 * W, W, UNIIR_EPI_BF16, b.bqkv, nullptr, nullptr, st, f16));
        TRY(attention_fwd_impl(l.qkv, l.ao, l.lse, p.row_off, p.M, p.T, p.H, t->is_text ? 1 : 0, f16, st));
        TRY(linear_fwd(l.ao, b.wo16, l.x2, R, W, W, UNIIR_EPI_RESID_F32, b.bo, x, nullptr, st, f16));
        TRY(layernorm_fwd_impl(l.x2, W, b.ln2_w, b.ln2_b, l.h2, nullptr, R, W, 1e-5f, f16, st));
        void* g = p.stash_act ? l.g : (void*)(ws + p.g);
        if (p.save)       // f (pre-activation) is stashed for the backward; a forward-only pass writes act(f) alone
            TRY(linear_fwd(l.h2, b.wfc16, l.f, R, 4 * W, W, UNIIR_EPI_BIAS_ACT, b.bfc, nullptr, g, st, f16));
        else
            TRY(linear_fwd(l.h2, b.wfc16, g, R, 4 * W, W, UNIIR_EPI_ACT_ONLY, b.bfc, nullptr, nullptr, st, f16));
        TRY(linear_fwd(g, b.wproj16, xn, R, W, 4 * W, UNIIR_EPI_RESID_F32, b.bproj, l.x2, nullptr, st, f16));
    }
    return UNIIR_OK;
}


// ---- the LAST block on the pooled rows (uniir_clip_tower.pool_last_block) -----------------------------------------------------
// Only ONE row per item leaves the tower: the class token (VisionTransformer.forward: ln_post(x[:, 0, :])) or the EOT row
// (CLIP.encode_text: x[arange, text.argmax(-1)]), clip_sf.py:44,47.  In the last residual block every other row's attention output,
// out-projection and MLP are computed by the reference and thrown away; what the pooled row needs from the other rows is their KEYS
// and VALUES only.  So the last block runs ln_1 and the K | V two thirds of in_proj on every row, and the Q third, the attention
// (one query row per item), out_proj, ln_2 and the MLP on the M pooled rows: 10 of the block's 12 WxW GEMM units and nearly all of
// its attention leave the step -- 1/24 x ~0.85 of the vision tower -- with the same embedding (the same dot products in the same
// order on the rows that matter) and, in the backward, the same parameter gradients without their exact-zero terms.
struct PoolIdx {
    const int32_t* idx;       // pooled row of item m = m * seq + idx[m] (idx NULL: 0)
    int seq;
    const int32_t* klen;      // dense text: keys 0 .. eot (klen[m] + 1 of them)
    int klen_add;
    const int32_t* kv_row_off;
};
PoolIdx pool_idx(const uniir_clip_tower* t, const Plan& p, char* ws) {
    PoolIdx x = {nullptr, p.T, nullptr, 0, nullptr};
    if (t->is_text) {
        x.idx = (const int32_t*)(ws + p.eot);
        if (p.row_off) { x.seq = 0; x.kv_row_off = p.row_off; }          // packed: absolute EOT rows; an item's keys are its own rows
        else { x.klen = x.idx; x.klen_add = 1; }
    }
    return x;
}
int last_block_fwd_pooled(const uniir_clip_tower* t, const Plan& p, char* ws, void* st) {
    const int R = p.R, W = p.W, M = p.M, i = p.L - 1;
    const int f16 = t->dtype16 != 0;
    const uniir_clip_block& b = t->blocks[i];
    Lay l = layer_bufs(p, ws, i);
    float* x = stream_in(p, ws, i);
    const PoolIdx px = pool_idx(t, p, ws);
    TRY(layernorm_fwd_impl(x, W, b.ln1_w, b.ln1_b, l.h1, nullptr, R, W, 1e-5f, f16, st));
    uniir_gemm_desc d;
    base_desc(d);       // K | V of every row -> columns W .. 3W of the qkv buffer
    if (f16) d.dtype = UNIIR_DT_F16;
    d.A = l.h1; d.B = (const char*)b.wqkv16 + (int64_t)W * W * 2; d.C = (char*)l.qkv + (int64_t)W * 2; d.bias = b.bqkv + W;
    d.M = R; d.N = 2 * W; d.K = W; d.lda = W; d.ldb = W; d.ldc = 3 * W; d.epilogue = UNIIR_EPI_BF16;
    TRY(uniir_gemm(&d, st));
    TRY(rows_copy_impl(l.h1, px.idx, ws + p.pl_h1, M, px.seq, W * 2, (int64_t)W * 2, (int64_t)W * 2, 0, st));
    TRY(linear_fwd(ws + p.pl_h1, b.wqkv16, ws + p.pl_q, M, W, W, UNIIR_EPI_BF16, b.bqkv, nullptr, nullptr, st, f16));
    TRY(attention_pooled_fwd_impl(ws + p.pl_q, (char*)l.qkv + (int64_t)W * 2, (char*)l.qkv + (int64_t)W * 4, 3 * W, ws + p.pl_ao,
                                  (float*)(ws + p.pl_lse), px.klen, px.klen_add, px.kv_row_off, M, p.T, p.H, f16, st));
    TRY(rows_copy_impl(x, px.idx, ws + p.pl_x, M, px.seq, W * 4, (int64_t)W * 4, (int64_t)W * 4, 0, st));
    TRY(linear_fwd(ws + p.pl_ao, b.wo16, ws + p.pl_x2, M, W, W, UNIIR_EPI_RESID_F32, b.bo, (float*)(ws + p.pl_x), nullptr, st, f16));
    TRY(layernorm_fwd_impl((float*)(ws + p.pl_x2), W, b.ln2_w, b.ln2_b, ws + p.pl_h2, nullptr, M, W, 1e-5f, f16, st));
    if (p.save)
        TRY(linear_fwd(ws + p.pl_h2, b.wfc16, ws + p.pl_f, M, 4 * W, W, UNIIR_EPI_BIAS_ACT, b.bfc, nullptr, ws + p.pl_g, st, f16));
    else
        TRY(linear_fwd(ws + p.pl_h2, b.wfc16, ws + p.pl_g, M, 4 * W, W, UNIIR_EPI_ACT_ONLY, b.bfc, nullptr, nullptr, st, f16));
    // the block's output on the pooled rows IS the tower's pooled-row buffer
    return linear_fwd(ws + p.pl_g, b.wproj16, ws + p.rows, M, W, 4 * W, UNIIR_EPI_RESID_F32, b.bproj, (float*)(ws + p.pl_x2), nullptr, st, f16);
}

// the tower output stream after blocks_fwd
float* stream_out(const Plan& p, char* ws) {
    if (p.save || (p.L & 1) == 0) return p.save ? (float*)(ws + p.x_last) : (float*)(ws + p.lay0 + p.o_x);
    return (float*)(ws + p.x_last);
}

// packed calls: a text tower, row_off given, batch <= live rows <= batch x tokens
int check_packed(const uniir_clip_tower* t, int batch, const int32_t* row_off, int rows) {
    if (!t->is_text || !row_off) return UNIIR_EINVAL;
    if (rows < batch || rows > batch * t->tokens) return UNIIR_ESHAPE;
    return UNIIR_OK;
}

int tower_fwd(const uniir_clip_tower* t, const void* input, int32_t batch, const int32_t* row_off, int rows, float* emb_out,
              void* workspace, int64_t workspace_bytes, int32_t save_for_backward, void* stream) {
    TRY(check_tower(t, batch));
    if (!input || !emb_out || !workspace) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if ((uintptr_t)workspace & 255) return UNIIR_EALIGN;
    const int f16 = t->dtype16 != 0;
    if (f16 && save_for_backward) return UNIIR_EUNSUPPORTED;       // the fp16 towers are the embedder's forward; training is bf16
    const Plan p = plan(t, batch, save_for_backward != 0, rows, row_off);
    if (workspace_bytes < p.total) return UNIIR_EINVAL;
    char* ws = (char*)workspace;
    const int M = p.M, T = p.T, W = p.W, R = p.R;
    float* x_in = stream_in(p, ws, 0);
    const int32_t* eot = nullptr;
    if (!t->is_text) {
        TRY(patchify_impl((const float*)input, ws + p.patches, M, t->resolution, t->patch, t->kpad, f16, stream));
        TRY(linear_fwd(ws + p.patches, t->conv16, ws + p.po, M * p.G, W, t->kpad, UNIIR_EPI_BF16, nullptr, nullptr, nullptr, stream, f16));
        TRY(vit_assemble_impl(ws + p.po, t->class_emb, t->pos_emb, (float*)(ws + p.x0), M, T, W, f16, stream));
        TRY(uniir_layernorm_fwd((float*)(ws + p.x0), W, t->ln_pre_w, t->ln_pre_b, nullptr, x_in, R, W, 1e-5f, stream));
    } else if (row_off) {    // packed: only the rows up to each caption's EOT exist; p.eot holds every item's last (= EOT) row
        TRY(uniir_text_embed_packed((const int32_t*)input, t->token_emb, t->pos_emb, row_off, x_in, (int32_t*)(ws + p.eot), M, T, W,
                                    t->vocab, stream));
        eot = (const int32_t*)(ws + p.eot);
    } else {
        TRY(uniir_text_embed((const int32_t*)input, t->token_emb, t->pos_emb, x_in, (int32_t*)(ws + p.eot), M, T, W, t->vocab,
                             stream));
        eot = (const int32_t*)(ws + p.eot);
    }
    TRY(blocks_fwd(t, p, ws, stream));
    if (p.pool_last) {
        TRY(last_block_fwd_pooled(t, p, ws, stream));      // writes the pooled rows itself
    } else {
        float* x_out = p.save ? (float*)(ws + p.x_last) : stream_out(p, ws);
        TRY(uniir_gather_rows(x_out, eot, (float*)(ws + p.rows), M, row_off ? 0 : T, W, stream));     // (packed: absolute row indices)
    }
    TRY(layernorm_fwd_impl((float*)(ws + p.rows), W, t->ln_post_w, t->ln_post_b, ws + p.pooled, nullptr, M, W, 1e-5f, f16, stream));
    uniir_gemm_desc d;
    base_desc(d);
    if (f16) d.dtype = UNIIR_DT_F16;
    d.A = ws + p.pooled; d.B = t->proj16; d.C = emb_out;
    d.M = M; d.N = p.E; d.K = W; d.lda = W; d.ldb = p.E; d.ldc = p.E; d.b_tmaj = 1; d.epilogue = UNIIR_EPI_F32;
    return uniir_gemm(&d, stream);
}

// backward, stage 1: projection, ln_post / ln_final, scatter of the pooled row's gradient into the token stream
int tower_bwd_head(const uniir_clip_tower* t, const float* demb, int32_t batch, const int32_t* row_off, int rows, void* workspace,
                   int64_t workspace_bytes, void* stream) {
    TRY(check_tower(t, batch));
    if (!demb || !workspace || !t->g_proj || !t->g_ln_post_w || !t->g_ln_post_b) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    const Plan p = plan(t, batch, true, rows, row_off);
    if (workspace_bytes < p.total) return UNIIR_EINVAL;
    char* ws = (char*)workspace;
    const int M = p.M, T = p.T, W = p.W, R = p.R, E = p.E;
    TRY(uniir_cast_f32_to_bf16(demb, ws + p.demb16, (int64_t)M * E, stream));
    uniir_gemm_desc d;
    base_desc(d);       // dproj[W,E] += pooled^T @ demb
    d.A = ws + p.pooled; d.B = ws + p.demb16; d.C = t->g_proj;
    d.M = W; d.N = E; d.K = M; d.lda = W; d.ldb = E; d.ldc = E; d.a_tmaj = 1; d.b_tmaj = 1; d.epilogue = UNIIR_EPI_ATOMIC_F32;
    TRY(uniir_gemm(&d, stream));
    base_desc(d);       // dpooled[M,W] = demb @ proj^T
    d.A = ws + p.demb16; d.B = t->proj16; d.C = ws + p.dpooled;
    d.M = M; d.N = W; d.K = E; d.lda = E; d.ldb = E; d.ldc = W; d.epilogue = UNIIR_EPI_BF16;
    TRY(uniir_gemm(&d, stream));
    TRY(uniir_layernorm_bwd((float*)(ws + p.rows), W, t->ln_post_w, ws + p.dpooled, 0, nullptr, (float*)(ws + p.drows), W, nullptr,
                            t->g_ln_post_w, t->g_ln_post_b, nullptr, M, W, 1e-5f, stream));
    if (p.pool_last) {      // the gradient stays on the pooled rows through the last block (last_block_bwd_pooled)
        TRY(uniir_cast_f32_to_bf16((float*)(ws + p.drows), ws + p.pl_dxb, (int64_t)M * W, stream));
        return uniir_colsum_bf16(ws + p.pl_dxb, W, t->blocks[p.L - 1].g_bproj, M, W, stream);
    }
    if (hipMemsetAsync(ws + p.dx, 0, (size_t)R * W * 4, (hipStream_t)stream) != hipSuccess) return UNIIR_ELAUNCH;
    TRY(uniir_scatter_rows((float*)(ws + p.drows), t->is_text ? (const int32_t*)(ws + p.eot) : nullptr, (float*)(ws + p.dx), M,
                           row_off ? 0 : T, W, stream));
    TRY(uniir_cast_f32_to_bf16((float*)(ws + p.dx), ws + p.dxb, (int64_t)R * W, stream));
    // bias gradient of the last block's c_proj: column sums of the incoming gradient (the other blocks get theirs from the
    // LayerNorm backward that produces their incoming gradient)
    return uniir_colsum_bf16(ws + p.dxb, W, t->blocks[p.L - 1].g_bproj, R, W, stream);
}


// backward of last_block_fwd_pooled: the incoming gradient lives on the pooled rows (ws + p.drows fp32, p.pl_dxb bf16); leaves the
// block's input gradient in dx / dxb like every other block
int last_block_bwd_pooled(const uniir_clip_tower* t, const Plan& p, char* ws, void* st) {
    const int R = p.R, W = p.W, M = p.M, i = p.L - 1;
    const uniir_clip_block& b = t->blocks[i];
    Lay l = layer_bufs(p, ws, i);
    const PoolIdx px = pool_idx(t, p, ws);
    float* dx = (float*)(ws + p.dx);
    float* dx2 = (float*)(ws + p.dx2);
    void *dxb = ws + p.dxb, *dh = ws + p.dh, *dqkv = ws + p.dqkv;
    void *pdxb = ws + p.pl_dxb, *pdf = ws + p.pl_df, *pdh = ws + p.pl_dh, *pdqkv = ws + p.pl_dqkv;
    // MLP and out_proj on the M pooled rows
    TRY(linear_dgrad(pdxb, b.wproj16, pdf, M, W, 4 * W, ws + p.pl_f, nullptr, b.g_bfc, st));
    TRY(linear_wgrad(t, pdxb, ws + p.pl_g, b.g_wproj, M, W, 4 * W, st));
    TRY(linear_wgrad(t, pdf, ws + p.pl_h2, b.g_wfc, M, 4 * W, W, st));
    TRY(linear_dgrad(pdf, b.wfc16, pdh, M, 4 * W, W, nullptr, nullptr, nullptr, st));
    TRY(uniir_layernorm_bwd((float*)(ws + p.pl_x2), W, b.ln2_w, pdh, 0, (float*)(ws + p.drows), (float*)(ws + p.pl_dx2), W, pdxb,
                            b.g_ln2_w, b.g_ln2_b, b.g_bo, M, W, 1e-5f, st));
    TRY(linear_wgrad(t, pdxb, ws + p.pl_ao, b.g_wo, M, W, W, st));
    TRY(linear_dgrad(pdxb, b.wo16, pdh, M, W, W, nullptr, nullptr, nullptr, st));
    // one query row per item: dq -> columns 0 .. W of the pooled [M, 3W] gradient, dk | dv of EVERY row -> columns W .. 3W of dqkv
    TRY(attention_pooled_bwd_impl(ws + p.pl_q, (char*)l.qkv + (int64_t)W * 2, (char*)l.qkv + (int64_t)W * 4, 3 * W, ws + p.pl_ao, pdh,
                                  (float*)(ws + p.pl_lse), px.klen, px.klen_add, px.kv_row_off, pdqkv, 3 * W,
                                  (char*)dqkv + (int64_t)W * 2, (char*)dqkv + (int64_t)W * 4, 3 * W, M, p.T, p.H, st));
    uniir_gemm_desc d;
    base_desc(d);       // in_proj weight / bias gradient, K | V rows: over every row
    d.A = (char*)dqkv + (int64_t)W * 2; d.B = l.h1; d.C = b.g_wqkv + (int64_t)W * W; d.a_rowsum = b.g_bqkv + W;
    d.M = 2 * W; d.N = W; d.K = R; d.lda = 3 * W; d.ldb = W; d.ldc = W; d.a_tmaj = 1; d.b_tmaj = 1;
    d.epilogue = UNIIR_EPI_ATOMIC_F32;
    d.k_splits = wgrad_splits(R, ((2 * W + 255) / 256) * ((W + 255) / 256));
    if (d.k_splits > 1 && t->splitk_ws && t->splitk_ws_bytes >= (int64_t)4 * d.k_splits * 2 * W * W) {
        d.splitk_ws = t->splitk_ws;
        d.splitk_ws_bytes = t->splitk_ws_bytes;
    }
    TRY(uniir_gemm(&d, st));
    base_desc(d);       // ... Q rows: over the pooled rows
    d.A = pdqkv; d.B = ws + p.pl_h1; d.C = b.g_wqkv; d.a_rowsum = b.g_bqkv;
    d.M = W; d.N = W; d.K = M; d.lda = 3 * W; d.ldb = W; d.ldc = W; d.a_tmaj = 1; d.b_tmaj = 1; d.epilogue = UNIIR_EPI_ATOMIC_F32;
    TRY(uniir_gemm(&d, st));
    base_desc(d);       // d ln_1 out of every row: its K | V part ...
    d.A = (char*)dqkv + (int64_t)W * 2; d.B = (const char*)b.wqkv16 + (int64_t)W * W * 2; d.C = dh;
    d.M = R; d.N = W; d.K = 2 * W; d.lda = 3 * W; d.ldb = W; d.ldc = W; d.b_tmaj = 1; d.epilogue = UNIIR_EPI_BF16;
    TRY(uniir_gemm(&d, st));
    // ... and, for the pooled rows, the whole K = 3W product in one accumulation (what the dense call computes for them)
    TRY(rows_copy_impl((char*)dqkv + (int64_t)W * 2, px.idx, (char*)pdqkv + (int64_t)W * 2, M, px.seq, 2 * W * 2, (int64_t)3 * W * 2,
                       (int64_t)3 * W * 2, 0, st));
    base_desc(d);
    d.A = pdqkv; d.B = b.wqkv16; d.C = ws + p.pl_dhq;
    d.M = M; d.N = W; d.K = 3 * W; d.lda = 3 * W; d.ldb = W; d.ldc = W; d.b_tmaj = 1; d.epilogue = UNIIR_EPI_BF16;
    TRY(uniir_gemm(&d, st));
    TRY(rows_copy_impl(ws + p.pl_dhq, px.idx, dh, M, px.seq, W * 2, (int64_t)W * 2, (int64_t)W * 2, 1, st));
    // the residual gradient entering ln_1's backward is zero except on the pooled rows
    if (hipMemsetAsync(dx2, 0, (size_t)R * W * 4, (hipStream_t)st) != hipSuccess) return UNIIR_ELAUNCH;
    TRY(uniir_scatter_rows((float*)(ws + p.pl_dx2), px.idx, dx2, M, px.seq, W, st));
    return uniir_layernorm_bwd(l.x, W, b.ln1_w, dh, 0, dx2, dx, W, dxb, b.g_ln1_w, b.g_ln1_b,
                               i > 0 ? t->blocks[i - 1].g_bproj : nullptr, R, W, 1e-5f, st);
}

// backward, stage 2: residual blocks layer_hi-1 ... layer_lo (call with descending ranges that cover [0, layers))
int tower_bwd_blocks(const uniir_clip_tower* t, int32_t batch, const int32_t* row_off, int rows, int32_t layer_lo, int32_t layer_hi,
                     void* workspace, int64_t workspace_bytes, void* stream) {
    TRY(check_tower(t, batch));
    if (!workspace || layer_lo < 0 || layer_hi > t->layers || layer_lo > layer_hi) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    const Plan p = plan(t, batch, true, rows, row_off);
    if (workspace_bytes < p.total) return UNIIR_EINVAL;
    char* ws = (char*)workspace;
    const int W = p.W, R = p.R;
    float* dx = (float*)(ws + p.dx);
    float* dx2 = (float*)(ws + p.dx2);
    void *dxb = ws + p.dxb, *g = ws + p.g, *df = ws + p.df, *dh = ws + p.dh, *dqkv = ws + p.dqkv;
    for (int i = layer_hi - 1; i >= layer_lo; --i) {
        const uniir_clip_block& b = t->blocks[i];
        if (!b.g_wqkv || !b.g_bqkv || !b.g_wo || !b.g_bo || !b.g_wfc || !b.g_bfc || !b.g_wproj || !b.g_bproj || !b.g_ln1_w ||
            !b.g_ln1_b || !b.g_ln2_w || !b.g_ln2_b)
            return UNIIR_EINVAL;
        if (p.pool_last && i == p.L - 1) {
            TRY(last_block_bwd_pooled(t, p, ws, stream));
            continue;
        }
        Lay l = layer_bufs(p, ws, i);
        // d(mlp): df = (dx @ Wproj) * act'(f); the same epilogue re-materialises g = act(f) for dWproj and sums df's columns
        // into the c_fc bias gradient
        // (stash_act: l.g holds act(f) since the forward -- the epilogue writes no second output)
        TRY(linear_dgrad(dxb, b.wproj16, df, R, W, 4 * W, l.f, p.stash_act ? nullptr : g, b.g_bfc, stream));
        TRY(linear_wgrad(t, dxb, p.stash_act ? l.g : g, b.g_wproj, R, W, 4 * W, stream));
        TRY(linear_wgrad(t, df, l.h2, b.g_wfc, R, 4 * W, W, stream));
        TRY(linear_dgrad(df, b.wfc16, dh, R, 4 * W, W, nullptr, nullptr, nullptr, stream));            // d ln_2 out
        TRY(uniir_layernorm_bwd(l.x2, W, b.ln2_w, dh, 0, dx, dx2, W, dxb, b.g_ln2_w, b.g_ln2_b, b.g_bo, R, W, 1e-5f, stream));
        TRY(linear_wgrad(t, dxb, l.ao, b.g_wo, R, W, W, stream));
        TRY(linear_dgrad(dxb, b.wo16, dh, R, W, W, nullptr, nullptr, nullptr, stream));                // d attention out
        if (row_off) TRY(uniir_attention_bwd_packed(l.qkv, l.ao, dh, l.lse, dqkv, row_off, p.M, p.T, p.H, t->is_text ? 1 : 0, stream));
        else TRY(uniir_attention_bwd(l.qkv, l.ao, dh, l.lse, dqkv, p.M, p.T, p.H, t->is_text ? 1 : 0, stream));
        TRY(linear_wgrad(t, dqkv, l.h1, b.g_wqkv, R, 3 * W, W, stream, b.g_bqkv));       // + the in_proj bias gradient
        TRY(linear_dgrad(dqkv, b.wqkv16, dh, R, 3 * W, W, nullptr, nullptr, nullptr, stream));         // d ln_1 out
        TRY(uniir_layernorm_bwd(l.x, W, b.ln1_w, dh, 0, dx2, dx, W, dxb, b.g_ln1_w, b.g_ln1_b,
                                i > 0 ? t->blocks[i - 1].g_bproj : nullptr, R, W, 1e-5f, stream));
    }
    return UNIIR_OK;
}

// backward, stage 3: what feeds the first block (ln_pre + patch embedding, or the token / positional embeddings)
int tower_bwd_stem(const uniir_clip_tower* t, const void* input, int32_t batch, const int32_t* row_off, int rows, void* workspace,
                   int64_t workspace_bytes, void* stream) {
    TRY(check_tower(t, batch));
    if (!workspace || !input) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    const Plan p = plan(t, batch, true, rows, row_off);
    if (workspace_bytes < p.total) return UNIIR_EINVAL;
    char* ws = (char*)workspace;
    const int M = p.M, T = p.T, W = p.W, R = p.R;
    if (t->is_text) {
        if (!t->g_token || !t->g_pos) return UNIIR_EINVAL;
        if (row_off)
            return uniir_text_embed_bwd_packed((const int32_t*)input, (float*)(ws + p.dx), row_off, t->g_token, t->g_pos, M, T, W,
                                               t->vocab, stream);
        return uniir_text_embed_bwd((const int32_t*)input, (float*)(ws + p.dx), t->g_token, t->g_pos, M, T, W, t->vocab, stream);
    }
    if (!t->g_conv || !t->g_class || !t->g_pos || !t->g_ln_pre_w || !t->g_ln_pre_b) return UNIIR_EINVAL;
    TRY(uniir_layernorm_bwd((float*)(ws + p.x0), W, t->ln_pre_w, ws + p.dx, 1, nullptr, (float*)(ws + p.dx0), W, nullptr,
                            t->g_ln_pre_w, t->g_ln_pre_b, nullptr, R, W, 1e-5f, stream));
    TRY(uniir_vit_assemble_bwd((float*)(ws + p.dx0), ws + p.dpo, t->g_class, t->g_pos, M, T, W, stream));
    if (hipMemsetAsync(ws + p.dconv, 0, (size_t)W * p.kpad * 4, (hipStream_t)stream) != hipSuccess) return UNIIR_ELAUNCH;
    TRY(linear_wgrad(t, ws + p.dpo, ws + p.patches, (float*)(ws + p.dconv), M * p.G, W, p.kpad, stream));
    return uniir_unpad_add((float*)(ws + p.dconv), t->g_conv, W, 3 * t->patch * t->patch, p.kpad, stream);
}

}  // namespace

extern "C" int64_t uniir_clip_tower_workspace_bytes(const uniir_clip_tower* t, int32_t batch, int32_t save_for_backward) {
    if (check_tower(t, batch)) return -1;
    return plan(t, batch, save_for_backward != 0).total;
}
extern "C" int uniir_clip_tower_fwd(const uniir_clip_tower* t, const void* input, int32_t batch, float* emb_out,
                                    void* workspace, int64_t workspace_bytes, int32_t save_for_backward, void* stream) {
    return tower_fwd(t, input, batch, nullptr, -1, emb_out, workspace, workspace_bytes, save_for_backward, stream);
}
extern "C" int uniir_clip_tower_bwd_head(const uniir_clip_tower* t, const float* demb, int32_t batch, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
    return tower_bwd_head(t, demb, batch, nullptr, -1, workspace, workspace_bytes, stream);
}
extern "C" int uniir_clip_tower_bwd_blocks(const uniir_clip_tower* t, int32_t batch, int32_t layer_lo, int32_t layer_hi,
                                           void* workspace, int64_t workspace_bytes, void* stream) {
    return tower_bwd_blocks(t, batch, nullptr, -1, layer_lo, layer_hi, workspace, workspace_bytes, stream);
}
extern "C" int uniir_clip_tower_bwd_stem(const uniir_clip_tower* t, const void* input, int32_t batch, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
    return tower_bwd_stem(t, input, batch, nullptr, -1, workspace, workspace_bytes, stream);
}
extern "C" int uniir_clip_tower_bwd(const uniir_clip_tower* t, const void* input, const float* demb, int32_t batch,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
    TRY(uniir_clip_tower_bwd_head(t, demb, batch, workspace, workspace_bytes, stream));
    TRY(uniir_clip_tower_bwd_blocks(t, batch, 0, t->layers, workspace, workspace_bytes, stream));
    return uniir_clip_tower_bwd_stem(t, input, batch, workspace, workspace_bytes, stream);
}

// ---- the text tower on PACKED rows (exact): only the tokens up to and including each caption's EOT are rows of the residual stream.
// Under the causal mask nothing behind the EOT reaches the pooled feature (clip_sf.py:43-44; upstream pools at argmax(tokens)), so the
// embeddings and every activation gradient are bitwise those of the dense call; weight gradients are the same sums over fewer (all
// the non-zero) terms, i.e. equal up to the order of fp32 additions.  row_off: device int32 [batch + 1], prefix sums of the live
// lengths (argmax + 1 per caption); live_rows = row_off[batch], known to the host (it sizes the GEMMs and the workspace).
extern "C" int64_t uniir_clip_tower_workspace_bytes_packed(const uniir_clip_tower* t, int32_t batch, int32_t live_rows,
                                                           int32_t save_for_backward) {
    if (check_tower(t, batch) || !t->is_text || live_rows < batch || live_rows > batch * t->tokens) return -1;
    return plan(t, batch, save_for_backward != 0, live_rows).total;
}
extern "C" int uniir_clip_tower_fwd_packed(const uniir_clip_tower* t, const void* tokens, int32_t batch, const int32_t* row_off,
                                           int32_t live_rows, float* emb_out, void* workspace, int64_t workspace_bytes,
                                           int32_t save_for_backward, void* stream) {
    TRY(check_tower(t, batch));
    TRY(check_packed(t, batch, row_off, live_rows));
    return tower_fwd(t, tokens, batch, row_off, live_rows, emb_out, workspace, workspace_bytes, save_for_backward, stream);
}
extern "C" int uniir_clip_tower_bwd_head_packed(const uniir_clip_tower* t, const float* demb, int32_t batch, const int32_t* row_off,
                                                int32_t live_rows, void* workspace, int64_t workspace_bytes, void* stream) {
    TRY(check_tower(t, batch));
    TRY(check_packed(t, batch, row_off, live_rows));
    return tower_bwd_head(t, demb, batch, row_off, live_rows, workspace, workspace_bytes, stream);
}
extern "C" int uniir_clip_tower_bwd_blocks_packed(const uniir_clip_tower* t, int32_t batch, const int32_t* row_off, int32_t live_rows,
                                                  int32_t layer_lo, int32_t layer_hi, void* workspace, int64_t workspace_bytes,
                                                  void* stream) {
    TRY(check_tower(t, batch));
    TRY(check_packed(t, batch, row_off, live_rows));
    return tower_bwd_blocks(t, batch, row_off, live_rows, layer_lo, layer_hi, workspace, workspace_bytes, stream);
}
extern "C" int uniir_clip_tower_bwd_stem_packed(const uniir_clip_tower* t, const void* tokens, int32_t batch, const int32_t* row_off,
                                                int32_t live_rows, void* workspace, int64_t workspace_bytes, void* stream) {
    TRY(check_tower(t, batch));
    TRY(check_packed(t, batch, row_off, live_rows));
    return tower_bwd_stem(t, tokens, batch, row_off, live_rows, workspace, workspace_bytes, stream);
}
