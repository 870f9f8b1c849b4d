/*
 * CPU ORACLE (test infrastructure, NOT the product path) -- plain C restatement of the integer / exact-order parts
 * of the hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Built by oracle/Makefile with -O2 -ffp-contract=off (no implicit fma: the summation orders below are exact).
 *
 * PARITY PINNING
 *  - oracle_infonce_*: restates UniIR src/models/uniir_clip/clip_scorefusion/clip_sf.py:133-144; pinned against
 *    the reference class itself through tests/golden/g1_infonce_w1.npz / g2_infonce_w2.npz (tests/test_oracle.py).
 *  - oracle_topk: restates what src/common/mbeir_retriever.py:76,85-103,188-232 asks of faiss
 *    (faiss.normalize_L2 + IndexFlatIP under IndexIDMap + search).  faiss-gpu is a third-party dependency that is
 *    absent from the reference tree and from this image (src/common/faiss_env.yml:10, no version pinned) and the
 *    reference holds no test or golden vector for it: PARITY UNPINNED against FAISS itself.  Its published
 *    algorithm is restated: fvec_renorm_L2 (x *= 1/sqrt(sum x^2) when the sum is > 0), exact inner products,
 *    k largest, descending; ties broken by ascending id (FAISS's own tie order is implementation-defined).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* score[i][j] = (fma chain over k of q[i][k]*p[j][k]) * scale  -- the order of v_mfma_f32_16x16x4_f32 */
void oracle_infonce_scores(const float* q, const float* p, float scale, int b, int B, int dim, float* score) {
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < B; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < dim; ++k) acc = fmaf(q[(size_t)i * dim + k], p[(size_t)j * dim + k], acc);
            score[(size_t)i * B + j] = acc * scale;
        }
}

/* loss = mean_i(logsumexp(score_i) - score_i[t_i]), acc = mean_i(first-argmax == t_i), t_i = toff + i (double math) */
void oracle_infonce_loss(const float* score, int b, int B, int toff, double* loss, double* acc, double* row_lse) {
    double l = 0.0, a = 0.0;
    for (int i = 0; i < b; ++i) {
        const float* r = score + (size_t)i * B;
        int am = 0;
        for (int j = 1; j < B; ++j)
            if (r[j] > r[am]) am = j;
        double s = 0.0;
        for (int j = 0; j < B; ++j) s += exp((double)r[j] - (double)r[am]);
        const double lse = (double)r[am] + log(s);
        if (row_lse) row_lse[i] = lse;
        l += lse - (double)r[toff + i];
        a += (am == toff + i) ? 1.0 : 0.0;
    }
    *loss = l / b;
    *acc = a / b;
}

static float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { ++e; man <<= 1; } while ((man & 0x400u) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* inv[i] = 1/sqrt(sum_j x_ij^2) sequential fp32 (0 when the row is all zero: FAISS leaves such rows untouched) */
void oracle_inv_norms(const uint16_t* x_f16, int64_t n, int dim, float* inv) {
    for (int64_t i = 0; i < n; ++i) {
        float s = 0.0f;
        for (int j = 0; j < dim; ++j) {
            const float v = half_to_float(x_f16[i * dim + j]);
            s = s + v * v;
        }
        inv[i] = s > 0.0f ? (float)(1.0 / (double)sqrtf(s)) : 0.0f; /* faiss fvec_renorm_L2: 1.0 / sqrtf(nr) */
    }
}

typedef struct { float s; int64_t id; } ent_t;
static int ent_cmp(const void* a, const void* b) {
    const ent_t *x = a, *y = b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->id > y->id) - (x->id < y->id);
}

/* exact top-k: normalise pool and queries like faiss.normalize_L2, inner product sequential fp32, k best by
 * (score desc, id asc); missing results are (-inf, -1) like FAISS. pool/queries are fp16 bit patterns. */
void oracle_topk(const uint16_t* pool, const int64_t* ids, int64_t n, int dim, const uint16_t* queries, int nq, int k,
                 float* out_scores, int64_t* out_ids) {
    float* pinv = malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float* qinv = malloc(sizeof(float) * (size_t)(nq > 0 ? nq : 1));
    float* qn = malloc(sizeof(float) * (size_t)dim);
    ent_t* e = malloc(sizeof(ent_t) * (size_t)(n > 0 ? n : 1));
    oracle_inv_norms(pool, n, dim, pinv);
    oracle_inv_norms(queries, nq, dim, qinv);
    for (int q = 0; q < nq; ++q) {
        for (int j = 0; j < dim; ++j) {
            float v = half_to_float(queries[(size_t)q * dim + j]);
            if (qinv[q] != 0.0f) v = v * qinv[q];
            qn[j] = v;
        }
        for (int64_t c = 0; c < n; ++c) {
            float s = 0.0f;
            for (int j = 0; j < dim; ++j) {
                float v = half_to_float(pool[c * dim + j]);
                if (pinv[c] != 0.0f) v = v * pinv[c];
                s = s + qn[j] * v;
            }
            e[c].s = s;
            e[c].id = ids ? ids[c] : c;
        }
        qsort(e, (size_t)n, sizeof(ent_t), ent_cmp);
        for (int j = 0; j < k; ++j) {
            if (j < n) { out_scores[(size_t)q * k + j] = e[j].s; out_ids[(size_t)q * k + j] = e[j].id; }
            else { out_scores[(size_t)q * k + j] = -INFINITY; out_ids[(size_t)q * k + j] = -1; }
        }
    }
    free(pinv); free(qinv); free(qn); free(e);
}
