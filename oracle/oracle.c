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

/* ------------------------------------------------------------------------------------------------------------
 * Image preprocessing of the encoders' input pipeline: upstream clip._transform, reached
 * from src/models/uniir_clip/clip_scorefusion/clip_sf.py:25-26 (clip.load -> preprocess) and applied per item in
 * src/data/mbeir_dataset.py:92-100; BLIP's eval transform (backbone/transform/blip_transform.py:41-48) is the same chain
 * with a square resize and no crop.  The arithmetic lives in third-party code that is NOT in /root/reference:
 * Pillow (Image.resize(BICUBIC) = libImaging/Resample.c, 8-bit path) and torchvision (Resize / CenterCrop / ToTensor /
 * Normalize).  Restated here from their published behaviour and pinned against Pillow 12.2.0 run in the build
 * container (tests/golden/g14_image.npz, made by tests/golden/make_golden_image.py).
 *   resample: per output coordinate a window [xmin, xmin + n) of source pixels around center = (x + 0.5) * scale with
 *   support 2 * max(scale, 1), Keys cubic (a = -0.5) weights evaluated at (i + xmin - center + 0.5) / max(scale, 1),
 *   normalised in double, rounded half away from zero to 22-bit fixed point; horizontal pass first, each pass
 *   accumulates ints from 2^21, arithmetic-shifts by 22 and clamps to 0..255 into a uint8 intermediate.
 * ------------------------------------------------------------------------------------------------------------ */
#define RS_BITS 22
static double cubic_keys(double x) {
    const double a = -0.5;
    if (x < 0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
/* -> window size; bounds[2*o] = first source index, bounds[2*o+1] = count; kk[o*ksize + i] fixed-point weights */
int oracle_resample_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int kk_cap) {
    const double scale = (double)in_size / out_size;
    const double fscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * fscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if (!bounds || !kk) return ksize;
    if ((long)ksize * out_size > kk_cap) return -1;
    double* w = (double*)malloc(sizeof(double) * ksize);
    for (int o = 0; o < out_size; ++o) {
        const double center = (o + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        const int n = xmax - xmin;
        double ww = 0.0;
        for (int i = 0; i < n; ++i) {
            w[i] = cubic_keys((i + xmin - center + 0.5) * (1.0 / fscale));
            ww += w[i];
        }
        for (int i = 0; i < ksize; ++i) {
            double v = i < n ? (ww != 0.0 ? w[i] / ww : w[i]) : 0.0;
            kk[(long)o * ksize + i] = v < 0 ? (int32_t)(-0.5 + v * (1 << RS_BITS)) : (int32_t)(0.5 + v * (1 << RS_BITS));
        }
        bounds[2 * o] = xmin;
        bounds[2 * o + 1] = n;
    }
    free(w);
    return ksize;
}
static uint8_t rs_clip8(int32_t acc) {
    const int32_t v = acc >> RS_BITS;      /* arithmetic shift: floor */
    return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v);
}
/* src uint8 [h][w][3] -> dst uint8 [oh][ow][3] (PIL Image.resize((ow, oh), BICUBIC) of an RGB image) */
int oracle_resize_bicubic_rgb8(const uint8_t* src, int h, int w, uint8_t* dst, int oh, int ow) {
    const int kx = oracle_resample_coeffs(w, ow, 0, 0, 0), ky = oracle_resample_coeffs(h, oh, 0, 0, 0);
    int32_t* bx = (int32_t*)malloc(sizeof(int32_t) * 2 * ow);
    int32_t* by = (int32_t*)malloc(sizeof(int32_t) * 2 * oh);
    int32_t* cx = (int32_t*)malloc(sizeof(int32_t) * (size_t)kx * ow);
    int32_t* cy = (int32_t*)malloc(sizeof(int32_t) * (size_t)ky * oh);
    oracle_resample_coeffs(w, ow, bx, cx, kx * ow);
    oracle_resample_coeffs(h, oh, by, cy, ky * oh);
    const uint8_t* hsrc = src;
    uint8_t* tmp = 0;
    if (ow != w) {                       /* horizontal pass (skipped when the width is unchanged, like Pillow) */
        tmp = (uint8_t*)malloc((size_t)h * ow * 3);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < ow; ++x)
                for (int c = 0; c < 3; ++c) {
                    int32_t acc = 1 << (RS_BITS - 1);
                    for (int i = 0; i < bx[2 * x + 1]; ++i)
                        acc += (int32_t)src[((size_t)y * w + bx[2 * x] + i) * 3 + c] * cx[(size_t)x * kx + i];
                    tmp[((size_t)y * ow + x) * 3 + c] = rs_clip8(acc);
                }
        hsrc = tmp;
    }
    if (oh != h) {
        for (int y = 0; y < oh; ++y)
            for (int x = 0; x < ow; ++x)
                for (int c = 0; c < 3; ++c) {
                    int32_t acc = 1 << (RS_BITS - 1);
                    for (int i = 0; i < by[2 * y + 1]; ++i)
                        acc += (int32_t)hsrc[((size_t)(by[2 * y] + i) * ow + x) * 3 + c] * cy[(size_t)y * ky + i];
                    dst[((size_t)y * ow + x) * 3 + c] = rs_clip8(acc);
                }
    } else {
        memcpy(dst, hsrc, (size_t)oh * ow * 3);
    }
    free(bx); free(by); free(cx); free(cy); free(tmp);
    return 0;
}
/* crop [top, top + n) x [left, left + n) of uint8 [h][w][3], ToTensor (x / 255 in fp32) and Normalize ((v - mean) / std):
 * out fp32 [3][n][n] */
void oracle_crop_normalize(const uint8_t* img, int h, int w, int top, int left, int n, const float* mean, const float* std,
                           float* out) {
    (void)h;
    for (int c = 0; c < 3; ++c)
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) {
                const float v = (float)img[((size_t)(top + y) * w + left + x) * 3 + c] / 255.0f;
                out[((size_t)c * n + y) * n + x] = (v - mean[c]) / std[c];
            }
}
