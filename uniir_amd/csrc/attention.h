// Shared definitions of the attention kernels (attention.hip: the general 16-row-tile kernels; attention_pair.hip: the persistent
// pair-tile kernels for plain self-attention): argument block, LDS tile layout, fragment helpers.
#pragma once
#include "common.h"
#include "../../include/uniir_hip.h"
#include <stdlib.h>

// K / V / Q / dO of a head are read by exactly one workgroup: -DUNIIR_ATT_NT=1 stages them with the nt policy (A/B build).
// MEASURED (round 3, same box, interleaved twice): step 634.5 / 635.6 ms plain, 637.7 / 637.2 ms nt -> stays off.
#ifndef UNIIR_ATT_NT
#define UNIIR_ATT_NT 0
#endif
#if UNIIR_ATT_NT
#define ATT_LD(p) __builtin_nontemporal_load(p)
#else
#define ATT_LD(p) (*(p))
#endif
#define ATT_D 64
#define SCALE_LOG2E 0.18033688011112042f  // (1/sqrt(64)) * log2(e)
#define ATT_SCALE 0.125f
#define LN2F 0.6931471805599453f
#define LOG2EF 1.4426950408889634f

#define ATT_DEFER 8.0f      // attn_fwd_kernel: a row's reference maximum moves when a logit exceeds it by more than this (log2 units)

DEVINL int swz_off(int row, int col) {  // byte offset of element (row, col) in a swizzled [rows][64] bf16 tile
    return row * 128 + ((((col >> 3) ^ (row & 7)) << 4) | ((col & 7) << 1));
}

#define ATT_THREADS 512
#define ATT_WAVES (ATT_THREADS / 64)

// legacy staging (A/B switch UNIIR_ATTN_LEGACY_STAGE=1): one slice, loads 4 deep, one HBM round trip per 2048 chunks
template <int NT>
DEVINL void stage_head(char* lds, const unsigned short* __restrict__ src, long ld, int T, int Tp, int tid) {
    const int total = Tp * 8;
    for (int c0 = 0; c0 < total; c0 += 4 * NT) {
        u32x4_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * NT + tid;
            const int row = min(c >> 3, T - 1), kc = c & 7;
            v[u] = ATT_LD(reinterpret_cast<const u32x4_t*>(src + (long)row * ld + kc * 8));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * NT + tid;
            const int row = c >> 3, kc = c & 7;
            if (c < total) {
                const u32x4_t z = {0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4_t*>(lds + row * 128 + ((kc ^ (row & 7)) << 4)) = (row < T) ? v[u] : z;
            }
        }
    }
}
// stage rows [0, Tp) of TWO [T][64] bf16 head slices (row stride `ld` elements each) into swizzled LDS, zero padded.
// ALL loads of both slices are issued before the first LDS store (one HBM round trip per staging instead of one per 2048
// chunks and slice -- four at 257 tokens; the PMC anatomy in profiles/r02_attention_pmc.txt shows the waves parked on
// exactly these waits); loads come from clamped (always valid) addresses, the zero-select happens at the LDS store.
// NL = 16-B loads per thread and slice: ceil(Tp * 8 / NT) (<= 8 for Tp <= 512 at 512 threads; the 384-thread backward only
// runs up to 128 tokens).
// `mid` runs between the loads and the LDS stores: independent work (the backward's per-row statistics with their own global
// loads) that then shares the staging's HBM round trip instead of adding one.
template <int NT, int NL, class F>
DEVINL void stage_two_n(char* ldsA, const unsigned short* __restrict__ srcA, long ldA, char* ldsB,
                        const unsigned short* __restrict__ srcB, long ldB, int T, int Tp, int tid, F&& mid) {
    const int total = Tp * 8;
    u32x4_t va[NL], vb[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int c = u * NT + tid;
        const int row = min(c >> 3, T - 1), kc = c & 7;
        va[u] = ATT_LD(reinterpret_cast<const u32x4_t*>(srcA + (long)row * ldA + kc * 8));
        vb[u] = ATT_LD(reinterpret_cast<const u32x4_t*>(srcB + (long)row * ldB + kc * 8));
    }
    mid();
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int c = u * NT + tid;
        const int row = c >> 3, kc = c & 7;
        if (c < total) {
            const u32x4_t z = {0u, 0u, 0u, 0u};
            const int off = row * 128 + ((kc ^ (row & 7)) << 4);
            *reinterpret_cast<u32x4_t*>(ldsA + off) = (row < T) ? va[u] : z;
            *reinterpret_cast<u32x4_t*>(ldsB + off) = (row < T) ? vb[u] : z;
        }
    }
}
template <int NT, class F>
DEVINL void stage_two(char* ldsA, const unsigned short* __restrict__ srcA, long ldA, char* ldsB,
                      const unsigned short* __restrict__ srcB, long ldB, int T, int Tp, int tid, F&& mid) {
    const int nl = (Tp * 8 + NT - 1) / NT;       // wave-uniform
    if (nl <= 2) stage_two_n<NT, 2>(ldsA, srcA, ldA, ldsB, srcB, ldB, T, Tp, tid, mid);
    else if (nl <= 5) stage_two_n<NT, 5>(ldsA, srcA, ldA, ldsB, srcB, ldB, T, Tp, tid, mid);
    else stage_two_n<NT, 8>(ldsA, srcA, ldA, ldsB, srcB, ldB, T, Tp, tid, mid);
}
template <int NT>
DEVINL void stage_two(char* ldsA, const unsigned short* __restrict__ srcA, long ldA, char* ldsB,
                      const unsigned short* __restrict__ srcB, long ldB, int T, int Tp, int tid) {
    stage_two<NT>(ldsA, srcA, ldA, ldsB, srcB, ldB, T, Tp, tid, [] {});
}
// b128 fragment: 8 consecutive d (k-step s) of row r0 + (lane&15)
DEVINL bf16x8_t frag_rows(const char* lds, int r0, int s, int lane) {
    const int row = r0 + (lane & 15), kc = s * 4 + (lane >> 4);
    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(lds + row * 128 + ((kc ^ (row & 7)) << 4)));
}
// same fragment straight from global (rows >= T read as zero)
DEVINL bf16x8_t frag_rows_global(const unsigned short* __restrict__ src, long ld, int r0, int s, int lane, int T) {
    const int row = r0 + (lane & 15);
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (row < T) v = *reinterpret_cast<const u32x4_t*>(src + (long)row * ld + s * 32 + (lane >> 4) * 8);
    return __builtin_bit_cast(bf16x8_t, v);
}
// transposed fragment for column tile dt (16 cols) over the 32-row block starting at rb:
// lane (i = col = lane&15, g = lane>>4) gets rows {rb+4g+0..3, rb+16+4g+0..3} of column 16*dt + (lane&15)
DEVINL bf16x8_t frag_cols_tr(const char* lds, int rb, int dt, int lane) {
    const int t = lane & 15, g = lane >> 4;
    const int row = rb + 4 * g + (t >> 2), col = 16 * dt + 4 * (t & 3);
    const s16x4_t lo = lds_read_tr16(lds + swz_off(row, col));
    const s16x4_t hi = lds_read_tr16(lds + swz_off(row + 16, col));
    const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
    const u32x4_t r = {l2[0], l2[1], h2[0], h2[1]};
    return __builtin_bit_cast(bf16x8_t, r);
}
DEVINL bf16x8_t pack8(const f32x4_t a, const f32x4_t b) {
    const u32x4_t r = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]),
                       pack_bf16x2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8_t, r);
}
DEVINL f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// the fp16 forward (attn_fwd_kernel<.., F16 = true>): the same fragments, the 16 bits read as fp16
template <bool F16>
DEVINL f32x4_t mfma16x(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    if (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
DEVINL bf16x8_t pack8x(const f32x4_t a, const f32x4_t b) {
    const u32x4_t r = {pack16x2<F16>(a[0], a[1]), pack16x2<F16>(a[2], a[3]), pack16x2<F16>(b[0], b[1]), pack16x2<F16>(b[2], b[3])};
    return __builtin_bit_cast(bf16x8_t, r);
}
// 16 rows x 64 columns of gradients (lane = row li, registers: column 16 dt + 4 g + r) -> bf16, 64 contiguous bytes per row and
// store: v_permlane16_swap exchanges the 8-byte pieces of column tiles (0,1) / (2,3) between the lane rows g = (0,1) / (2,3), so
// that every lane ends up with 8 consecutive columns starting at 32 p + {0, 16, 8, 24}[g].
template <bool F16 = false>
DEVINL void att_store_tile(const f32x4_t (&acc)[4], float scale, unsigned short* rowp, bool valid, int g) {
    u32x2_t pk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const f32x4_t x = acc[dt] * scale;
        pk[dt] = u32x2_t{pack16x2<F16>(x[0], x[1]), pack16x2<F16>(x[2], x[3])};
    }
    const int dstart = ((g & 1) << 4) | ((g & 2) << 2);      // {0, 16, 8, 24}[g]
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const auto x = __builtin_amdgcn_permlane16_swap(pk[2 * p][0], pk[2 * p + 1][0], false, false);
        const auto y = __builtin_amdgcn_permlane16_swap(pk[2 * p][1], pk[2 * p + 1][1], false, false);
        const u32x4_t v = {x[0], y[0], x[1], y[1]};
        if (valid) *reinterpret_cast<u32x4_t*>(rowp + 32 * p + dstart) = v;
    }
}

// Generalised argument block: self-attention on the packed in_proj layout (q|k|v per token, CLIP / BLIP ViT / BERT self)
// and rectangular cross-attention (BLIP MED: Tq text tokens attending to Tk image tokens) share the kernels.
struct AttnArgs {
    const unsigned short* q;      // [batch][Tq] rows, row stride q_ld, head h at column h*64
    const unsigned short* k;      // [batch][Tk] rows, row stride kv_ld
    const unsigned short* v;
    long q_ld, kv_ld;
    unsigned short* out;          // [batch*Tq][out_ld]
    long out_ld;
    float* lse;                   // [batch][H][Tq]
    const int* klen;              // optional [batch]: keys >= klen[m] are masked (BERT padding mask)
    int Tq, Tk, H, causal;
    const unsigned short* dout;   // backward only: [batch*Tq][out_ld]
    unsigned short* dq;           // [batch][Tq] rows, stride dq_ld
    unsigned short* dk;           // [batch][Tk] rows, stride dkv_ld
    unsigned short* dv;
    long dq_ld, dkv_ld;
    // T5-style attention (CLIP_FF fusion stack): logits = scale * q.k + rel_emb[rel_bucket[key - query + Tq - 1]][h]
    float scale;                  // 1/8 for CLIP / BLIP, 1 for T5
    const float* rel_emb;         // optional [buckets][H] fp32
    const int* rel_bucket;        // [Tq + Tk - 1] bucket of every key - query offset
    float* drel;                  // backward, optional: [buckets][H] += d loss / d rel_emb
    int nbuckets;
    // attention-probability dropout (BERT attention_probs_dropout_prob, T5 dropout_rate): P V uses P * mask / keep with
    // mask(seed, ((m H + h) Tq + q) Tk + key) (common.h drop_hash); the softmax statistics stay those of the full P
    float drop_p;
    unsigned drop_seed;
    int legacy_stage;             // 1 = one slice at a time, statistics before it (see launch_attn_bwd for when)
    const int* row_off;           // optional [batch + 1]: packed rows -- item m = rows row_off[m] .. row_off[m + 1] - 1 (Tq = Tk = its
                                  // length <= the Tq given, which stays the stride of lse); no relative-position bias
    int row_off_q_only;           // with row_off: only the QUERY side is packed (cross-attention; K / V dense [batch][Tk], klen allowed)
    const int* kv_row_off;        // optional [batch + 1]: the KEY / VALUE side alone is packed (item m = rows kv_row_off[m] ..): the
                                  // pooled-row attention of a packed text tower's last block (tower.hip): one dense query row per item
    int klen_add;                 // added to klen[m] (the EOT index + 1 = the key count of a pooled causal query)
#ifdef UNIIR_EXP_BUILD
    unsigned long long* stamps;   // timing build: [64 workgroups][8 waves][8] s_memtime stamps
    int exp;                      // knock-out bits: 1 phase 1 / compute, 2 phase 2, 4 stage A, 8 stage B, 16 stores
#endif
};
#ifdef UNIIR_EXP_BUILD
extern unsigned long long* g_att_stamps;     // defined in attention.hip (uniir_exp_attn_set)
extern int g_att_exp;
#define ATT_STAMP(i)                                                                                          \
    do {                                                                                                      \
        if (a.stamps) {                                                                                       \
            const int wg_ = (int)blockIdx.x - (int)(gridDim.x / 2);                                           \
            if (wg_ >= 0 && wg_ < 64 && (threadIdx.x & 63) == 0)                                              \
                a.stamps[(wg_ * 8 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime();           \
        }                                                                                                     \
    } while (0)
#define ATT_EXP(bit) (a.exp & (bit))
#else
#define ATT_STAMP(i) do {} while (0)
#define ATT_EXP(bit) 0
#endif

