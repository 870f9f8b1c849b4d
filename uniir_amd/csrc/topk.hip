// Exact brute-force inner-product top-k over an fp16 candidate pool: the MI355X replacement of
// faiss.normalize_L2 + index_factory("IDMap,Flat", METRIC_INNER_PRODUCT) + index.search
// (UniIR src/common/mbeir_retriever.py:76,85-103,188-232).
//
//   coarse : fp16 MFMA scan (gemm_core.h main loop, fp32 accumulate) of a [128 queries] x [pool slice] panel
//            with a threshold-select epilogue: a score enters the per-query candidate buffer only when it
//            beats the current kc-th best, so after the first tiles almost nothing is appended.
//   rescore: the shortlist is re-scored in fp32 in the oracle's exact summation order (sequential, no fma;
//            oracle/topk_oracle.c) and sorted by (score desc, id asc) -> bit-exact distances, identical ids.
//
// That is the op-level pair (uniir_topk_coarse / uniir_topk_rescore).  The one-call search uniir_topk_ip (end of this file) runs,
// per sweep of <= 1024 queries, THREE launches:
//   scan  : gmax[q][g] = best approximate score of the 16 pool rows of group g.  <= 64 queries: topk_stream2_kernel (queries
//           in registers, pool streamed through wave-private LDS-DMA rings); 65..256: topk_stream5_kernel (the ring shared by
//           2 / 4 waves of 64 register-resident queries each); more: topk_gmax_pp_kernel (ping-pong GEMM core);
//   tail 1: topk_tail_select_rescore_kernel: the candidate groups of a query (hierarchically from the scan's per-wave maxima: every
//           group within the proven rounding bound of the k-th best group maximum, topk_select.h GselBound -- round 5; the fixed
//           k + 8 best groups before), the query's inverse norm, the exact re-score of those groups' rows (LDS-DMA gather);
//   tail 2: topk_tail_sort_kernel: (score desc, id asc) by rank counting.
// This file: inverse norms, the scans, the selection kernels, the one-call search.  topk_tail.hip: re-score, sorts, merges and the
// fused tail; topk_select.h: the group-selection templates both use; topk.h: shared constants.
// Exactness: an fp16-product / fp32-accumulate score differs from the oracle's only by rounding (bounded: topk_select.h), so the
// true top k rows lie in the groups whose maximum is within that bound of the k-th best group maximum (the fused tail), a fortiori
// in the k + 8 best groups on any non-adversarial input (the op-level selection kernels; ties at the threshold keep up to 2 (k + 8)
// groups in both); the re-score then reproduces the oracle bit for bit.  No run-time switches: the variants that lost their A/B (filtered scan, one-wave rolling-register scan,
// default-policy pool streams, ...) are described in experiments/topk/README.md.
#include "topk.h"
#include "topk_select.h"

// inv_norm[i] = 1/sqrt(sum_j x_j^2), sequential fp32 without fma (matches oracle); 0 for zero rows.
// A lane owns a row (the sum is one chain in element order), but the rows are FETCHED by the wave: 64 rows x 128 B per step, eight
// lanes side by side on a row (whole cache lines, each read once), through a wave-private LDS slab from which every lane then takes
// its own row's 64 elements.  (Round 3's one-thread-one-row loads at a 1 536-B stride fetched 6.5 GB for a 1.075-GB pool: 0.97 ms.)
#define TKN_ROW 144      // slab row stride: 128 B + 16 (the lanes' 16-byte reads of their own rows fall on distinct banks)
__global__ __launch_bounds__(256) void inv_norm_kernel(const unsigned short* __restrict__ x, long n, int dim,
                                                       float* __restrict__ inv) {
    __shared__ __attribute__((aligned(16))) char stage[4][64 * TKN_ROW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long row0 = ((long)blockIdx.x * 4 + w) * 64;
    if (row0 >= n) return;                                  // (whole wave; no workgroup barrier below)
    char* st = stage[w];
    const int sub = lane >> 3, ch = lane & 7;
    float s = 0.f;
    for (int c0 = 0; c0 < dim; c0 += 64) {
        const bool live = c0 + ch * 8 < dim;                // the last step of a dim that is not a multiple of 64
        u32x4_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long gr = min(row0 + sub + 8 * i, n - 1);
            v[i] = live ? *reinterpret_cast<const u32x4_t*>(x + gr * dim + c0 + ch * 8) : u32x4_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4_t*>(st + (sub + 8 * i) * TKN_ROW + ch * 16) = v[i];
        const int nch = min(8, (dim - c0) / 8);
        for (int c = 0; c < nch; ++c) {
            const u32x4_t q = *reinterpret_cast<const u32x4_t*>(st + lane * TKN_ROW + c * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = f16_to_f32((unsigned short)(q[e] & 0xffffu));
                const float hi = f16_to_f32((unsigned short)(q[e] >> 16));
                s = __fadd_rn(s, __fmul_rn(lo, lo));
                s = __fadd_rn(s, __fmul_rn(hi, hi));
            }
        }
    }
    // FAISS fvec_renorm_L2: inv_nr = 1.0 / sqrtf(nr) (double division of a correctly rounded float sqrt). The f32
    // sqrt is taken through f64 (exactly the correctly rounded f32 result) so it cannot be lowered to v_rsq/v_sqrt.
    if (row0 + lane < n) inv[row0 + lane] = s > 0.f ? (float)(1.0 / (double)(float)sqrt((double)s)) : 0.f;
}

extern "C" int uniir_pool_inv_norms(const void* x_f16, int64_t n, int32_t dim, float* inv_norm, void* stream) {
    if (!x_f16 || !inv_norm || n < 0 || dim <= 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    if (dim % 8) return UNIIR_ESHAPE;
    if ((uintptr_t)x_f16 & 15) return UNIIR_EALIGN;
    hipLaunchKernelGGL(inv_norm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x_f16, (long)n, dim, inv_norm);       // 4 waves x 64 rows per workgroup
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

DEVINL bool better(float s1, int i1, float s2, int i2) { return s1 > s2 || (s1 == s2 && i1 < i2); }

// One wave selects the best `kc` of buf[0..cnt) (cnt <= TK_CAP) by (score desc, idx asc).  Lane j ends up
// holding the j-th best in (*my_s, *my_i); returns the new count and the kc-th score through tau_out
// (-inf when fewer than kc entries exist).  Does not write memory.
DEVINL int wave_select(const TkEntry* buf, int cnt, int kc, int lane, float* tau_out, float* my_s, int* my_i) {
    float s[TK_CAP / 64];
    int ix[TK_CAP / 64];
#pragma unroll
    for (int e = 0; e < TK_CAP / 64; ++e) {
        const int p = lane + 64 * e;
        if (p < cnt) { s[e] = buf[p].score; ix[e] = buf[p].idx; } else { s[e] = -INFINITY; ix[e] = 0x7fffffff; }
    }
    const int keep = cnt < kc ? cnt : kc;
    float last = -INFINITY, ms = -INFINITY;
    int mi = -1;
    for (int j = 0; j < keep; ++j) {
        float bs = s[0]; int bi = ix[0]; int be = 0;
#pragma unroll
        for (int e = 1; e < TK_CAP / 64; ++e)
            if (better(s[e], ix[e], bs, bi)) { bs = s[e]; bi = ix[e]; be = e; }
        float ws = bs; int wi = bi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(ws, o, 64);
            const int oi = __shfl_xor(wi, o, 64);
            if (better(os, oi, ws, wi)) { ws = os; wi = oi; }
        }
        if (wi == bi && bi != 0x7fffffff) {  // this lane owns the winner (idx is unique within a buffer)
#pragma unroll
            for (int e = 0; e < TK_CAP / 64; ++e)
                if (e == be) { s[e] = -INFINITY; ix[e] = 0x7fffffff; }
        }
        if (lane == j) { ms = ws; mi = wi; }
        last = ws;
    }
    *tau_out = (keep == kc) ? last : -INFINITY;
    *my_s = ms;
    *my_i = mi;
    return keep;
}

// grid = (nslices, nqtiles). Block scans pool rows [slice*rows_per_slice, ...) for queries [qt*128, +128).
// MFMA tile = [256 candidates] x [128 queries] x dim via the LDS-DMA main loop <2,2,32> (48 KiB LDS, 4 waves, up to
// three workgroups per CU so one workgroup's tile prologue / select epilogue hides under the others' streaming).
using TkShape = GldsShape<2, 2, 32>;

__global__ __launch_bounds__(256, 2) void topk_coarse_kernel(const unsigned short* __restrict__ pool,
                                                             const float* __restrict__ pinv, long rows, int dim,
                                                             const unsigned short* __restrict__ queries, int nq,
                                                             int kc, long rows_per_slice,
                                                             TkEntry* __restrict__ bufs,      // [blocks][128][CAP]
                                                             TkEntry* __restrict__ partial) { // [nslices][nq][kc]
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* tau = reinterpret_cast<float*>(lds + TkShape::LDS_BYTES);          // [128]
    int* cnt = reinterpret_cast<int*>(lds + TkShape::LDS_BYTES + 512);        // [128]
    int* flag = reinterpret_cast<int*>(lds + TkShape::LDS_BYTES + 1024);      // [1]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int slice = blockIdx.x, qt = blockIdx.y;
    const int q0 = qt * TK_QT;
    const long r_begin = (long)slice * rows_per_slice;
    const long r_end = min(rows, r_begin + rows_per_slice);
    TkEntry* mybuf = bufs + ((long)blockIdx.y * gridDim.x + blockIdx.x) * TK_QT * TK_CAP;
    if (tid < TK_QT) { tau[tid] = -INFINITY; cnt[tid] = 0; }
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    const int wm = (w >> 1) * 128, wn = (w & 1) * 64;
    const int li = lane & 15, lg = lane >> 4;
    for (long n0 = r_begin; n0 < r_end; n0 += TK_CT) {
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // A operand = pool rows [n0, n0+256) (rows >= r_end are clamped duplicates, masked below); B = queries
        glds_mainloop<ElemF16, false, false, 2, 2, 32>(pool, dim, (int)r_end, queries, dim, nq, (int)n0, q0, 0, dim, lds, acc);
        // threshold-select epilogue: lane owns queries wn + j*16 + 4*lg + r of candidate row wm + i*16 + li
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = wn + j * 16 + 4 * lg + r;
                t[j][r] = (q0 + ql < nq) ? tau[ql] : INFINITY;   // +inf: padded queries never append
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long n = n0 + wm + i * 16 + li;
            const bool nok = n < r_end;
            const float iv = nok ? pinv[n] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sc = acc[i][j][r] * iv;
                    if (nok && sc > t[j][r]) {
                        const int ql = wn + j * 16 + 4 * lg + r;
                        const int pos = atomicAdd(&cnt[ql], 1);
                        if (pos < TK_CAP) { mybuf[ql * TK_CAP + pos].score = sc; mybuf[ql * TK_CAP + pos].idx = (int)n; }
                    }
                }
        }
        __syncthreads();
        if (tid < TK_QT && cnt[tid] > TK_CAP - TK_CT) flag[0] = 1;
        __syncthreads();
        if (flag[0]) {
            for (int ql = w; ql < TK_QT; ql += 4) {
                int c = cnt[ql];
                if (c > TK_CAP) c = TK_CAP;
                if (c > kc) {
                    float tt, ms;
                    int mi;
                    const int keep = wave_select(mybuf + ql * TK_CAP, c, kc, lane, &tt, &ms, &mi);
                    if (lane < keep) { mybuf[ql * TK_CAP + lane].score = ms; mybuf[ql * TK_CAP + lane].idx = mi; }
                    if (lane == 0) { cnt[ql] = keep; tau[ql] = tt; }
                }
            }
            __syncthreads();
            if (tid == 0) flag[0] = 0;
            __syncthreads();
        }
    }
    // final selection -> partial[slice][q][kc]
    __syncthreads();
    for (int ql = w; ql < TK_QT; ql += 4) {
        const int q = q0 + ql;
        if (q >= nq) continue;
        int c = cnt[ql];
        if (c > TK_CAP) c = TK_CAP;
        float tt, ms;
        int mi;
        const int keep = wave_select(mybuf + ql * TK_CAP, c, kc, lane, &tt, &ms, &mi);
        TkEntry* dst = partial + ((long)slice * nq + q) * kc;
        if (lane < kc) {
            TkEntry e;
            if (lane < keep) { e.score = ms; e.idx = mi; } else { e.score = -INFINITY; e.idx = -1; }
            dst[lane] = e;
        }
    }
}

// merge the per-slice partial lists: one block per query, iterative selection of kc best.
__global__ __launch_bounds__(256) void topk_merge_partial_kernel(const TkEntry* __restrict__ partial, int nslices,
                                                                 int nq, int kc, int* __restrict__ cand_idx,
                                                                 float* __restrict__ cand_score) {
    __shared__ float ss[4];
    __shared__ int si[4];
    __shared__ float wsel;
    __shared__ int isel;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int total = nslices * kc;
    float last_s = INFINITY;
    int last_i = -1;
    for (int j = 0; j < kc; ++j) {
        // best entry strictly after (last_s, last_i) in (score desc, idx asc) order
        float bs = -INFINITY; int bi = 0x7fffffff;
        for (int e = tid; e < total; e += 256) {
            const int sl = e / kc, p = e - sl * kc;
            const TkEntry en = partial[((long)sl * nq + q) * kc + p];
            if (en.idx < 0) continue;
            const bool after = (en.score < last_s) || (en.score == last_s && en.idx > last_i);
            if (after && better(en.score, en.idx, bs, bi)) { bs = en.score; bi = en.idx; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (better(os, oi, bs, bi)) { bs = os; bi = oi; }
        }
        if (lane == 0) { ss[w] = bs; si[w] = bi; }
        __syncthreads();
        if (tid == 0) {
            float fs = ss[0]; int fi = si[0];
            for (int k = 1; k < 4; ++k) if (better(ss[k], si[k], fs, fi)) { fs = ss[k]; fi = si[k]; }
            wsel = fs; isel = fi;
            cand_idx[(long)q * kc + j] = (fi == 0x7fffffff) ? -1 : fi;
            if (cand_score) cand_score[(long)q * kc + j] = fs;
        }
        __syncthreads();
        last_s = wsel; last_i = isel;
        if (last_i == 0x7fffffff) { last_s = -INFINITY; }
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------------------------
// Group-max path (nq <= TK_GPATH_MAXQ): the streaming kernel keeps NO per-query state.  Per 16-candidate group
// (one MFMA row tile) and query it writes the group's best approximate score; a selection kernel then finds the
// kc-th best group per query and hands every member of the qualifying groups to the exact re-score.  A true top-k
// candidate always sits in a group whose maximum is at least its own score, so the result is exact (up to the same
// near-tie margin kc - k as the buffered path).  No atomics, no compaction, no buffers in the HBM-bound sweep.

template <int WM>   // WM = 1: 128-candidate tiles, 2 waves, 32 KiB LDS (many workgroups per CU); WM = 2: 256 / 4 / 48 KiB
__global__ __launch_bounds__(128 * WM, 2) void topk_gmax_kernel(const unsigned short* __restrict__ pool,
                                                                const float* __restrict__ pinv, long rows, int dim,
                                                                const unsigned short* __restrict__ queries, int nq,
                                                                long rows_per_slice, float* __restrict__ gmax,
                                                                long ngroups) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int CT = 128 * WM;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int slice = blockIdx.x, q0 = blockIdx.y * TK_QT;
    const long r_begin = (long)slice * rows_per_slice;
    const long r_end = min(rows, r_begin + rows_per_slice);
    const int wm = (w >> 1) * 128, wn = (w & 1) * 64;
    const int li = lane & 15, lg = lane >> 4;
    for (long n0 = r_begin; n0 < r_end; n0 += CT) {
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        glds_mainloop<ElemF16, false, false, WM, 2, 32>(pool, dim, (int)r_end, queries, dim, nq, (int)n0, q0, 0, dim, lds, acc);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long n = n0 + wm + i * 16 + li;
            const bool nok = n < r_end;
            const float iv = nok ? pinv[n] : 0.f;
            const long group = (n0 + wm + i * 16) >> 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4_t v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = nok ? acc[i][j][r] * iv : -INFINITY;
                    x = fmaxf(x, __shfl_xor(x, 1, 64));     // max over the 16 candidate rows of the group (lanes li)
                    x = fmaxf(x, __shfl_xor(x, 2, 64));
                    x = fmaxf(x, __shfl_xor(x, 4, 64));
                    x = fmaxf(x, __shfl_xor(x, 8, 64));
                    v[r] = x;
                }
                if (li == 0 && group < ngroups) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = q0 + wn + j * 16 + 4 * lg + r;
                        if (q < nq) gmax[(long)q * ngroups + group] = v[r];
                    }
                }
            }
        }
    }
}

template <int BS, bool REG = false>   // threads per query (256: many queries; 1024: <= 64 queries)
__global__ __launch_bounds__(BS) void topk_gsel_kernel(const float* __restrict__ gmax, long ngroups, long rows, int nq,
                                                        int kc, int gcap, int* __restrict__ cand_idx) {
    const int q = blockIdx.x;
    gsel_body<BS, REG>(gmax + (long)q * ngroups, ngroups, rows, kc, gcap, cand_idx + (long)q * gcap * TK_G, [] {});
}

static void coarse_plan(int nq, long rows, int* nqt, int* nslices, long* rows_per_slice) {
    *nqt = (nq + TK_QT - 1) / TK_QT;
    long tiles = (rows + TK_CT - 1) / TK_CT;
    long want = (768 + *nqt - 1) / *nqt;        // aim at ~768 workgroups (3 per CU)
    if (want < 1) want = 1;
    if (want > tiles) want = tiles;
    long tps = (tiles + want - 1) / want;       // candidate tiles per slice
    if (tps < 1) tps = 1;
    *rows_per_slice = tps * TK_CT;
    *nslices = (int)((rows + *rows_per_slice - 1) / *rows_per_slice);
}

extern "C" int32_t uniir_topk_ncand(int32_t nq, int32_t kc) {
    return nq <= TK_GPATH_MAXQ ? TK_GMULT * kc * TK_G : kc;
}

struct ElemF16Direct {   // operands in the order the main loop hands them over (B fragment, A fragment) -> mfma(A, B)
    static DEVINL f32x4_t mfma(u32x4_t b, u32x4_t a, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};


// -------------------------------------------------------------------------------------------------------------
// Group-max scan for many queries (129 .. 1024 per call: the MFMA-bound regime): one 256-candidate x 256-query tile per
// workgroup on the ping-pong LDS-DMA loop of the training GEMMs (gemm_core_pp.h, fp16 operands, K = dim), group maxima
// straight from the accumulators.  Accumulator map: acc[4h+i][2h'+j][r] = candidate 128h + 64wr + 16i + (lane & 15),
// query 128h' + 32wc + 16j + 4 (lane >> 4) + r.
// HALFN: <= 128 queries, the second half of the query tile is padding -> half the MFMA work (gemm_core_pp.h); AUXA: cache policy
// of the pool stream (2 = nt: every pool row is read once per sweep)
template <bool HALFN, int AUXA>
__global__ __launch_bounds__(512, 2) void topk_gmax_pp_kernel(const unsigned short* __restrict__ pool,
                                                              const float* __restrict__ pinv, long rows, int dim,
                                                              const unsigned short* __restrict__ queries, int nq,
                                                              float* __restrict__ gmax, long ngroups, int tiles_q) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = id / tiles_q, qt = id - ct * tiles_q;       // consecutive workgroups share the candidate tile
    const int m0 = ct * 256, q0 = qt * 256;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // un-swapped MFMA operands here: D[row = 4 (lane >> 4) + r -> candidate][col = lane & 15 -> query], so the maximum over
    // the 16 candidates of a group is 3 in-lane max + 2 cross-row exchanges per 16x16 tile (instead of 16 DPP steps)
    glds_mainloop_pp<ElemF16Direct, false, false, false, HALFN, AUXA>(pool, dim, (int)rows, queries, dim, nq, m0, q0, 0, dim, lds, acc);
    // group maxima -> LDS image [256 queries][16 groups] (the ring is idle now), then 32-B runs per query to gmax
    float* stage = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rl = (i >> 2) * 128 + (w >> 2) * 64 + (i & 3) * 16;     // first candidate row of this accumulator tile
        const long n0r = (long)m0 + rl + 4 * lg;                          // this lane's 4 candidates
        f32x4_t iv = {0.f, 0.f, 0.f, 0.f};
        if (n0r + 3 < rows) iv = *reinterpret_cast<const f32x4_t*>(pinv + n0r);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n0r + r < rows) iv[r] = pinv[n0r + r];
        }
        const bool full = n0r + 3 < rows;
#pragma unroll
        for (int j = 0; j < (HALFN ? 2 : 4); ++j) {
            const int ql = (j >> 1) * 128 + (w & 3) * 32 + (j & 1) * 16 + li;
            const f32x4_t v = acc[i][j] * iv;
            float x;
            if (full) x = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            else {
                x = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n0r + r < rows) x = fmaxf(x, v[r]);
            }
            x = group_max(x);
            if (lg == 0) stage[ql * 16 + (rl >> 4)] = x;
        }
    }
    __syncthreads();
    {
        const int ql = threadIdx.x >> 1, half = threadIdx.x & 1;
        const int q = q0 + ql;
        const long g0 = ((long)m0 >> 4) + half * 8;
        if (q < nq) {
            float* dst = gmax + (long)q * ngroups + g0;
            const float* src = stage + ql * 16 + half * 8;
#pragma unroll
            for (int g = 0; g < 8; ++g)
                if (g0 + g < ngroups) dst[g] = src[g];
        }
    }
}

// -------------------------------------------------------------------------------------------------------------
// Streaming group-max scan for nq <= 64 (the interactive regime: HBM-bound).  The queries
// (<= 64 x dim fp16, <= 96 KiB) live in LDS for the whole kernel; every wave streams its own 16-candidate tiles
// straight from HBM into registers in the MFMA A-fragment layout (lane -> row lane & 15, 16 B at k = 32 s + 8 (lane >> 4))
// -- no LDS staging of the pool, no barrier in the loop, two tiles of loads (2 x dim/32 x 1 KiB per wave) in flight
// while the previous tile's MFMAs run: 8 waves x ~36 KiB keep ~290 KiB per CU in flight, enough for the HBM
// latency-bandwidth product.  Output: the same gmax[q][group] matrix as topk_gmax_kernel (group = 16 candidates).
#define TKS_CH 8     // tiles (groups) per work item: one 32-B run of gmax per query
#ifndef TKS_LOAD
#define TKS_LOAD(p) (*(p))   // plain loads measured 8 % faster than nontemporal here
#endif
template <int NK>    // dim = 32 * NK
__global__ __launch_bounds__(512) void topk_stream_kernel(const unsigned short* __restrict__ pool,
                                                          const float* __restrict__ pinv, long rows,
                                                          const unsigned short* __restrict__ queries, int nq,
                                                          float* __restrict__ gmax, long ngroups, long nchunks) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int DIM = 32 * NK, QS = DIM * 2 + 16;    // padded query row: 16 rows of a B fragment hit 16 different bank slots
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    char* qs = lds;
    float* stage = reinterpret_cast<float*>(lds + 64 * QS) + w * (64 * TKS_CH);
    for (int c = tid; c < 64 * (DIM / 8); c += 512) {
        const int q = c / (DIM / 8), kc = c - q * (DIM / 8);
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (q < nq) v = *reinterpret_cast<const u32x4_t*>(queries + (long)q * DIM + kc * 8);
        *reinterpret_cast<u32x4_t*>(qs + q * QS + kc * 16) = v;
    }
    __syncthreads();
    const char* qb = qs + li * QS + lg * 16;     // + j * 16 * QS + s * 64
    // Every wave owns ONE contiguous range of tiles, sizes differing by at most one tile (43 750 tiles over 2 048 waves: 21 or 22
    // each).  Strided chunks of 8 tiles left 2.67 chunks per wave, i.e. a third round that only 2/3 of the waves took part in
    // (89 % balance), and drained the two-tile load pipeline at every chunk start; here the pipeline runs through the whole range
    // and only the 32-byte gmax runs are flushed every TKS_CH tiles.
    const long gw = (long)blockIdx.x * 8 + w, nw = (long)gridDim.x * 8;
    const long lo = gw * ngroups / nw, hi = (gw + 1) * ngroups / nw;
    u32x4_t a[2][NK];
    auto load_tile = [&](long tile, u32x4_t (&dst)[NK]) {
        long row = tile * 16 + li;
        if (row > rows - 1) row = rows - 1;
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(pool + row * DIM) + lg;
#pragma unroll
        for (int s = 0; s < NK; ++s) dst[s] = TKS_LOAD(src + 4 * s);
    };
    auto do_tile = [&](long tile, int t, const u32x4_t (&af)[NK]) {
        f32x4_t acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NK; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4_t b = *reinterpret_cast<const u32x4_t*>(qb + j * 16 * QS + s * 64);
                acc[j] = ElemF16::mfma(af[s], b, acc[j]);
            }
        }
        // D: lane -> query j*16 + li, candidates 4 lg + r of the tile
        const long r0 = tile * 16 + 4 * lg;
        f32x4_t iv = {0.f, 0.f, 0.f, 0.f};
        if (r0 + 3 < rows) iv = *reinterpret_cast<const f32x4_t*>(pinv + r0);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (r0 + r < rows) iv[r] = pinv[r0 + r];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, (r0 + r < rows) ? acc[j][r] * iv[r] : -INFINITY);
            m = group_max(m);
            if (lg == 0) stage[(j * 16 + li) * TKS_CH + t] = m;
        }
    };
    auto flush = [&](long g0, int n) {       // lane q writes its n <= TKS_CH consecutive groups starting at g0
        if (lane < nq) {                      // (wave-private staging: program order suffices)
            float* dst = gmax + (long)lane * ngroups + g0;
            for (int g = 0; g < n; ++g) dst[g] = stage[lane * TKS_CH + g];
        }
    };
    if (lo >= hi) return;
    load_tile(lo, a[0]);
    long g0 = lo;
    int slot = 0;
    for (long ta = lo; ta + 1 < hi; ta += 2) {              // two tiles per trip: the register buffers keep static indices
        load_tile(ta + 1, a[1]);
        do_tile(ta, slot, a[0]);
        load_tile(ta + 2 < hi ? ta + 2 : hi - 1, a[0]);      // always issued (the last one re-reads a tile): no branch in the loop
        do_tile(ta + 1, slot + 1, a[1]);
        slot += 2;
        if (slot == TKS_CH) {
            flush(g0, TKS_CH);
            g0 += TKS_CH;
            slot = 0;
        }
    }
    if ((hi - lo) & 1) {                                      // odd range: its last tile is the one a[0] holds
        do_tile(hi - 1, slot, a[0]);
        ++slot;
    }
    if (slot) flush(g0, slot);
}

// -------------------------------------------------------------------------------------------------------------
// Streaming group-max scan, second generation (nq <= 64, dim = 768): THE QUERIES LIVE IN REGISTERS, THE POOL STREAMS THROUGH LDS.
//   * One wave per SIMD (256-thread workgroups, one per CU).  Every wave holds the MFMA B fragments of all 64 queries for the whole
//     kernel: 4 query tiles x 24 k-steps x 4 registers = 384 of its 512 VGPRs -- no LDS reads for the queries at all (the first
//     generation re-read 96 KiB of query fragments from LDS per 24.5-KiB pool tile).
//   * The pool goes HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): 8 consecutive lanes fetch one whole 128-byte line, i.e.
//     every request is a full line (the register-direct A-fragment loads of the first generation were 64-byte pieces at a
//     1536-byte stride, two requests per line).  Each wave owns a private ring of three half-tiles (16 rows x 384 dims = 12 KiB
//     = 12 DMA instructions); two half-tiles (24 KiB per wave, 96 KiB per CU) are in flight while the third is multiplied.  The
//     ring is wave-private, so the only ordering needed is the issuing wave's own counted vmcnt: no barrier anywhere.
//   * A fragments come back from LDS with ds_read_b128; the swizzle g(row) = (row >> 1) & 7 is applied on the SOURCE side (which
//     16-byte chunk of its line a lane fetches) and undone in the read address: conflict-free for the 4 x 16-lane service groups of
//     ds_read_b128 (checked exhaustively in tools/r3/check_stream2_layout.py).
//   * The 16 inverse norms of a tile ride the same DMA queue (one dword LDS-DMA per tile), so vmcnt counting stays exact.
// LDS image of a half-tile: instruction j (0..11) writes 1 KiB = [8 rows][8 chunks]: rows 8 (j & 1) + (lane >> 3), column block
// j >> 1 (128 bytes), position lane & 7 holds chunk (lane & 7) ^ g(row).
// Round 4: templated on NK = dim / 32 (24: the 768-wide M-BEIR pools of the large models; 16: the 512-wide pools of the CLIP base
// models, which used to fall back to the first-generation scan): a row is NK * 64 bytes, a half-tile 16 rows x NK * 16 dims =
// NK / 2 DMA instructions and NK / 2 k-steps, the queries take 4 * NK * 4 registers.
#define TKR_HALF_BYTES 12288          // dim 768 (the shared-ring scans below are 768-only)
template <int NK>
struct Tkr {
    static constexpr int ROW = NK * 64;                  // bytes per pool row
    static constexpr int NH = NK / 2;                    // DMA instructions = k-steps per half-tile
    static constexpr int HALF = NK * 512;                // bytes per half-tile (16 rows x ROW / 2)
    static constexpr int PINV_OFF = 3 * HALF;
    static constexpr int STAGE_OFF = PINV_OFF + 512;
    static constexpr int WAVE_LDS = 3 * HALF + 512 + 2048;
};
// group-max stores of the streaming scans: -DUNIIR_GMAX_NT=1 builds them as non-temporal stores.  MEASURED (round 3, same box, whole
// search): 64 queries 0.2233 / 0.2059 ms (default) vs 0.2288 / 0.2095 (nt); 128 queries 0.2360 / 0.2236 vs 0.2652 / 0.2450 -- the
// 32-byte runs want the L2's write combining; default stays.
#if defined(UNIIR_GMAX_NT) && UNIIR_GMAX_NT
#define TK_GST(ptr, v) __builtin_nontemporal_store((v), (ptr))
#else
#define TK_GST(ptr, v) (*(ptr) = (v))
#endif
// v_max_f32 without the canonicalising self-max hipcc puts in front of fmaxf on values it cannot see the origin of (swap results)
DEVINL float tk5_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
struct TkrState {
    __amdgpu_buffer_rsrc_t rp, ri;     // pool rows (bounds = rows * 1536 bytes: rows past the end read as zeros), inverse norms
    unsigned vb0, vb1, vpi;            // per-lane source byte offsets inside a tile (even / odd DMA instruction), inverse norms
    unsigned la0, la1;                 // per-lane LDS byte offsets of the A fragment reads (even / odd k-step)
    unsigned lbase;
    char* my;
};
template <int NK, int HALF, int AUX>      // AUX: cache policy of the pool stream (0 default, 2 = nt: read-once data)
DEVINL void tkr_issue(const TkrState& st, long tile, int slot) {
    using K = Tkr<NK>;
    // the position inside the row goes into the SCALAR offset: it is excluded from the bounds check (so the check is exactly "is
    // this row inside the shard": voffset = first line of the row) and, unlike the instruction's immediate offset, is not added
    // to the LDS address as well
    const unsigned tb = (unsigned)(tile * (16 * K::ROW));
    const unsigned v0 = st.vb0 + tb, v1 = st.vb1 + tb;
    char* dst = st.my + slot * K::HALF;
#define TKR_DMA(J)                                                                                                       \
    if ((J) < K::NH)                                                                                                     \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rp, (void __attribute__((address_space(3)))*)(dst + (J) * 1024), 16, \
                                                 ((J) & 1) ? v1 : v0, ((J) >> 1) * 128 + HALF * (K::ROW / 2), 0, AUX);
    TKR_DMA(0) TKR_DMA(1) TKR_DMA(2) TKR_DMA(3) TKR_DMA(4) TKR_DMA(5) TKR_DMA(6) TKR_DMA(7) TKR_DMA(8) TKR_DMA(9) TKR_DMA(10) TKR_DMA(11)
#undef TKR_DMA
    if (HALF == 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(st.ri, (void __attribute__((address_space(3)))*)(st.my + K::PINV_OFF + (int)(tile & 1) * 256),
                                                 4, st.vpi + (unsigned)(tile * 64), 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}
template <int NK, int HALF>
DEVINL void tkr_process(const TkrState& st, int slot, const u32x4_t (&qf)[4][NK], f32x4_t (&acc)[4]) {
    using K = Tkr<NK>;
    const unsigned sb = st.lbase + slot * K::HALF;
    const unsigned a0 = sb + st.la0, a1 = sb + st.la1;
    u32x4_t a[K::NH];
    a[0] = asm_ds_read_b128<0 * 2048>(a0);  a[1] = asm_ds_read_b128<0 * 2048>(a1);
    a[2] = asm_ds_read_b128<1 * 2048>(a0);  a[3] = asm_ds_read_b128<1 * 2048>(a1);
    a[4] = asm_ds_read_b128<2 * 2048>(a0);  a[5] = asm_ds_read_b128<2 * 2048>(a1);
    a[6] = asm_ds_read_b128<3 * 2048>(a0);  a[7] = asm_ds_read_b128<3 * 2048>(a1);
    if constexpr (K::NH > 8) {
        a[8] = asm_ds_read_b128<4 * 2048>(a0);  a[9] = asm_ds_read_b128<4 * 2048>(a1);
        a[10] = asm_ds_read_b128<5 * 2048>(a0); a[11] = asm_ds_read_b128<5 * 2048>(a1);
    }
    __builtin_amdgcn_sched_barrier(0);
#define TKR_STEP(SH)                                                                                  \
    if constexpr ((SH) < K::NH) {                                                                     \
        asm_wait_lgkm<K::NH - 1 - (SH) < 0 ? 0 : K::NH - 1 - (SH)>();                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[j] = ElemF16::mfma(a[SH], qf[j][K::NH * HALF + (SH)], acc[j]); \
    }
    TKR_STEP(0) TKR_STEP(1) TKR_STEP(2) TKR_STEP(3) TKR_STEP(4) TKR_STEP(5)
    TKR_STEP(6) TKR_STEP(7) TKR_STEP(8) TKR_STEP(9) TKR_STEP(10) TKR_STEP(11)
#undef TKR_STEP
    __builtin_amdgcn_sched_barrier(0);
}
// everything a streaming-scan wave sets up before its first pool tile: DMA descriptors / per-lane offsets, and the query fragments
// (all 64 queries, 96 x 16 bytes per lane) staged through LDS.  Contains two workgroup barriers.
template <int NK, int WAVE_LDS>
DEVINL void tkr_prepare(TkrState& st, u32x4_t (&qf)[4][NK], char* lds, const unsigned short* __restrict__ pool,
                        const float* __restrict__ pinv, long rows, const unsigned short* __restrict__ queries, int nq, int w,
                        int lane) {
    constexpr unsigned ROW = NK * 64;
    const int li = lane & 15, lg = lane >> 4;
    st.my = lds + w * WAVE_LDS;
    st.lbase = lds_addr32(st.my);
    st.rp = __builtin_amdgcn_make_buffer_rsrc((void*)pool, 0, (int)(rows * ROW), 0x00020000);
    st.ri = __builtin_amdgcn_make_buffer_rsrc((void*)pinv, 0, (int)(rows * 4), 0x00020000);
    {
        const int r8 = lane >> 3, c8 = lane & 7;
        const int row0 = r8, row1 = 8 + r8;                                   // even / odd DMA instruction
        st.vb0 = (unsigned)(row0 * ROW + ((c8 ^ ((row0 >> 1) & 7)) << 4));
        st.vb1 = (unsigned)(row1 * ROW + ((c8 ^ ((row1 >> 1) & 7)) << 4));
        st.vpi = (unsigned)((lane & 15) * 4);
        const int g = (li >> 1) & 7;
        st.la0 = (unsigned)((li >> 3) * 1024 + (li & 7) * 128 + (((0 + lg) ^ g) << 4));
        st.la1 = (unsigned)((li >> 3) * 1024 + (li & 7) * 128 + (((4 + lg) ^ g) << 4));
    }
    // The query fragments: 96 x 16 bytes per lane.  As plain global loads hipcc serialises them (load, wait, move to an AGPR, 96
    // times: ~45 us per wave, measured as 244 vs 202 us between 64 and 16 queries' worth of ... the same loads).  So the 64 x 1536
    // bytes of queries are first copied into LDS by LDS-DMA (the rings are not in use yet; 16-byte chunk index ^= (query & 15) on
    // the source side, so that the 16 queries x 4 chunks of a fragment read are conflict-free), then every wave reads all of them.
    {
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)queries, 0, nq * (int)ROW, 0x00020000);
        // 98 304 bytes = 96 DMA instructions of 1 KiB over 4 waves: instruction i covers LDS bytes [1024 i, +1024): query i * 2 / 3 ...
        // lane -> LDS byte b = 1024 i + 16 lane -> query b / 1536, position p = (b % 1536) / 16, source chunk p ^ (query & 15)
        // (the XOR stays inside the row: 96 chunks = 6 blocks of 16)
#pragma unroll
        for (int ii = 0; ii < NK; ++ii) {
            const int i = ii * 4 + w;
            const unsigned b = 1024u * i + 16u * lane;
            const unsigned qq = b / ROW, p = (b % ROW) >> 4;
            const unsigned src = qq * ROW + (((p & ~15u) | ((p ^ qq) & 15u)) << 4);        // queries >= nq: out of bounds -> zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (void __attribute__((address_space(3)))*)(lds + 1024 * i), 16, src, 0, 0, 0);
        }
        __syncthreads();                 // hipcc drains the LDS-DMA in front of the barrier
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = j * 16 + li;
#pragma unroll
            for (int s = 0; s < NK; ++s) {
                const int c = 4 * s + lg;
                qf[j][s] = *reinterpret_cast<const u32x4_t*>(lds + q * ROW + (((c & ~15) | ((c ^ q) & 15)) << 4));
            }
        }
        __syncthreads();                 // every wave has its fragments: the rings may be filled
    }
}

template <int NK, int AUX>
__global__ __launch_bounds__(256, 1) void topk_stream2_kernel(const unsigned short* __restrict__ pool,
                                                             const float* __restrict__ pinv, long rows,
                                                             const unsigned short* __restrict__ queries, int nq,
                                                             float* __restrict__ gmax, long ngroups,
                                                             float* __restrict__ wmax) {      // optional [nq][waves]: per-wave maxima
    using K = Tkr<NK>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // one contiguous range of tiles (groups of 16 rows) per wave, sizes differing by at most one
    const long gw = (long)blockIdx.x * 4 + w, nw = (long)gridDim.x * 4;
    const long lo = gw * ngroups / nw, hi = (gw + 1) * ngroups / nw;      // never empty: the launcher asks for >= 2048 groups
    TkrState st;
    u32x4_t qf[4][NK];
    tkr_prepare<NK, K::WAVE_LDS>(st, qf, lds, pool, pinv, rows, queries, nq, w, lane);
    tkr_issue<NK, 0, AUX>(st, lo, 0);
    tkr_issue<NK, 1, AUX>(st, lo, 1);
    f32x4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float wave_best = -INFINITY;       // maximum over this wave's whole range, per query (gsel_hier starts from these)
    const unsigned stg = st.lbase + K::STAGE_OFF + lane * 32;
    const unsigned qoff = (unsigned)(((long)lane * ngroups) & 3);     // 16-byte alignment of the stores (octets start on it)
    auto finish_tile = [&](long tile) {
        // D: lane -> query j * 16 + li, candidates 4 lg + r of the tile
        const u32x4_t ivb = asm_ds_read_b128<0>(st.lbase + K::PINV_OFF + (unsigned)(tile & 1) * 256 + lg * 16);
        asm_wait_lgkm<0>();
        const f32x4_t iv = __builtin_bit_cast(f32x4_t, ivb);
        float m[4];
        if (tile * 16 + 16 <= rows) {                              // every tile but a ragged last one: no row masks
#pragma unroll
            for (int j = 0; j < 4; ++j)
                m[j] = tk5_max(tk5_max(acc[j][0] * iv[0], acc[j][1] * iv[1]), tk5_max(acc[j][2] * iv[2], acc[j][3] * iv[3]));
        } else {
            const long r0 = tile * 16 + 4 * lg;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) x = fmaxf(x, (r0 + r < rows) ? acc[j][r] * iv[r] : -INFINITY);
                m[j] = x;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // the four row reductions as one reduce-scatter (see topk_stream5_kernel): lane -> the maximum of query lg * 16 + li = lane
        const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m[0]), __float_as_uint(m[2]), false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m[1]), __float_as_uint(m[3]), false, false);
        const float r02 = tk5_max(__uint_as_float(s02[0]), __uint_as_float(s02[1]));
        const float r13 = tk5_max(__uint_as_float(s13[0]), __uint_as_float(s13[1]));
        const auto sq = __builtin_amdgcn_permlane16_swap(__float_as_uint(r02), __float_as_uint(r13), false, false);
        const float mine = tk5_max(__uint_as_float(sq[0]), __uint_as_float(sq[1]));
        wave_best = fmaxf(wave_best, mine);
        // Eight consecutive group maxima of a query leave as two 16-byte stores to one 32-byte run (one 4-byte store per lane and
        // tile is 64 scattered requests per tile, 2.8 M per sweep).  The lane's 32 bytes of LDS serve as an indexed register file
        // (asm accesses: the compiler must not order them against the LDS-DMA stream); the octet phase is per lane, so that the
        // stores are 16-byte aligned whatever (query * ngroups) % 4 is.
        const unsigned k = ((unsigned)tile + qoff) & 7u;
        asm volatile("ds_write_b32 %0, %1" ::"v"(stg + k * 4u), "v"(mine) : "memory");
        if (k == 7u || tile == hi - 1) {
            const u32x4_t s0 = asm_ds_read_b128<0>(stg), s1 = asm_ds_read_b128<16>(stg);
            asm_wait_lgkm<0>();
            const f32x4_t v0 = __builtin_bit_cast(f32x4_t, s0), v1 = __builtin_bit_cast(f32x4_t, s1);
            const long g0 = tile - k;                        // first group of this lane's octet
            if (lane < nq) {
                float* dst = gmax + (long)lane * ngroups + g0;
                if (k == 7u && g0 >= lo) {
                    TK_GST(reinterpret_cast<f32x4_t*>(dst), v0);
                    TK_GST(reinterpret_cast<f32x4_t*>(dst + 4), v1);
                } else {                                     // head of the wave's range / end of the range: the groups this wave computed
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i <= (int)k && g0 + i >= lo) dst[i] = i < 4 ? v0[i] : v1[i - 4];
                }
            }
        }
    };
    // half-tile h = 2 (tile - lo) + half lives in ring slot h % 3; two half-tiles stay in flight (NK / 2 + NK / 2 + 1 DMA instructions:
    // the inverse norms travel with the first half), so every wait is vmcnt(NK + 1).  The group-max stores also count in vmcnt: they can
    // only make a wait stricter, never looser.
    int slot = 0;
    long t = lo;
    for (; t + 1 < hi; ++t) {
        tkr_issue<NK, 0, AUX>(st, t + 1, slot == 0 ? 2 : slot - 1);       // h + 2 -> slot (h + 2) % 3
        tkr_wait_vm<NK + 1>();
        tkr_process<NK, 0>(st, slot, qf, acc);
        slot = slot == 2 ? 0 : slot + 1;
        tkr_issue<NK, 1, AUX>(st, t + 1, slot == 0 ? 2 : slot - 1);
        tkr_wait_vm<NK + 1>();
        tkr_process<NK, 1>(st, slot, qf, acc);
        slot = slot == 2 ? 0 : slot + 1;
        finish_tile(t);
    }
    tkr_wait_vm<NK / 2>();
    tkr_process<NK, 0>(st, slot, qf, acc);
    slot = slot == 2 ? 0 : slot + 1;
    tkr_wait_vm<0>();
    tkr_process<NK, 1>(st, slot, qf, acc);
    finish_tile(t);
    if (wmax && lane < nq) wmax[(long)lane * nw + gw] = wave_best;
}

// -------------------------------------------------------------------------------------------------------------
// Streaming scan for 65 .. 256 queries (topk_stream5_kernel): the stream2 kernel with the pool ring SHARED by the QW waves of a
// workgroup.  Every wave still keeps 64 queries in registers (wave w: queries 64 w ..), so one half-tile of the pool that lands in
// LDS is multiplied with 64 QW queries: the pool is read from HBM once for up to 256 queries, and no query fragment is ever re-read
// from LDS (the ping-pong GEMM scan re-reads 64 KiB of them per 256-row pool tile).
//   QW = 2: 128-thread workgroups, two per CU (2 x 512-register waves each); QW = 4: 256-thread workgroups, one per CU.
// A half-tile (12 DMA instructions) is fetched by all QW waves together (12 / QW instructions each), so the ring needs the
// workgroup barrier the private rings did without.  The inverse norms travel with the first half of a tile, one private copy per
// wave, a ring of 8 tiles.  Past the end of the range the loop keeps issuing DMAs whose offset is out of bounds (no memory
// traffic, zeros into a free slot): the vmcnt arithmetic stays exact without a tail loop.
// The per-"wave" maxima for gsel_hier are written per VIRTUAL wave: the workgroup's range is the union of VR = 1024 / workgroups
// consecutive ranges of the same  v * ngroups / NV  partition the 64-query kernel uses.
// (A first version -- topk_stream4_kernel, round 3 -- waited, passed the barrier, issued the next DMA and only then read and
// multiplied the half-tile: 128 queries 203 us, but 256 queries 332 us vs 322 us for the GEMM-shaped scan, because one wave per
// SIMD cannot hide its own LDS latency that way.  The rolling-register loop below replaced it: 181 / ~275 us.)
// Tk4 only carries the DMA split (PER) the issue helper needs.
template <int QW>
struct Tk4 {
    static constexpr int PER = 12 / QW;          // DMA instructions per wave and half-tile
};
struct Tk4State {
    __amdgpu_buffer_rsrc_t rp, ri;
    unsigned ve, vo, vpi;              // source offsets of this wave's even / odd instruction of a half-tile; inverse norms
    unsigned so0;                      // scalar offset of this wave's first instruction (HALF 0)
    unsigned la0, la1;
    unsigned rbase, pbase;             // LDS addresses: ring, this wave's private area
    char* ring;
    char* priv;
    int w;
};
#define TK4_DUMMY 0x80000000u          // >= every buffer bound used here (shards are < 2^31 bytes): the DMA fetches nothing
template <int QW, int HALF, int AUX>
DEVINL void tk4_issue(const Tk4State& st, long tile, bool live, int slot) {
    constexpr int PER = Tk4<QW>::PER;
    const unsigned tb = live ? (unsigned)(tile * (16 * 1536)) : TK4_DUMMY;
    const unsigned ve = st.ve + tb, vo = st.vo + tb;
    char* dst = st.ring + slot * TKR_HALF_BYTES + st.w * (PER * 1024);
    // instruction J = w PER + i: LDS piece J, source line (J >> 1) * 128 + HALF * 768 of rows 8 (J & 1) ...
#define TK4_DMA(I)                                                                                                       \
    if ((I) < PER)                                                                                                       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rp, (void __attribute__((address_space(3)))*)(dst + (I) * 1024), 16,  \
                                                 ((I) & 1) ? vo : ve, (((st.w * PER + (I)) >> 1) * 128) + HALF * 768, 0, AUX);
    TK4_DMA(0) TK4_DMA(1) TK4_DMA(2) TK4_DMA(3) TK4_DMA(4) TK4_DMA(5)
    TK4_DMA(6) TK4_DMA(7) TK4_DMA(8) TK4_DMA(9) TK4_DMA(10) TK4_DMA(11)
#undef TK4_DMA
    if (HALF == 0) {
        if ((threadIdx.x & 63) < 16)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(st.ri, (void __attribute__((address_space(3)))*)(st.priv + (int)(tile & 7) * 64), 4,
                                                     live ? st.vpi + (unsigned)(tile * 64) : TK4_DUMMY, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}
// c += a x q (v_mfma_f32_16x16x32_f16) with the query fragment in an AGPR (IN_A) or a VGPR
template <bool IN_A>
DEVINL void tk5_mfma(const u32x4_t& a, const u32x4_t& q, f32x4_t& c) {
    if constexpr (IN_A) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(q));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(q));
}
// The shared-ring scan with ROLLING fragment registers (see the loop): a ring of NS half-tiles, D = NS - 1 of them in flight, ONE
// barrier per half-tile, in the middle of its MFMAs:
//     [reads of k-steps 6..11 of h] [MFMAs 0..5 of h] | wait: MY part of h + 1 has landed | s_barrier: everybody's has, and
//     everybody is done with h - 1 | issue h + D into the slot of h - 1 | [reads of k-steps 0..5 of h + 1] [MFMAs 6..11 of h]
// D is even: at the wait h + 2 .. h + D - 1 may stay in flight = D - 2 half-tiles with (D - 2) / 2 inverse-norm DMAs -- a constant.
template <int QW>
struct Tk5 {
    static constexpr int NS = QW == 1 ? 3 : QW == 2 ? 5 : 9;          // QW = 1: four single-wave workgroups per CU, private rings
    static constexpr int D = NS - 1;
    static constexpr int PER = 12 / QW;
    static constexpr int WAIT = (D - 2) * PER + (D - 2) / 2;
    static constexpr int WAIT0 = (D - 1) * PER + (D - 2) / 2;        // prologue: half-tile 0 of 0 .. D - 1
    static constexpr int RING = NS * TKR_HALF_BYTES;
    static constexpr int PRIV = 512 + 2048;
    static constexpr int LDS = RING + QW * PRIV;
    static constexpr int VR = QW;
};
template <int QW, int AUX>
__global__ __launch_bounds__(64 * QW, 1) void topk_stream5_kernel(const unsigned short* __restrict__ pool,
                                                                 const float* __restrict__ pinv, long rows,
                                                                 const unsigned short* __restrict__ queries, int nq,
                                                                 float* __restrict__ gmax, long ngroups,
                                                                 float* __restrict__ wmax) {     // optional [nq][4 ncu] maxima
    using C = Tk5<QW>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // 32-bit range arithmetic: ngroups < 2^31 / 1536 / 16 and nv <= 1024 (launcher), so v * ngroups < 2^27
    const unsigned nv = gridDim.x * C::VR, ng32 = (unsigned)ngroups;
    const unsigned v0 = blockIdx.x * C::VR;
    const long lo = v0 * ng32 / nv, hi = (v0 + C::VR) * ng32 / nv;             // >= VR tiles (the launcher asks for >= 2048 groups)
    Tk4State st;
    st.w = w;
    st.ring = lds;
    st.priv = lds + C::RING + w * C::PRIV;
    st.rbase = lds_addr32(st.ring);
    st.pbase = lds_addr32(st.priv);
    st.rp = __builtin_amdgcn_make_buffer_rsrc((void*)pool, 0, (int)(rows * 1536), 0x00020000);
    st.ri = __builtin_amdgcn_make_buffer_rsrc((void*)pinv, 0, (int)(rows * 4), 0x00020000);
    {
        const int r8 = lane >> 3, c8 = lane & 7;
        const int row0 = r8, row1 = 8 + r8;
        const unsigned vb0 = (unsigned)(row0 * 1536 + ((c8 ^ ((row0 >> 1) & 7)) << 4));
        const unsigned vb1 = (unsigned)(row1 * 1536 + ((c8 ^ ((row1 >> 1) & 7)) << 4));
        const bool odd_first = ((w * C::PER) & 1) != 0;          // QW = 4: waves 1 and 3 start on an odd instruction
        st.ve = odd_first ? vb1 : vb0;
        st.vo = odd_first ? vb0 : vb1;
        st.vpi = (unsigned)((lane & 15) * 4);
        const int g = (li >> 1) & 7;
        st.la0 = (unsigned)((li >> 3) * 1024 + (li & 7) * 128 + (((0 + lg) ^ g) << 4));
        st.la1 = (unsigned)((li >> 3) * 1024 + (li & 7) * 128 + (((4 + lg) ^ g) << 4));
    }
    // this wave's 64 queries -> registers, 16 at a time through a private 24-KiB piece of the (still unused) ring
    u32x4_t qf[4][24];
    {
        const unsigned qb = 64u * w;                               // queries >= nq are out of bounds: zeros
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)queries, 0, nq * 1536, 0x00020000);
        char* mine = lds + w * 24576;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const unsigned b = 1024u * i + 16u * lane;
                const unsigned qq = b / 1536u, p = (b % 1536u) >> 4;                // query 16 j + qq of this wave, chunk position p
                const unsigned src = (qb + 16u * j + qq) * 1536u + (((p & ~15u) | ((p ^ qq) & 15u)) << 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (void __attribute__((address_space(3)))*)(mine + 1024 * i), 16, src, 0, 0, 0);
            }
            tkr_wait_vm<0>();                                      // (hipcc does not order plain LDS loads behind LDS-DMA by itself)
            // plain LDS loads: the compiler knows when their results are valid (they move on into AGPRs)
#pragma unroll
            for (int s = 0; s < 24; ++s) {
                const int c = 4 * s + lg;
                qf[j][s] = *reinterpret_cast<const u32x4_t*>(mine + li * 1536 + (((c & ~15) | ((c ^ li) & 15)) << 4));
            }
            asm_wait_lgkm<0>();                                    // ... and the next round's DMA must not overtake them
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                                  // every wave is done with its staging piece: the ring may fill
    __builtin_amdgcn_sched_barrier(0);
    const long nh = 2 * (hi - lo);
    // half-tiles 0 .. D - 1 (D even: whole tiles)
#pragma unroll
    for (int h = 0; h < C::D; ++h) {
        const long tile = lo + (h >> 1);
        if (h & 1) tk4_issue<QW, 1, AUX>(st, tile, h < nh, h);
        else tk4_issue<QW, 0, AUX>(st, tile, h < nh, h);
    }
    // two accumulator sets: the tile being multiplied and the previous one, whose maxima are taken UNDER the current tile's MFMAs
    f32x4_t accA[4], accB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) accA[j] = accB[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int qi = 64 * w + lane;                                  // this lane's query in the group-max matrix
    float best = -INFINITY;
    unsigned vcur = v0;
    long vhi = (v0 + 1) * ng32 / nv;
    const unsigned stg = st.pbase + 512 + lane * 32;
    const unsigned qoff = (unsigned)(((long)qi * ngroups) & 3);
    // maxima over the four 16-lane rows for the four query tiles at once, as a reduce-scatter: after the 32-lane swap the lower half
    // of the wave owns tiles 0 / 1, the upper half tiles 2 / 3; after the 16-lane swap row lg owns tile lg, i.e. the lane holds the
    // group maximum of query 16 lg + li = its own query (6 instructions instead of 4 full reductions and a select)
    auto rows_to_mine = [&](const float (&m)[4]) {
        const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m[0]), __float_as_uint(m[2]), false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m[1]), __float_as_uint(m[3]), false, false);
        const float r02 = tk5_max(__uint_as_float(s02[0]), __uint_as_float(s02[1]));
        const float r13 = tk5_max(__uint_as_float(s13[0]), __uint_as_float(s13[1]));
        const auto sq = __builtin_amdgcn_permlane16_swap(__float_as_uint(r02), __float_as_uint(r13), false, false);
        return tk5_max(__uint_as_float(sq[0]), __uint_as_float(sq[1]));
    };
    // what follows a tile's group maximum `mine`: the virtual wave's running maximum, the octet staging, the 32-byte stores
    auto fin_tail = [&](long tile, float mine) {
        best = fmaxf(best, mine);
        if (tile == vhi - 1) {                                     // end of a virtual wave's range
            if (wmax && qi < nq) wmax[(long)qi * nv + vcur] = best;
            best = -INFINITY;
            ++vcur;
            vhi = (vcur + 1) * ng32 / nv;
        }
        const unsigned k = ((unsigned)tile + qoff) & 7u;
        asm volatile("ds_write_b32 %0, %1" ::"v"(stg + k * 4u), "v"(mine) : "memory");
        if (k == 7u || tile == hi - 1) {
            const u32x4_t s0 = asm_ds_read_b128<0>(stg), s1 = asm_ds_read_b128<16>(stg);
            asm_wait_lgkm<0>();
            const f32x4_t v0 = __builtin_bit_cast(f32x4_t, s0), v1 = __builtin_bit_cast(f32x4_t, s1);
            const long g0 = tile - k;
            if (qi < nq) {
                float* dst = gmax + (long)qi * ngroups + g0;
                if (k == 7u && g0 >= lo) {
                    TK_GST(reinterpret_cast<f32x4_t*>(dst), v0);
                    TK_GST(reinterpret_cast<f32x4_t*>(dst + 4), v1);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i <= (int)k && g0 + i >= lo) dst[i] = i < 4 ? v0[i] : v1[i - 4];
                }
            }
        }
    };
    // the whole epilogue of a tile in one piece (the last tile of the range -- nothing left to hide it under; the only one that can
    // be ragged).  The MFMAs are inline asm: the compiler's hazard recogniser does not see them -- their results are read >= 100
    // cycles later, behind this LDS round trip.
    auto fin_full = [&](f32x4_t (&ac)[4], long tile) {
        const u32x4_t ivb = asm_ds_read_b128<0>(st.pbase + (unsigned)(tile & 7) * 64 + lg * 16);
        asm volatile("s_nop 7\n\ts_nop 7");
        asm_wait_lgkm<0>();
        const f32x4_t iv = __builtin_bit_cast(f32x4_t, ivb);
        float m[4];
        if (tile * 16 + 16 <= rows) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                m[j] = tk5_max(tk5_max(ac[j][0] * iv[0], ac[j][1] * iv[1]), tk5_max(ac[j][2] * iv[2], ac[j][3] * iv[3]));
        } else {
            const long r0 = tile * 16 + 4 * lg;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) x = fmaxf(x, (r0 + r < rows) ? ac[j][r] * iv[r] : -INFINITY);
                m[j] = x;
            }
        }
        fin_tail(tile, rows_to_mine(m));
    };
    // Rolling fragment registers: a[0..5] (k-steps 0..5 of a half-tile) are re-loaded for half-tile h + 1 as soon as the MFMAs of
    // half-tile h have issued past them, a[6..11] at the top of h: LDS latency, the DMA issue and the barrier all sit under
    // MFMAs of the same wave (one wave per SIMD: nobody else could hide them).
    u32x4_t a[12];
#define TK5_READ6(F, SB)                                                                                       \
    a[6 * F + 0] = asm_ds_read_b128<(3 * F + 0) * 2048>((SB) + st.la0); a[6 * F + 1] = asm_ds_read_b128<(3 * F + 0) * 2048>((SB) + st.la1); \
    a[6 * F + 2] = asm_ds_read_b128<(3 * F + 1) * 2048>((SB) + st.la0); a[6 * F + 3] = asm_ds_read_b128<(3 * F + 1) * 2048>((SB) + st.la1); \
    a[6 * F + 4] = asm_ds_read_b128<(3 * F + 2) * 2048>((SB) + st.la0); a[6 * F + 5] = asm_ds_read_b128<(3 * F + 2) * 2048>((SB) + st.la1); \
    __builtin_amdgcn_sched_barrier(0);
    // The MFMAs are written out: 64 of the 96 query fragments are pinned to AGPRs and go into the MFMA as its B operand directly
    // (left to itself hipcc parks ~230 registers of them in AGPRs as well but copies four registers back per use:
    // v_accvgpr_read x 4 in front of most MFMAs -- issue slots a one-wave-per-SIMD kernel does not have to spare).
#define TK5_MFMA(ACC, J, HALF, SH) tk5_mfma<((J) * 24 + 12 * (HALF) + (SH) < 64)>(a[SH], qf[J][12 * (HALF) + (SH)], ACC[J]);
#define TK5_STEP(ACC, HALF, SH, W)                                                                              \
    asm_wait_lgkm<W>();                                                                                        \
    TK5_MFMA(ACC, 0, HALF, SH) TK5_MFMA(ACC, 1, HALF, SH) TK5_MFMA(ACC, 2, HALF, SH) TK5_MFMA(ACC, 3, HALF, SH)
    // The previous tile's epilogue, in pieces that sit between the MFMA groups of the current tile (plain VALU code between two
    // scheduling barriers: it issues in the shadow of the four MFMAs next to it).  PREV holds the finished tile; its inverse norms
    // were requested at the top of the first half (TK5_PINV: one more LDS read in flight there, hence X = 1 on that half's
    // counted waits) and have arrived, in order, before the second half's first wait.
#define TK5_PINV(TILE) ivb_d = asm_ds_read_b128<0>(st.pbase + (unsigned)((TILE) & 7) * 64 + lg * 16); __builtin_amdgcn_sched_barrier(0);
#define TK5_FIN(PREV, J)                                                                                        \
    md[J] = tk5_max(tk5_max(PREV[J][0] * iv_d[0], PREV[J][1] * iv_d[1]), tk5_max(PREV[J][2] * iv_d[2], PREV[J][3] * iv_d[3])); \
    PREV[J] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // one half-tile: `slot` holds it, its first six fragments are already on their way (or there)
#define TK5_HALF(ACC, HALF, TILE_NEXT, LIVE, X, PRE, F0, F1, F2, F3, F4)                                        \
    {                                                                                                          \
        const unsigned sb = st.rbase + slot * TKR_HALF_BYTES;                                                  \
        TK5_READ6(1, sb)                                                                                       \
        PRE                                                                                                    \
        TK5_STEP(ACC, HALF, 0, 6 + X) F0 TK5_STEP(ACC, HALF, 1, 6 + X) F1 TK5_STEP(ACC, HALF, 2, 6 + X) F2         \
        TK5_STEP(ACC, HALF, 3, 6 + X) F3 TK5_STEP(ACC, HALF, 4, 6 + X) F4 TK5_STEP(ACC, HALF, 5, 6 + X)            \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        tkr_wait_vm<C::WAIT>();                                                                                \
        pp_barrier();                                                                                          \
        tk4_issue<QW, HALF, AUX>(st, TILE_NEXT, LIVE, slot == 0 ? C::NS - 1 : slot - 1);                       \
        slot = slot == C::NS - 1 ? 0 : slot + 1;                                                               \
        const unsigned sn = st.rbase + slot * TKR_HALF_BYTES;                                                  \
        TK5_READ6(0, sn)                                                                                       \
        TK5_STEP(ACC, HALF, 6, 11 + X) TK5_STEP(ACC, HALF, 7, 10 + X) TK5_STEP(ACC, HALF, 8, 9 + X)               \
        TK5_STEP(ACC, HALF, 9, 8 + X) TK5_STEP(ACC, HALF, 10, 7 + X) TK5_STEP(ACC, HALF, 11, 6 + X)               \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
    // one tile into ACC while PREV (tile T - 1, if the range has one) is reduced; ends with PREV's stores
#define TK5_TILE(ACC, PREV, T)                                                                                  \
    {                                                                                                          \
        u32x4_t ivb_d;                                                                                         \
        float md[4], mine_d;                                                                                   \
        TK5_HALF(ACC, 0, (T) + C::D / 2, h + C::D < nh, 1, TK5_PINV((T) - 1), , , , , )                         \
        const f32x4_t iv_d = __builtin_bit_cast(f32x4_t, ivb_d);                                               \
        TK5_HALF(ACC, 1, (T) + C::D / 2, h + 1 + C::D < nh, 0, , TK5_FIN(PREV, 0), TK5_FIN(PREV, 1), TK5_FIN(PREV, 2), \
                 TK5_FIN(PREV, 3), mine_d = rows_to_mine(md);)                                        \
        if ((T) > lo) fin_tail((T) - 1, mine_d);                                                                   \
        h += 2;                                                                                                \
    }
    int slot = 0;
    long h = 0;
    tkr_wait_vm<C::WAIT0>();
    pp_barrier();
    {
        const unsigned s0 = st.rbase;
        TK5_READ6(0, s0)
        asm_wait_lgkm<0>();
    }
    long t = lo;
    for (; t + 1 < hi; t += 2) {
        TK5_TILE(accA, accB, t)
        TK5_TILE(accB, accA, t + 1)
    }
    if (t < hi) {                                                  // odd range: one more tile into accA, then it is the last
        TK5_TILE(accA, accB, t)
        fin_full(accA, t);
    } else {
        fin_full(accB, hi - 1);
    }
#undef TK5_TILE
#undef TK5_HALF
#undef TK5_FIN
#undef TK5_PINV
#undef TK5_STEP
#undef TK5_MFMA
#undef TK5_READ6
    tkr_wait_vm<0>();                                              // the trailing dummies (and the last stores)
}

// The group-max scan of <= 1024 queries over the shard: gmax[q][group] = best approximate score of the 16 rows of the group.
// Returns 1 when a 1024-thread selection is the matching follow-up (streaming / ping-pong scans), 0 for the 256-thread one,
// negative on error.
// wmax / nw_out (optional): room for [nq][1024] per-wave maxima; *nw_out = the number of waves when the scan wrote them, else 0.
// does the shared-ring streaming scan (topk_stream5_kernel<2 / 4>) take a sweep of nq (65 .. 256) queries over this shard?
static bool stream_shared_ok(int dim, int64_t rows, int nq) {
    const long ngroups = (rows + TK_G - 1) / TK_G;
    return nq > 64 && nq <= 256 && dim == 768 && rows * 1536 < (1L << 31) && ngroups >= 2048;
}
static int tk_cu_count() {
    static int ncu = 0;
    if (!ncu) {
        int d = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&d) != hipSuccess || hipGetDeviceProperties(&prop, d) != hipSuccess) return -1;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return ncu;
}
template <int QW>
static void launch_stream5(int nv, hipStream_t st0, const void* pool_f16, const float* pool_inv_norm, long rows,
                           const void* queries_f16, int nq, float* gmax, long ngroups, float* wm) {
    static PerDeviceOnce attr;
    if (attr.first())
        (void)hipFuncSetAttribute((const void*)topk_stream5_kernel<QW, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, Tk5<QW>::LDS);
    hipLaunchKernelGGL((topk_stream5_kernel<QW, 2>), dim3(nv / QW), dim3(64 * QW), Tk5<QW>::LDS, st0, (const unsigned short*)pool_f16,
                       pool_inv_norm, rows, (const unsigned short*)queries_f16, nq, gmax, ngroups, wm);
}
// The scan of a sweep, by shape (every pool stream is read with the nt policy where a row is read exactly once per sweep:
// measured 0.2729 -> 0.2466 ms per 64-query search):
//   <= 64 queries, dim 768 / 512, a shard the 31-bit buffer bound addresses, >= 2048 groups : topk_stream2_kernel<dim / 32>
//   <= 64 queries, dim 768 / 512 otherwise (small shards)                                    : topk_stream_kernel (queries in LDS)
//   65 .. 256 queries where the first line's conditions hold                           : topk_stream5_kernel<2 / 4>
//   more queries or other shapes, dim a multiple of 64 and >= 192                      : topk_gmax_pp_kernel (ping-pong GEMM core)
//   anything else                                                                      : topk_gmax_kernel
static int launch_gmax_scan(const void* pool_f16, const float* pool_inv_norm, int64_t rows, int32_t dim,
                            const void* queries_f16, int32_t nq, float* gmax, hipStream_t st0, float* wmax = nullptr,
                            int* nw_out = nullptr) {
    if (nw_out) *nw_out = 0;
    const long ngroups = (rows + TK_G - 1) / TK_G;
    int nqt, nsl; long rps;
    coarse_plan(nq, rows, &nqt, &nsl, &rps);
    const bool big_stream = (dim == 768 || dim == 512) && rows * dim * 2 < (1L << 31) && ngroups >= 2048;
    if (nq <= 64 && big_stream) {
        const int ncu = tk_cu_count();
        if (ncu < 0) return UNIIR_ELAUNCH;
        // the per-wave maxima feed the hierarchical selection of the fused tail
        float* wm = (wmax && nw_out && ncu * 4 <= 1024 && (ngroups + ncu * 4 - 1) / (ncu * 4) <= 64) ? wmax : nullptr;
        if (wm) *nw_out = ncu * 4;
#define TKS2_LAUNCH(NK)                                                                                                             \
    do {                                                                                                                            \
        static PerDeviceOnce attr_s2;                                                                                               \
        if (attr_s2.first())                                                                                                        \
            (void)hipFuncSetAttribute((const void*)topk_stream2_kernel<NK, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,          \
                                      4 * Tkr<NK>::WAVE_LDS);                                                                       \
        hipLaunchKernelGGL((topk_stream2_kernel<NK, 2>), dim3(ncu), dim3(256), 4 * Tkr<NK>::WAVE_LDS, st0,                          \
                           (const unsigned short*)pool_f16, pool_inv_norm, (long)rows, (const unsigned short*)queries_f16, nq, gmax, \
                           ngroups, wm);                                                                                            \
    } while (0)
        if (dim == 768) TKS2_LAUNCH(24);
        else TKS2_LAUNCH(16);
#undef TKS2_LAUNCH
        HIP_LAUNCH_CHECK();
        return 1;
    }
    if (nq <= 64 && (dim == 768 || dim == 512)) {
        const long nchunks = (ngroups + TKS_CH - 1) / TKS_CH;
        const size_t sms = 64 * (dim * 2 + 16) + 8 * 64 * TKS_CH * 4;
        int grid = 256;
        if (nchunks < (long)grid * 8) grid = (int)((nchunks + 7) / 8);
        if (dim == 768) {
            (void)hipFuncSetAttribute((const void*)topk_stream_kernel<24>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sms);
            hipLaunchKernelGGL(topk_stream_kernel<24>, dim3(grid), dim3(512), sms, st0, (const unsigned short*)pool_f16,
                               pool_inv_norm, (long)rows, (const unsigned short*)queries_f16, nq, gmax, ngroups, nchunks);
        } else {
            (void)hipFuncSetAttribute((const void*)topk_stream_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sms);
            hipLaunchKernelGGL(topk_stream_kernel<16>, dim3(grid), dim3(512), sms, st0, (const unsigned short*)pool_f16,
                               pool_inv_norm, (long)rows, (const unsigned short*)queries_f16, nq, gmax, ngroups, nchunks);
        }
        HIP_LAUNCH_CHECK();
        return 1;
    }
    // 65 .. 256 queries, dim 768: the shared-ring streaming scan (queries in registers, 64 per wave)
    // MEASURED (round 3, 700 k rows, whole search): 128 queries 0.224-0.239 ms vs 0.280-0.291 (ping-pong GEMM scan), 256 queries
    // 0.318-0.345 vs 0.376
    if (stream_shared_ok(dim, rows, nq)) {
        const int ncu = tk_cu_count();
        if (ncu < 0) return UNIIR_ELAUNCH;
        const int nv = ncu * 4;
        float* wm = (wmax && nw_out && nv <= 1024 && (ngroups + nv - 1) / nv <= 64) ? wmax : nullptr;
        if (wm) *nw_out = nv;
        if (nq <= 128) launch_stream5<2>(nv, st0, pool_f16, pool_inv_norm, (long)rows, queries_f16, nq, gmax, ngroups, wm);
        else launch_stream5<4>(nv, st0, pool_f16, pool_inv_norm, (long)rows, queries_f16, nq, gmax, ngroups, wm);
        HIP_LAUNCH_CHECK();
        return 1;
    }
    if (nq > 64 && dim % 64 == 0 && dim >= 192) {
        const int tiles_q = (nq + 255) / 256;
        const long tiles_c = (rows + 255) / 256;
        const bool halfn = nq <= 128;         // half-width query tile
        const bool nt = nq <= 256;            // one query tile: every pool row is read exactly once
#define TKPP_LAUNCH(H, A)                                                                                                      \
    do {                                                                                                                       \
        static PerDeviceOnce attr;                                                                                             \
        if (attr.first())                                                                                                      \
            (void)hipFuncSetAttribute((const void*)topk_gmax_pp_kernel<H, A>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
        hipLaunchKernelGGL((topk_gmax_pp_kernel<H, A>), dim3((unsigned)(tiles_c * tiles_q)), dim3(512), 131072, st0,          \
                           (const unsigned short*)pool_f16, pool_inv_norm, (long)rows, dim, (const unsigned short*)queries_f16, \
                           nq, gmax, ngroups, tiles_q);                                                                        \
    } while (0)
        if (halfn) TKPP_LAUNCH(true, 2);
        else if (nt) TKPP_LAUNCH(false, 2);
        else TKPP_LAUNCH(false, 0);
#undef TKPP_LAUNCH
        HIP_LAUNCH_CHECK();
        return 1;
    }
    const long ct = 256;
    long tiles = (rows + ct - 1) / ct;
    long want = (768 + nqt - 1) / nqt;
    if (want > tiles) want = tiles;
    if (want < 1) want = 1;
    const long tps = (tiles + want - 1) / want;
    rps = tps * ct;
    nsl = (int)((rows + rps - 1) / rps);
    const size_t smg = GldsShape<2, 2, 32>::LDS_BYTES;
    (void)hipFuncSetAttribute((const void*)topk_gmax_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smg);
    hipLaunchKernelGGL(topk_gmax_kernel<2>, dim3(nsl, nqt), dim3(256), smg, st0, (const unsigned short*)pool_f16,
                       pool_inv_norm, (long)rows, dim, (const unsigned short*)queries_f16, nq, rps, gmax, ngroups);
    HIP_LAUNCH_CHECK();
    return 0;
}

// the stand-alone group selection behind a scan (`sel` = launch_gmax_scan's return value)
static int topk_select_after_scan(int sel, const float* gmax, int64_t rows, int32_t nq, int32_t kc, int32_t* cand_idx,
                                  hipStream_t st0) {
    const long ngroups = (rows + TK_G - 1) / TK_G;
    if (sel == 1 && ngroups % 2 == 0 && ngroups <= 1024L * 2 * TK_SELREG)       // register-resident selection
        hipLaunchKernelGGL((topk_gsel_kernel<1024, true>), dim3(nq), dim3(1024), 0, st0, gmax, ngroups, (long)rows, nq,
                           kc, TK_GMULT * kc, cand_idx);
    else if (sel == 1 && nq <= 64)
        hipLaunchKernelGGL((topk_gsel_kernel<1024, false>), dim3(nq), dim3(1024), 0, st0, gmax, ngroups, (long)rows, nq,
                           kc, TK_GMULT * kc, cand_idx);
    else
        hipLaunchKernelGGL(topk_gsel_kernel<256>, dim3(nq), dim3(256), 0, st0, gmax, ngroups, (long)rows, nq, kc,
                           TK_GMULT * kc, cand_idx);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

#define TK_WMAX_BYTES(nq) ((int64_t)((nq) < 64 ? 64 : (nq)) * 1024 * 4 + 256)     // per-wave maxima of the streaming scans: [nq][<= 1024 waves] fp32
extern "C" int64_t uniir_topk_workspace_bytes(int32_t nq, int32_t kc, int64_t rows) {
    if (nq <= 0 || kc <= 0 || rows <= 0) return 0;
    if (nq <= TK_GPATH_MAXQ) {
        const int64_t ngroups = (rows + TK_G - 1) / TK_G;
        const int64_t dense = (int64_t)nq * ngroups * 4 + 256 + TK_WMAX_BYTES(nq <= 256 ? nq : 64);
        return dense;
    }
    int nqt, nsl; long rps;
    coarse_plan(nq, rows, &nqt, &nsl, &rps);
    return (int64_t)nqt * nsl * TK_QT * TK_CAP * (int64_t)sizeof(TkEntry) + (int64_t)nsl * nq * kc * (int64_t)sizeof(TkEntry) + 256;
}

extern "C" int uniir_topk_coarse(const void* pool_f16, const float* pool_inv_norm, int64_t rows, int32_t dim,
                                 const void* queries_f16, int32_t nq, int32_t kc, int32_t* cand_idx,
                                 float* cand_score, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!pool_f16 || !pool_inv_norm || !queries_f16 || !cand_idx || !workspace) return UNIIR_EINVAL;
    if (rows <= 0 || nq <= 0 || kc <= 0) return UNIIR_EINVAL;
    if (kc > TK_MAXKC || dim % 32 || dim <= 0 || rows > 0x7fffffffL || rows * (int64_t)dim * 2 >= ((int64_t)1 << 31)) return UNIIR_ESHAPE;
    if (((uintptr_t)pool_f16 & 15) || ((uintptr_t)queries_f16 & 15) || ((uintptr_t)pool_inv_norm & 15) ||
        ((uintptr_t)workspace & 15))
        return UNIIR_EALIGN;
    if (workspace_bytes < uniir_topk_workspace_bytes(nq, kc, rows)) return UNIIR_EINVAL;
    int nqt, nsl; long rps;
    coarse_plan(nq, rows, &nqt, &nsl, &rps);
    hipStream_t st0 = (hipStream_t)stream;
    if (nq <= TK_GPATH_MAXQ) {
        const long ngroups = (rows + TK_G - 1) / TK_G;
        float* gmax = (float*)workspace;
        const int sel = launch_gmax_scan(pool_f16, pool_inv_norm, rows, dim, queries_f16, nq, gmax, st0);
        if (sel < 0) return sel;
        return topk_select_after_scan(sel, gmax, rows, nq, kc, cand_idx, st0);
    }
    TkEntry* bufs = (TkEntry*)workspace;
    TkEntry* partial = bufs + (long)nqt * nsl * TK_QT * TK_CAP;
    const size_t sm = TkShape::LDS_BYTES + 1024 + 64;
    static PerDeviceOnce attr;
    if (attr.first())
        (void)hipFuncSetAttribute((const void*)topk_coarse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(topk_coarse_kernel, dim3(nsl, nqt), dim3(256), sm, st, (const unsigned short*)pool_f16,
                       pool_inv_norm, (long)rows, dim, (const unsigned short*)queries_f16, nq, kc, rps, bufs, partial);
    hipLaunchKernelGGL(topk_merge_partial_kernel, dim3(nq), dim3(256), 0, st, partial, nsl, nq, kc, cand_idx,
                       cand_score);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// -------------------------------------------------------------------------------------------------------------
// uniir_topk_ip: the whole search_index of one pool shard in one call (mbeir_retriever.py:188-232 = normalise the queries,
// exact inner-product top-k): query inverse norms, then per chunk of <= 1024 queries one sweep of the shard (group-max scan),
// group selection, exact re-score, sort.  k <= 56 (k + 8 <= TK_MAXKC groups per query); larger k is assembled from slices by
// the caller.  (A single fused select + re-score + sort launch per query was measured and dropped: with 64 queries the tail is
// latency-bound and one 1024-thread workgroup per query serialises what the three launches spread over the whole chip:
// 98 us instead of 80 us behind the 217-us scan; at 1024 queries it made no difference.)
#define TKI_CHUNK_MAX 1024
static int g_tki_chunk = 0;             // 0 = automatic
// queries per sweep of uniir_topk_ip.  0 = automatic: 256 where the 4-wave streaming scan applies (dim 768, a shard the buffer
// bound can address, >= 2048 groups), else 1024 (the most the GEMM-shaped scan takes).  MEASURED (round 3, 700 k rows, same box,
// twice): 16 384 queries 20.14 / 20.24 ms in 256-query sweeps vs 20.91 / 20.87 in 1024-query sweeps; 1024 queries 1.267 / 1.270 vs
// 1.380 / 1.303 -- four streaming sweeps re-read the pool three more times (HBM is not the limit there) and still beat one GEMM-
// shaped sweep.  An explicit value only changes the number of sweeps (tests of the sweep loop; tuning); results never depend on it.
// Host-side setting, not thread-safe.
extern "C" int uniir_topk_set_chunk(int32_t queries_per_sweep) {
    if (queries_per_sweep < 0 || queries_per_sweep > TKI_CHUNK_MAX) return UNIIR_EINVAL;
    g_tki_chunk = queries_per_sweep;
    return UNIIR_OK;
}
extern "C" int32_t uniir_topk_ip_sweep_queries(int32_t dim, int64_t rows) {
    if (g_tki_chunk) return g_tki_chunk;
    return stream_shared_ok(dim, rows, 256) ? 256 : TKI_CHUNK_MAX;
}
static int64_t tki_ws_bytes(int32_t nq, int32_t k, int64_t rows, int chunk_max) {
    const int chunk = nq < chunk_max ? nq : chunk_max;
    const int kc = k + 8 < TK_MAXKC ? k + 8 : TK_MAXKC;
    const int64_t ncand = uniir_topk_ncand(chunk, kc);
    return uniir_topk_workspace_bytes(chunk, kc, rows) + (int64_t)nq * 4 + 256 + 2 * ((int64_t)chunk * ncand * 4 + 256);
}
extern "C" int64_t uniir_topk_ip_workspace_bytes(int32_t nq, int32_t k, int64_t rows) {
    if (nq <= 0 || k <= 0 || rows <= 0) return 0;
    if (g_tki_chunk) return tki_ws_bytes(nq, k, rows, g_tki_chunk);
    const int64_t a = tki_ws_bytes(nq, k, rows, 256), b = tki_ws_bytes(nq, k, rows, TKI_CHUNK_MAX);     // dim is not known here
    return a > b ? a : b;
}
// the exact requirement of a search of this shape: the sweep width is a function of (dim, rows), so up to 256 queries on a dim-768
// shard need a quarter of the group-maxima region the dim-agnostic bound reserves (45 MB instead of 179 MB at 700 k rows)
extern "C" int64_t uniir_topk_ip_workspace_bytes_ex(int32_t nq, int32_t k, int64_t rows, int32_t dim) {
    if (nq <= 0 || k <= 0 || rows <= 0 || dim <= 0) return 0;
    return tki_ws_bytes(nq, k, rows, uniir_topk_ip_sweep_queries(dim, rows));
}
extern "C" int uniir_topk_ip(const void* pool_f16, const float* pool_inv_norm, const int64_t* pool_ids, int64_t rows,
                             int32_t dim, const void* queries_f16, int32_t nq, int32_t k, float* out_scores,
                             int64_t* out_ids, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!pool_f16 || !pool_inv_norm || !queries_f16 || !out_scores || !out_ids || !workspace) return UNIIR_EINVAL;
    if (rows <= 0 || nq <= 0 || k <= 0) return UNIIR_EINVAL;
    if (k + 8 > TK_MAXKC || dim % 64 || dim <= 0 || rows > 0x7fffffffL) return UNIIR_ESHAPE;
    // a shard is addressed through 31-bit buffer offsets: >= 2 GiB of rows is searched as equal sub-shards by the caller
    // (uniir_topk_ip_multi below: the 5.6 M x 768 pool on one GPU = 8 logical sub-shards), never silently by a slower path
    if (rows * (int64_t)dim * 2 >= ((int64_t)1 << 31)) return UNIIR_ESHAPE;
    if (((uintptr_t)pool_f16 & 15) || ((uintptr_t)queries_f16 & 15) || ((uintptr_t)pool_inv_norm & 15) ||
        ((uintptr_t)workspace & 255))
        return UNIIR_EALIGN;
    if (workspace_bytes < uniir_topk_ip_workspace_bytes_ex(nq, k, rows, dim)) return UNIIR_EINVAL;
    const int chunk_max = uniir_topk_ip_sweep_queries(dim, rows);
    const int chunk = nq < chunk_max ? nq : chunk_max;
    const int kc = k + 8;
    const int ncand = uniir_topk_ncand(chunk, kc);
    char* ws = (char*)workspace;
    float* gmax = (float*)ws;
    const int64_t gbytes = (uniir_topk_workspace_bytes(chunk, kc, rows) + 255) & ~(int64_t)255;
    float* qinv = (float*)(ws + gbytes);
    int32_t* cand = (int32_t*)(ws + gbytes + (((int64_t)nq * 4 + 255) & ~(int64_t)255));
    float* exact = (float*)((char*)cand + (((int64_t)chunk * ncand * 4 + 255) & ~(int64_t)255));
    int rc = UNIIR_OK;
    bool have_qinv = false;
    // (One tail per BATCH of sweeps -- the scans of four 256-query sweeps writing their maxima side by side, then one fused tail and
    // one sort for the 1024 queries -- was built and measured in round 4: 1024 queries 1.262 ms vs 1.24-1.27, 100 000 queries 123.8 vs
    // 124.3 ms.  The tail is throughput-bound, not launch-bound; dropped.)
    for (int lo = 0; lo < nq; lo += chunk) {
        const int n = nq - lo < chunk ? nq - lo : chunk;
        const unsigned short* qp = (const unsigned short*)queries_f16 + (long)lo * dim;
        if (n <= TK_GPATH_MAXQ) {       // group-max scan, then the fused tail (selection + query norm + exact re-score | sort)
            // the per-wave maxima live behind the group maxima of this sweep (uniir_topk_workspace_bytes reserves the room)
            const int64_t ngr = (rows + TK_G - 1) / TK_G;
            float* wmax = (float*)(((uintptr_t)(gmax + (int64_t)n * ngr) + 255) & ~(uintptr_t)255);
            int nw = 0;
            const int sel = launch_gmax_scan(pool_f16, pool_inv_norm, rows, dim, qp, n, gmax, (hipStream_t)stream, wmax, &nw);
            if (sel < 0) return sel;
            if (sel >= 1 && launch_fused_tail(pool_f16, pool_inv_norm, pool_ids, rows, dim, qp, n, kc, k, gmax, cand, exact,
                                              out_scores + (long)lo * k, out_ids + (long)lo * k, (hipStream_t)stream, wmax, nw)) {
                HIP_LAUNCH_CHECK();
                continue;
            }
            rc = topk_select_after_scan(sel, gmax, rows, n, kc, cand, (hipStream_t)stream);
            if (rc) return rc;
        } else {
            rc = uniir_topk_coarse(pool_f16, pool_inv_norm, rows, dim, qp, n, kc, cand, nullptr, gmax, gbytes, stream);
            if (rc) return rc;
        }
        if (!have_qinv) {               // the unfused tail reads the query norms from a separate pass
            rc = uniir_pool_inv_norms(queries_f16, nq, dim, qinv, stream);
            if (rc) return rc;
            have_qinv = true;
        }
        rc = uniir_topk_rescore(pool_f16, pool_inv_norm, pool_ids, rows, dim, qp, qinv + lo, n, cand, ncand, k, exact,
                                out_scores + (long)lo * k, out_ids + (long)lo * k, stream);
        if (rc) return rc;
    }
    return UNIIR_OK;
}

// -------------------------------------------------------------------------------------------------------------
// uniir_topk_ip_multi (round 5): the search of ONE resident shard of any size -- the whole 5.6 M x 768 M-BEIR pool on one GPU
// (mbeir_retriever.py:196-206 with a single visible device; README.md:126).  The streaming scans and the gather address rows through
// 31-bit buffer offsets and the fused tail selects from at most 786 432 rows, so the shard is cut into equal LOGICAL sub-shards
// (uniir_topk_subshard_rows: a multiple of 32 rows, below both bounds; nothing is copied) and per sweep of queries
//   one scan launch PER sub-shard (its own 64-bit base, its own group / wave maxima region)  -- the pool is still read exactly once,
//   ONE fused tail launch for all sub-shards (workgroup (query, part, sub-shard), TkMulti),
//   ONE sort launch (a k-list per (sub-shard, query)),
//   ONE merge launch on (score desc, id asc) -- uniir_topk_merge's kernel; ids are unique, so this IS the search of the whole shard.
// Round 4 ran the three launches of uniir_topk_ip per sub-shard from the host and merged at the end: 5 x (27 + 8) us of latency-bound
// tails behind 5 scans (64 queries: 1.88 ms = 0.571 of 8 TB/s over the 8.6-GB pool); here the tails of all sub-shards run side by side.
// A shape the batched tail does not take (a last sub-shard with an odd number of groups, k + 8 > 64 ...) falls back to exactly that
// loop -- same results.  A shard below both bounds is one uniir_topk_ip call.
// -------------------------------------------------------------------------------------------------------------
extern "C" int64_t uniir_topk_subshard_rows(int64_t rows, int32_t dim) {
    if (rows <= 0 || dim <= 0) return 0;
    if (rows * (int64_t)dim * 2 < ((int64_t)1 << 31)) return rows;
    const int64_t fused = 1024L * 2 * TK_SELREG * TK_G;                       // the fused tail's register-resident selection
    int64_t cap = ((((int64_t)1 << 31) - 1) / ((int64_t)dim * 2)) / 32 * 32;
    if (cap > fused) cap = fused;
    const int64_t parts = (rows + cap - 1) / cap;
    return ((rows + parts - 1) / parts + 31) / 32 * 32;
}
static int64_t tkm_align(int64_t x) { return (x + 255) & ~(int64_t)255; }
struct TkmPlan {
    int64_t per; int nsub; int chunk; int kc; int ncand;
    int64_t g_bytes, w_bytes, c_bytes, o_bytes, inner_bytes;     // per sub-shard regions (aligned); inner = one uniir_topk_ip call's scratch
};
static void tkm_plan(int32_t nq, int32_t k, int64_t rows, int32_t dim, TkmPlan* p) {
    p->per = uniir_topk_subshard_rows(rows, dim);
    p->nsub = (int)((rows + p->per - 1) / p->per);
    const int chunk_max = uniir_topk_ip_sweep_queries(dim, p->per);
    p->chunk = nq < chunk_max ? nq : chunk_max;
    p->kc = k + 8;
    p->ncand = uniir_topk_ncand(p->chunk, p->kc);
    const int64_t ngr = (p->per + TK_G - 1) / TK_G;
    p->g_bytes = tkm_align((int64_t)p->chunk * ngr * 4);
    p->w_bytes = tkm_align(TK_WMAX_BYTES(p->chunk <= 256 ? p->chunk : 64));
    p->c_bytes = tkm_align((int64_t)p->chunk * p->ncand * 4);
    p->o_bytes = tkm_align((int64_t)p->chunk * k * 8);
    p->inner_bytes = tkm_align(uniir_topk_ip_workspace_bytes_ex(nq, k, p->per, dim));
}
extern "C" int64_t uniir_topk_ip_multi_workspace_bytes(int32_t nq, int32_t k, int64_t rows, int32_t dim) {
    if (nq <= 0 || k <= 0 || rows <= 0 || dim <= 0) return 0;
    if (rows * (int64_t)dim * 2 < ((int64_t)1 << 31)) return uniir_topk_ip_workspace_bytes_ex(nq, k, rows, dim);
    TkmPlan p;
    tkm_plan(nq, k, rows, dim, &p);
    const int64_t batched = p.nsub * (p.g_bytes + p.w_bytes + 2 * p.c_bytes + 2 * p.o_bytes);
    const int64_t fallback = p.inner_bytes + 2 * p.nsub * tkm_align((int64_t)nq * k * 8);       // per-sub-shard lists of all queries
    return (batched > fallback ? batched : fallback) + 256;
}
extern "C" int uniir_topk_ip_multi(const void* pool_f16, const float* pool_inv_norm, const int64_t* pool_ids, int64_t rows,
                                   int32_t dim, const void* queries_f16, int32_t nq, int32_t k, float* out_scores,
                                   int64_t* out_ids, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!pool_f16 || !pool_inv_norm || !queries_f16 || !out_scores || !out_ids || !workspace) return UNIIR_EINVAL;
    if (rows <= 0 || nq <= 0 || k <= 0) return UNIIR_EINVAL;
    if (rows * (int64_t)dim * 2 < ((int64_t)1 << 31))
        return uniir_topk_ip(pool_f16, pool_inv_norm, pool_ids, rows, dim, queries_f16, nq, k, out_scores, out_ids, workspace,
                             workspace_bytes, stream);
    if (k + 8 > TK_MAXKC || dim % 64 || dim <= 0 || rows > 0x7fffffffL) return UNIIR_ESHAPE;
    if (((uintptr_t)pool_f16 & 15) || ((uintptr_t)queries_f16 & 15) || ((uintptr_t)pool_inv_norm & 15) || ((uintptr_t)workspace & 255))
        return UNIIR_EALIGN;
    if (workspace_bytes < uniir_topk_ip_multi_workspace_bytes(nq, k, rows, dim)) return UNIIR_EINVAL;
    TkmPlan p;
    tkm_plan(nq, k, rows, dim, &p);
    hipStream_t st = (hipStream_t)stream;
    const int64_t last_rows = rows - (int64_t)(p.nsub - 1) * p.per;
    const int64_t last_groups = (last_rows + TK_G - 1) / TK_G;
    // the batched tail wants every sub-shard on a streaming scan with wave maxima and on the fused tail (<= 256 queries per sweep)
    const bool big = (dim == 768 || dim == 512) && last_groups >= 2048 && last_groups % 2 == 0 && p.chunk <= 256 &&
                     (p.chunk <= 64 || stream_shared_ok(dim, p.per, p.chunk)) && (p.chunk <= 64 || stream_shared_ok(dim, last_rows, p.chunk)) &&
                     fused_tail_ok(p.per, dim, p.kc) && fused_tail_ok(last_rows, dim, p.kc) && p.nsub <= 64;
    if (!pool_ids) return UNIIR_EINVAL;          // (local row numbers of different sub-shards cannot be merged)
    char* ws = (char*)workspace;
    // the per-sub-shard loop of round 4, on the device side of the ABI: lists [nsub][nq][k], one merge.  Also the way out when the
    // batched path refuses a shape (same results: both end in the exact re-score and the same merge rule)
    auto per_sub_shard = [&]() -> int {
        float* ls = (float*)(ws + p.inner_bytes);
        int64_t* li = (int64_t*)(ws + p.inner_bytes + p.nsub * tkm_align((int64_t)nq * k * 8));
        for (int z = 0; z < p.nsub; ++z) {
            const int64_t lo = (int64_t)z * p.per, n = z == p.nsub - 1 ? last_rows : p.per;
            const int rc = uniir_topk_ip((const unsigned short*)pool_f16 + lo * dim, pool_inv_norm + lo, pool_ids ? pool_ids + lo : nullptr, n,
                                         dim, queries_f16, nq, k, ls + (int64_t)z * nq * k, li + (int64_t)z * nq * k, ws, p.inner_bytes, stream);
            if (rc) return rc;
        }
        return uniir_topk_merge(ls, li, p.nsub, nq, k, out_scores, out_ids, stream);
    };
    if (!big) return per_sub_shard();
    // the per-wave maxima (hierarchical selection) exist only when EVERY sub-shard's scan writes them: the rule depends on the
    // sub-shard's group count and the device's CU count, and the shorter last sub-shard may disagree with the full ones
    const int ncu = tk_cu_count();
    if (ncu < 0) return UNIIR_ELAUNCH;
    auto waves_ok = [&](int64_t r) { const long ng = (r + TK_G - 1) / TK_G; return ncu * 4 <= 1024 && (ng + ncu * 4 - 1) / (ncu * 4) <= 64; };
    const bool use_wmax = waves_ok(p.per) && waves_ok(last_rows);
    float* gmax = (float*)ws;
    float* wmax = (float*)(ws + p.nsub * p.g_bytes);
    int32_t* cand = (int32_t*)(ws + p.nsub * (p.g_bytes + p.w_bytes));
    float* exact = (float*)((char*)cand + p.nsub * p.c_bytes);
    float* ls = (float*)((char*)exact + p.nsub * p.c_bytes);
    int64_t* li = (int64_t*)((char*)ls + p.nsub * p.o_bytes);
    TkMulti mu;
    mu.per = p.per;
    mu.rows_total = rows;
    mu.g_stride = p.g_bytes / 4;
    mu.w_stride = p.w_bytes / 4;
    mu.c_stride = p.c_bytes / 4;
    for (int lo = 0; lo < nq; lo += p.chunk) {
        const int n = nq - lo < p.chunk ? nq - lo : p.chunk;
        const unsigned short* qp = (const unsigned short*)queries_f16 + (long)lo * dim;
        int nw = 0;
        for (int z = 0; z < p.nsub; ++z) {
            const int64_t r0 = (int64_t)z * p.per, nr = z == p.nsub - 1 ? last_rows : p.per;
            int nwz = 0;
            const int sel = launch_gmax_scan((const unsigned short*)pool_f16 + r0 * dim, pool_inv_norm + r0, nr, dim, qp, n,
                                             gmax + z * mu.g_stride, st, use_wmax ? wmax + z * mu.w_stride : nullptr, &nwz);
            if (sel < 0) return sel;
            if (sel < 1 || (z > 0 && nwz != nw)) return per_sub_shard();       // a scan the batched tail cannot follow: start over
            if (z == 0) nw = nwz;
        }
        mu.o_stride = (long)n * k;
        if (!launch_fused_tail(pool_f16, pool_inv_norm, pool_ids, p.per, dim, qp, n, p.kc, k, gmax, cand, exact, ls, li, st,
                               nw > 0 ? wmax : nullptr, nw, &mu, p.nsub))
            return per_sub_shard();
        HIP_LAUNCH_CHECK();
        const int rc = uniir_topk_merge(ls, li, p.nsub, n, k, out_scores + (long)lo * k, out_ids + (long)lo * k, stream);
        if (rc) return rc;
    }
    return UNIIR_OK;
}
