// The tail of the exact top-k search (see topk.hip for the whole picture): exact fp32 re-score in the oracle's summation order,
// (score desc, id asc) sorts, the k-way merge of per-shard results, and the fused tail (selection + query norm + re-score | sort)
// behind a group-max scan.  Reference: mbeir_retriever.py:188-232 (search_index), :96-100 (sharded index).
#include "topk.h"
#include "topk_select.h"

// exact re-score: one thread per (query, candidate); then one thread per query sorts its shortlist.
__global__ __launch_bounds__(256) void rescore_kernel(const unsigned short* __restrict__ pool,
                                                      const float* __restrict__ pinv,
                                                      const unsigned short* __restrict__ queries,
                                                      const float* __restrict__ qinv, int nq, int dim,
                                                      const int* __restrict__ cand_idx, int ncand,
                                                      float* __restrict__ exact) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)nq * ncand) return;
    const int q = (int)(t / ncand);
    const int ci = cand_idx[t];
    if (ci < 0) { exact[t] = -INFINITY; return; }
    const unsigned short* qr = queries + (long)q * dim;
    const unsigned short* cr = pool + (long)ci * dim;
    const float iq = qinv[q], ic = pinv[ci];
    float s = 0.f;
    for (int c = 0; c < dim; c += 8) {
        const u32x4_t a = *reinterpret_cast<const u32x4_t*>(qr + c);
        const u32x4_t b = *reinterpret_cast<const u32x4_t*>(cr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float qa = f16_to_f32((unsigned short)(a[e] & 0xffffu)), ca = f16_to_f32((unsigned short)(b[e] & 0xffffu));
            float qb = f16_to_f32((unsigned short)(a[e] >> 16)), cb = f16_to_f32((unsigned short)(b[e] >> 16));
            // FAISS renorm: x[i] *= inv_nr (skipped for all-zero rows, where inv == 0 and x == 0 anyway)
            if (iq != 0.f) { qa = __fmul_rn(qa, iq); qb = __fmul_rn(qb, iq); }
            if (ic != 0.f) { ca = __fmul_rn(ca, ic); cb = __fmul_rn(cb, ic); }
            s = __fadd_rn(s, __fmul_rn(qa, ca));
            s = __fadd_rn(s, __fmul_rn(qb, cb));
        }
    }
    exact[t] = s;
}
// Same arithmetic, coalesced gathers (dim % 64 == 0): a wave owns 64 (query, candidate) pairs.  The candidate rows are
// fetched 128 B at a time by 8 lanes per row (8 rows per load instruction instead of 64 rows x 16 B), parked in LDS
// ([64 rows][128 B + 16 pad] per wave) and each lane then walks ITS row's 64 elements in order: the sum is still one
// sequential fp32 chain per pair, in the oracle's order.  The next 128-B slice is already in flight during the walk (two slices
// of look-ahead measured 34 us instead of 29 us at 64 queries: more registers per thread, no less exposed latency).
#define RSC_PITCH 144
__global__ __launch_bounds__(256) void rescore_coalesced_kernel(const unsigned short* __restrict__ pool,
                                                                const float* __restrict__ pinv,
                                                                const unsigned short* __restrict__ queries,
                                                                const float* __restrict__ qinv, int nq, int dim,
                                                                const int* __restrict__ cand_idx, int ncand,
                                                                float* __restrict__ exact) {
    __shared__ __attribute__((aligned(16))) char stage[4][64 * RSC_PITCH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long pairs = (long)nq * ncand;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = t < pairs;
    const int q = live ? (int)(t / ncand) : 0;
    const int ci = live ? cand_idx[t] : -1;
    const unsigned short* qr = queries + (long)q * dim;
    const float iq = qinv[q], ic = ci >= 0 ? pinv[ci] : 0.f;
    // lane -> (row r = 8 i + lane / 8 of the wave's 64 rows, 16-B piece lane % 8) for the cooperative loads
    int rows8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rows8[i] = __shfl(ci, 8 * i + (lane >> 3), 64);
    const int piece = lane & 7;
    char* mine = &stage[w][0];
    u32x4_t pre[8];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u32x4_t z = {0u, 0u, 0u, 0u};
            pre[i] = rows8[i] >= 0 ? *reinterpret_cast<const u32x4_t*>(pool + (long)rows8[i] * dim + c0 + piece * 8) : z;
        }
    };
    float s = 0.f;
    fetch(0);
    for (int c0 = 0; c0 < dim; c0 += 64) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<u32x4_t*>(mine + (8 * i + (lane >> 3)) * RSC_PITCH + piece * 16) = pre[i];
        __builtin_amdgcn_wave_barrier();       // the staging area is wave-private and a wave's LDS operations execute in issue
        if (c0 + 64 < dim) fetch(c0 + 64);     // order: no workgroup barrier (the four waves used to wait for each other's gathers)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u32x4_t a = *reinterpret_cast<const u32x4_t*>(qr + c0 + 8 * u);
            const u32x4_t b = *reinterpret_cast<const u32x4_t*>(mine + lane * RSC_PITCH + u * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float qa = f16_to_f32((unsigned short)(a[e] & 0xffffu)), ca = f16_to_f32((unsigned short)(b[e] & 0xffffu));
                float qb = f16_to_f32((unsigned short)(a[e] >> 16)), cb = f16_to_f32((unsigned short)(b[e] >> 16));
                if (iq != 0.f) { qa = __fmul_rn(qa, iq); qb = __fmul_rn(qb, iq); }
                if (ic != 0.f) { ca = __fmul_rn(ca, ic); cb = __fmul_rn(cb, ic); }
                s = __fadd_rn(s, __fmul_rn(qa, ca));
                s = __fadd_rn(s, __fmul_rn(qb, cb));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (live) exact[t] = ci >= 0 ? s : -INFINITY;
}
__global__ __launch_bounds__(256) void final_sort_kernel(const float* __restrict__ exact,
                                                         const int* __restrict__ cand_idx,
                                                         const long long* __restrict__ ids, int nq, int ncand,
                                                         int k, float* __restrict__ out_s,
                                                         long long* __restrict__ out_i) {
    // one block per query: k rounds of "best entry strictly after the previous one" in (score desc, id asc) order
    __shared__ float ss[4];
    __shared__ long long si[4];
    __shared__ float wsel;
    __shared__ long long isel;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* es = exact + (long)q * ncand;
    const int* ci = cand_idx + (long)q * ncand;
    if (ncand <= TK_RANKCAP) {
        // short shortlists (the group path hands over 2 (k + 8) groups of 16 rows, about half of them live): compact the live
        // entries into LDS and give each its rank by counting -- (score desc, id asc) is a strict total order, so ranks are
        // unique and ranks < k are the answer.  One pass instead of k rounds of workgroup-wide arg-max (15 -> ~6 us at k = 10).
        __shared__ __attribute__((aligned(16))) float ls[TK_RANKCAP + 16];
        __shared__ long long lid[TK_RANKCAP];
        __shared__ int nlive;
        if (tid == 0) nlive = 0;
        __syncthreads();
        for (int c = tid; c < ncand; c += 256) {
            const int row = ci[c];
            if (row >= 0) {
                const int pos = atomicAdd(&nlive, 1);
                ls[pos] = es[c];
                lid[pos] = ids ? ids[row] : (long long)row;
            }
        }
        __syncthreads();
        const int n = nlive, n16 = (n + 15) & ~15;
        if (tid < 16) ls[n + tid] = -INFINITY;          // pad to the 16-wide compare loop (never better than, never equal to a live score)
        __syncthreads();
        for (int e = tid; e < n; e += 256) {
            const float sc = ls[e];
            int rank = 0, ties = 0;
            // 16 scores per trip as four 16-byte LDS reads (uniform addresses: broadcasts), all in flight together: the one
            // entry per trip form of this loop was LDS-latency-bound (47 us)
            for (int u = 0; u < n16; u += 16) {
                f32x4_t v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4_t*>(ls + u + 4 * j);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        rank += v[j][r] > sc ? 1 : 0;
                        ties += v[j][r] == sc ? 1 : 0;
                    }
            }
            const long long id = lid[e];
            if (ties > 1) {                               // exact score ties (duplicate rows): the id decides
                for (int u = 0; u < n; ++u) rank += (ls[u] == sc && lid[u] < id) ? 1 : 0;
            }
            if (rank < k) {
                out_s[(long)q * k + rank] = sc;
                out_i[(long)q * k + rank] = id;
            }
        }
        for (int t = n + tid; t < k; t += 256) {       // FAISS pads missing results with -inf distance / id -1
            out_s[(long)q * k + t] = -INFINITY;
            out_i[(long)q * k + t] = -1;
        }
        return;
    }
    float last_s = INFINITY;
    long long last_id = -1;
    // the thread's shortlist entries live in registers for all k rounds (ncand <= 2048 = 8 per thread; more fall back to
    // re-reading): the rounds were re-fetching score / row / id from L2 every time
    float rs[8];
    long long rid[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = tid + 256 * u;
        const bool ok = c < ncand && ci[c] >= 0;
        rs[u] = ok ? es[c] : -INFINITY;
        rid[u] = ok ? (ids ? ids[ci[c]] : (long long)ci[c]) : 0x7fffffffffffffffLL;
    }
    for (int j = 0; j < k; ++j) {
        float bs = -INFINITY;
        long long bid = 0x7fffffffffffffffLL;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float s = rs[u];
            const long long id = rid[u];
            const bool after = id != 0x7fffffffffffffffLL && ((s < last_s) || (s == last_s && id > last_id));
            if (after && (s > bs || (s == bs && id < bid))) { bs = s; bid = id; }
        }
        for (int c = tid + 2048; c < ncand; c += 256) {
            if (ci[c] < 0) continue;
            const float s = es[c];
            const long long id = ids ? ids[ci[c]] : (long long)ci[c];
            const bool after = (s < last_s) || (s == last_s && id > last_id);
            if (after && (s > bs || (s == bs && id < bid))) { bs = s; bid = id; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o, 64);
            const long long oi = __shfl_xor(bid, o, 64);
            if (os > bs || (os == bs && oi < bid)) { bs = os; bid = oi; }
        }
        if (lane == 0) { ss[w] = bs; si[w] = bid; }
        __syncthreads();
        if (tid == 0) {
            float fs = ss[0]; long long fi = si[0];
            for (int kk = 1; kk < 4; ++kk) if (ss[kk] > fs || (ss[kk] == fs && si[kk] < fi)) { fs = ss[kk]; fi = si[kk]; }
            const bool found = fi != 0x7fffffffffffffffLL;
            // FAISS pads missing results with -inf distance / id -1 for inner product
            out_s[(long)q * k + j] = found ? fs : -INFINITY;
            out_i[(long)q * k + j] = found ? fi : -1;
            wsel = found ? fs : -INFINITY; isel = found ? fi : 0x7fffffffffffffffLL;
        }
        __syncthreads();
        last_s = wsel; last_id = isel;
        __syncthreads();
    }
}

extern "C" int uniir_topk_rescore(const void* pool_f16, const float* pool_inv_norm, const int64_t* pool_ids,
                                  int64_t rows, int32_t dim, const void* queries_f16, const float* query_inv_norm,
                                  int32_t nq, const int32_t* cand_idx, int32_t ncand, int32_t k, float* exact_ws,
                                  float* out_scores, int64_t* out_ids, void* stream) {
    if (!pool_f16 || !pool_inv_norm || !queries_f16 || !query_inv_norm || !cand_idx || !exact_ws || !out_scores ||
        !out_ids)
        return UNIIR_EINVAL;
    if (rows <= 0 || nq <= 0 || ncand <= 0 || k <= 0) return UNIIR_EINVAL;
    if (dim % 8) return UNIIR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    const long pairs = (long)nq * ncand;
    if (dim % 64 == 0)
        hipLaunchKernelGGL(rescore_coalesced_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st,
                           (const unsigned short*)pool_f16, pool_inv_norm, (const unsigned short*)queries_f16,
                           query_inv_norm, nq, dim, cand_idx, ncand, exact_ws);
    else
        hipLaunchKernelGGL(rescore_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st,
                           (const unsigned short*)pool_f16, pool_inv_norm, (const unsigned short*)queries_f16,
                           query_inv_norm, nq, dim, cand_idx, ncand, exact_ws);
    hipLaunchKernelGGL(final_sort_kernel, dim3(nq), dim3(256), 0, st, exact_ws, cand_idx,
                       (const long long*)pool_ids, nq, ncand, k, out_scores, (long long*)out_ids);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// k-way merge of per-shard final results (score desc, id asc); ids are unique across shards.
// merge_rank_kernel (round 5): one 256-thread workgroup per query, the nshard * kin entries in LDS, every live entry counts the
// entries that beat it; rank < k is its place.  The chain kernel below (one THREAD per query walking all lists k times, every step a
// dependent global load) took 194 us for 64 queries x 8 lists of 10 and 279 us for 256 -- a quarter of a 256-query sweep over
// eight sub-shards; it remains for merges of more than MERGE_CAP entries per query.
#define MERGE_CAP 4096
__global__ __launch_bounds__(256) void merge_rank_kernel(const float* __restrict__ s, const long long* __restrict__ ids, int nshard,
                                                         int nq, int kin, int k, float* __restrict__ os, long long* __restrict__ oi) {
    __shared__ float ls[MERGE_CAP];
    __shared__ long long li[MERGE_CAP];
    __shared__ int nlive;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n = nshard * kin;
    if (tid == 0) nlive = 0;
    __syncthreads();
    int mine = 0;
    for (int e = tid; e < n; e += 256) {
        const int sh = e / kin, c = e - sh * kin;
        const long o = ((long)sh * nq + q) * kin + c;
        const long long id = ids[o];
        ls[e] = id >= 0 ? s[o] : -INFINITY;
        li[e] = id;
        mine += id >= 0 ? 1 : 0;
    }
    if (mine) atomicAdd(&nlive, mine);
    __syncthreads();
    for (int e = tid; e < n; e += 256) {
        const long long id = li[e];
        if (id < 0) continue;
        const float v = ls[e];
        int rank = 0;
        for (int t = 0; t < n; ++t) {
            const long long oid = li[t];
            const float ov = ls[t];
            rank += (oid >= 0 && (ov > v || (ov == v && oid < id))) ? 1 : 0;
        }
        if (rank < k) { os[(long)q * k + rank] = v; oi[(long)q * k + rank] = id; }
    }
    for (int t = nlive + tid; t < k; t += 256) { os[(long)q * k + t] = -INFINITY; oi[(long)q * k + t] = -1; }      // FAISS padding
}
__global__ __launch_bounds__(256) void merge_shards_kernel(const float* __restrict__ s, const long long* __restrict__ ids,
                                                           int nshard, int nq, int kin, int k, float* __restrict__ os,
                                                           long long* __restrict__ oi) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    float last_s = INFINITY; long long last_id = -1;
    for (int j = 0; j < k; ++j) {
        float bs = -INFINITY; long long bid = 0x7fffffffffffffffLL; bool found = false;
        for (int sh = 0; sh < nshard; ++sh)
            for (int c = 0; c < kin; ++c) {
                const long o = ((long)sh * nq + q) * kin + c;
                const long long id = ids[o];
                if (id < 0) continue;
                const float v = s[o];
                const bool after = (v < last_s) || (v == last_s && id > last_id);
                if (!after) continue;
                if (!found || v > bs || (v == bs && id < bid)) { bs = v; bid = id; found = true; }
            }
        if (found) { os[(long)q * k + j] = bs; oi[(long)q * k + j] = bid; last_s = bs; last_id = bid; }
        else { os[(long)q * k + j] = -INFINITY; oi[(long)q * k + j] = -1; last_s = -INFINITY; last_id = 0x7fffffffffffffffLL; }
    }
}
static void launch_merge(const float* scores, const int64_t* ids, int nshard, int nq, int kin, int k, float* out_scores,
                         int64_t* out_ids, hipStream_t st) {
    if ((long)nshard * kin <= MERGE_CAP)
        hipLaunchKernelGGL(merge_rank_kernel, dim3(nq), dim3(256), 0, st, scores, (const long long*)ids, nshard, nq, kin, k, out_scores,
                           (long long*)out_ids);
    else
        hipLaunchKernelGGL(merge_shards_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, scores, (const long long*)ids, nshard, nq,
                           kin, k, out_scores, (long long*)out_ids);
}
extern "C" int uniir_topk_merge(const float* scores, const int64_t* ids, int32_t nshard, int32_t nq, int32_t k,
                                float* out_scores, int64_t* out_ids, void* stream) {
    if (!scores || !ids || !out_scores || !out_ids || nshard <= 0 || nq <= 0 || k <= 0) return UNIIR_EINVAL;
    launch_merge(scores, ids, nshard, nq, k, k, out_scores, out_ids, (hipStream_t)stream);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
// the same with lists of k_in entries per shard merged into the k_out best (k_out may exceed k_in: a large-k search
// assembled from slices of the pool, uniir_amd/retrieval.py)
extern "C" int uniir_topk_merge_ex(const float* scores, const int64_t* ids, int32_t nshard, int32_t nq, int32_t k_in,
                                   int32_t k_out, float* out_scores, int64_t* out_ids, void* stream) {
    if (!scores || !ids || !out_scores || !out_ids || nshard <= 0 || nq <= 0 || k_in <= 0 || k_out <= 0) return UNIIR_EINVAL;
    launch_merge(scores, ids, nshard, nq, k_in, k_out, out_scores, out_ids, (hipStream_t)stream);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// -------------------------------------------------------------------------------------------------------------
// Fused tail of a group-max search (round 3): after the scan, TWO launches instead of four (query norms, group selection, exact
// re-score, sort) and no separate pass over the queries.
//   topk_tail_select_rescore_kernel, grid (nq, PARTS), 1024 threads: every workgroup of a query runs the register-resident group
//     selection on the query's row of group maxima (the PARTS copies are redundant on purpose: 175 KB from L2 per copy buys a
//     four times wider re-score), while wave 0 computes the query's inverse norm -- the oracle's sequential fp32 chain -- inside the
//     round trip of the selection's loads; then the workgroup re-scores the groups of rank == part (mod PARTS) exactly
//     (rescore_wave: the arithmetic of rescore_coalesced_kernel, the oracle's summation order) and writes exact[q][slot] /
//     cand[q][slot].  With PARTS = 4 the 64 queries of the interactive regime occupy all 256 CUs, ~1.3 re-score waves per CU.
//   topk_tail_sort_kernel, grid nq, 256 threads: rank by counting over the <= 1024 slots held in LDS, ids fetched for the k winners only.
#define TKT_THREADS 1024
DEVINL float rescore_wave(const unsigned short* __restrict__ pool, const unsigned short* __restrict__ qr, float iq, float ic,
                          int ci, int dim, char* mine, int lane) {
    // lane -> (row 8 i + lane / 8 of the wave's 64 rows, 16-B piece lane % 8) for the cooperative 128-byte gathers
    int rows8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rows8[i] = __shfl(ci, 8 * i + (lane >> 3), 64);
    const int piece = lane & 7;
    u32x4_t pre[8];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u32x4_t z = {0u, 0u, 0u, 0u};
            pre[i] = rows8[i] >= 0 ? *reinterpret_cast<const u32x4_t*>(pool + (long)rows8[i] * dim + c0 + piece * 8) : z;
        }
    };
    float s = 0.f;
    fetch(0);
    for (int c0 = 0; c0 < dim; c0 += 64) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<u32x4_t*>(mine + (8 * i + (lane >> 3)) * RSC_PITCH + piece * 16) = pre[i];
        __builtin_amdgcn_wave_barrier();
        if (c0 + 64 < dim) fetch(c0 + 64);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u32x4_t a = *reinterpret_cast<const u32x4_t*>(qr + c0 + 8 * u);
            const u32x4_t b = *reinterpret_cast<const u32x4_t*>(mine + lane * RSC_PITCH + u * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float qa = f16_to_f32((unsigned short)(a[e] & 0xffffu)), ca = f16_to_f32((unsigned short)(b[e] & 0xffffu));
                float qb = f16_to_f32((unsigned short)(a[e] >> 16)), cb = f16_to_f32((unsigned short)(b[e] >> 16));
                if (iq != 0.f) { qa = __fmul_rn(qa, iq); qb = __fmul_rn(qb, iq); }
                if (ic != 0.f) { ca = __fmul_rn(ca, ic); cb = __fmul_rn(cb, ic); }
                s = __fadd_rn(s, __fmul_rn(qa, ca));
                s = __fadd_rn(s, __fmul_rn(qb, cb));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    return s;
}

// The same exact re-score with the candidate rows gathered by LDS-DMA (buffer_load_dwordx4 ... lds) into a wave-private ring of
// DEPTH slices (64 rows x 128 bytes = 8 KiB each): DEPTH - 1 slices are in flight while one is walked, at no register cost -- the
// register-staged gather above exposes one HBM round trip per 128-byte slice (12 per row: ~2 us each, 25 us per re-score).  Layout:
// DMA instruction i covers rows 8 i + (lane >> 3); position p of a row holds its 16-byte piece p ^ ((row >> 1) & 7) (source-side
// swizzle, conflict-free for the row-per-lane ds_read_b128).  qn = the query already scaled by its inverse norm, fp32, in LDS
// (computed once per workgroup: the oracle's qn[j]); the candidate's scaling by ic is applied unconditionally (ic == 0 only for
// an all-zero row, where x * 0 == x bit for bit).  Same summation order as the oracle: one sequential fp32 chain per pair.
template <int DEPTH>
DEVINL float rescore_wave_dma(__amdgpu_buffer_rsrc_t rp, unsigned qn32, float ic, int ci, int dim, char* ring, int lane) {
    const unsigned ring32 = lds_addr32(ring);
    unsigned vo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rloc = 8 * i + (lane >> 3);
        const int r = __shfl(ci, rloc, 64);
        const unsigned piece = (unsigned)((lane & 7) ^ ((rloc >> 1) & 7));
        vo[i] = r >= 0 ? (unsigned)r * (unsigned)(dim * 2) + piece * 16u : 0xffffff00u;       // out of bounds -> zeros
    }
    const unsigned rd = ring32 + lane * 128;                       // this lane's row inside a slice
    const unsigned sw = (unsigned)((lane >> 1) & 7);
    // slice t >= nsl: the same 8 instructions with out-of-bounds offsets (zeros, no memory traffic) -- the DMA count per trip stays
    // constant, so one counted vmcnt serves the whole loop and no separate tail code exists
    auto issue = [&](int t, int slot, bool real) {
        char* dst = ring + slot * 8192;
        const unsigned so = real ? (unsigned)t * 128u : 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (void __attribute__((address_space(3)))*)(dst + i * 1024), 16,
                                                     real ? vo[i] : 0xffffff00u, so, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    float s = 0.f;
    auto consume = [&](int t, int slot) {
        const unsigned base = rd + slot * 8192;
        u32x4_t b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = asm_ds_read_b128<0>(base + (((unsigned)u ^ sw) << 4));
        const unsigned qa = qn32 + (unsigned)t * 256u;             // 64 floats of qn per slice
        u32x4_t q0 = asm_ds_read_b128<0>(qa), q1 = asm_ds_read_b128<16>(qa);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm_wait_lgkm<0>();
            const f32x4_t a0 = __builtin_bit_cast(f32x4_t, q0), a1 = __builtin_bit_cast(f32x4_t, q1);
            const u32x4_t bb = b[u];
            if (u < 7) {
                q0 = asm_ds_read_b128<0>(qa + (u + 1) * 32);
                q1 = asm_ds_read_b128<16>(qa + (u + 1) * 32);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ca = __fmul_rn(f16_to_f32((unsigned short)(bb[e] & 0xffffu)), ic);
                const float cb = __fmul_rn(f16_to_f32((unsigned short)(bb[e] >> 16)), ic);
                const float x0 = e < 2 ? a0[2 * e] : a1[2 * e - 4], x1 = e < 2 ? a0[2 * e + 1] : a1[2 * e - 3];
                s = __fadd_rn(s, __fmul_rn(x0, ca));
                s = __fadd_rn(s, __fmul_rn(x1, cb));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const int nsl = dim / 64;
#pragma unroll
    for (int t = 0; t < DEPTH - 1; ++t) issue(t, t, t < nsl);
    int slot = 0;
    for (int t = 0; t < nsl; ++t) {
        issue(t + DEPTH - 1, slot == 0 ? DEPTH - 1 : slot - 1, t + DEPTH - 1 < nsl);       // slot (t + DEPTH - 1) % DEPTH
        tkr_wait_vm<(DEPTH - 1) * 8>();
        consume(t, slot);
        slot = slot == DEPTH - 1 ? 0 : slot + 1;
    }
    tkr_wait_vm<0>();            // the trailing dummies: nothing may be in flight towards the ring when the caller reuses it
    return s;
}

// PARTS workgroups per query; RW waves of a workgroup re-score at a time, each with a DEPTH-slice gather ring (RW x DEPTH x 8 KiB)
template <int PARTS, int RW, int DEPTH>
__global__ __launch_bounds__(TKT_THREADS) void topk_tail_select_rescore_kernel(
    const unsigned short* __restrict__ pool, const float* __restrict__ pinv, long rows, int dim,
    const unsigned short* __restrict__ queries, const float* __restrict__ gmax, long ngroups, int kc, int gcap,
    int* __restrict__ cand, float* __restrict__ exact, const float* __restrict__ wmax, int nw, TkMulti mu, int k) {
    if (mu.per > 0) {        // batched tail: blockIdx.z = sub-shard (rows [z per, ..) of one resident pool), everything re-based onto it
        const long z = blockIdx.z;
        pool += z * mu.per * dim;
        pinv += z * mu.per;
        rows = min(mu.per, mu.rows_total - z * mu.per);
        ngroups = (rows + TK_G - 1) / TK_G;
        gmax += z * mu.g_stride;
        if (wmax) wmax += z * mu.w_stride;
        cand += z * mu.c_stride;
        exact += z * mu.c_stride;
    }
    extern __shared__ __attribute__((aligned(16))) char dyn[];       // the gather rings
    __shared__ int sel[2 * TK_MAXKC * TK_G];                         // the selection's output: gcap * 16 row indices, -1 = empty
    __shared__ __attribute__((aligned(16))) unsigned short qrow[4096];
    __shared__ __attribute__((aligned(16))) float qn[4096];
    __shared__ float s_iq;
    __shared__ int s_nkept;
    const int q = blockIdx.x, part = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned short* qr = queries + (long)q * dim;
    // the query's inverse norm (FAISS fvec_renorm_L2 as restated by the oracle: sequential fp32 sum of squares, no fma; see
    // inv_norm_kernel) by lane 0 of wave 0, from a wave-private LDS copy of the query, while the selection's loads are in flight
    bool normed = false;
    auto qnorm = [&] {
        if (normed) return;              // (the selection may run twice: the hierarchical one can bail out)
        normed = true;
        if (w != 0) return;
        // the 768 squares in parallel (64 lanes), then ONE lane adds them in the oracle's order: the only serial part is the chain of
        // fp32 additions (the all-in-one-lane form of this took ~7 us -- conversions, multiplies and LDS latency inside the chain --
        // and was the critical path of the kernel's first phase).  qn[] doubles as the buffer of squares until the scaling below.
        for (int c = lane; c < dim / 8; c += 64) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(qr + c * 8);
            *reinterpret_cast<u32x4_t*>(qrow + c * 8) = v;
            f32x4_t a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = f16_to_f32((unsigned short)(v[e] & 0xffffu));
                const float hi = f16_to_f32((unsigned short)(v[e] >> 16));
                const float l2 = __fmul_rn(lo, lo), h2 = __fmul_rn(hi, hi);
                if (e < 2) { a[2 * e] = l2; a[2 * e + 1] = h2; } else { b[2 * e - 4] = l2; b[2 * e - 3] = h2; }
            }
            *reinterpret_cast<f32x4_t*>(qn + c * 8) = a;
            *reinterpret_cast<f32x4_t*>(qn + c * 8 + 4) = b;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            float s = 0.f;
            for (int c = 0; c < dim; c += 16) {          // dim % 64 == 0
                const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(qn + c), x1 = *reinterpret_cast<const f32x4_t*>(qn + c + 4);
                const f32x4_t x2 = *reinterpret_cast<const f32x4_t*>(qn + c + 8), x3 = *reinterpret_cast<const f32x4_t*>(qn + c + 12);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __fadd_rn(s, x0[e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __fadd_rn(s, x1[e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __fadd_rn(s, x2[e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __fadd_rn(s, x3[e]);
            }
            s_iq = s > 0.f ? (float)(1.0 / (double)(float)sqrt((double)s)) : 0.f;
        }
    };
    // the selection ends with a barrier; behind the streaming scans it starts from the per-wave maxima (gsel_hier)
    // candidates: every group within the proven rounding bound of the k-th best group maximum (GselBound, topk_select.h); they fill a
    // prefix of the slots
    GselBound gb;
    gb.k = k;
    gb.slack_cos = tk_slack_cos(dim);
    gb.iq = &s_iq;
    gb.nkept = &s_nkept;
    if (!(wmax && gsel_hier<TKT_THREADS>(gmax + (long)q * ngroups, wmax + (long)q * nw, nw, ngroups, rows, kc, gcap, sel, qnorm, &gb)))
        gsel_body<TKT_THREADS, true>(gmax + (long)q * ngroups, ngroups, rows, kc, gcap, sel, qnorm, &gb);
    const int nkept = s_nkept;
    // the slots behind the kept groups stay empty for the sort (this workgroup writes the -1 / -inf of its share of them below)
    const int ngrp = (nkept - part + PARTS - 1) / PARTS;     // this workgroup's share: groups of rank part, part + PARTS, ...
    const int nth = ngrp * TK_G;                             // thread t -> member t % 16 of its (t / 16)-th group
    {
        const float iq = s_iq;                 // the normalised query, once per workgroup (the oracle's qn[j])
        for (int j = tid; j < dim; j += TKT_THREADS) {
            const float v = f16_to_f32(qrow[j]);
            qn[j] = iq != 0.f ? __fmul_rn(v, iq) : v;
        }
    }
    __syncthreads();
    if (part == 0)                            // the slots no group was kept for: empty for the sort
        for (int e = nkept * TK_G + tid; e < gcap * TK_G; e += TKT_THREADS) {
            cand[(long)q * gcap * TK_G + e] = -1;
            exact[(long)q * gcap * TK_G + e] = -INFINITY;
        }
    if (w >= RW) return;                                              // no barrier follows
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)pool, 0, (int)(rows * dim * 2), 0x00020000);
    const unsigned qn32 = lds_addr32(reinterpret_cast<const char*>(qn));
    for (int base = 0; base + w * 64 < nth; base += RW * 64) {
        const int t = base + tid;
        const int slot = t < nth ? ((t >> 4) * PARTS + part) * TK_G + (t & 15) : -1;
        const int ci = slot >= 0 ? sel[slot] : -1;
        const float ic = ci >= 0 ? pinv[ci] : 0.f;
        const float sc = rescore_wave_dma<DEPTH>(rp, qn32, ic, ci, dim, dyn + w * (DEPTH * 8192), lane);
        if (slot >= 0) {
            const long o = (long)q * gcap * TK_G + slot;
            cand[o] = ci;
            exact[o] = ci >= 0 ? sc : -INFINITY;
        }
    }
}

// one workgroup per query: the live slots are compacted into LDS (about half of the gcap * 16 slots are: the groups beyond the
// kc-th best and its ties stay empty), then every live entry counts the entries that beat it (score desc, id asc); rank < k is the
// answer.  1024 threads: one entry per thread, ~n / 16 trips of 16 compares each (the 256-thread, uncompacted form of this kernel
// took 22 us: 576^2 compares on four waves).
#define TKT_SORTCAP (2 * TK_MAXKC * TK_G)     // 2048 slots at most (gcap <= 128 groups)
__global__ __launch_bounds__(1024) void topk_tail_sort_kernel(const float* __restrict__ exact, const int* __restrict__ cand,
                                                             const long long* __restrict__ ids, int ncand, int k,
                                                             float* __restrict__ out_s, long long* __restrict__ out_i, TkMulti mu) {
    if (mu.per > 0) {        // batched tail: blockIdx.y = sub-shard; its list goes to out[y][q][k]
        const long z = blockIdx.y;
        exact += z * mu.c_stride;
        cand += z * mu.c_stride;
        if (ids) ids += z * mu.per;
        out_s += z * mu.o_stride;
        out_i += z * mu.o_stride;
    }
    __shared__ __attribute__((aligned(16))) float ls[TKT_SORTCAP + 16];
    __shared__ int lrow[TKT_SORTCAP];
    __shared__ int nlive;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const float* es = exact + (long)q * ncand;
    const int* ci = cand + (long)q * ncand;
    if (tid == 0) nlive = 0;
    __syncthreads();
    for (int c0 = 0; c0 < ncand; c0 += 1024) {           // <= 2 trips; wave-uniform trip count
        const int c = c0 + tid;
        const int row = c < ncand ? ci[c] : -1;
        const float sc = row >= 0 ? es[c] : 0.f;
        const unsigned long long m = __ballot(row >= 0);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&nlive, __popcll(m));
        base = __shfl(base, 0, 64);
        if (row >= 0) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            ls[pos] = sc;
            lrow[pos] = row;
        }
    }
    __syncthreads();
    const int n = nlive, n16 = (n + 15) & ~15;
    if (tid < 16) ls[n + tid] = -INFINITY;                // pad to the 16-wide compare loop (never better than, never equal to a live score)
    __syncthreads();
    for (int e = tid; e < n; e += 1024) {
        const float sc = ls[e];
        int rank = 0, ties = 0;
        for (int u = 0; u < n16; u += 16) {
            f32x4_t v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4_t*>(ls + u + 4 * j);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    rank += v[j][r] > sc ? 1 : 0;
                    ties += v[j][r] == sc ? 1 : 0;
                }
        }
        if (rank >= k) continue;
        const int row = lrow[e];
        const long long id = ids ? ids[row] : (long long)row;
        if (ties > 1) {                                   // exact score ties (duplicate rows): the id decides
            for (int u = 0; u < n; ++u)
                if (u != e && ls[u] == sc) rank += (ids ? ids[lrow[u]] : (long long)lrow[u]) < id ? 1 : 0;
        }
        if (rank < k) {
            out_s[(long)q * k + rank] = sc;
            out_i[(long)q * k + rank] = id;
        }
    }
    for (int t = n + tid; t < k; t += 1024) {             // FAISS pads missing results with -inf distance / id -1
        out_s[(long)q * k + t] = -INFINITY;
        out_i[(long)q * k + t] = -1;
    }
}

// can the fused tail (selection + query norm + exact re-score | sort) serve this search?
bool fused_tail_ok(int64_t rows, int32_t dim, int32_t kc) {
    const long ngroups = (rows + TK_G - 1) / TK_G;
    const int gcap = TK_GMULT * kc;
    if (dim % 64 || dim > 4096 || ngroups % 2 || ngroups > 1024L * 2 * TK_SELREG || gcap * TK_G > TKT_SORTCAP || gcap > 2 * TK_MAXKC)
        return false;
    return rows * dim * 2 < (1L << 31);                                // the gather's 31-bit buffer bound
}
// selection + exact re-score + sort behind a finished scan (dense gmax [+ wave maxima]); false when the shape does not fit the
// fused kernels
bool launch_fused_tail(const void* pool_f16, const float* pinv, const int64_t* pool_ids, int64_t rows, int32_t dim,
                              const void* queries_f16, int32_t nq, int32_t kc, int32_t k, const float* gmax, int32_t* cand,
                              float* exact, float* out_scores, int64_t* out_ids, hipStream_t st, const float* wmax, int nw,
                              const TkMulti* mu, int nsub) {
    // rows: of the (first) sub-shard -- in a batched tail every sub-shard but the last has this many (the caller checked the last one)
    const long ngroups = (rows + TK_G - 1) / TK_G;
    const int gcap = TK_GMULT * kc;
    if (!fused_tail_ok(rows, dim, kc)) return false;
    TkMulti one = {};
    if (!mu) { mu = &one; nsub = 1; }
    // one workgroup per (query, part, sub-shard): the interactive regime spreads a query's re-score over PARTS workgroups; with
    // several sub-shards in one launch the sub-shards already fill the chip
    const int parts = nq * nsub <= 64 ? 4 : nq * nsub <= 128 ? 2 : 1;
    const dim3 g(nq, parts, nsub), b(TKT_THREADS);
#define TKT_LAUNCH(P, RW, DEPTH)                                                                                       \
    do {                                                                                                               \
        static PerDeviceOnce attr;                                                                                     \
        if (attr.first())                                                                                              \
            (void)hipFuncSetAttribute((const void*)topk_tail_select_rescore_kernel<P, RW, DEPTH>,                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, RW * DEPTH * 8192);                  \
        hipLaunchKernelGGL((topk_tail_select_rescore_kernel<P, RW, DEPTH>), g, b, RW * DEPTH * 8192, st,               \
                           (const unsigned short*)pool_f16, pinv, (long)rows, dim, (const unsigned short*)queries_f16, \
                           gmax, ngroups, kc, gcap, cand, exact, nw > 0 ? wmax : nullptr, nw, *mu, k);                 \
    } while (0)
    // rings: 3 waves x 4 slices (the interactive regime: 144 slots per workgroup at k = 10) or 5 waves x 2 slices = 96 / 80 KiB
    if (parts == 4) TKT_LAUNCH(4, 3, 4);
    else if (parts == 2) TKT_LAUNCH(2, 5, 2);
    else TKT_LAUNCH(1, 5, 2);
#undef TKT_LAUNCH
    hipLaunchKernelGGL(topk_tail_sort_kernel, dim3(nq, nsub), dim3(1024), 0, st, exact, cand, (const long long*)pool_ids,
                       gcap * TK_G, k, out_scores, (long long*)out_ids, *mu);
    return true;
}
