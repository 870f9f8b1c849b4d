// Group selection of the group-max search (device templates): used by topk_gsel_kernel (topk.hip) and by the fused tail kernel
// (topk_tail.hip).
#pragma once
#include "topk.h"

// one block per query: the best groups by (group max desc, group index asc): the kc best plus ties of the kc-th
// value (at most gcap groups); every member row of those groups becomes a re-score candidate.
//   pass 1: per-thread maxima -> tau0 = kc-th largest of the 256 thread maxima (a valid lower bound of the kc-th
//           largest group value: that many distinct groups reach it)
//   pass 2: groups >= tau0 are collected in LDS (a few dozen), ranked exactly, the best gcap kept.
// If the collection overflows (massive exact ties) the kernel falls back to one-extraction-per-round selection.
// The rigorous candidate rule of the fused tail (round 5).  The scan's group maxima are APPROXIMATE scores (fp16 x fp16 products are
// exact in fp32, the MFMA chain and the scaling by the row's inverse norm round: |approx - true| <= d_a = 793 x 2^-23 of |q| in the
// scan's units, which leave the query un-normalised); the tail's re-score is the oracle's fp32 chain (|exact - true| <= d_e = 771 x
// 2^-24 in cosine units).  Let tau_k be the k-th largest group maximum: k distinct rows reach it, so the k-th best EXACT score is at
// least tau_k - (d_a + d_e), and a row of the exact top-k has an approximate score -- hence a group maximum -- of at least
// tau_k - 2 (d_a + d_e) = tau_k - 2.8e-4 |q|.  Every group within TK_SLACK_COS = 5e-4 (x |q|) of tau_k becomes a candidate, nothing
// else: ~11 groups at k = 10 on random data instead of the fixed k + 8 = 18 (which is no bound at all: nine near-ties within the
// rounding error would defeat it) -- 40 % fewer rows gathered and re-scored.  More than gcap = 2 (k + 8) qualifying groups (massive
// near-ties: duplicated rows) keep the best gcap, as before.  The kept groups occupy a PREFIX of the slots; *nkept is their number.
// Both error bounds grow with the length of the fp32 chain: d_a = (dim + 25) x 2^-23, d_e = (dim + 3) x 2^-24.  The slack follows dim
// (a constant 5e-4 stops being a proven bound above dim ~ 1300 while the fused tail accepts dim <= 4096): 1.78 x the bound
// 2 (d_a + d_e) -- the margin 5e-4 / 2.81e-4 of the 768 case -- and never below the 5e-4 that dims <= 768 were measured with.
#define TK_SLACK_COS 5.0e-4f
static inline __host__ __device__ float tk_slack_cos(int dim) {
    const float bound = 2.f * ((float)(dim + 25) * 1.1920929e-7f + (float)(dim + 3) * 5.9604645e-8f);
    const float s = 1.78f * bound;
    return s > TK_SLACK_COS ? s : TK_SLACK_COS;
}
struct GselBound {
    int k;              // entries wanted
    float slack_cos;    // tk_slack_cos(dim)
    const float* iq;    // LDS: the query's inverse norm, written by the caller's mid() (read after the barrier that follows it)
    int* nkept;         // LDS: groups kept
};
#define TK_SELCAP 1024
#define TK_SELREG 24    // float2 loads per thread of the register-resident variant: ngroups <= 1024 * 2 * 24
// REG (the interactive <= 64-query path, one 1024-thread block per query): the query's group maxima are read ONCE, as
// back-to-back 8-byte loads that all stay in flight, and both passes (thread maxima, collection above the threshold) run
// on registers -- the two dependent strided passes over global memory were 40 of the kernel's 62 us at 700 k rows.
// `out` (gcap * TK_G row indices, -1 = empty) may be global memory (topk_gsel_kernel) or LDS (the fused tail kernel); every
// thread of the block returns from this function (no early exit: the fused kernel goes on to the re-score).
template <int BS, bool REG, class F>
DEVINL void gsel_body(const float* __restrict__ g, long ngroups, long rows, int kc, int gcap, int* out, F&& mid,
                      const GselBound* gb = nullptr) {
    const int kk = gb ? gb->k : kc;          // the rank whose value anchors the thresholds
    __shared__ float tmax[BS];
    __shared__ float bval[TK_SELCAP];
    __shared__ int bgrp[TK_SELCAP];
    __shared__ int bcnt;
    __shared__ float tau0, tau;
    __shared__ float ss[BS / 64];
    __shared__ long long si[BS / 64];
    __shared__ float wsel;
    __shared__ long long isel;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int e = tid; e < gcap * TK_G; e += BS) out[e] = -1;
    float mx = -INFINITY;
    f32x2_t rv[REG ? TK_SELREG : 1];
    if (REG) {
#pragma unroll
        for (int it = 0; it < TK_SELREG; ++it) {
            const long e = 2L * (it * BS + tid);
            rv[it] = e < ngroups ? *reinterpret_cast<const f32x2_t*>(g + e) : f32x2_t{-INFINITY, -INFINITY};   // ngroups is even
        }
        mid();      // independent work of the caller that rides the round trip of the loads above (the fused tail: the query norm)
#pragma unroll
        for (int it = 0; it < TK_SELREG; ++it) mx = fmaxf(mx, fmaxf(rv[it][0], rv[it][1]));
    } else {
        mid();
        for (long e = tid; e < ngroups; e += BS) mx = fmaxf(mx, g[e]);
    }
    if (tid == 0) { bcnt = 0; tau0 = -INFINITY; tau = -INFINITY; if (gb) *gb->nkept = 0; }
    if (BS == 1024) {
        // threshold = the kc-th largest of the 64 quarter-wave maxima (disjoint subsets, so at least kc entries reach it;
        // ~20-30 entries do at kc = 18).  Ranking all 1024 thread maxima against each other was 1 M compares per block
        // -- 40 of the kernel's 50 us.
        const float qm = row16_max(mx);
        if ((tid & 15) == 0) tmax[tid >> 4] = qm;
        __syncthreads();
        if (tid < 64) {
            const float v = tmax[tid];
            int rank = 0;
            for (int t = 0; t < 64; ++t) {
                const float o = tmax[t];
                rank += (o > v || (o == v && t < tid)) ? 1 : 0;
            }
            if (rank == min(kk, 64) - 1) tau0 = v;
        }
    } else {
        tmax[tid] = mx;
        __syncthreads();
        int rank = 0;
        for (int t = 0; t < BS; ++t) {
            const float o = tmax[t];
            rank += (o > mx || (o == mx && t < tid)) ? 1 : 0;
        }
        if (rank == kk - 1) tau0 = mx;   // exactly one thread has this rank
    }
    __syncthreads();
    // (the barrier above also publishes *gb->iq, written inside mid())
    const float slack = (gb && *gb->iq > 0.f) ? gb->slack_cos / *gb->iq : 0.f;      // in the scan's units (scores x |q|)
    const float t0 = tau0 - slack;
    if (REG) {
#pragma unroll
        for (int it = 0; it < TK_SELREG; ++it)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float v = rv[it][h];
                if (v >= t0 && v > -INFINITY) {
                    const int pos = atomicAdd(&bcnt, 1);
                    if (pos < TK_SELCAP) { bval[pos] = v; bgrp[pos] = (int)(2L * (it * BS + tid) + h); }
                }
            }
    } else {
        for (long e = tid; e < ngroups; e += BS) {
            const float v = g[e];
            if (v >= t0 && v > -INFINITY) {
                const int pos = atomicAdd(&bcnt, 1);
                if (pos < TK_SELCAP) { bval[pos] = v; bgrp[pos] = (int)e; }
            }
        }
    }
    __syncthreads();
    const int n = bcnt;
    if (n <= TK_SELCAP) {
        // exact rank of every collected entry by (value desc, group asc)
        for (int e = tid; e < n; e += BS) {
            const float v = bval[e];
            const int gi = bgrp[e];
            int rank = 0;
            for (int t = 0; t < n; ++t) {
                const float o = bval[t];
                const int og = bgrp[t];
                rank += (o > v || (o == v && og < gi)) ? 1 : 0;
            }
            if (rank == min(kk, n) - 1) tau = v;
        }
        __syncthreads();
        const float tt = tau - slack;
        for (int e = tid; e < n; e += BS) {
            const float v = bval[e];
            const int gi = bgrp[e];
            int rank = 0;
            for (int t = 0; t < n; ++t) {
                const float o = bval[t];
                const int og = bgrp[t];
                rank += (o > v || (o == v && og < gi)) ? 1 : 0;
            }
            if (rank < gcap && v >= tt) {
                for (int m = 0; m < TK_G; ++m) {
                    const long row = (long)gi * TK_G + m;
                    out[rank * TK_G + m] = row < rows ? (int)row : -1;
                }
                if (gb) atomicMax(gb->nkept, rank + 1);
            }
        }
    } else {
    // fallback: one extraction per round (value desc, group asc), stop after the kc-th value's ties or gcap groups
    float last_s = INFINITY, tk = -INFINITY;
    long long last_g = -1;
    for (int j = 0; j < gcap; ++j) {
        float bs = -INFINITY;
        long long bg = 0x7fffffffffffffffLL;
        for (long e = tid; e < ngroups; e += BS) {
            const float v = g[e];
            const bool after = (v < last_s) || (v == last_s && (long long)e > last_g);
            if (after && (v > bs || (v == bs && (long long)e < bg))) { bs = v; bg = e; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o, 64);
            const long long og = __shfl_xor(bg, o, 64);
            if (os > bs || (os == bs && og < bg)) { bs = os; bg = og; }
        }
        if (lane == 0) { ss[w] = bs; si[w] = bg; }
        __syncthreads();
        if (tid == 0) {
            float fs = ss[0]; long long fg = si[0];
            for (int k = 1; k < BS / 64; ++k) if (ss[k] > fs || (ss[k] == fs && si[k] < fg)) { fs = ss[k]; fg = si[k]; }
            wsel = fs; isel = fg;
        }
        __syncthreads();
        last_s = wsel; last_g = isel;
        __syncthreads();
        if (last_g == 0x7fffffffffffffffLL || last_s == -INFINITY) break;
        if (j == kk - 1) tk = last_s - slack;
        if (j >= kk && last_s < tk) break;
        if (tid < TK_G) {
            const long row = last_g * TK_G + tid;
            out[j * TK_G + tid] = row < rows ? (int)row : -1;
        }
        if (gb && tid == 0) *gb->nkept = j + 1;
    }
    }
    __syncthreads();
}

// Hierarchical selection behind the stream2 scan (<= 64 queries): the scan also leaves the maximum of every WAVE's range of groups
// (wmax[q][nw], nw <= 1024 waves of ~43 consecutive groups).  One value per thread instead of 48: the threshold comes from the
// wave maxima (the kc-th largest quarter-wave maximum: at least kc waves, hence at least kc groups, reach it), the ~20 waves that
// reach it hand in their groups (~900 values), the ~20 of those above the threshold are ranked exactly.  Same output as gsel_body
// (both collect every group >= a valid lower bound of the kc-th best value and rank the collection by (value desc, group asc)).
// Returns false (workgroup-uniform, nothing written but the -1 fill) when a cap overflows: the caller then runs gsel_body.
#define TK_HWAVES 128        // candidate waves kept
template <int BS, class F>
DEVINL bool gsel_hier(const float* __restrict__ g, const float* __restrict__ wm, int nw, long ngroups, long rows, int kc, int gcap,
                      int* out, F&& mid, const GselBound* gb = nullptr) {
    const int kk = gb ? gb->k : kc;
    __shared__ float qmax[64];
    __shared__ int cwave[TK_HWAVES];
    __shared__ float hval[TK_SELCAP];
    __shared__ int hgrp[TK_SELCAP];
    __shared__ int ccnt, hcnt;
    __shared__ float htau0, htau;
    const int tid = threadIdx.x;
    for (int e = tid; e < gcap * TK_G; e += BS) out[e] = -1;
    const float v = tid < nw ? wm[tid] : -INFINITY;
    mid();
    if (tid == 0) { ccnt = 0; hcnt = 0; htau0 = -INFINITY; htau = -INFINITY; if (gb) *gb->nkept = 0; }
    const float qm = row16_max(v);
    if ((tid & 15) == 0) qmax[tid >> 4] = qm;
    __syncthreads();
    if (tid < 64) {
        const float x = qmax[tid];
        int rank = 0;
        for (int t = 0; t < 64; ++t) {
            const float o = qmax[t];
            rank += (o > x || (o == x && t < tid)) ? 1 : 0;
        }
        if (rank == min(kk, 64) - 1) htau0 = x;
    }
    __syncthreads();
    const float slack = (gb && *gb->iq > 0.f) ? gb->slack_cos / *gb->iq : 0.f;      // (*gb->iq: published by the barriers above)
    const float t0 = htau0 - slack;
    if (v >= t0 && v > -INFINITY) {
        const int pos = atomicAdd(&ccnt, 1);
        if (pos < TK_HWAVES) cwave[pos] = tid;
    }
    __syncthreads();
    const int nc = ccnt;
    if (nc > TK_HWAVES) return false;
    // 16 candidate waves per trip: thread -> (candidate tid / 64, group lo + tid % 64) -- a wave's range holds <= 64 groups
    for (int c0 = 0; c0 < nc; c0 += BS / 64) {
        const int c = c0 + (tid >> 6);
        if (c < nc) {
            const long wv = cwave[c];
            const long lo = wv * ngroups / nw, hi = (wv + 1) * ngroups / nw;
            const long e = lo + (tid & 63);
            if (e < hi) {
                const float x = g[e];
                if (x >= t0 && x > -INFINITY) {
                    const int pos = atomicAdd(&hcnt, 1);
                    if (pos < TK_SELCAP) { hval[pos] = x; hgrp[pos] = (int)e; }
                }
            }
        }
    }
    __syncthreads();
    const int n = hcnt;
    if (n > TK_SELCAP) return false;
    for (int e = tid; e < n; e += BS) {
        const float x = hval[e];
        const int gi = hgrp[e];
        int rank = 0;
        for (int t = 0; t < n; ++t) {
            const float o = hval[t];
            const int og = hgrp[t];
            rank += (o > x || (o == x && og < gi)) ? 1 : 0;
        }
        if (rank == min(kk, n) - 1) htau = x;
    }
    __syncthreads();
    const float tt = htau - slack;
    for (int e = tid; e < n; e += BS) {
        const float x = hval[e];
        const int gi = hgrp[e];
        int rank = 0;
        for (int t = 0; t < n; ++t) {
            const float o = hval[t];
            const int og = hgrp[t];
            rank += (o > x || (o == x && og < gi)) ? 1 : 0;
        }
        if (rank < gcap && x >= tt) {
            for (int m = 0; m < TK_G; ++m) {
                const long row = (long)gi * TK_G + m;
                out[rank * TK_G + m] = row < rows ? (int)row : -1;
            }
            if (gb) atomicMax(gb->nkept, rank + 1);
        }
    }
    __syncthreads();
    return true;
}
