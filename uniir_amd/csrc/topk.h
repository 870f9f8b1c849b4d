// Shared by topk.hip (scans, selection kernels, the one-call search) and topk_tail.hip (exact re-score, sorts, merges, fused tail).
#pragma once
#include "gemm_core.h"
#include "gemm_core256.h"
#include "gemm_core_pp.h"
#include "../../include/uniir_hip.h"
#include <stdlib.h>
#include <type_traits>

#define TK_QT 128        // queries per block tile (GEMM N)
#define TK_CT 256        // candidates per MFMA tile (GEMM M): pool rows stream through the 256-row LDS-DMA operand
#define TK_CAP 512       // candidate buffer entries per (block, query) (>= 2 * TK_CT)
#define TK_MAXKC 64
#define TK_RANKCAP 1024    // final sort by rank counting up to this many shortlist entries

struct TkEntry { float score; int idx; };

#define TK_G 16              // pool rows per group (one MFMA row tile): the scans keep one maximum per (query, group)
#define TK_GMULT 2            // groups kept per query = TK_GMULT * kc
#define TK_GPATH_MAXQ 1024

// counted wait on the wave's own LDS-DMA / load queue (the streaming scans and the gather rings order themselves with it)
template <int N>
DEVINL void tkr_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N));
    __builtin_amdgcn_sched_barrier(0);
}

// topk_tail.hip: can the fused tail (selection + query norm + exact re-score | sort) serve this search / launch it behind a finished
// scan (dense gmax [+ wave maxima]); false when the shape does not fit the fused kernels
// Batched tail (round 5, uniir_topk_ip_multi): ONE tail launch + ONE sort launch serve the scans of all sub-shards of a resident pool
// that is larger than the 31-bit buffer bound -- workgroup (query, part, z) works on sub-shard z = rows [z per, min((z + 1) per,
// rows_total)) with its own descriptors (every sub-shard stays below 2 GiB), its own group / wave maxima and candidate regions; the
// sort writes one k-list per (sub-shard, query) and uniir_topk_merge's kernel combines them.  per == 0: a plain single-shard tail.
struct TkMulti {
    long per;          // rows per sub-shard (a multiple of 32)
    long rows_total;
    long g_stride;     // floats between the sub-shards' group-maxima regions  [nq][ngroups of a full sub-shard]
    long w_stride;     // floats between their wave-maxima regions
    long c_stride;     // entries between their cand / exact regions           [nq][gcap * 16]
    long o_stride;     // entries between their output lists                   [nq][k]
};
bool fused_tail_ok(int64_t rows, int32_t dim, int32_t kc);
bool launch_fused_tail(const void* pool_f16, const float* pinv, const int64_t* pool_ids, int64_t rows, int32_t dim,
                       const void* queries_f16, int32_t nq, int32_t kc, int32_t k, const float* gmax, int32_t* cand,
                       float* exact, float* out_scores, int64_t* out_ids, hipStream_t st, const float* wmax, int nw,
                       const TkMulti* mu = nullptr, int nsub = 1);
