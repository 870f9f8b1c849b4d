// Fused multi-head attention forward / backward for the CLIP towers (head_dim 64; seq 257 / 77 / 50).
// Replaces nn.MultiheadAttention's softmax(QK^T/sqrt(64) [+ causal mask])V inside openai/CLIP
// ResidualAttentionBlock.attention (model.py), reached from clip_sf.py:43-47.
//
// One workgroup (4 waves) per (item, head); the whole K/V (fwd) or Q/dO then K/V (bwd) of that head sits
// in LDS ([row][64] bf16, 128-B rows, 16-B chunk index XOR (row & 7): conflict-free both for the
// ds_read_b128 fragment reads and for the ds_read_b64_tr_b16 transposing reads).
//
// Everything is computed "transposed" so that softmax statistics are lane-local:
//   S^T = K Q^T   -> lane holds S[q = lane&15][key = 4*(lane>>4) + r]   (v_mfma_f32_16x16x32_bf16)
//   O^T = V^T P^T -> A operand = V^T via transposing LDS reads, B operand = P^T straight from the S^T
//                    accumulators (k-slots permuted consistently on both operands).
#include "attention.h"
#ifdef UNIIR_EXP_BUILD
unsigned long long* g_att_stamps = nullptr;
int g_att_exp = 0;
extern "C" void uniir_exp_attn_set(void* stamps, int mode) { g_att_stamps = (unsigned long long*)stamps; g_att_exp = mode; }
#endif

// F16: q / k / v / out are fp16 instead of bf16 (plain or causal forward of the fp16 embedding towers; P <= 2^8 by the deferred
// maximum, well inside fp16's range)
template <bool REL, bool DROP, bool F16 = false>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int H = a.H, causal = a.causal;
    const int m = blockIdx.x / H, h = blockIdx.x % H;
    // packed rows (a.row_off): item m owns the rows row_off[m] .. row_off[m + 1] - 1 of every tensor (its own length); the
    // statistics keep the dense [item][head][a.Tq] layout
    int Tq = a.Tq, Tk = a.Tk;
    long qr0 = (long)m * Tq, kr0 = (long)m * Tk;
    if (a.row_off) {
        qr0 = a.row_off[m];
        Tq = a.row_off[m + 1] - a.row_off[m];
        if (!a.row_off_q_only) { kr0 = qr0; Tk = Tq; }      // (q_only: rectangular cross-attention, K / V stay dense per item)
    }
    if (a.kv_row_off) {
        kr0 = a.kv_row_off[m];
        Tk = a.kv_row_off[m + 1] - a.kv_row_off[m];
    }
    const int Tkp = (Tk + 31) & ~31;
    char* ldsK = lds;
    char* ldsV = lds + Tkp * 128;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar loop control
    const unsigned short* qbase = a.q + qr0 * a.q_ld + h * ATT_D;
    const unsigned short* kbase = a.k + kr0 * a.kv_ld + h * ATT_D;
    const unsigned short* vbase = a.v + kr0 * a.kv_ld + h * ATT_D;
    const int kvalid = a.klen ? min(Tk, a.klen[m] + a.klen_add) : Tk;
    const float sl2 = REL ? a.scale * LOG2EF : SCALE_LOG2E;
    const unsigned dth = DROP ? drop_threshold(a.drop_p) : 0u;
    const float dks = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    float* dbias = reinterpret_cast<float*>(lds + 2 * Tkp * 128);    // bias of every diagonal key - query (x log2 e)
    if (REL)
        for (int d = tid; d < Tq + Tk - 1; d += ATT_THREADS) dbias[d] = a.rel_emb[a.rel_bucket[d] * H + h] * LOG2EF;
    const int nqt = (Tq + 15) >> 4;
    const int qi = lane & 15, g = lane >> 4;
    // the query fragments of a tile come straight from global memory: the first tile's ride the staging round trip, every
    // further tile's are fetched while the previous tile computes
    bf16x8_t qnext[2];
    qnext[0] = frag_rows_global(qbase, a.q_ld, w * 16, 0, lane, Tq);
    qnext[1] = frag_rows_global(qbase, a.q_ld, w * 16, 1, lane, Tq);
    ATT_STAMP(0);
    if (ATT_EXP(4)) {
    } else if (a.legacy_stage) {
        stage_head<ATT_THREADS>(ldsK, kbase, a.kv_ld, Tk, Tkp, tid);
        stage_head<ATT_THREADS>(ldsV, vbase, a.kv_ld, Tk, Tkp, tid);
    } else {
        stage_two<ATT_THREADS>(ldsK, kbase, a.kv_ld, ldsV, vbase, a.kv_ld, Tk, Tkp, tid);
    }
    ATT_STAMP(1);
    __syncthreads();
    ATT_STAMP(2);
    // (17 tiles over 4 waves are 5 + 4 + 4 + 4 with wave 0 carrying the fifth.  Dealing the tiles from a start that rotates with the
    // head, so that every SIMD hosts a five-tile wave equally often, was measured in round 4: 257 tokens 0.61 -> 0.69-0.74 ms,
    // 197 tokens 0.44 -> 0.52.  The first-dispatched wave of a SIMD wins the issue arbitration; the extra tile belongs there.)
    for (int qt = w; qt < (ATT_EXP(1) ? 0 : nqt); qt += ATT_WAVES) {
        const int q0 = qt * 16, q = q0 + qi;
        bf16x8_t qf[2];
        qf[0] = qnext[0];
        qf[1] = qnext[1];
        if (qt + ATT_WAVES < nqt) {
            qnext[0] = frag_rows_global(qbase, a.q_ld, q0 + 16 * ATT_WAVES, 0, lane, Tq);
            qnext[1] = frag_rows_global(qbase, a.q_ld, q0 + 16 * ATT_WAVES, 1, lane, Tq);
        }
        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        float m_run = -1e30f;
        f32x4_t l4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int kmax = causal ? min(kvalid, q0 + 16) : kvalid;
        const int nkb = max(1, (kmax + 31) >> 5);
        // S^T of a key block is computed one block ahead (its MFMAs overlap the previous block's softmax); the loop is
        // unrolled by two with the two logit buffers swapping roles, so nothing is copied between iterations
        auto s_block = [&](int kb, f32x4_t (&sx)[2]) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                sx[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) sx[kt] = mfma16x<F16>(frag_rows(ldsK, kb * 32 + kt * 16, s, lane), qf[s], sx[kt]);
            }
        };
        auto softmax_pv = [&](int kb, f32x4_t (&st)[2]) {
            // The softmax is VALU-bound (4 waves share a SIMD): logits stay raw (the 1/8 log2 e scale is folded into the
            // exp2 argument with one fma), masks are applied only on blocks that touch a boundary (wave-uniform
            // test), and the reference maximum moves -- and the running output is rescaled -- only when a row needs it.
            const bool edge = kb * 32 + 32 > kvalid || (causal && kb * 32 + 31 > q0);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kb * 32 + kt * 16 + 4 * g + r;
                    if (REL) st[kt][r] = st[kt][r] * sl2 + dbias[min(max(key - q + Tq - 1, 0), Tq + Tk - 2)];
                }
            if (edge) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kb * 32 + kt * 16 + 4 * g + r;
                        if (key >= kvalid || (causal && key > q)) st[kt][r] = -1e30f;
                    }
            }
            // DEFERRED MAXIMUM (round 4).  A row's reference m_run only has to be common to the row's four lanes and close enough to
            // the true maximum that exp2 cannot overflow; it moves only when one of the ROW'S OWN logits exceeds it by more than
            // ATT_DEFER (2^8: P <= 256, bf16 keeps its relative precision, the fp32 sums have room).  The test is lane-local (4
            // v_max3 / v_max, one fma, one compare); the ballot folds the four lane groups into one bit per row on the SCALAR unit, and only
            // a wave with a flagged row runs the cross-lane maximum and the rescale -- after its first block practically never
            // (round 3 took both on 85-100 % of the blocks: with 16 rows per wave SOME row's maximum nearly always moved).  The
            // decision is per row, so a row's result does not depend on which rows share its tile (the packed text tower relies on it).
            float mx = -1e30f;           // (this chain form compiles to four v_max3; a balanced tree gets canonicalising self-maxima)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
            const float es = REL ? 1.0f : sl2;
            const unsigned long long over = __ballot(__builtin_fmaf(mx, es, -m_run) > ATT_DEFER);
            unsigned rows_over = (unsigned)(over | (over >> 32));
            rows_over = (rows_over | (rows_over >> 16)) & 0xffffu;
            if (rows_over) {
                const float gm = group_max(mx);
                const float m_new = ((rows_over >> qi) & 1u) ? fmaxf(m_run, gm * es) : m_run;
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // exactly 1 for the rows that keep their reference
                l4 = l4 * alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = o[dt] * alpha;
                m_run = m_new;
            }
            // masked logits (-1e30) underflow to exactly 0; key 0 is valid for every row (klen >= 1, causal includes the
            // diagonal), so m_run is finite from the first block on.  Vector expressions: v_pk_fma_f32 / v_pk_add_f32.
            // The row sum stays a per-lane vector of four partials until the tile ends.
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const f32x4_t arg = __builtin_elementwise_fma(st[kt], f32x4_t{es, es, es, es}, f32x4_t{-m_run, -m_run, -m_run, -m_run});
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kt][r] = __builtin_amdgcn_exp2f(arg[r]);
                l4 = l4 + st[kt];
            }
            if (DROP) {
                const unsigned rowbase = (unsigned)((((long)m * H + h) * a.Tq + q) * a.Tk);      // dense coordinates: packed rows draw the dense call's mask
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        st[kt][r] *= drop_scale(rowbase + (unsigned)(kb * 32 + kt * 16 + 4 * g + r), a.drop_seed, dth, dks);
            }
            const bf16x8_t pf = pack8x<F16>(st[0], st[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16x<F16>(frag_cols_tr(ldsV, kb * 32, dt, lane), pf, o[dt]);
        };
        f32x4_t sa[2], sb[2];
        s_block(0, sa);
        for (int kb = 0; kb < nkb; kb += 2) {
            if (kb + 1 < nkb) s_block(kb + 1, sb);
            softmax_pv(kb, sa);
            if (kb + 1 >= nkb) break;
            if (kb + 2 < nkb) s_block(kb + 2, sa);
            softmax_pv(kb + 1, sb);
        }
        const float l_run = group_sum((l4[0] + l4[1]) + (l4[2] + l4[3]));
        // rows leave as 16-byte pieces, 64 contiguous bytes per row and store (att_store_tile; round 4: the 8-byte stores at a
        // row stride cost the 257-token forward 0.14 of its 0.70 ms)
        const bool live = q < Tq && !ATT_EXP(16);
        att_store_tile<F16>(o, 1.0f / l_run, a.out + (qr0 + min(q, Tq - 1)) * a.out_ld + h * ATT_D, live, g);
        if (live && g == 0) a.lse[((long)m * H + h) * a.Tq + q] = m_run * LN2F + __logf(l_run);
    }
    ATT_STAMP(3);
}

// Backward. Phase 1 (per 16-key tile): dV^T += dO^T P, dK^T += Q^T dS with S = Q K^T oriented [q][key].
// Phase 2 (per 16-query tile): dQ^T += K^T dS^T with S^T = K Q^T oriented [key][q].
// CAUSAL (CLIP text tower, 77 tokens): compile-time; its short loops are mostly boundary tiles, so masks are always on
// NT threads: 512 (8 waves, 128 registers) or 384 (6 waves, 168 registers: no spills, and 5 key tiles of a 77-token head keep 5 of
// 6 waves busy instead of 5 of 8) -- the launcher picks 384 up to 128 tokens (measured: 50 tokens 0.335 -> 0.261 ms, 77 causal
// 0.518 -> 0.386 ms at 1024 items; 197 / 257 tokens lose 15-20 % with fewer waves).
template <bool REL, bool DROP, bool CAUSAL, int NT>
__global__ __launch_bounds__(NT, NT == 512 ? 4 : 3) void attn_bwd_kernel(AttnArgs a) {
    constexpr int NWAVES = NT / 64;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int H = a.H;
    constexpr bool causal = CAUSAL;
    const int m = blockIdx.x / H, h = blockIdx.x % H;
    int Tq = a.Tq, Tk = a.Tk;              // packed rows: see attn_fwd_kernel
    long qr0 = (long)m * Tq, kr0 = (long)m * Tk;
    if (a.row_off) {
        qr0 = a.row_off[m];
        Tq = a.row_off[m + 1] - a.row_off[m];
        if (!a.row_off_q_only) { kr0 = qr0; Tk = Tq; }
    }
    if (a.kv_row_off) {
        kr0 = a.kv_row_off[m];
        Tk = a.kv_row_off[m + 1] - a.kv_row_off[m];
    }
    const int Tqp = (Tq + 31) & ~31, Tkp = (Tk + 31) & ~31;
    const int Tmax = max(Tqp, Tkp);
    char* bufA = lds;                 // Q, later K
    char* bufB = lds + Tmax * 128;    // dO, later V
    float* lse2 = reinterpret_cast<float*>(lds + 2 * Tmax * 128);
    float* Dq = lse2 + Tqp;
    float* dbias = Dq + Tqp;               // [Tq + Tk - 1] bias per diagonal (x log2 e), only with rel_emb
    float* ddiag = dbias + (Tq + Tk);      // [Tq + Tk - 1] gradient per diagonal
    const float sl2 = REL ? a.scale * LOG2EF : SCALE_LOG2E;
    const float oscale = REL ? a.scale : ATT_SCALE;
    const unsigned dth = DROP ? drop_threshold(a.drop_p) : 0u;
    const float dks = DROP ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar loop control
    const unsigned headbase = (unsigned)((((long)m * H + h) * a.Tq) * a.Tk);      // dense coordinates (see attn_fwd_kernel)
    const unsigned short* qbase = a.q + qr0 * a.q_ld + h * ATT_D;
    const unsigned short* kbase = a.k + kr0 * a.kv_ld + h * ATT_D;
    const unsigned short* vbase = a.v + kr0 * a.kv_ld + h * ATT_D;
    const unsigned short* obase = a.out + qr0 * a.out_ld + h * ATT_D;
    const unsigned short* dobase = a.dout + qr0 * a.out_ld + h * ATT_D;
    unsigned short* dqbase = a.dq + qr0 * a.dq_ld + h * ATT_D;
    unsigned short* dkbase = a.dk + kr0 * a.dkv_ld + h * ATT_D;
    unsigned short* dvbase = a.dv + kr0 * a.dkv_ld + h * ATT_D;
    const int kvalid = a.klen ? min(Tk, a.klen[m] + a.klen_add) : Tk;

    auto row_stats = [&] {          // D[q] = dO[q] . O[q], lse2[q] = lse[q] * log2 e; the bias / gradient rows of a T5 head
        for (int r = tid; r < Tqp; r += NT) {
            float d = 0.f, l = 0.f;
            if (r < Tq) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const u32x4_t x = *reinterpret_cast<const u32x4_t*>(obase + (long)r * a.out_ld + c * 8);
                    const u32x4_t y = *reinterpret_cast<const u32x4_t*>(dobase + (long)r * a.out_ld + c * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        d += __uint_as_float(x[e] << 16) * __uint_as_float(y[e] << 16);
                        d += __uint_as_float(x[e] & 0xffff0000u) * __uint_as_float(y[e] & 0xffff0000u);
                    }
                }
                l = a.lse[((long)m * H + h) * a.Tq + r] * LOG2EF;
            }
            Dq[r] = d;
            lse2[r] = l;
        }
        if (REL)
            for (int d = tid; d < Tq + Tk - 1; d += NT) {
                dbias[d] = a.rel_emb[a.rel_bucket[d] * H + h] * LOG2EF;
                ddiag[d] = 0.f;
            }
    };
    ATT_STAMP(0);
    if (ATT_EXP(4)) {
    } else if (a.legacy_stage) {
        row_stats();
        stage_head<NT>(bufA, qbase, a.q_ld, Tq, Tqp, tid);
        stage_head<NT>(bufB, dobase, a.out_ld, Tq, Tqp, tid);
    } else {
        stage_two<NT>(bufA, qbase, a.q_ld, bufB, dobase, a.out_ld, Tq, Tqp, tid, row_stats);
    }
    ATT_STAMP(1);
    __syncthreads();
    ATT_STAMP(2);

    const int li = lane & 15, g = lane >> 4;
    const int nktile = (Tk + 15) >> 4, nqtile = (Tq + 15) >> 4;
    const int nqblk = Tqp >> 5;
    // The loops below are issue-bound (VALU + LDS + MFMA of 4 waves share a SIMD), so: the lane parts of the swizzled LDS
    // fragment addresses are computed once (row0 is a multiple of 16 / 32, so (row & 7) is a lane constant and the block
    // offset is a scalar add with immediate sub-offsets); masks are applied only on tiles that touch a boundary
    // (wave-uniform test); the per-query statistics are read as 16-byte vectors.
    int offR[2], offT[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) offR[s] = li * 128 + (((s * 4 + g) ^ (li & 7)) << 4);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const int row = 4 * g + (li >> 2), col = 16 * dt + 4 * (li & 3);
        offT[dt] = row * 128 + ((((col >> 3) ^ (row & 7)) << 4) | ((col & 7) << 1));
    }
    const char* rA[2] = {bufA + offR[0], bufA + offR[1]};
    const int dAB = Tmax * 128;      // bufB - bufA (scalar): the dO / V fragments sit at the same lane offsets
    const char* tA[4] = {bufA + offT[0], bufA + offT[1], bufA + offT[2], bufA + offT[3]};
    auto rows_frag = [&](const char* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(p)); };
    auto cols_frag = [&](const char* p) {     // 32-row block: rows 4g.., 16 + 4g.. of one column tile
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lds_read_tr16(p));
        const u32x2_t h2 = __builtin_bit_cast(u32x2_t, lds_read_tr16(p + 2048));
        const u32x4_t r = {l2[0], l2[1], h2[0], h2[1]};
        return __builtin_bit_cast(bf16x8_t, r);
    };
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    // ---------------- phase 1: dK, dV ----------------
    for (int kt = w; kt < (ATT_EXP(1) ? 0 : nktile); kt += NWAVES) {
        const int k0 = kt * 16, key = k0 + li;
        bf16x8_t kf[2], vf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kf[s] = frag_rows_global(kbase, a.kv_ld, k0, s, lane, Tk);
            vf[s] = frag_rows_global(vbase, a.kv_ld, k0, s, lane, Tk);
        }
        f32x4_t dv[4], dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dv[dt] = zero4;
            dk[dt] = zero4;
        }
        const bool kedge = k0 + 16 > kvalid;
        const int qb0 = causal ? (k0 >> 5) : 0;
        for (int qb = qb0; qb < nqblk; ++qb) {
            const int blk = qb * 4096;
            const bool edge = CAUSAL || kedge || (REL && qb * 32 + 32 > Tq);
            f32x4_t pt[2], dst[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4_t sa = zero4, dp = zero4;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    sa = mfma16(rows_frag(rA[s] + blk + qt * 2048), kf[s], sa);
                    dp = mfma16(rows_frag(rA[s] + dAB + blk + qt * 2048), vf[s], dp);
                }
                // A rows = queries, B cols = keys -> acc[r] = S[q = 4g + r][key = li]
                const int qv = qb * 32 + qt * 16 + 4 * g;
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse2 + qv);
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(Dq + qv);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qv + r;
                    const int dg = min(max(key - q + Tq - 1, 0), Tq + Tk - 2);
                    float p = REL ? __builtin_amdgcn_exp2f(sa[r] * sl2 + dbias[dg] - l4[r])
                                  : __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], sl2, -l4[r]));
                    pt[qt][r] = p;
                }
                if (edge) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key >= kvalid || (causal && key > qv + r) || (REL && qv + r >= Tq)) pt[qt][r] = 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qv + r;
                    const int dg = min(max(key - q + Tq - 1, 0), Tq + Tk - 2);
                    const float p = pt[qt][r];
                    const float mk = DROP ? drop_scale(headbase + (unsigned)q * (unsigned)a.Tk + (unsigned)key, a.drop_seed, dth, dks) : 1.0f;
                    pt[qt][r] = DROP ? p * mk : p;
                    dst[qt][r] = p * ((DROP ? dp[r] * mk : dp[r]) - d4[r]);
                }
                if (REL && a.drel) {
                    // d bias = d logits summed along the diagonals.  An LDS float atomic costs ~155 cycles per 64-lane instruction
                    // (measured: 3.1 of the 4.9 ms of this kernel at 334 tokens were the four per-element atomics), so the 4 x 16
                    // block of this 16-lane row is first summed along its diagonals with DPP row shifts: lane j ends with
                    // diagonal (key_j - (qv + 3)), i.e. d(r = 3, j) + d(2, j - 1) + d(1, j - 2) + d(0, j - 3); the six elements
                    // that fall off the row's right edge (diagonals of lanes 13 .. 15 at r = 0) go into a second value held by
                    // lanes 13 .. 15 only: 64 + 12 lane-atomics instead of 256.
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (key < Tk && qv + r < Tq) ? dst[qt][r] : 0.f;
                    auto shr1 = [](float x) {
                        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true));
                    };
                    float t = v[0];
                    t = v[1] + shr1(t);
                    t = v[2] + shr1(t);
                    t = v[3] + shr1(t);
                    const int dgt = key - (qv + 3) + Tq - 1;
                    if (dgt >= 0 && dgt <= Tq + Tk - 2 && t != 0.f) atomicAdd(&ddiag[dgt], t);
                    const float w1 = li >= 14 ? v[1] : 0.f, w2 = li == 15 ? v[2] : 0.f;
                    float wv = li >= 13 ? v[0] : 0.f;
                    wv += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, w1), 0x101, 0xf, 0xf, true));   // row_shl:1
                    wv += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, w2), 0x102, 0xf, 0xf, true));   // row_shl:2
                    const int dgw = key - qv + Tq - 1;
                    if (li >= 13 && dgw >= 0 && dgw <= Tq + Tk - 2 && wv != 0.f) atomicAdd(&ddiag[dgw], wv);
                }
            }
            const bf16x8_t pf = pack8(pt[0], pt[1]), dsf = pack8(dst[0], dst[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = mfma16(cols_frag(tA[dt] + dAB + blk), pf, dv[dt]);
                dk[dt] = mfma16(cols_frag(tA[dt] + blk), dsf, dk[dt]);
            }
        }
        {
            const bool live = key < Tk && !ATT_EXP(16);
            const long krow = (long)min(key, Tk - 1) * a.dkv_ld;
            att_store_tile(dk, oscale, dkbase + krow, live, g);
            att_store_tile(dv, 1.0f, dvbase + krow, live, g);
        }
    }
    ATT_STAMP(3);
    __syncthreads();
    ATT_STAMP(4);
    if (ATT_EXP(8)) {
    } else if (a.legacy_stage) {
        stage_head<NT>(bufA, kbase, a.kv_ld, Tk, Tkp, tid);
        stage_head<NT>(bufB, vbase, a.kv_ld, Tk, Tkp, tid);
    } else {
        stage_two<NT>(bufA, kbase, a.kv_ld, bufB, vbase, a.kv_ld, Tk, Tkp, tid);
    }
    ATT_STAMP(5);
    __syncthreads();
    ATT_STAMP(6);
    // ---------------- phase 2: dQ ----------------
    for (int qt = w; qt < (ATT_EXP(2) ? 0 : nqtile); qt += NWAVES) {
        const int q0 = qt * 16, q = q0 + li;
        bf16x8_t qf[2], dof[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[s] = frag_rows_global(qbase, a.q_ld, q0, s, lane, Tq);
            dof[s] = frag_rows_global(dobase, a.out_ld, q0, s, lane, Tq);
        }
        const float my_lse = lse2[q0 + li], my_D = Dq[q0 + li];
        f32x4_t dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = zero4;
        const int kmax = causal ? min(kvalid, q0 + 16) : kvalid;
        const int nkb = (kmax + 31) >> 5;
        for (int kb = 0; kb < nkb; ++kb) {
            const int blk = kb * 4096;
            const bool edge = CAUSAL || kb * 32 + 32 > kvalid;
            f32x4_t dst[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x4_t sa = zero4, dp = zero4;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    sa = mfma16(rows_frag(rA[s] + blk + kt * 2048), qf[s], sa);
                    dp = mfma16(rows_frag(rA[s] + dAB + blk + kt * 2048), dof[s], dp);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kb * 32 + kt * 16 + 4 * g + r;
                    float p = REL ? __builtin_amdgcn_exp2f(sa[r] * sl2 + dbias[min(max(key - q + Tq - 1, 0), Tq + Tk - 2)] - my_lse)
                                  : __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], sl2, -my_lse));
                    dst[kt][r] = p;
                }
                if (edge) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kb * 32 + kt * 16 + 4 * g + r;
                        if (key >= kvalid || (causal && key > q)) dst[kt][r] = 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kb * 32 + kt * 16 + 4 * g + r;
                    const float mk = DROP ? drop_scale(headbase + (unsigned)q * (unsigned)a.Tk + (unsigned)key, a.drop_seed, dth, dks) : 1.0f;
                    dst[kt][r] = dst[kt][r] * ((DROP ? dp[r] * mk : dp[r]) - my_D);
                }
            }
            const bf16x8_t dsf = pack8(dst[0], dst[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma16(cols_frag(tA[dt] + blk), dsf, dq[dt]);
        }
        att_store_tile(dq, oscale, dqbase + (long)min(q, Tq - 1) * a.dq_ld, q < Tq && !ATT_EXP(16), g);
    }
    ATT_STAMP(7);
    if (REL && a.drel) {      // diagonals -> buckets -> global (one atomic per touched bucket and workgroup)
        __syncthreads();
        float* bsum = lse2;     // phase 2 is over: reuse
        for (int b = tid; b < a.nbuckets; b += NT) bsum[b] = 0.f;
        __syncthreads();
        for (int d = tid; d < Tq + Tk - 1; d += NT) atomicAdd(&bsum[a.rel_bucket[d]], ddiag[d]);
        __syncthreads();
        for (int b = tid; b < a.nbuckets; b += NT)
            if (bsum[b] != 0.f) atomicAdd(a.drel + b * H + h, bsum[b]);
    }
}

// Staging policy (A/B on one MI355X, 1024 items, tools/microbench.py MB_ONLY=attn; legacy -> batched):
//   forward : 257 tokens 0.783 -> 0.760 ms, 197: 0.488 -> 0.487, 77 causal: 0.146 -> 0.132, 50: 0.116 -> 0.095   => always batched
//   backward: 257 tokens 2.026 -> 2.021 ms, 197: 1.399 -> 1.446 (worse), 77: 0.388 -> 0.359, 50: 0.336 -> 0.294   => batched up to 128 tokens
static int attn_legacy_stage(bool backward, int tmax) { return backward && tmax > 128; }
static int launch_attn_fwd(const AttnArgs& a0, int batch, hipStream_t st, bool f16 = false) {
    AttnArgs a = a0;
#ifdef UNIIR_EXP_BUILD
    a.stamps = g_att_stamps; a.exp = g_att_exp;
#endif
    a.legacy_stage = attn_legacy_stage(false, a.Tk);
    const int Tkp = (a.Tk + 31) & ~31;
    const int sm = 2 * Tkp * 128 + (a.rel_emb ? (a.Tq + a.Tk) * 4 : 0);
    static PerDeviceOnce attr;
    if (attr.first()) {
        const int big = 2 * 512 * 128 + 1024 * 4;
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    }
    const dim3 g(batch * a.H), b(ATT_THREADS);
    const bool drop = a.drop_p > 0.f;
    if (f16) {          // the fp16 forward of the embedding towers: plain / causal / packed rows only
        if (a.rel_emb || drop) return UNIIR_EUNSUPPORTED;
        static PerDeviceOnce attr16;
        if (attr16.first())
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      2 * 512 * 128 + 1024 * 4);
        hipLaunchKernelGGL((attn_fwd_kernel<false, false, true>), g, b, sm, st, a);
        HIP_LAUNCH_CHECK();
        return UNIIR_OK;
    }
    if (a.rel_emb && drop) hipLaunchKernelGGL((attn_fwd_kernel<true, true>), g, b, sm, st, a);
    else if (a.rel_emb) hipLaunchKernelGGL((attn_fwd_kernel<true, false>), g, b, sm, st, a);
    else if (drop) hipLaunchKernelGGL((attn_fwd_kernel<false, true>), g, b, sm, st, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, false>), g, b, sm, st, a);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
template <int NT>
static int launch_attn_bwd_nt(const AttnArgs& a, int batch, int sm, hipStream_t st) {
    static PerDeviceOnce attr;
    if (attr.first()) {
        const int big = 2 * 512 * 128 + 2 * 512 * 4 + 2 * 1024 * 4;
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<false, false, false, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<false, true, false, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<true, false, false, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<true, true, false, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<false, false, true, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<false, true, true, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    }
    const dim3 g(batch * a.H), b(NT);
    const bool drop = a.drop_p > 0.f;
    if (a.causal && a.rel_emb) return UNIIR_ESHAPE;
    if (a.causal && drop) hipLaunchKernelGGL((attn_bwd_kernel<false, true, true, NT>), g, b, sm, st, a);
    else if (a.causal) hipLaunchKernelGGL((attn_bwd_kernel<false, false, true, NT>), g, b, sm, st, a);
    else if (a.rel_emb && drop) hipLaunchKernelGGL((attn_bwd_kernel<true, true, false, NT>), g, b, sm, st, a);
    else if (a.rel_emb) hipLaunchKernelGGL((attn_bwd_kernel<true, false, false, NT>), g, b, sm, st, a);
    else if (drop) hipLaunchKernelGGL((attn_bwd_kernel<false, true, false, NT>), g, b, sm, st, a);
    else hipLaunchKernelGGL((attn_bwd_kernel<false, false, false, NT>), g, b, sm, st, a);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
int launch_attn_bwd_pair(const AttnArgs& a, int batch, hipStream_t st);      // attention_pair.hip; 1 = shape not taken
static int launch_attn_bwd(const AttnArgs& a0, int batch, hipStream_t st) {
    AttnArgs a = a0;
#ifdef UNIIR_EXP_BUILD
    a.stamps = g_att_stamps; a.exp = g_att_exp;
#endif
    {   // plain self-attention of 193 .. 288 tokens (CLIP ViT-L/14, BLIP ViT): the persistent pair-tile kernel
        const int rc = launch_attn_bwd_pair(a, batch, st);
        if (rc != 1) return rc;
    }
    const int tmax = a.Tq > a.Tk ? a.Tq : a.Tk;
    a.legacy_stage = attn_legacy_stage(true, tmax);
    const int Tqp = (a.Tq + 31) & ~31, Tkp = (a.Tk + 31) & ~31;
    const int Tmax = Tqp > Tkp ? Tqp : Tkp;
    const int sm = 2 * Tmax * 128 + 2 * Tqp * 4 + (a.rel_emb ? 2 * (a.Tq + a.Tk) * 4 : 0);
    bool six = tmax <= 128;
    if (six && (Tmax * 8 + 383) / 384 > 8) six = false;      // stage_two holds <= 8 loads per thread and slice: 384 threads stop at 384 tokens
    return six ? launch_attn_bwd_nt<384>(a, batch, sm, st) : launch_attn_bwd_nt<512>(a, batch, sm, st);
}

// tower.hip's forward: row_off = nullptr for dense rows; f16 = the 16-bit tensors are fp16
int attention_fwd_impl(const void* qkv, void* out, float* lse, const int32_t* row_off, int32_t batch, int32_t seq, int32_t heads,
                       int32_t causal, int f16, void* stream) {
    if (!qkv || !out || !lse || batch < 0 || heads <= 0) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (seq < 1 || seq > 512) return UNIIR_ESHAPE;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) return UNIIR_EALIGN;
    const long W = (long)heads * ATT_D;
    AttnArgs a = {};
    a.q = (const unsigned short*)qkv; a.k = a.q + W; a.v = a.q + 2 * W;
    a.q_ld = a.kv_ld = 3 * W;
    a.out = (unsigned short*)out; a.out_ld = W; a.lse = lse; a.klen = nullptr; a.row_off = row_off;
    a.Tq = a.Tk = seq; a.H = heads; a.causal = causal; a.scale = ATT_SCALE;
    return launch_attn_fwd(a, batch, (hipStream_t)stream, f16 != 0);
}
// tower.hip's last block when only one row per item is pooled (the class token / the EOT row): ONE query row per item ([batch][W],
// dense) against the item's keys and values.  K / V are columns of the block's qkv buffer (row stride kv_ld): dense [batch][tk]
// rows with an optional key count klen[m] + klen_add (the causal text tower: the EOT index + 1), or the item's own packed rows
// (kv_row_off).  A query row's result does not depend on the rows that share its tile, so out equals that row of the full call.
int attention_pooled_fwd_impl(const void* q, const void* k, const void* v, int64_t kv_ld, void* out, float* lse, const int32_t* klen,
                              int32_t klen_add, const int32_t* kv_row_off, int32_t batch, int32_t tk, int32_t heads, int f16,
                              void* stream) {
    if (!q || !k || !v || !out || !lse || batch < 0 || heads <= 0) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (tk < 1 || tk > 512 || (kv_ld % 8)) return UNIIR_ESHAPE;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)out & 15)) return UNIIR_EALIGN;
    const long W = (long)heads * ATT_D;
    AttnArgs a = {};
    a.q = (const unsigned short*)q; a.k = (const unsigned short*)k; a.v = (const unsigned short*)v;
    a.q_ld = W; a.kv_ld = kv_ld; a.out = (unsigned short*)out; a.out_ld = W; a.lse = lse;
    a.klen = klen; a.klen_add = klen_add; a.kv_row_off = kv_row_off;
    a.Tq = 1; a.Tk = tk; a.H = heads; a.causal = 0; a.scale = ATT_SCALE;
    return launch_attn_fwd(a, batch, (hipStream_t)stream, f16 != 0);
}
int attention_pooled_bwd_impl(const void* q, const void* k, const void* v, int64_t kv_ld, const void* out, const void* dout,
                              const float* lse, const int32_t* klen, int32_t klen_add, const int32_t* kv_row_off, void* dq,
                              int64_t dq_ld, void* dk, void* dv, int64_t dkv_ld, int32_t batch, int32_t tk, int32_t heads, void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || batch < 0 || heads <= 0) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (tk < 1 || tk > 512 || (kv_ld % 8) || (dkv_ld % 8) || (dq_ld % 8)) return UNIIR_ESHAPE;
    if (((uintptr_t)dq & 15) || ((uintptr_t)dk & 15) || ((uintptr_t)dv & 15)) return UNIIR_EALIGN;
    const long W = (long)heads * ATT_D;
    AttnArgs a = {};
    a.q = (const unsigned short*)q; a.k = (const unsigned short*)k; a.v = (const unsigned short*)v;
    a.q_ld = W; a.kv_ld = kv_ld; a.out = (unsigned short*)out; a.out_ld = W; a.lse = const_cast<float*>(lse);
    a.klen = klen; a.klen_add = klen_add; a.kv_row_off = kv_row_off;
    a.Tq = 1; a.Tk = tk; a.H = heads; a.causal = 0; a.scale = ATT_SCALE;
    a.dout = (const unsigned short*)dout;
    a.dq = (unsigned short*)dq; a.dk = (unsigned short*)dk; a.dv = (unsigned short*)dv;
    a.dq_ld = dq_ld; a.dkv_ld = dkv_ld;
    return launch_attn_bwd(a, batch, (hipStream_t)stream);
}

extern "C" int uniir_attention_fwd(const void* qkv, void* out, float* lse, int32_t batch, int32_t seq,
                                   int32_t heads, int32_t causal, void* stream) {
    if (!qkv || !out || !lse || batch < 0 || heads <= 0) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (seq < 1 || seq > 512) return UNIIR_ESHAPE;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) return UNIIR_EALIGN;
    const long W = (long)heads * ATT_D;
    AttnArgs a = {};
    a.q = (const unsigned short*)qkv; a.k = a.q + W; a.v = a.q + 2 * W;
    a.q_ld = a.kv_ld = 3 * W;
    a.out = (unsigned short*)out; a.out_ld = W; a.lse = lse; a.klen = nullptr;
    a.Tq = a.Tk = seq; a.H = heads; a.causal = causal; a.scale = ATT_SCALE;
    return launch_attn_fwd(a, batch, (hipStream_t)stream);
}

extern "C" int uniir_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                                   void* dqkv, int32_t batch, int32_t seq, int32_t heads, int32_t causal,
                                   void* stream) {
    if (!qkv || !out || !dout || !lse || !dqkv || batch < 0 || heads <= 0) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (seq < 1 || seq > 512) return UNIIR_ESHAPE;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15))
        return UNIIR_EALIGN;
    const long W = (long)heads * ATT_D;
    AttnArgs a = {};
    a.q = (const unsigned short*)qkv; a.k = a.q + W; a.v = a.q + 2 * W;
    a.q_ld = a.kv_ld = 3 * W;
    a.out = (unsigned short*)out; a.out_ld = W; a.lse = const_cast<float*>(lse); a.klen = nullptr;
    a.Tq = a.Tk = seq; a.H = heads; a.causal = causal; a.scale = ATT_SCALE;
    a.dout = (const unsigned short*)dout;
    a.dq = (unsigned short*)dqkv; a.dk = a.dq + W; a.dv = a.dq + 2 * W;
    a.dq_ld = a.dkv_ld = 3 * W;
    return launch_attn_bwd(a, batch, (hipStream_t)stream);
}

// packed rows: item m owns rows row_off[m] .. row_off[m + 1] - 1 of qkv / out (lengths <= max_seq); lse stays [batch][heads][max_seq].
// The CLIP text tower on the live rows of its captions only (rows behind a caption's EOT never reach its pooled feature under the
// causal mask, clip_sf.py:43-44): every live row's result is bitwise that of the dense call.
extern "C" int uniir_attention_fwd_packed(const void* qkv, void* out, float* lse, const int32_t* row_off, int32_t batch,
                                          int32_t max_seq, int32_t heads, int32_t causal, void* stream) {
    if (!qkv || !out || !lse || !row_off || batch < 0 || heads <= 0) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (max_seq < 1 || max_seq > 512) return UNIIR_ESHAPE;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) return UNIIR_EALIGN;
    const long W = (long)heads * ATT_D;
    AttnArgs a = {};
    a.q = (const unsigned short*)qkv; a.k = a.q + W; a.v = a.q + 2 * W;
    a.q_ld = a.kv_ld = 3 * W;
    a.out = (unsigned short*)out; a.out_ld = W; a.lse = lse; a.klen = nullptr; a.row_off = row_off;
    a.Tq = a.Tk = max_seq; a.H = heads; a.causal = causal; a.scale = ATT_SCALE;
    return launch_attn_fwd(a, batch, (hipStream_t)stream);
}
extern "C" int uniir_attention_bwd_packed(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                          const int32_t* row_off, int32_t batch, int32_t max_seq, int32_t heads, int32_t causal,
                                          void* stream) {
    if (!qkv || !out || !dout || !lse || !dqkv || !row_off || batch < 0 || heads <= 0) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (max_seq < 1 || max_seq > 512) return UNIIR_ESHAPE;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15)) return UNIIR_EALIGN;
    const long W = (long)heads * ATT_D;
    AttnArgs a = {};
    a.q = (const unsigned short*)qkv; a.k = a.q + W; a.v = a.q + 2 * W;
    a.q_ld = a.kv_ld = 3 * W;
    a.out = (unsigned short*)out; a.out_ld = W; a.lse = const_cast<float*>(lse); a.klen = nullptr; a.row_off = row_off;
    a.Tq = a.Tk = max_seq; a.H = heads; a.causal = causal; a.scale = ATT_SCALE;
    a.dout = (const unsigned short*)dout;
    a.dq = (unsigned short*)dqkv; a.dk = a.dq + W; a.dv = a.dq + 2 * W;
    a.dq_ld = a.dkv_ld = 3 * W;
    return launch_attn_bwd(a, batch, (hipStream_t)stream);
}

// general form: separate Q and K/V tensors (cross-attention), optional per-item key length (padding mask)
extern "C" int uniir_attention_fwd_ex(const void* q, int64_t q_ld, const void* k, const void* v, int64_t kv_ld,
                                      void* out, int64_t out_ld, float* lse, const int32_t* key_len, int32_t batch,
                                      int32_t tq, int32_t tk, int32_t heads, int32_t causal, float drop_p,
                                      uint32_t drop_seed, void* stream) {
    if (!q || !k || !v || !out || !lse || batch < 0 || heads <= 0) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (tq < 1 || tk < 1 || tq > 512 || tk > 512) return UNIIR_ESHAPE;
    if (causal && tq != tk) return UNIIR_ESHAPE;
    if ((q_ld % 8) || (kv_ld % 8) || (out_ld % 8)) return UNIIR_EALIGN;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)out & 15)) return UNIIR_EALIGN;
    AttnArgs a = {};
    a.q = (const unsigned short*)q; a.k = (const unsigned short*)k; a.v = (const unsigned short*)v;
    a.q_ld = q_ld; a.kv_ld = kv_ld; a.out = (unsigned short*)out; a.out_ld = out_ld; a.lse = lse; a.klen = key_len;
    a.Tq = tq; a.Tk = tk; a.H = heads; a.causal = causal; a.scale = ATT_SCALE;
    if (drop_p < 0.f || drop_p >= 1.f) return UNIIR_EINVAL;
    a.drop_p = drop_p; a.drop_seed = drop_seed;
    return launch_attn_fwd(a, batch, (hipStream_t)stream);
}

extern "C" int uniir_attention_bwd_ex(const void* q, int64_t q_ld, const void* k, const void* v, int64_t kv_ld,
                                      const void* out, const void* dout, int64_t out_ld, const float* lse,
                                      const int32_t* key_len, void* dq, int64_t dq_ld, void* dk, void* dv,
                                      int64_t dkv_ld, int32_t batch, int32_t tq, int32_t tk, int32_t heads,
                                      int32_t causal, float drop_p, uint32_t drop_seed, void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || batch < 0 || heads <= 0) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (tq < 1 || tk < 1 || tq > 512 || tk > 512) return UNIIR_ESHAPE;
    if (causal && tq != tk) return UNIIR_ESHAPE;
    if ((q_ld % 8) || (kv_ld % 8) || (out_ld % 8) || (dq_ld % 8) || (dkv_ld % 8)) return UNIIR_EALIGN;      // 16-byte pieces
    if (((uintptr_t)dq & 15) || ((uintptr_t)dk & 15) || ((uintptr_t)dv & 15)) return UNIIR_EALIGN;
    AttnArgs a = {};
    a.q = (const unsigned short*)q; a.k = (const unsigned short*)k; a.v = (const unsigned short*)v;
    a.q_ld = q_ld; a.kv_ld = kv_ld; a.out = (unsigned short*)out; a.out_ld = out_ld;
    a.lse = const_cast<float*>(lse); a.klen = key_len;
    a.Tq = tq; a.Tk = tk; a.H = heads; a.causal = causal; a.scale = ATT_SCALE;
    a.dout = (const unsigned short*)dout;
    a.dq = (unsigned short*)dq; a.dk = (unsigned short*)dk; a.dv = (unsigned short*)dv;
    a.dq_ld = dq_ld; a.dkv_ld = dkv_ld;
    if (drop_p < 0.f || drop_p >= 1.f) return UNIIR_EINVAL;
    a.drop_p = drop_p; a.drop_seed = drop_seed;
    return launch_attn_bwd(a, batch, (hipStream_t)stream);
}

// The general form on PACKED QUERY ROWS (the BLIP MED BERT on the tokens up to each caption's valid length only): item m owns the rows
// q_row_off[m] .. q_row_off[m + 1] - 1 of q / out / dout / dq (its length <= tq).  kv_packed != 0: self-attention, K / V (and dk /
// dv) are rows of the same packed numbering and the item's key count is its own length (the dense call's key_len); kv_packed == 0:
// cross-attention, K / V stay dense ([batch][tk] rows, optional key_len).  lse keeps the dense [batch][heads][tq] layout and the
// dropout mask is drawn at the DENSE coordinates ((m H + h) tq + q) tk + key, so every live row's result -- train mode included -- is
// bitwise that of uniir_attention_fwd_ex / _bwd_ex on the padded batch (masked keys contribute exactly 0 there, med.py:687-688).
extern "C" int uniir_attention_fwd_rows(const void* q, int64_t q_ld, const void* k, const void* v, int64_t kv_ld, void* out,
                                        int64_t out_ld, float* lse, const int32_t* q_row_off, int32_t kv_packed,
                                        const int32_t* key_len, int32_t batch, int32_t tq, int32_t tk, int32_t heads, float drop_p,
                                        uint32_t drop_seed, void* stream) {
    if (!q || !k || !v || !out || !lse || !q_row_off || batch < 0 || heads <= 0) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (tq < 1 || tk < 1 || tq > 512 || tk > 512) return UNIIR_ESHAPE;
    if (kv_packed && (tq != tk || key_len)) return UNIIR_EINVAL;
    if ((q_ld % 8) || (kv_ld % 8) || (out_ld % 8)) return UNIIR_EALIGN;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)out & 15)) return UNIIR_EALIGN;
    if (drop_p < 0.f || drop_p >= 1.f) return UNIIR_EINVAL;
    AttnArgs a = {};
    a.q = (const unsigned short*)q; a.k = (const unsigned short*)k; a.v = (const unsigned short*)v;
    a.q_ld = q_ld; a.kv_ld = kv_ld; a.out = (unsigned short*)out; a.out_ld = out_ld; a.lse = lse; a.klen = key_len;
    a.Tq = tq; a.Tk = tk; a.H = heads; a.causal = 0; a.scale = ATT_SCALE;
    a.row_off = q_row_off; a.row_off_q_only = kv_packed ? 0 : 1;
    a.drop_p = drop_p; a.drop_seed = drop_seed;
    return launch_attn_fwd(a, batch, (hipStream_t)stream);
}
extern "C" int uniir_attention_bwd_rows(const void* q, int64_t q_ld, const void* k, const void* v, int64_t kv_ld, const void* out,
                                        const void* dout, int64_t out_ld, const float* lse, const int32_t* q_row_off,
                                        int32_t kv_packed, const int32_t* key_len, void* dq, int64_t dq_ld, void* dk, void* dv,
                                        int64_t dkv_ld, int32_t batch, int32_t tq, int32_t tk, int32_t heads, float drop_p,
                                        uint32_t drop_seed, void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !q_row_off || batch < 0 || heads <= 0) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (tq < 1 || tk < 1 || tq > 512 || tk > 512) return UNIIR_ESHAPE;
    if (kv_packed && (tq != tk || key_len)) return UNIIR_EINVAL;
    if ((q_ld % 8) || (kv_ld % 8) || (out_ld % 8) || (dq_ld % 8) || (dkv_ld % 8)) return UNIIR_EALIGN;
    if (((uintptr_t)dq & 15) || ((uintptr_t)dk & 15) || ((uintptr_t)dv & 15)) return UNIIR_EALIGN;
    if (drop_p < 0.f || drop_p >= 1.f) return UNIIR_EINVAL;
    AttnArgs a = {};
    a.q = (const unsigned short*)q; a.k = (const unsigned short*)k; a.v = (const unsigned short*)v;
    a.q_ld = q_ld; a.kv_ld = kv_ld; a.out = (unsigned short*)out; a.out_ld = out_ld;
    a.lse = const_cast<float*>(lse); a.klen = key_len;
    a.Tq = tq; a.Tk = tk; a.H = heads; a.causal = 0; a.scale = ATT_SCALE;
    a.row_off = q_row_off; a.row_off_q_only = kv_packed ? 0 : 1;
    a.dout = (const unsigned short*)dout;
    a.dq = (unsigned short*)dq; a.dk = (unsigned short*)dk; a.dv = (unsigned short*)dv;
    a.dq_ld = dq_ld; a.dkv_ld = dkv_ld;
    a.drop_p = drop_p; a.drop_seed = drop_seed;
    return launch_attn_bwd(a, batch, (hipStream_t)stream);
}

// T5-style self-attention for the CLIP_FF fusion stack: logits = scale * q.k + rel_emb[rel_bucket[key - query + seq - 1]][head]
// (transformers T5Attention: scale 1, bucketed relative position bias shared by all layers), packed qkv like
// uniir_attention_fwd.  bwd adds d loss / d rel_emb into drel (fp32 [buckets][heads], zero it once per step).
extern "C" int uniir_attention_rel_fwd(const void* qkv, void* out, float* lse, const float* rel_emb,
                                       const int32_t* rel_bucket, int32_t nbuckets, float scale, int32_t batch,
                                       int32_t seq, int32_t heads, float drop_p, uint32_t drop_seed, void* stream) {
    if (!qkv || !out || !lse || !rel_emb || !rel_bucket || batch < 0 || heads <= 0 || nbuckets <= 0 || nbuckets > 64)
        return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (seq < 1 || seq > 512) return UNIIR_ESHAPE;
    const long W = (long)heads * ATT_D;
    AttnArgs a = {};
    a.q = (const unsigned short*)qkv; a.k = a.q + W; a.v = a.q + 2 * W;
    a.q_ld = a.kv_ld = 3 * W; a.out = (unsigned short*)out; a.out_ld = W; a.lse = lse;
    a.Tq = a.Tk = seq; a.H = heads; a.causal = 0; a.scale = scale;
    a.rel_emb = rel_emb; a.rel_bucket = rel_bucket; a.nbuckets = nbuckets;
    if (drop_p < 0.f || drop_p >= 1.f) return UNIIR_EINVAL;
    a.drop_p = drop_p; a.drop_seed = drop_seed;
    return launch_attn_fwd(a, batch, (hipStream_t)stream);
}

extern "C" int uniir_attention_rel_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                       const float* rel_emb, const int32_t* rel_bucket, int32_t nbuckets, float scale,
                                       float* drel, int32_t batch, int32_t seq, int32_t heads, float drop_p,
                                       uint32_t drop_seed, void* stream) {
    if (!qkv || !out || !dout || !lse || !dqkv || !rel_emb || !rel_bucket || batch < 0 || heads <= 0 || nbuckets <= 0 ||
        nbuckets > 64)
        return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (seq < 1 || seq > 512) return UNIIR_ESHAPE;
    const long W = (long)heads * ATT_D;
    AttnArgs a = {};
    a.q = (const unsigned short*)qkv; a.k = a.q + W; a.v = a.q + 2 * W;
    a.q_ld = a.kv_ld = 3 * W; a.out = (unsigned short*)const_cast<void*>(out); a.out_ld = W;
    a.lse = const_cast<float*>(lse); a.dout = (const unsigned short*)dout;
    a.dq = (unsigned short*)dqkv; a.dk = a.dq + W; a.dv = a.dq + 2 * W; a.dq_ld = a.dkv_ld = 3 * W;
    a.Tq = a.Tk = seq; a.H = heads; a.causal = 0; a.scale = scale;
    a.rel_emb = rel_emb; a.rel_bucket = rel_bucket; a.nbuckets = nbuckets; a.drel = drel;
    if (drop_p < 0.f || drop_p >= 1.f) return UNIIR_EINVAL;
    a.drop_p = drop_p; a.drop_seed = drop_seed;
    return launch_attn_bwd(a, batch, (hipStream_t)stream);
}
