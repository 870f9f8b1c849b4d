// Persistent pair-tile attention BACKWARD for PLAIN self-attention (no mask, no dropout, no relative bias, Tq = Tk): the CLIP / BLIP
// ViT towers' 257- and 197-token layers (openai/CLIP ResidualAttentionBlock.attention via clip_sf.py:43-47; BLIP
// vit.py:86-106).  Everything else -- and every forward -- stays on the general kernels of attention.hip.
//
// Why a second structure (measured, round 4, tools/r4/attn_diag.py on the general backward at 257 tokens x 16 heads x 1024 items,
// 1.98 ms): with the arithmetic knocked out the kernel still takes 0.80 ms (two stagings + statistics: every operand crosses the
// fabric twice, 5.07 GB fetched for 2.70 GB of operands), the strided 8-byte gradient stores cost 0.34 ms, and the arithmetic
// alone 1.30 ms with the 17 sixteen-row tiles dealt 3 / 2 / 2 / .. to the 8 waves.  Here:
//   * ONE persistent 8-wave workgroup per CU walks over (item, head) pairs; Q, dO, K, V of a head are ALL resident in LDS
//     (4 x 36 KiB at 257 tokens) and arrive by LDS-DMA (buffer_load .. lds, issued piece by piece from inside the phase loops) one
//     phase ahead of their use: K, V while phase 1 computes on Q, dO; the next head's Q, dO while phase 2 computes on K, V.
//     Every operand is fetched once: rocprofv3 FETCH_SIZE x 2 = 2.70 GB per launch = the algorithmic reads (was 5.07).
//   * A wave owns a PAIR of 16-row tiles (32 keys in phase 1, 32 queries in phase 2): each LDS fragment feeds two MFMAs (half
//     the LDS and MFMA-operand instructions per flop of the one-tile kernels) and 16 tiles = 8 pairs balance over the 8 waves; the
//     odd last tile (one valid row at 257 tokens) is split over waves 0..3 along its inner loop, its partial sums meet in LDS.
//   * Gradient rows leave as 16-byte pieces, 64 contiguous bytes per row and instruction (att_store_tile).
// Result: 1.54-1.59 ms (general kernel 2.0-2.25 ms in the same process), 197 tokens 1.11 vs 1.57-1.66 ms.
// What bounds it now (rocprofv3 --pmc, tools/r4/attn_pmc.sh): the matrix pipe is busy 33.5 % of the kernel, the vector issue 47 %,
// LDS 23 %, no bank conflicts -- and the costs ADD: a SIMD with two waves spends 16 cycles per 16x16x32 MFMA, ~4.5 per vector or
// LDS instruction, ~8 per v_exp, one after the other (per 32 x 32 block and wave: 512 + ~540 in phase 1), so the floor of this
// structure is the instruction count per logit (2 exp + ~7 vector + 3.5 LDS instructions per 64 logits and phase).  Variants
// that were measured and did not pay: two-group ping-pong barriers (1.88 ms), LDS reads one MFMA group ahead (kept: 1.62 ->
// 1.57), de-serialised MFMA pairs (kept, no change: the pipe is not the limit), a forward in the same structure
// (experiments/attention_pair/fwd_pair.h: 0.70 vs 0.66 ms, the forward's softmax instructions do not shrink with pairing).
// Arithmetic, fragment layouts and the order of every accumulation are those of attention.hip's backward; D = rowsum(dO * O) is
// summed as a tree (8 lanes per row) instead of a chain and the odd tile's rows are sums of eight partials: one-bf16-ulp
// differences in < 0.1 % of the elements (tests/test_kernels_gpu.py).
#include "attention.h"

#ifdef UNIIR_EXP_BUILD
#define AP_STAMP(i)                                                                                                     \
    do {                                                                                                                \
        if (a.stamps && it == 2) {                                                                                      \
            const int wg_ = (int)blockIdx.x - 64;                                                                       \
            if (wg_ >= 0 && wg_ < 64 && (threadIdx.x & 63) == 0)                                                        \
                a.stamps[(wg_ * 8 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memtime();                     \
        }                                                                                                               \
    } while (0)
#else
#define AP_STAMP(i) do {} while (0)
#endif
#define AP_THREADS 512
#define AP_WAVES 8

struct ApGeom {
    int T;            // tokens
    int nblk;         // 32-row blocks (TP / 32)
    int npair;        // 16-row tile pairs = waves with a pair
    int left;         // 1: an odd last tile (rows 32 npair .. T - 1), split over the waves
    int nvl;          // its valid rows (1 .. 16)
    int total_heads;  // items x heads
};

// ---- LDS-DMA from inline asm: the compiler must not know about these loads (it would put vmcnt(0) in front of every later ds_read);
// their completion is awaited explicitly (ap_wait_vm0) before the barrier that publishes the rows.
DEVINL u32x4_t ap_make_srd(const void* base, unsigned bytes) {
    const unsigned long long p = (unsigned long long)base;
    u32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu);    // stride 0: raw buffer
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
DEVINL void ap_dma16(u32x4_t srd, unsigned lds_addr, unsigned voff) {         // 64 lanes x 16 B -> LDS [lds_addr, +1024)
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(srd)
                 : "memory");
}
DEVINL void ap_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
DEVINL unsigned ap_lds_addr(const char* p) { return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p; }
template <class V>
DEVINL void ap_pin(V& v) { asm volatile("" : "+v"(v)); }       // the value is complete here (the compiler's own wait lands before)

// this wave's share of the LDS-DMA of two [T][64] bf16 head slices (row strides ldA / ldB elements) into two swizzled LDS slices:
// instruction j moves rows 8 j .. 8 j + 7 (lane -> row 8 j + (lane >> 3), 16-B slot lane & 7 holds source chunk slot ^ (row & 7));
// rows >= T are fetched out of the descriptor's bounds and arrive as zeros.
DEVINL void ap_stage2(u32x4_t srdA, long ldA, u32x4_t srdB, long ldB, unsigned ldsA, unsigned ldsB, int T, int w, int lane) {
    const int ninst = (T + 7) >> 3;
    const int rsub = lane >> 3, slot = lane & 7;
    for (int j = w; j < ninst; j += AP_WAVES) {
        const int row = j * 8 + rsub;
        const unsigned src = (unsigned)((slot ^ (row & 7)) * 16);
        const unsigned va = row < T ? (unsigned)row * (unsigned)(ldA * 2) + src : 0x7ffffff0u;
        const unsigned vb = row < T ? (unsigned)row * (unsigned)(ldB * 2) + src : 0x7ffffff0u;
        ap_dma16(srdA, ldsA + (unsigned)j * 1024u, va);
        ap_dma16(srdB, ldsB + (unsigned)j * 1024u, vb);
    }
}

// The same transfer issued piece by piece from inside a phase loop (one K + one V piece, or Q + dO, per inner block): the ~250 ticks
// an LDS-DMA instruction occupies its wave then fall into the partner wave's compute instead of a serial section in front of the
// phase (measured, round 4: 2.3 k of a 53 k-tick head per issue section with all eight waves issuing at once).
struct ApDma {
    u32x4_t sa, sb;
    unsigned lda2, ldb2, la, lb;
    int j, ninst, T;
    DEVINL void start(u32x4_t srdA, long ldA, u32x4_t srdB, long ldB, unsigned ldsA, unsigned ldsB, int T_, int w) {
        sa = srdA; sb = srdB; lda2 = (unsigned)(ldA * 2); ldb2 = (unsigned)(ldB * 2); la = ldsA; lb = ldsB;
        T = T_; ninst = (T_ + 7) >> 3; j = w;
    }
    DEVINL void idle() { j = 0; ninst = 0; }
    DEVINL void step(int lane) {                 // one piece of each slice, if any is left (wave-uniform)
        if (j < ninst) {
            const int row = j * 8 + (lane >> 3);
            const unsigned src = (unsigned)(((lane & 7) ^ (row & 7)) * 16);
            const unsigned va = row < T ? (unsigned)row * lda2 + src : 0x7ffffff0u;
            const unsigned vb = row < T ? (unsigned)row * ldb2 + src : 0x7ffffff0u;
            // (readfirstlane: the values ARE wave-uniform, but the asm operands must be provably so)
            ap_dma16(sa, __builtin_amdgcn_readfirstlane(la + (unsigned)j * 1024u), va);
            ap_dma16(sb, __builtin_amdgcn_readfirstlane(lb + (unsigned)j * 1024u), vb);
            j += AP_WAVES;
        }
    }
    DEVINL void drain(int lane) {
        while (j < ninst) step(lane);
    }
};

// row fragment of rows r0 .. r0 + 15 straight from global memory, rows >= T as zeros (branch-free: clamped address, select)
DEVINL bf16x8_t ap_frag_global(const unsigned short* __restrict__ src, long ld, int r0, int s, int lane, int T) {
    const int row = r0 + (lane & 15);
    const int rc = min(row, T - 1);
    u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + (long)rc * ld + s * 32 + (lane >> 4) * 8);
    const u32x4_t z = {0u, 0u, 0u, 0u};
    v = row < T ? v : z;
    return __builtin_bit_cast(bf16x8_t, v);
}

// D[q] = dO[q] . O[q] and lse2[q] = lse[q] log2 e of one head -> stats[0 .. TP) = lse2, stats[TP .. 2 TP) = -D (negated: the
// consumers then ADD it, which the compiler turns into v_pk_add_f32; a vector subtraction it splits into scalar v_sub_f32).  Thread t takes the
// 16-B chunks t, t + 512, ..: the 8 lanes of a row sit side by side (coalesced 128-B rows), their partial dots meet by DPP.
// Two halves: the loads are issued before a barrier, the arithmetic and the LDS writes run behind it.
template <int TP>
struct ApStats {
    static constexpr int NL = (TP * 8 + AP_THREADS - 1) / AP_THREADS;
    u32x4_t xo[NL], xd[NL];
    float lv[NL];
    DEVINL void load(const unsigned short* __restrict__ obase, const unsigned short* __restrict__ dobase, long ld,
                     const float* __restrict__ lse, int T, int tid) {
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int c = u * AP_THREADS + tid;
            const int row = min(c >> 3, T - 1), ch = c & 7;
            xo[u] = *reinterpret_cast<const u32x4_t*>(obase + (long)row * ld + ch * 8);
            xd[u] = *reinterpret_cast<const u32x4_t*>(dobase + (long)row * ld + ch * 8);
            lv[u] = lse[row];
        }
    }
    DEVINL void finish(int T, float* stats, int tid) const {
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int c = u * AP_THREADS + tid;
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d += __uint_as_float(xo[u][e] << 16) * __uint_as_float(xd[u][e] << 16);
                d += __uint_as_float(xo[u][e] & 0xffff0000u) * __uint_as_float(xd[u][e] & 0xffff0000u);
            }
            auto dpp_add = [](float v, auto ctrl) {
                const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true);
                return v + __builtin_bit_cast(float, o);
            };
            d = dpp_add(d, std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
            d = dpp_add(d, std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
            d = dpp_add(d, std::integral_constant<int, 0x141>{});    // row_half_mirror: the other quad of the 8 lanes
            const int row = c >> 3;
            if ((c & 7) == 0 && row < T) {
                stats[row] = lv[u] * LOG2EF;
                stats[TP + row] = -d;
            }
        }
    }
};

// lane-constant byte offsets of the fragment reads inside a swizzled [rows][64] slice (block offsets are multiples of 32 rows =
// 4096 B and do not touch the swizzle)
struct ApOff {
    unsigned R[2];    // row fragment, k-step s: row li, chunk (4 s + g) ^ (li & 7)
    unsigned Tt[4];   // transposed fragment, column tile dt: rows 4 g + (li >> 2) (+ 16: +2048 B), columns 16 dt + 4 (li & 3)
    DEVINL void init(int lane) {
        const int li = lane & 15, g = lane >> 4;
#pragma unroll
        for (int s = 0; s < 2; ++s) R[s] = li * 128 + (((s * 4 + g) ^ (li & 7)) << 4);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int row = 4 * g + (li >> 2), col = 16 * dt + 4 * (li & 3);
            Tt[dt] = row * 128 + ((((col >> 3) ^ (row & 7)) << 4) | ((col & 7) << 1));
        }
    }
};
// fragment addresses of one slice as opaque VGPRs: the loops then keep ONE running register per fragment class (base + block
// offset) and every other displacement (k-step half, second slice, second 16 rows) goes into the instruction's offset field
struct ApBase {
    unsigned r[2], t[4];
    DEVINL void init(unsigned slice, const ApOff& of);
};
// LDS reads take 32-bit LDS byte addresses (a VGPR base + a compile-time offset folds into the instruction's offset field)
DEVINL void ApBase::init(unsigned slice, const ApOff& of) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        r[s] = slice + of.R[s];
        asm volatile("" : "+v"(r[s]));
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        t[dt] = slice + of.Tt[dt];
        asm volatile("" : "+v"(t[dt]));
    }
}
#define AP_LDS(T, addr) (*reinterpret_cast<const __attribute__((address_space(3))) T*>((unsigned long)(addr)))
DEVINL bf16x8_t ap_rows(unsigned addr) { return __builtin_bit_cast(bf16x8_t, AP_LDS(u32x4_t, addr)); }
DEVINL bf16x8_t ap_cols(unsigned addr) {      // 32-row block: rows 4 g .., 16 + 4 g .. of one column tile
    typedef __attribute__((address_space(3))) s16x4_t* lp;
    const u32x2_t l2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(unsigned long)addr));
    const u32x2_t h2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(unsigned long)(addr + 2048)));
    const u32x4_t r = {l2[0], l2[1], h2[0], h2[1]};
    return __builtin_bit_cast(bf16x8_t, r);
}

// ---- phase 1 of KT key tiles (keys k0 .. k0 + 16 KT - 1, lane = key column) over the query blocks b_lo, b_lo + b_step, .. < b_hi:
//   S = Q K^T, dP = dO V^T, P = exp2(c S - lse2), dS = P (dP - D);  dV^T += dO^T P,  dK^T += Q^T dS
// MASK: some key column of these tiles is padding (the odd tile; pairs are always full, see ap_geom)
// Every LDS read is issued one MFMA group ahead of its use (measured, round 4: a wave issues ~1 instruction per 4.5 cycles and with
// the reads placed next to their uses the ~14 LDS round trips per block, ~1000 cycles, were the larger half of a block's 1600-2200):
//   [statistics + transposed fragments of block b]  S / dP MFMAs (row fragments of b: read during the previous block)
//   [row fragments of block b + step into the registers the MFMAs have just read]  softmax arithmetic  dV / dK MFMAs
template <int TP, int KT, bool MASK>
DEVINL void ap_phase1(unsigned lq, unsigned stats, const ApOff& of, const bf16x8_t (&kf)[KT][2], const bf16x8_t (&vf)[KT][2],
                      int k0, int T, int b_lo, int b_hi, int b_step, f32x4_t (&dv)[KT][4], f32x4_t (&dk)[KT][4], int lane,
                      ApDma& dma) {
    constexpr int SB = TP * 128;
    const int li = lane & 15, g = lane >> 4;
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    bool kval[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) kval[kt] = k0 + kt * 16 + li < T;
    ApBase ba;
    ba.init(lq, of);
    unsigned sbase = stats + 16 * g;
    asm volatile("" : "+v"(sbase));
    bf16x8_t RQ[2][2], RD[2][2];
    auto load_rows = [&](int b) {
        const unsigned blk = (unsigned)b * 4096u;
        const unsigned r0 = ba.r[0] + blk, r1 = ba.r[1] + blk;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            RQ[qt][0] = ap_rows(r0 + qt * 2048);
            RQ[qt][1] = ap_rows(r1 + qt * 2048);
            RD[qt][0] = ap_rows(r0 + (SB + qt * 2048));
            RD[qt][1] = ap_rows(r1 + (SB + qt * 2048));
        }
    };
    if (b_lo < b_hi) load_rows(b_lo);
    auto body = [&](int b, auto with_dma) {
        const unsigned blk = (unsigned)b * 4096u;
        const unsigned st = sbase + (unsigned)b * 128u;
        f32x4_t l4[2], d4[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            l4[qt] = AP_LDS(f32x4_t, st + qt * 64);
            d4[qt] = AP_LDS(f32x4_t, st + (TP * 4 + qt * 64));
        }
        bf16x8_t TQ[4], TD[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const unsigned tp = ba.t[dt] + blk;
            TQ[dt] = ap_cols(tp);
            TD[dt] = ap_cols(tp + SB);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the two k-steps of a product are a dependent pair: all first steps, then all second steps (left to itself the compiler
        // chains every pair back to back through one temporary -- 16 serialised MFMAs at the full latency, measured: the matrix
        // pipe then paces the loop at 32 instead of 16 cycles per MFMA)
        f32x4_t sa[2][KT], dp[2][KT];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                sa[qt][kt] = mfma16(RQ[qt][0], kf[kt][0], zero4);
                dp[qt][kt] = mfma16(RD[qt][0], vf[kt][0], zero4);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                sa[qt][kt] = mfma16(RQ[qt][1], kf[kt][1], sa[qt][kt]);
                dp[qt][kt] = mfma16(RD[qt][1], vf[kt][1], dp[qt][kt]);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (decltype(with_dma)::value) dma.step(lane);
        if (b + b_step < b_hi) load_rows(b + b_step);
        __builtin_amdgcn_sched_barrier(0);
        // A rows = queries, B columns = keys -> acc[r] = S[q = 32 b + 16 qt + 4 g + r][key = k0 + 16 kt + li]
        bf16x8_t pf[KT], dsf[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            // vector expressions: the compiler emits v_pk_fma / v_pk_add / v_pk_mul (two elements per issue slot; the loops are
            // bound by the number of instructions a SIMD can issue, ~1 per 4.5 cycles, not by any pipe)
            f32x4_t p[2], ds[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {     // rows q >= T carry lse2 = 1e30: P = 0 without a mask
                const f32x4_t arg = __builtin_elementwise_fma(sa[qt][kt], f32x4_t{SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E}, -l4[qt]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(arg[r]);
                    p[qt][r] = MASK ? (kval[kt] ? e : 0.f) : e;
                }
                ds[qt] = p[qt] * (dp[qt][kt] + d4[qt]);          // d4 = -D
            }
            pf[kt] = pack8(p[0], p[1]);
            dsf[kt] = pack8(ds[0], ds[1]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                dv[kt][dt] = mfma16(TD[dt], pf[kt], dv[kt][dt]);
                dk[kt][dt] = mfma16(TQ[dt], dsf[kt], dk[kt][dt]);
            }
    };
    // the DMA pieces go out with the first blocks; the remaining blocks run a loop body without the issue logic (the loops are bound
    // by instruction issue: ~20 scalar instructions per block for a piece that is not there)
    int b = b_lo;
    for (; b < b_hi && dma.j < dma.ninst; b += b_step) body(b, std::true_type{});
    for (; b < b_hi; b += b_step) body(b, std::false_type{});
}

// ---- phase 2 of QT query tiles (lane = query column) over the key blocks b_lo, b_lo + b_step, .. < nblk:
//   S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - D);  dQ^T += K^T dS^T
// only the last block (nblk - 1) holds padded keys (T % 32 != 0, see ap_geom); LDS reads one MFMA group ahead as in phase 1
template <int TP, int QT>
DEVINL void ap_phase2(unsigned lk, const ApOff& of, const bf16x8_t (&qf)[QT][2], const bf16x8_t (&dof)[QT][2],
                      const float (&mylse)[QT], const float (&myD)[QT], int T, int b_lo, int nblk, int b_step,
                      f32x4_t (&dq)[QT][4], int lane, ApDma& dma) {
    constexpr int SB = TP * 128;
    const int g = lane >> 4;
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    ApBase ba;
    ba.init(lk, of);
    bf16x8_t RK[2][2], RV[2][2];
    auto load_rows = [&](int b) {
        const unsigned blk = (unsigned)b * 4096u;
        const unsigned r0 = ba.r[0] + blk, r1 = ba.r[1] + blk;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            RK[kt][0] = ap_rows(r0 + kt * 2048);
            RK[kt][1] = ap_rows(r1 + kt * 2048);
            RV[kt][0] = ap_rows(r0 + (SB + kt * 2048));
            RV[kt][1] = ap_rows(r1 + (SB + kt * 2048));
        }
    };
    if (b_lo < nblk) load_rows(b_lo);
    auto body = [&](int b, auto with_dma, auto is_edge) {
        const unsigned blk = (unsigned)b * 4096u;
        bf16x8_t TK[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) TK[dt] = ap_cols(ba.t[dt] + blk);
        __builtin_amdgcn_sched_barrier(0);
        f32x4_t sa[2][QT], dp[2][QT];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                sa[kt][qt] = mfma16(RK[kt][0], qf[qt][0], zero4);
                dp[kt][qt] = mfma16(RV[kt][0], dof[qt][0], zero4);
            }
        __builtin_amdgcn_sched_barrier(0);       // (first k-steps, then second k-steps: see phase 1)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                sa[kt][qt] = mfma16(RK[kt][1], qf[qt][1], sa[kt][qt]);
                dp[kt][qt] = mfma16(RV[kt][1], dof[qt][1], dp[kt][qt]);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (decltype(with_dma)::value) dma.step(lane);
        if (b + b_step < nblk) load_rows(b + b_step);
        __builtin_amdgcn_sched_barrier(0);
        // acc[r] = S^T[key = 32 b + 16 kt + 4 g + r][q = lane column]
        constexpr bool edge = decltype(is_edge)::value;
        const int krem = T - (b * 32 + 4 * g);               // key 16 kt + r of this lane's rows is valid iff < krem
        bf16x8_t dsf[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4_t ds[2];
            const f32x4_t sc4 = {SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E};
            const f32x4_t nl4 = {-mylse[qt], -mylse[qt], -mylse[qt], -mylse[qt]}, D4 = {myD[qt], myD[qt], myD[qt], myD[qt]};   // myD = -D
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const f32x4_t arg = __builtin_elementwise_fma(sa[kt][qt], sc4, nl4);
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[kt][r] = __builtin_amdgcn_exp2f(arg[r]);
            }
            if (edge) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[kt][r] = (kt * 16 + r < krem) ? ds[kt][r] : 0.f;
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) ds[kt] = ds[kt] * (dp[kt][qt] + D4);
            dsf[qt] = pack8(ds[0], ds[1]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) dq[qt][dt] = mfma16(TK[dt], dsf[qt], dq[qt][dt]);
    };
    int b = b_lo;
    for (; b < nblk - 1 && dma.j < dma.ninst; b += b_step) body(b, std::true_type{}, std::false_type{});
    for (; b < nblk - 1; b += b_step) body(b, std::false_type{}, std::false_type{});
    if (b == nblk - 1) body(b, std::true_type{}, std::true_type{});       // the one block with padded keys
}

// MEASURED AND DROPPED (round 4): two-group ping-pong forms of the pair loops (waves 0..3 / 4..7 one s_barrier apart, every block =
// [MFMA interval | LDS + softmax interval]): correct, but 1.88 ms vs 1.62 ms for the free-running loops at 257 tokens x 16 heads x 1024
// items -- the vector interval (~150 instructions + a DMA piece at ~4.5 cycles per instruction) sets the pace, not the matrix pipe.
// the waves' partial sums of the odd tile -> its (<= 16) gradient rows: thread t sums 8 consecutive columns of row (t >> 3) % nvl
// of matrix t / (8 nvl) over the 8 waves, in wave order, and stores them as one 16-byte piece
DEVINL void ap_reduce_rows(const float* part, int nmat, int nvl, float scale0, unsigned short* const (&dst)[2], long ld, int row0,
                           int tid) {
    if (tid >= nmat * nvl * 8) return;
    const int mat = tid / (nvl * 8), rr = (tid >> 3) % nvl, c = tid & 7;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    for (int w = 0; w < AP_WAVES; ++w) {
        const float* p = part + ((w * nmat + mat) * nvl + rr) * 64 + c * 8;
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p), b = *reinterpret_cast<const f32x4_t*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[e] += a[e];
            s[4 + e] += b[e];
        }
    }
    const float sc = mat == 0 ? scale0 : 1.0f;
    const u32x4_t v = {pack_bf16x2(s[0] * sc, s[1] * sc), pack_bf16x2(s[2] * sc, s[3] * sc), pack_bf16x2(s[4] * sc, s[5] * sc),
                       pack_bf16x2(s[6] * sc, s[7] * sc)};
    *reinterpret_cast<u32x4_t*>(dst[mat] + (long)(row0 + rr) * ld + c * 8) = v;
}
// this wave's partial of the odd tile (lane = row li, registers: column 16 dt + 4 g + r) -> part[(w nmat + mat) nvl + li][64]
DEVINL void ap_write_partial(float* part, int w, int nmat, int mat, int nvl, const f32x4_t (&acc)[4], int lane) {
    const int li = lane & 15, g = lane >> 4;
    if (li < nvl) {
        float* p = part + ((w * nmat + mat) * nvl + li) * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4_t*>(p + 16 * dt) = acc[dt];
    }
}

// this wave's fragments of its key pair and of the odd tile, straight from global memory (the K / V slices of the head are still
// on their way into LDS when phase 1 starts)
struct ApKeyFrags {
    bf16x8_t kf[2][2], vf[2][2], kf1[1][2], vf1[1][2];
    DEVINL void load(const unsigned short* __restrict__ kbase, const unsigned short* __restrict__ vbase, long ld, int k0, int kL0,
                     int lane, int T) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                kf[kt][s] = ap_frag_global(kbase, ld, k0 + kt * 16, s, lane, T);
                vf[kt][s] = ap_frag_global(vbase, ld, k0 + kt * 16, s, lane, T);
            }
            kf1[0][s] = ap_frag_global(kbase, ld, kL0, s, lane, T);
            vf1[0][s] = ap_frag_global(vbase, ld, kL0, s, lane, T);
        }
    }
    DEVINL void pin() {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            ap_pin(kf[0][s]); ap_pin(kf[1][s]); ap_pin(vf[0][s]); ap_pin(vf[1][s]); ap_pin(kf1[0][s]); ap_pin(vf1[0][s]);
        }
    }
};

// Per head (barriers B1, B2):
//   next head's statistics requested | phase 1 on Q, dO (odd-tile share, then the pair; one K + V DMA piece issued per inner block)
//   | the wave's phase-2 operands leave LDS | next statistics -> the other statistics buffer | dK, dV stored | B1 | odd-tile rows
//   reduced | phase 2 on K, V (one next-Q + next-dO piece per inner block) | next head's key fragments requested | dQ stored | B2 |
//   odd-tile rows reduced
// The compiler sees none of the DMA: every wait for it is an explicit vmcnt(0) placed where no compiler-visible load is pending
// behind it, and a compiler-visible load is consumed either before the first DMA piece of a phase or after its last block (where
// the pieces, issued in the first half of the phase, have long landed) -- the compiler's counted waits for ITS loads would
// otherwise drain the DMA queue in the middle of a phase.
template <int TP>
__global__ __launch_bounds__(AP_THREADS, 2) void attn_bwd_pair_kernel(AttnArgs a, ApGeom gm) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int SB = TP * 128;
    const int T = gm.T, H = a.H, nblk = gm.nblk, npair = gm.npair, nvl = gm.nvl;
    char* lQ = lds;                     // Q, dO = + SB
    char* lK = lds + 2 * SB;            // K, V = + SB
    float* stats = reinterpret_cast<float*>(lds + 4 * SB);      // 2 heads x (lse2 [TP], D [TP])
    float* part1 = stats + 4 * TP;                              // [8 waves][dK, dV][nvl][64]
    float* part2 = part1 + AP_WAVES * 2 * nvl * 64;             // [8 waves][nvl][64]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned aQ = ap_lds_addr(lQ), aK = ap_lds_addr(lK), aS = ap_lds_addr(reinterpret_cast<const char*>(stats));
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    const int k0 = 32 * w;                                      // first row of this wave's pair (keys in phase 1, queries in phase 2)
    const int kL0 = 32 * npair;                                 // first row of the odd tile
    // the odd tile's inner blocks: to the waves without a pair when there are any; else to waves 0..3 -- the first-dispatched wave of
    // each SIMD wins the issue arbitration against its partner (measured: pair loops of waves 0..3 finish ~20 % earlier), so the
    // extra blocks go where the slack is
    const int idle = AP_WAVES - npair;
    const int lb_lo = gm.left ? (idle > 0 ? (w >= npair ? w - npair : nblk) : (w < 4 ? w : nblk)) : nblk;
    const int lb_step = idle > 0 ? idle : 4;

    // rows T .. TP - 1 of the four slices stay zero for the whole kernel (the DMA never writes whole padded pieces; padded rows
    // inside a piece arrive as zeros); padded statistics: lse2 = 1e30 (P = 0), D = 0
    {
        const int npad = (TP - T) * 8;
        const u32x4_t z = {0u, 0u, 0u, 0u};
        for (int c = tid; c < 4 * npad; c += AP_THREADS) {
            const int sl = c / npad, r = c % npad;
            *reinterpret_cast<u32x4_t*>(lds + sl * SB + T * 128 + r * 16) = z;
        }
        for (int r = T + tid; r < TP; r += AP_THREADS) {
            stats[r] = stats[2 * TP + r] = 1e30f;
            stats[TP + r] = stats[3 * TP + r] = 0.f;
        }
    }
    const unsigned qbytes = (unsigned)((long)(T - 1) * a.q_ld * 2 + 128), kbytes = (unsigned)((long)(T - 1) * a.kv_ld * 2 + 128),
                   obytes = (unsigned)((long)(T - 1) * a.out_ld * 2 + 128);
    int hd = blockIdx.x;
    if (hd >= gm.total_heads) return;
    ApKeyFrags kfr;
    {   // first head: statistics, key fragments, Q and dO
        const int m = hd / H, h = hd % H;
        const unsigned short* qbase = a.q + (long)m * T * a.q_ld + h * ATT_D;
        const unsigned short* obase = a.out + (long)m * T * a.out_ld + h * ATT_D;
        const unsigned short* dobase = a.dout + (long)m * T * a.out_ld + h * ATT_D;
        {
            ApStats<TP> st;
            st.load(obase, dobase, a.out_ld, a.lse + ((long)m * H + h) * T, T, tid);
            st.finish(T, stats, tid);
        }
        kfr.load(a.k + (long)m * T * a.kv_ld + h * ATT_D, a.v + (long)m * T * a.kv_ld + h * ATT_D, a.kv_ld, k0, kL0, lane, T);
        ap_wait_vm0();
        kfr.pin();
        ap_stage2(ap_make_srd(qbase, qbytes), a.q_ld, ap_make_srd(dobase, obytes), a.out_ld, aQ, aQ + SB, T, w, lane);
        ap_wait_vm0();
        __syncthreads();
    }
    int it = 0;
    for (; hd < gm.total_heads; hd += gridDim.x, ++it) {
        const int m = hd / H, h = hd % H;
        AP_STAMP(0);
        const unsigned short* kbase = a.k + (long)m * T * a.kv_ld + h * ATT_D;
        const unsigned short* vbase = a.v + (long)m * T * a.kv_ld + h * ATT_D;
        unsigned short* dqbase = a.dq + (long)m * T * a.dq_ld + h * ATT_D;
        unsigned short* dkbase = a.dk + (long)m * T * a.dkv_ld + h * ATT_D;
        unsigned short* dvbase = a.dv + (long)m * T * a.dkv_ld + h * ATT_D;
        const int nh = hd + gridDim.x;
        const bool more = nh < gm.total_heads;
        const int m2 = more ? nh / H : m, h2 = more ? nh % H : h;          // (the last head re-reads itself: harmless, unused)
        const unsigned sCur = aS + (unsigned)(it & 1) * (2 * TP * 4);
        float* statsCur = stats + (it & 1) * (2 * TP);
        float* statsNxt = stats + ((it + 1) & 1) * (2 * TP);
        // ---------------- phase 1: dK, dV ----------------
        ap_wait_vm0();          // the key fragments requested at the end of the last phase 2 (and that head's stores) are complete
        kfr.pin();
        // Lane-dependent addresses outside the phase loops are rebuilt from an opaque copy of the thread id in every section: hoisted
        // out of the head loop they would be dozens of registers live across both phases, i.e. spills -- and every spill reload is a
        // compiler-visible load whose wait would drain the DMA queue in the middle of a phase.
        int tid1 = tid;
        asm volatile("" : "+v"(tid1));
        const int lane1 = tid1 & 63;
        ApOff of1;
        of1.init(lane1);
        // K, V of this head arrive while phase 1 computes (their slices were released by the barrier that ended the last phase 2)
        ApDma dma;
        dma.start(ap_make_srd(kbase, kbytes), a.kv_ld, ap_make_srd(vbase, kbytes), a.kv_ld, aK, aK + SB, T, w);
        AP_STAMP(1);
        if (gm.left) {
            f32x4_t dv1[1][4], dk1[1][4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dv1[0][dt] = dk1[0][dt] = zero4;
            ap_phase1<TP, 1, true>(aQ, sCur, of1, kfr.kf1, kfr.vf1, kL0, T, lb_lo, nblk, lb_step, dv1, dk1, lane1, dma);
            ap_write_partial(part1, w, 2, 0, nvl, dk1[0], lane1);
            ap_write_partial(part1, w, 2, 1, nvl, dv1[0], lane1);
        }
        AP_STAMP(2);
        f32x4_t dv[2][4], dk[2][4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dv[0][dt] = dv[1][dt] = dk[0][dt] = dk[1][dt] = zero4;
        if (w < npair) ap_phase1<TP, 2, false>(aQ, sCur, of1, kfr.kf, kfr.vf, k0, T, 0, nblk, 1, dv, dk, lane1, dma);
        AP_STAMP(3);
        int tid2 = tid;
        asm volatile("" : "+v"(tid2));
        const int lane2 = tid2 & 63, li2 = lane2 & 15, g2 = lane2 >> 4;
        dma.drain(lane2);                   // (waves whose loops were shorter than their share of the pieces)
        // this wave's phase-2 operands: its query rows and their statistics leave LDS before the slices are released
        bf16x8_t qf[2][2], dof[2][2], qf1[1][2], dof1[1][2];
        float mylse[2], myD[2], mylse1[1], myD1[1];
        ApOff of2;
        of2.init(lane2);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                qf[qt][s] = ap_rows(aQ + of2.R[s] + (k0 + qt * 16) * 128);
                dof[qt][s] = ap_rows(aQ + SB + of2.R[s] + (k0 + qt * 16) * 128);
            }
            qf1[0][s] = ap_rows(aQ + of2.R[s] + kL0 * 128);
            dof1[0][s] = ap_rows(aQ + SB + of2.R[s] + kL0 * 128);
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            mylse[qt] = statsCur[k0 + qt * 16 + li2];
            myD[qt] = statsCur[TP + k0 + qt * 16 + li2];
        }
        mylse1[0] = statsCur[kL0 + li2];
        myD1[0] = statsCur[TP + kL0 + li2];
        ap_wait_vm0();                      // this wave's K / V pieces have landed
        AP_STAMP(4);
        if (w < npair) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int key = k0 + kt * 16 + li2;
                att_store_tile(dk[kt], ATT_SCALE, dkbase + (long)key * a.dkv_ld, key < T, g2);
                att_store_tile(dv[kt], 1.0f, dvbase + (long)key * a.dkv_ld, key < T, g2);
            }
        }
        AP_STAMP(5);
        __syncthreads();                    // B1: K, V visible; Q, dO slices released; partials complete
        AP_STAMP(6);
        int tid3 = tid;
        asm volatile("" : "+v"(tid3));
        const int lane3 = tid3 & 63;
        ApOff of3;
        of3.init(lane3);
        if (gm.left) {
            unsigned short* const dst[2] = {dkbase, dvbase};
            ap_reduce_rows(part1, 2, nvl, ATT_SCALE, dst, a.dkv_ld, kL0, tid3);
        }
        // ---------------- phase 2: dQ; the next head's Q, dO arrive under it ----------------
        if (more)
            dma.start(ap_make_srd(a.q + (long)m2 * T * a.q_ld + h2 * ATT_D, qbytes), a.q_ld,
                      ap_make_srd(a.dout + (long)m2 * T * a.out_ld + h2 * ATT_D, obytes), a.out_ld, aQ, aQ + SB, T, w);
        else
            dma.idle();
        ap_wait_vm0();                      // (the odd-tile stores: nothing the compiler tracks may be pending behind the first piece)
        // the next head's statistics are requested now and consumed behind the phase (their round trip costs nothing there; phase 2
        // has the registers to hold them, phase 1 does not)
        ApStats<TP> nst;
        nst.load(a.out + (long)m2 * T * a.out_ld + h2 * ATT_D, a.dout + (long)m2 * T * a.out_ld + h2 * ATT_D, a.out_ld,
                 a.lse + ((long)m2 * H + h2) * T, T, tid3);
        AP_STAMP(7);
        if (gm.left) {
            f32x4_t dq1[1][4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq1[0][dt] = zero4;
            ap_phase2<TP, 1>(aK, of3, qf1, dof1, mylse1, myD1, T, lb_lo, nblk, lb_step, dq1, lane3, dma);
            ap_write_partial(part2, w, 1, 0, nvl, dq1[0], lane3);
        }
        AP_STAMP(8);
        f32x4_t dq[2][4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[0][dt] = dq[1][dt] = zero4;
        if (w < npair) ap_phase2<TP, 2>(aK, of3, qf, dof, mylse, myD, T, 0, nblk, 1, dq, lane3, dma);
        AP_STAMP(9);
        int tid4 = tid;
        asm volatile("" : "+v"(tid4));
        const int lane4 = tid4 & 63, li4 = lane4 & 15, g4 = lane4 >> 4;
        dma.drain(lane4);
        // the next head's statistics -> the other buffer (read after B2).  Consumed on every path: a load left pending on one would
        // make the compiler wait for "it" -- i.e. for the DMA queue -- wherever the paths join
        nst.finish(more ? T : 0, statsNxt, tid4);
        ap_wait_vm0();                      // the next head's Q / dO pieces have landed
        AP_STAMP(10);
        if (w < npair) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int q = k0 + qt * 16 + li4;
                att_store_tile(dq[qt], ATT_SCALE, dqbase + (long)q * a.dq_ld, q < T, g4);
            }
        }
        // the next head's key fragments: their round trip lies under the barrier
        kfr.load(a.k + (long)m2 * T * a.kv_ld + h2 * ATT_D, a.v + (long)m2 * T * a.kv_ld + h2 * ATT_D, a.kv_ld, k0, kL0, lane4, T);
        AP_STAMP(11);
        __syncthreads();                    // B2: next Q, dO + statistics visible; K, V slices released; partials complete
        AP_STAMP(12);
        if (gm.left) {
            unsigned short* const dst[2] = {dqbase, dqbase};
            ap_reduce_rows(part2, 1, nvl, ATT_SCALE, dst, a.dq_ld, kL0, tid4);
        }
    }
}

// LDS bytes of the backward for TP padded rows and nvl valid rows in the odd tile
static int ap_bwd_lds(int TP, int nvl) { return 4 * TP * 128 + 4 * TP * 4 + AP_WAVES * 3 * nvl * 64 * 4; }

// can the pair kernels take this call?  plain self-attention (the caller checks mask / dropout / bias), 129 .. 288 tokens in one of
// the instantiated paddings, at most 8 pairs, and the odd tile's partial area within the LDS budget
static bool ap_geom(int T, ApGeom* gm, int* TP) {
    if (T <= 192 || T > 288) return false;
    const int tp = T <= 224 ? 224 : T <= 256 ? 256 : 288;
    if (tp == 256) return false;                          // (not instantiated: no such tower)
    const int nt = (T + 15) >> 4;
    gm->T = T;
    gm->nblk = tp / 32;
    gm->npair = nt >> 1;
    gm->left = nt & 1;
    gm->nvl = gm->left ? T - 32 * gm->npair : 1;
    if (gm->npair > AP_WAVES) return false;
    if (32 * gm->npair > T) return false;                 // every pair is full (no key mask in the pair loops) ...
    if (T % 32 == 0) return false;                        // ... and exactly the last 32-row block holds padding
    if (ap_bwd_lds(tp, gm->nvl) > 160 * 1024) return false;
    *TP = tp;
    return true;
}

int launch_attn_bwd_pair(const AttnArgs& a, int batch, hipStream_t st) {      // returns 1 when the shape is not taken
    ApGeom gm;
    int TP;
    if (a.Tq != a.Tk || a.causal || a.klen || a.row_off || a.kv_row_off || a.rel_emb || a.drop_p > 0.f || !ap_geom(a.Tq, &gm, &TP)) return 1;
    if ((a.q_ld | a.kv_ld | a.out_ld | a.dq_ld | a.dkv_ld) % 8) return 1;                  // 16-byte pieces
    if ((long)a.Tq * a.q_ld * 2 >= (1L << 31) || (long)a.Tq * a.kv_ld * 2 >= (1L << 31) || (long)a.Tq * a.out_ld * 2 >= (1L << 31)) return 1;
    gm.total_heads = batch * a.H;
    static int ncu = 0;
    if (!ncu) {
        int d = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&d) != hipSuccess || hipGetDeviceProperties(&prop, d) != hipSuccess) return UNIIR_ELAUNCH;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
#ifdef UNIIR_EXP_BUILD
    if (g_att_exp & 64) return 1;          // experiments: force the general kernel
#endif
    const int sm = ap_bwd_lds(TP, gm.nvl);
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_pair_kernel<288>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_bwd_pair_kernel<224>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const int grid = gm.total_heads < ncu ? gm.total_heads : ncu;
    if (TP == 288) hipLaunchKernelGGL(attn_bwd_pair_kernel<288>, dim3(grid), dim3(AP_THREADS), sm, st, a, gm);
    else hipLaunchKernelGGL(attn_bwd_pair_kernel<224>, dim3(grid), dim3(AP_THREADS), sm, st, a, gm);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
