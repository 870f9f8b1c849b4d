// 32x32x16-MFMA attention kernels for the plain case (no relative bias, no dropout, not causal): the CLIP / BLIP ViT towers'
// 257 / 197 / 577-token self-attention and rectangular cross-attention.  Included by attention.hip after AttnArgs.
//
// Why a second set of kernels: the 16x16x32 kernels above give every wave a 16-wide tile, so each K / V (forward) or Q / dO
// (backward) fragment read from LDS feeds one 16-column MFMA.  With 16 waves per CU that is 53 % LDS-pipe utilisation at 21 %
// matrix-pipe utilisation (profiles/r02_attention_pmc.txt): the LDS queue, not the arithmetic, sets the pace.  Here a wave owns
// a 32-wide tile: the same fragment feeds a 32x32x16 MFMA (twice the flops per LDS byte and per instruction), a workgroup is
// 4 waves (two workgroups = two heads per CU, LDS-capacity bound either way) and the 256-VGPR budget of 2 waves / SIMD pays for
// fragment prefetch one block ahead.
//
// Layouts (v_mfma_f32_32x32x16_bf16): A lane (m = lane & 31, h = lane >> 5) holds k = 8h .. 8h+7; B lane (n = lane & 31, h)
// likewise; C lane (n = lane & 31, h) register v holds row (v & 3) + 8 (v >> 2) + 4h.  An accumulator used as the next MFMA's
// operand therefore carries the k-slot order {16t + 4h + 0..3, 16t + 8 + 4h + 0..3} (t = k-step): the transposing LDS reads of
// the other operand fetch exactly those rows.
// LDS tiles are [row][64] bf16 with the 16-B chunk index XOR f(row), f = bit-reversed (row >> 1) & 7: conflict-free for the
// b128 row-fragment reads (lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) and for the b64 transposing reads, whose
// 32-lane halves touch rows R .. R+3 (R % 4 == 0) x 4 chunks: rows R and R + 2 must differ in chunk bit 2 -> row bit 1 -> f bit 2.
#pragma once

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define A32_THREADS 256
#define A32_WAVES 4

DEVINL int a32_f(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }
DEVINL f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
DEVINL float half_max(float v) {      // max with the lane 32 away (v_permlane32_swap: VALU, no LDS crossbar)
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
DEVINL float half_sum(float v) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
DEVINL bf16x8_t a32_lds128(const char* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(p)); }
DEVINL bf16x8_t a32_tr_pair(const char* p0, const char* p1) {     // k-slots 0..3 from p0, 4..7 from p1
    const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lds_read_tr16(p0));
    const u32x2_t h2 = __builtin_bit_cast(u32x2_t, lds_read_tr16(p1));
    const u32x4_t r = {l2[0], l2[1], h2[0], h2[1]};
    return __builtin_bit_cast(bf16x8_t, r);
}
// registers 8t .. 8t+7 of an accumulator as the bf16 operand of k-step t
DEVINL bf16x8_t a32_pack(const f32x16_t& x, int t) {
    const u32x4_t r = {pack_bf16x2(x[8 * t + 0], x[8 * t + 1]), pack_bf16x2(x[8 * t + 2], x[8 * t + 3]),
                       pack_bf16x2(x[8 * t + 4], x[8 * t + 5]), pack_bf16x2(x[8 * t + 6], x[8 * t + 7])};
    return __builtin_bit_cast(bf16x8_t, r);
}
// row fragment straight from global memory: lane (row r0 + (lane & 31), h) gets d = 16 s + 8 h .. +7 (rows >= T read as zero)
DEVINL bf16x8_t a32_rows_global(const unsigned short* __restrict__ src, long ld, int r0, int s, int lane, int T) {
    const int row = r0 + (lane & 31);
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (row < T) v = *reinterpret_cast<const u32x4_t*>(src + (long)row * ld + s * 16 + (lane >> 5) * 8);
    return __builtin_bit_cast(bf16x8_t, v);
}

// stage rows [0, Tp) of two [T][64] bf16 slices into swizzled LDS (zero padded); all loads of a batch are in flight before the
// first LDS store, `mid` runs between (independent work that shares the round trip)
template <int NL, class F>
DEVINL void a32_stage_batch(char* ldsA, const unsigned short* __restrict__ srcA, long ldA, char* ldsB,
                            const unsigned short* __restrict__ srcB, long ldB, int T, int Tp, int tid, int c0, F&& mid) {
    const int total = Tp * 8;
    u32x4_t va[NL], vb[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int c = c0 + u * A32_THREADS + tid;
        const int row = min(c >> 3, T - 1), kc = c & 7;
        va[u] = *reinterpret_cast<const u32x4_t*>(srcA + (long)row * ldA + kc * 8);
        vb[u] = *reinterpret_cast<const u32x4_t*>(srcB + (long)row * ldB + kc * 8);
    }
    mid();
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int c = c0 + u * A32_THREADS + tid;
        const int row = c >> 3, kc = c & 7;
        if (c < total) {
            const u32x4_t z = {0u, 0u, 0u, 0u};
            const int off = row * 128 + ((kc ^ a32_f(row)) << 4);
            *reinterpret_cast<u32x4_t*>(ldsA + off) = (row < T) ? va[u] : z;
            *reinterpret_cast<u32x4_t*>(ldsB + off) = (row < T) ? vb[u] : z;
        }
    }
}
template <class F>
DEVINL void a32_stage_two(char* ldsA, const unsigned short* __restrict__ srcA, long ldA, char* ldsB,
                          const unsigned short* __restrict__ srcB, long ldB, int T, int Tp, int tid, F&& mid) {
    const int nl = (Tp * 8 + A32_THREADS - 1) / A32_THREADS;       // wave-uniform
    if (nl <= 5) a32_stage_batch<5>(ldsA, srcA, ldA, ldsB, srcB, ldB, T, Tp, tid, 0, mid);
    else if (nl <= 9) a32_stage_batch<9>(ldsA, srcA, ldA, ldsB, srcB, ldB, T, Tp, tid, 0, mid);
    else {
        mid();
        for (int c0 = 0; c0 < Tp * 8; c0 += 8 * A32_THREADS)
            a32_stage_batch<8>(ldsA, srcA, ldA, ldsB, srcB, ldB, T, Tp, tid, c0, [] {});
    }
}

// lane-constant parts of the LDS fragment addresses (block row offsets are multiples of 32 rows = 4096 B and do not touch f)
struct A32Offsets {
    int R[4];        // row fragments, k-step s: row (lane & 31), chunk 2 s + h
    int T[2][2];     // transposed fragments, [32-wide column tile D][read 0 / 1]: rows 4 h + 8 rd + (tl >> 2), + 2048 per k-step t
};
DEVINL A32Offsets a32_offsets(int lane) {
    A32Offsets o;
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int s = 0; s < 4; ++s) o.R[s] = r * 128 + (((2 * s + hh) ^ a32_f(r)) << 4);
    const int tl = lane & 15, dh = (lane >> 4) & 1;
#pragma unroll
    for (int D = 0; D < 2; ++D)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int row = 4 * hh + 8 * rd + (tl >> 2);
            const int chunk = 4 * D + 2 * dh + ((tl & 3) >> 1);
            o.T[D][rd] = row * 128 + ((chunk ^ a32_f(row)) << 4) + (tl & 1) * 8;
        }
    return o;
}

#ifdef A32_STAMP
// timing build (experiments): s_memtime stamps of the first 64 workgroups' THIRD head, waves 0 and 7: [wg][wave][8]
__device__ unsigned long long a32_stamps[64 * 2 * 8];
#define A32_T(i)                                                                                   \
    do {                                                                                           \
        if (blockIdx.x < 64 && a32_iter == 2 && (w == 0 || w == 7) && lane == 0)                   \
            a32_stamps[(blockIdx.x * 2 + (w == 7)) * 8 + (i)] = __builtin_readcyclecounter();      \
    } while (0)
extern "C" int uniir_debug_a32_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(a32_stamps), sizeof(a32_stamps)) == hipSuccess ? 0 : -1;
}
#else
#define A32_T(i) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// forward: S^T = K Q^T (lane = query column: softmax statistics are lane-local up to one v_permlane32_swap),
// O^T = V^T P^T with P^T straight from the S^T accumulators.
//
// One persistent 8-wave workgroup per CU walking over (item, head) pairs g, g + gridDim.x, ...
//  * Two K / V buffers in LDS.  The next head's rows are fetched into registers at the start of the current head (NP 16-B
//    loads per thread: pieces of 8 rows, 1 KiB per wave) and written to the other buffer after pass 1, when they have landed: the
//    HBM round trip of the staging (28 % of a workgroup's life when exposed) is off the critical path.  (LDS-DMA was measured
//    first: ~400 clock ticks of issue per 1-KiB piece, 4 k of an 18 k-tick tile; and hipcc orders every later ds_read behind an
//    LDS-DMA it knows about with vmcnt(0).)  The wave's query fragments of the next head are requested before these loads
//    (in-order returns) and all of them branch-free: a conditional load makes the compiler's vmcnt bookkeeping fall back to
//    vmcnt(0) at the next use of any loaded register.
//  * Wave w owns the 32-query tile w; the tail tile (Tq % 32 rows: ONE row at 257 tokens) is split over the 8 waves by key
//    blocks, its query rows sit in LDS next to K / V, the partial (max, sum, O) are combined through a small LDS area by wave 0
//    -- 9 tiles on 8 waves would otherwise cost two tile times for 1.1 tiles of work.
//  * Two passes over the key blocks instead of an online softmax.  Pass 1: S^T and the row maxima only (4 MFMAs + 8 VALU per
//    block).  Pass 2: S^T again, P = exp2(c S^T - m) with the FINAL maximum -- no running maximum, no rescaling of O (that was 32
//    of 85 VALU instructions per block on a SIMD whose VALU, not its matrix pipe, was the busy unit: 38 % MFMA utilisation).  Pass 2
//    is a depth-2 software pipeline, one basic block per key block: the MFMAs of S^T(kb + 1) and P V(kb - 1) and the softmax of
//    block kb are mutually independent, so the compiler interleaves them (an in-order wave cannot overlap its own matrix and
//    vector work any other way).  The mathematics is the plain softmax of the reference.
//  * One barrier per head.
// Limits (the launcher sends everything else to the 16x16x32 kernels): Tq <= 287 (one full tile per wave), Tk <= 320 (two
// buffers), no key lengths.
// ---------------------------------------------------------------------------------------------------------------------------
#define A32F_THREADS 512
#define A32F_WAVES 8
#define A32_PART_STRIDE 34      // floats per (wave, query lane) partial: m, l, O[32]

template <int NPK>               // K (and V) pieces per wave: ceil(Tkp / 8 / 8) <= NPK
__global__ __launch_bounds__(A32F_THREADS, 2) void attn32_fwd_kernel(AttnArgs a, int total_heads) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int Tq = a.Tq, Tk = a.Tk, H = a.H;
    const int Tkp = (Tk + 31) & ~31;
    const int nfull = Tq >> 5, rem = Tq & 31;
    const int buf_bytes = 2 * Tkp * 128 + (rem ? 4096 : 0);  // K, V, then the 32 rows of the tail query tile
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const bool has_full = w < nfull;
    float* part_base = reinterpret_cast<float*>(lds + 2 * buf_bytes);
    const int part_floats = A32F_WAVES * 2 * rem * A32_PART_STRIDE;
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int npieces = Tkp >> 3;
    constexpr int NP = 2 * NPK + 1;
    auto q_of = [&](int g) { return a.q + (long)(g / H) * Tq * a.q_ld + (g % H) * ATT_D; };
    // piece j of this wave: K pieces w, w + 8, ..., then the same V pieces, then (waves 0..3) one piece of the tail query tile.
    // Loads are unconditional (clamped rows / pieces); the LDS write is what is predicated.
    // part 0: the K pieces, part 1: the V pieces, part 2: the tail-query piece -- issued at three different points of a head: all
    // 15 loads of the 8 waves at once (120 wave-instructions per CU) stall in the issue queue for ~2.5 us
    auto piece_load = [&](auto part, int g, int ln, u32x4_t (&pr)[NP]) {
        constexpr int PART = decltype(part)::value;
        const int gg = min(g, total_heads - 1);
        const int kc = (ln & 7) ^ a32_f((w & 1) * 8 + (ln >> 3));          // rows 8 p + (ln >> 3): f sees bit 3 = p & 1 = w & 1
        if (PART < 2) {
            const unsigned short* b_ = (PART == 0 ? a.k : a.v) + (long)(gg / H) * Tk * a.kv_ld + (gg % H) * ATT_D;
#pragma unroll
            for (int j = 0; j < NPK; ++j) {
                const int row = min((w + A32F_WAVES * j) * 8 + (ln >> 3), Tk - 1);
                pr[PART * NPK + j] = *reinterpret_cast<const u32x4_t*>(b_ + (long)row * a.kv_ld + kc * 8);
            }
        } else {
            const int qrow = min(nfull * 32 + (w & 3) * 8 + (ln >> 3), Tq - 1);
            pr[2 * NPK] = *reinterpret_cast<const u32x4_t*>(q_of(gg) + (long)qrow * a.q_ld + kc * 8);
        }
    };
    auto piece_store = [&](char* buf, int ln, const u32x4_t (&pr)[NP]) {
        // every piece load has landed from here on, also the ones this wave does not store (p >= npieces): without the explicit
        // wait the compiler protects their destination registers with vmcnt(0) at some later reuse, in the middle of pass 2
        __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0)
#pragma unroll
        for (int j = 0; j < NPK; ++j) {
            const int p = w + A32F_WAVES * j;
            if (p < npieces) {
                *reinterpret_cast<u32x4_t*>(buf + p * 1024 + ln * 16) = pr[j];
                *reinterpret_cast<u32x4_t*>(buf + Tkp * 128 + p * 1024 + ln * 16) = pr[NPK + j];
            }
        }
        if (rem && w < 4) *reinterpret_cast<u32x4_t*>(buf + 2 * Tkp * 128 + w * 1024 + ln * 16) = pr[2 * NPK];
    };
    auto load_q = [&](int g, bf16x8_t (&qx)[4], int ln) {
        const unsigned short* qb_ = q_of(min(g, total_heads - 1));
        const int row = min(w * 32 + (ln & 31), Tq - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qx[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(qb_ + (long)row * a.q_ld + s * 16 + (ln >> 5) * 8));
    };

    int g = blockIdx.x;
    if (g >= total_heads) return;
    bf16x8_t qn[4];                         // this wave's full tile of the next head
    {
        u32x4_t pr[NP];
        load_q(g, qn, lane);
        piece_load(std::integral_constant<int, 0>{}, g, lane, pr);
        piece_load(std::integral_constant<int, 1>{}, g, lane, pr);
        piece_load(std::integral_constant<int, 2>{}, g, lane, pr);
        piece_store(lds, lane, pr);
    }
    __syncthreads();
    int par = 0;
    int a32_iter = 0;
    for (; g < total_heads; g += gridDim.x, par ^= 1, ++a32_iter) {
        const int gn = g + gridDim.x;
        const int m = g / H, hd = g % H;
        A32_T(0);
        // per-lane global addresses are recomputed every head from an opaque copy of the lane id (hoisted out of the loop they
        // are dozens of live registers that get spilled)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        char* buf = lds + par * buf_bytes;
        char* nbuf = lds + (par ^ 1) * buf_bytes;
        const char* ldsK = buf;
        const char* ldsV = buf + Tkp * 128;
        float* part = part_base + par * part_floats;
        bf16x8_t qc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qc[s] = qn[s];
        load_q(gn, qn, lane_o);
        u32x4_t pr[NP];                      // the next head's pieces: requested during pass 1 of the full tile, one part per trip
        const int nkb = (Tk + 31) >> 5;
        A32_T(1);
        // the wave's full tile, then its share of the tail tile: two instances of the same code (a loop over the two would
        // make the compiler drain vmcnt on the back edge)
        auto run_item = [&](auto is_tail) {
            constexpr bool tail = decltype(is_tail)::value;
            int lane_i = lane;                              // (the LDS fragment offsets are rebuilt per item: not worth 8 registers
            asm volatile("" : "+v"(lane_i));                //  kept -- and spilled -- across the whole kernel)
            const A32Offsets off = a32_offsets(lane_i);
            const int q0 = (tail ? nfull : w) * 32, q = q0 + r;
            bf16x8_t qf[4];
            A32_T(2 + (tail ? 1 : 0));
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[s] = tail ? a32_lds128(buf + 2 * Tkp * 128 + off.R[s]) : qc[s];
            // key blocks of this item: all of them for a full tile, this wave's contiguous share for the tail
            const int kb_lo = tail ? (w * nkb) / A32F_WAVES : 0;
            const int kb_hi = tail ? ((w + 1) * nkb) / A32F_WAVES : nkb;
            const bool last_edge = kb_hi * 32 > Tk;         // only ever the last block of the range
            f32x16_t o[2] = {zero16, zero16};
            float m_row = -1e30f, l_run = 0.f;
            auto k_frags = [&](int kb, bf16x8_t (&kf)[4]) {
                const char* kp = ldsK + kb * 4096;
#pragma unroll
                for (int s = 0; s < 4; ++s) kf[s] = a32_lds128(kp + off.R[s]);
            };
            auto v_frags = [&](int kb, bf16x8_t (&vt)[2][2]) {
                const char* vp = ldsV + kb * 4096;
#pragma unroll
                for (int D = 0; D < 2; ++D)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        vt[D][t] = a32_tr_pair(vp + off.T[D][0] + t * 2048, vp + off.T[D][1] + t * 2048);
            };
            auto mask_edge = [&](int kb, f32x16_t& st) {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (kb * 32 + (v & 3) + 8 * (v >> 2) + 4 * hh >= Tk) st[v] = -1e30f;
            };
            if (kb_lo < kb_hi) {
                // ---- pass 1: row maxima.  Two blocks per trip (two independent MFMA chains); an odd count repeats the last block.
                {
                    float mx = -1e30f;
                    for (int kb = kb_lo; kb < kb_hi; kb += 2) {
                        const int kb1 = min(kb + 1, kb_hi - 1);
                        if (!tail) {                                // (static register indices, run-time trip test)
                            if (kb == kb_lo) piece_load(std::integral_constant<int, 0>{}, gn, lane_o, pr);
                            else if (kb == kb_lo + 2) piece_load(std::integral_constant<int, 1>{}, gn, lane_o, pr);
                            else if (kb == kb_lo + 4) piece_load(std::integral_constant<int, 2>{}, gn, lane_o, pr);
                        }
                        bf16x8_t k0[4], k1[4];
                        k_frags(kb, k0);
                        k_frags(kb1, k1);
                        f32x16_t s0 = mfma32(k0[0], qf[0], zero16), s1 = mfma32(k1[0], qf[0], zero16);
#pragma unroll
                        for (int s = 1; s < 4; ++s) {
                            s0 = mfma32(k0[s], qf[s], s0);
                            s1 = mfma32(k1[s], qf[s], s1);
                        }
                        if (last_edge && kb1 == kb_hi - 1) {
                            if (kb == kb1) mask_edge(kb, s0);
                            mask_edge(kb1, s1);
                        }
#pragma unroll
                        for (int v = 0; v < 16; v += 2) mx = fmaxf(mx, fmaxf(s0[v], s0[v + 1]));
#pragma unroll
                        for (int v = 0; v < 16; v += 2) mx = fmaxf(mx, fmaxf(s1[v], s1[v + 1]));
                    }
                    m_row = half_max(mx) * SCALE_LOG2E;
                }
                if (!tail) {
                    A32_T(6);
                    if (kb_hi - kb_lo <= 2) piece_load(std::integral_constant<int, 1>{}, gn, lane_o, pr);   // (short key loops)
                    if (kb_hi - kb_lo <= 4) piece_load(std::integral_constant<int, 2>{}, gn, lane_o, pr);
                    piece_store(nbuf, lane_o, pr);              // waits for the pieces: most of pass 1 lies behind the requests
                    A32_T(7);
                }
                // ---- pass 2
                struct Stage {
                    f32x16_t st;            // logits of the block whose softmax comes next
                    bf16x8_t pf[2];         // packed probabilities of the block whose P V comes next
                };
                bf16x8_t kf[4];             // K fragments of block kb + 1 at the start of step kb (one buffer, refilled late)
                auto softmax = [&](auto edge, int kb, f32x16_t& st, bf16x8_t (&pf)[2]) {
                    if (decltype(edge)::value) mask_edge(kb, st);
                    f32x16_t pv = st * SCALE_LOG2E - m_row;     // packed fma
#pragma unroll
                    for (int v = 0; v < 16; ++v) pv[v] = __builtin_amdgcn_exp2f(pv[v]);
                    // row-sum partial as a tree of packed adds (the two lanes of a query are added at the end)
                    typedef __attribute__((ext_vector_type(8))) float f32x8_t;
                    const f32x8_t s8 = __builtin_shufflevector(pv, pv, 0, 1, 2, 3, 4, 5, 6, 7) +
                                       __builtin_shufflevector(pv, pv, 8, 9, 10, 11, 12, 13, 14, 15);
                    const f32x4_t s4 = __builtin_shufflevector(s8, s8, 0, 1, 2, 3) + __builtin_shufflevector(s8, s8, 4, 5, 6, 7);
                    l_run += (s4[0] + s4[1]) + (s4[2] + s4[3]);
                    pf[0] = a32_pack(pv, 0);
                    pf[1] = a32_pack(pv, 1);
                };
                auto pv_mfma = [&](int kb, const Stage& x) {
                    bf16x8_t vt[2][2];
                    v_frags(kb, vt);
                    o[0] = mfma32(vt[0][0], x.pf[0], o[0]);
                    o[1] = mfma32(vt[1][0], x.pf[0], o[1]);
                    o[0] = mfma32(vt[0][1], x.pf[1], o[0]);
                    o[1] = mfma32(vt[1][1], x.pf[1], o[1]);
                };
                // cur: {S^T(kb), P(kb - 1)}  ->  nxt: {S^T(kb + 1), P(kb)};  kf: K(kb + 1) -> K(kb + 2);  first = no P V yet
                auto step = [&](auto edge, auto first, int kb, Stage& cur, Stage& nxt) {
                    bf16x8_t vt[2][2];
                    if (!decltype(first)::value) v_frags(kb - 1, vt);   // used by the P V MFMAs in the second half of the step
                    nxt.st = mfma32(kf[0], qf[0], zero16);
                    nxt.st = mfma32(kf[1], qf[1], nxt.st);
                    if (!decltype(first)::value) {
                        o[0] = mfma32(vt[0][0], cur.pf[0], o[0]);
                        nxt.st = mfma32(kf[2], qf[2], nxt.st);
                        o[1] = mfma32(vt[1][0], cur.pf[0], o[1]);
                        nxt.st = mfma32(kf[3], qf[3], nxt.st);
                        o[0] = mfma32(vt[0][1], cur.pf[1], o[0]);
                        o[1] = mfma32(vt[1][1], cur.pf[1], o[1]);
                    } else {
                        nxt.st = mfma32(kf[2], qf[2], nxt.st);
                        nxt.st = mfma32(kf[3], qf[3], nxt.st);
                    }
                    k_frags(min(kb + 2, kb_hi - 1), kf);        // after the MFMAs that read the old fragments were issued
                    softmax(edge, kb, cur.st, nxt.pf);
                };
                Stage A, B;
                k_frags(kb_lo, kf);
                A.st = mfma32(kf[0], qf[0], zero16);
#pragma unroll
                for (int s = 1; s < 4; ++s) A.st = mfma32(kf[s], qf[s], A.st);
                k_frags(min(kb_lo + 1, kb_hi - 1), kf);
                // steps kb_lo + 1 .. kb_hi - 2 are branch-free bodies; the first (no P V yet) and the last (key boundary) are peeled
                const std::false_type no{};
                const std::true_type yes{};
                if (kb_lo + 1 == kb_hi) {                       // a single block
                    if (last_edge) step(yes, yes, kb_lo, A, B);
                    else step(no, yes, kb_lo, A, B);
                    pv_mfma(kb_lo, B);
                } else {
                    step(no, yes, kb_lo, A, B);                 // B current from here on
                    int kb = kb_lo + 1;
                    for (; kb + 2 < kb_hi; kb += 2) {
                        step(no, no, kb, B, A);
                        step(no, no, kb + 1, A, B);
                    }
                    if (kb + 1 < kb_hi) {                       // one inner step left, then the last one: B -> A -> B
                        step(no, no, kb, B, A);
                        if (last_edge) step(yes, no, kb + 1, A, B);
                        else step(no, no, kb + 1, A, B);
                        pv_mfma(kb + 1, B);
                    } else {                                    // the last step: B -> A
                        if (last_edge) step(yes, no, kb, B, A);
                        else step(no, no, kb, B, A);
                        pv_mfma(kb, A);
                    }
                }
            }
            if (!tail) {
                l_run = half_sum(l_run);
                const float inv = 1.0f / l_run;
                unsigned short* orow = a.out + ((long)m * Tq + q) * a.out_ld + hd * ATT_D;
#pragma unroll
                for (int D = 0; D < 2; ++D)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const u32x2_t pk2 = {pack_bf16x2(o[D][4 * j] * inv, o[D][4 * j + 1] * inv),
                                             pack_bf16x2(o[D][4 * j + 2] * inv, o[D][4 * j + 3] * inv)};
                        *reinterpret_cast<u32x2_t*>(orow + 32 * D + 8 * j + 4 * hh) = pk2;
                    }
                if (hh == 0) a.lse[((long)m * H + hd) * Tq + q] = m_row * LN2F + __logf(l_run);
            } else if (r < rem) {        // partial of the tail rows over this wave's key blocks
                float* pp = part + ((w * rem + r) * 2 + hh) * A32_PART_STRIDE;
                pp[0] = m_row;
                pp[1] = l_run;
#pragma unroll
                for (int D = 0; D < 2; ++D)
#pragma unroll
                    for (int v = 0; v < 16; ++v) pp[2 + 16 * D + v] = o[D][v];
            }
        };
        if (has_full) run_item(std::false_type{});
        else {                                                   // waves without a full tile
            piece_load(std::integral_constant<int, 0>{}, gn, lane_o, pr);
            piece_load(std::integral_constant<int, 1>{}, gn, lane_o, pr);
            piece_load(std::integral_constant<int, 2>{}, gn, lane_o, pr);
            piece_store(nbuf, lane_o, pr);
        }
        if (rem) run_item(std::true_type{});
        // every LDS read of this head is done, the tail partials and the next head's rows are visible after the barrier
        A32_T(4);
        __syncthreads();
        A32_T(5);
        if (rem && w == 0) {         // combine: lane = output column d, one tail row at a time (coalesced 128-B row store)
            const int D = lane >> 5, hv = (lane >> 2) & 1, v = ((lane >> 3) & 3) * 4 + (lane & 3);
            for (int rr = 0; rr < rem; ++rr) {
                float mw[A32F_WAVES], lw[A32F_WAVES], ow[A32F_WAVES], mx = -1e30f;
#pragma unroll
                for (int x = 0; x < A32F_WAVES; ++x) {
                    const float* p0 = part + ((x * rem + rr) * 2) * A32_PART_STRIDE;
                    mw[x] = p0[0];
                    lw[x] = p0[1] + p0[A32_PART_STRIDE + 1];
                    ow[x] = p0[hv * A32_PART_STRIDE + 2 + 16 * D + v];
                    mx = fmaxf(mx, mw[x]);
                }
                float l = 0.f, acc = 0.f;
#pragma unroll
                for (int x = 0; x < A32F_WAVES; ++x) {
                    const float sc = __builtin_amdgcn_exp2f(mw[x] - mx);
                    l += lw[x] * sc;
                    acc += ow[x] * sc;
                }
                const int q = nfull * 32 + rr;
                a.out[((long)m * Tq + q) * a.out_ld + hd * ATT_D + lane] = f32_to_bf16(acc / l);
                if (lane == 0) a.lse[((long)m * H + hd) * Tq + q] = mx * LN2F + __logf(l);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward.  One 4-wave workgroup per (item, head), two workgroups per CU (8 waves at up to 256 registers; the two drift out of
// phase, so one computes while the other waits for its rows -- a persistent one-head-per-CU variant with all four slices in LDS
// was measured first: every workgroup stages at the same time, 27 k clock ticks of 90 k per head with the HBM idle in between).
//   stage Q, dO  ->  phase 1 (wave w = 32-key tiles w, w + 4, ..; lane = key column):  S = Q K^T, dP = dO V^T per 32-query block,
//            P = exp2(c S - lse), dS = P (dP - D);  dV^T += dO^T P,  dK^T += Q^T dS  (P / dS feed the next MFMA straight from
//            the accumulators);
//   stage K, V (same buffers)  ->  phase 2 (wave w = 32-query tiles w, w + 4, ..; lane = query column):  S^T = K Q^T,
//            dP^T = V dO^T per 32-key block, the same P / dS transposed (lse, D are lane scalars), dQ^T += K^T dS^T.
// The tail tiles (Tk % 32 keys / Tq % 32 queries: ONE row each at 257 tokens) are split over the 4 waves along the inner loop
// and reduced with LDS float atomics into a small area -- 9 tiles on 4 waves would otherwise cost 3 tile times for 2.25.
// Padded rows: slice rows beyond the end are copies of the last row (finite); lse2 = 1e30 (P = 0) and D = 0 beyond Tq; keys
// beyond Tk are zeroed in phase 2's last block and simply not stored in phase 1 (a lane owns a key there).
// Limits (the launcher sends everything else to the 16x16x32 kernels): no key lengths, tails <= 16 rows.
// ---------------------------------------------------------------------------------------------------------------------------
#define A32B_THREADS 256
#define A32B_WAVES 4

#ifdef A32_STAMP
#define B32_T(i)                                                                                   \
    do {                                                                                           \
        if (blockIdx.x < 64 && (w == 0 || w == 3) && lane == 0)                                    \
            a32_stamps[(blockIdx.x * 2 + (w == 3)) * 8 + (i)] = __builtin_readcyclecounter();      \
    } while (0)
#else
#define B32_T(i) do {} while (0)
#endif

__global__ __launch_bounds__(A32B_THREADS, 2) void attn32_bwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int Tq = a.Tq, Tk = a.Tk, H = a.H;
    const int Tqp = (Tq + 31) & ~31, Tkp = (Tk + 31) & ~31;
    const int Tmax = max(Tqp, Tkp);
    char* bufA = lds;                       // Q, later K
    char* bufB = lds + Tmax * 128;          // dO, later V
    float* lse2 = reinterpret_cast<float*>(lds + 2 * Tmax * 128);
    float* Dq = lse2 + Tqp;
    // per query 8 bf16: {x_hi, x_lo, 0, 0, n_hi, n_lo, 0, 0}, x = -lse2 / c and n = -D split into two bf16 each: phase 1 adds them to
    // S and dP inside the matrix pipe (a fifth k-step against ones) instead of reading 32 floats per lane and block from LDS
    unsigned* aug = reinterpret_cast<unsigned*>(Dq + Tqp);
    float* red = reinterpret_cast<float*>(aug + 4 * Tqp);   // tail reductions: phase 1 [2][remk][64], phase 2 [remq][64]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int nfq = Tq >> 5, remq = Tq & 31, nfk = Tk >> 5, remk = Tk & 31;
    const int nqb = Tqp >> 5, nkb = Tkp >> 5;
    float* red2 = red + 2 * remk * 64;
    const int red_floats = (2 * remk + remq) * 64;
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float oscale = ATT_SCALE;
    const int m = blockIdx.x / H, hd = blockIdx.x % H;
    const unsigned short* qbase = a.q + (long)m * Tq * a.q_ld + hd * ATT_D;
    const unsigned short* kbase = a.k + (long)m * Tk * a.kv_ld + hd * ATT_D;
    const unsigned short* vbase = a.v + (long)m * Tk * a.kv_ld + hd * ATT_D;
    const unsigned short* obase = a.out + (long)m * Tq * a.out_ld + hd * ATT_D;
    const unsigned short* dobase = a.dout + (long)m * Tq * a.out_ld + hd * ATT_D;
    unsigned short* dqbase = a.dq + (long)m * Tq * a.dq_ld + hd * ATT_D;
    unsigned short* dkbase = a.dk + (long)m * Tk * a.dkv_ld + hd * ATT_D;
    unsigned short* dvbase = a.dv + (long)m * Tk * a.dkv_ld + hd * ATT_D;
    const A32Offsets off = a32_offsets(lane);
    B32_T(0);
    // ---- stage Q, dO (all loads in flight, the row statistics' own loads ride the same round trip), reduction area
    a32_stage_two(bufA, qbase, a.q_ld, bufB, dobase, a.out_ld, Tq, Tqp, tid, [&] {
        for (int rr = tid; rr < Tqp; rr += A32B_THREADS) {          // D[q] = dO[q] . O[q],  lse2[q] = lse[q] log2 e
            float d = 0.f, l = 1e30f;
            if (rr < Tq) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const u32x4_t x = *reinterpret_cast<const u32x4_t*>(obase + (long)rr * a.out_ld + c * 8);
                    const u32x4_t y = *reinterpret_cast<const u32x4_t*>(dobase + (long)rr * a.out_ld + c * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        d += __uint_as_float(x[e] << 16) * __uint_as_float(y[e] << 16);
                        d += __uint_as_float(x[e] & 0xffff0000u) * __uint_as_float(y[e] & 0xffff0000u);
                    }
                }
                l = a.lse[((long)m * H + hd) * Tq + rr] * LOG2EF;
            }
            Dq[rr] = d;
            lse2[rr] = l;
            const float x = -l * (1.0f / SCALE_LOG2E), n = -d;
            const float xh = bf16_to_f32(f32_to_bf16(x)), nh = bf16_to_f32(f32_to_bf16(n));
            const u32x4_t av = {pack_bf16x2(xh, x - xh), 0u, pack_bf16x2(nh, n - nh), 0u};
            *reinterpret_cast<u32x4_t*>(aug + 4 * rr) = av;
        }
        for (int i = tid; i < red_floats; i += A32B_THREADS) red[i] = 0.f;
    });
    B32_T(1);
    __syncthreads();
    B32_T(2);

    // ---- phase 1: dK, dV of a key tile (full: all query blocks; tail: the wave's share of the query blocks)
    auto phase1 = [&](auto is_tail, int kt) {
        constexpr bool tail = decltype(is_tail)::value;
        const int k0 = kt * 32, key = k0 + r;
        const int qb_lo = tail ? (w * nqb) / A32B_WAVES : 0;
        const int qb_hi = tail ? ((w + 1) * nqb) / A32B_WAVES : nqb;
        if (qb_lo >= qb_hi) return;
        bf16x8_t kf[4], vf[4];                          // B operands: this lane's key row, from global (clamped)
        {
            const int krow = min(key, Tk - 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                kf[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(kbase + (long)krow * a.kv_ld + s * 16 + hh * 8));
                vf[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(vbase + (long)krow * a.kv_ld + s * 16 + hh * 8));
            }
        }
        f32x16_t dk[2] = {zero16, zero16}, dv[2] = {zero16, zero16};
        const u32x4_t z4 = {0u, 0u, 0u, 0u};
        const u32x4_t one4 = {hh ? 0u : 0x3F803F80u, 0u, 0u, 0u};     // B operand of the fifth k-step: ones in k-slots 0, 1
        const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, one4);
        (void)z4;
        // Per query block b:  R(b) row fragments of Q / dO (+ the augmented pair)  ->  M1(b): S + x, dP - D (10 MFMAs)  ->  V(b):
        // P = exp2(c S'), dS = P dP' packed for the matrix pipe  ->  T(b) transposed fragments  ->  M2(b): dV^T, dK^T (8 MFMAs).
        // One step runs V(b) next to M2(b - 1) (independent: the compiler interleaves the VALU and the MFMAs) and then M1(b + 1)
        // into the registers V(b) has just emptied.
        struct Packed { bf16x8_t pf[2], df[2]; };
        f32x16_t sx, dpx;
        auto rows_m1 = [&](int qb) {                        // R(qb) + M1(qb)
            const char* qp = bufA + qb * 4096;
            const char* dp_ = bufB + qb * 4096;
            bf16x8_t qa[4], da[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qa[s] = a32_lds128(qp + off.R[s]);
                da[s] = a32_lds128(dp_ + off.R[s]);
            }
            const u32x4_t ag = *reinterpret_cast<const u32x4_t*>(aug + 4 * (qb * 32 + r));
            const u32x4_t as4 = {hh ? 0u : ag[0], 0u, 0u, 0u}, ad4 = {hh ? 0u : ag[2], 0u, 0u, 0u};
            sx = mfma32(qa[0], kf[0], zero16);
            dpx = mfma32(da[0], vf[0], zero16);
#pragma unroll
            for (int s = 1; s < 4; ++s) {
                sx = mfma32(qa[s], kf[s], sx);
                dpx = mfma32(da[s], vf[s], dpx);
            }
            sx = mfma32(__builtin_bit_cast(bf16x8_t, as4), ones, sx);
            dpx = mfma32(__builtin_bit_cast(bf16x8_t, ad4), ones, dpx);
        };
        auto valu = [&](Packed& out) {                      // V: consumes sx, dpx
            f32x16_t pv = sx * SCALE_LOG2E;
#pragma unroll
            for (int v = 0; v < 16; ++v) pv[v] = __builtin_amdgcn_exp2f(pv[v]);
            const f32x16_t ds = pv * dpx;
            out.pf[0] = a32_pack(pv, 0);
            out.pf[1] = a32_pack(pv, 1);
            out.df[0] = a32_pack(ds, 0);
            out.df[1] = a32_pack(ds, 1);
        };
        auto m2 = [&](int qb, const Packed& in) {           // T(qb) + M2(qb)
            const char* qp = bufA + qb * 4096;
            const char* dp_ = bufB + qb * 4096;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int D = 0; D < 2; ++D) {
                    dv[D] = mfma32(a32_tr_pair(dp_ + off.T[D][0] + t * 2048, dp_ + off.T[D][1] + t * 2048), in.pf[t], dv[D]);
                    dk[D] = mfma32(a32_tr_pair(qp + off.T[D][0] + t * 2048, qp + off.T[D][1] + t * 2048), in.df[t], dk[D]);
                }
        };
        {
            Packed A, B;
            rows_m1(qb_lo);
            valu(A);                                        // block qb_lo (nothing to overlap with yet)
            int qb = qb_lo + 1;
            if (qb < qb_hi) rows_m1(qb);
            for (; qb + 1 < qb_hi; qb += 2) {               // A holds block qb - 1; sx / dpx hold block qb
                valu(B);
                m2(qb - 1, A);
                rows_m1(qb + 1);
                valu(A);
                m2(qb, B);
                if (qb + 2 < qb_hi) rows_m1(qb + 2);
            }
            if (qb < qb_hi) {                               // one block left: A = qb - 1, sx / dpx = qb
                valu(B);
                m2(qb - 1, A);
                m2(qb, B);
            } else {
                m2(qb - 1, A);
            }
        }
        // dK^T / dV^T: lane (key, hh), register v -> d = 32 D + (v & 3) + 8 (v >> 2) + 4 hh
        if (!tail) {
            unsigned short* krow = dkbase + (long)key * a.dkv_ld;
            unsigned short* vrow = dvbase + (long)key * a.dkv_ld;
#pragma unroll
            for (int D = 0; D < 2; ++D)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x2_t pk2 = {pack_bf16x2(dk[D][4 * j] * oscale, dk[D][4 * j + 1] * oscale),
                                         pack_bf16x2(dk[D][4 * j + 2] * oscale, dk[D][4 * j + 3] * oscale)};
                    *reinterpret_cast<u32x2_t*>(krow + 32 * D + 8 * j + 4 * hh) = pk2;
                    const u32x2_t pv2 = {pack_bf16x2(dv[D][4 * j], dv[D][4 * j + 1]), pack_bf16x2(dv[D][4 * j + 2], dv[D][4 * j + 3])};
                    *reinterpret_cast<u32x2_t*>(vrow + 32 * D + 8 * j + 4 * hh) = pv2;
                }
        } else if (r < remk) {
            float* rk = red + r * 64;
            float* rv = red + (remk + r) * 64;
#pragma unroll
            for (int D = 0; D < 2; ++D)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int d = 32 * D + (v & 3) + 8 * (v >> 2) + 4 * hh;
                    atomicAdd(rk + d, dk[D][v]);
                    atomicAdd(rv + d, dv[D][v]);
                }
        }
    };
    for (int kt = w; kt < nfk; kt += A32B_WAVES) phase1(std::false_type{}, kt);
    B32_T(3);
    if (remk) phase1(std::true_type{}, nfk);
    B32_T(4);
    __syncthreads();                     // every read of Q / dO is done; the tail sums are complete
    // ---- stage K, V into the same buffers; wave 0 writes the tail key rows meanwhile (lane = column)
    a32_stage_two(bufA, kbase, a.kv_ld, bufB, vbase, a.kv_ld, Tk, Tkp, tid, [&] {
        if (remk && w == 0)
            for (int rr = 0; rr < remk; ++rr) {
                dkbase[(long)(nfk * 32 + rr) * a.dkv_ld + lane] = f32_to_bf16(red[rr * 64 + lane] * oscale);
                dvbase[(long)(nfk * 32 + rr) * a.dkv_ld + lane] = f32_to_bf16(red[(remk + rr) * 64 + lane]);
            }
    });
    __syncthreads();
    B32_T(5);

    // ---- phase 2: dQ of a query tile (full: all key blocks; tail: the wave's share of the key blocks)
    auto phase2 = [&](auto is_tail, int qt) {
        constexpr bool tail = decltype(is_tail)::value;
        const int q0 = qt * 32, q = q0 + r;
        const int kb_lo = tail ? (w * nkb) / A32B_WAVES : 0;
        const int kb_hi = tail ? ((w + 1) * nkb) / A32B_WAVES : nkb;
        if (kb_lo >= kb_hi) return;
        bf16x8_t qf[4], dof[4];                         // B operands: this lane's query row, from global (clamped)
        {
            const int qrow = min(q, Tq - 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qf[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(qbase + (long)qrow * a.q_ld + s * 16 + hh * 8));
                dof[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(dobase + (long)qrow * a.out_ld + s * 16 + hh * 8));
            }
        }
        const float my_lse = lse2[q], my_D = Dq[q];     // (q < Tqp; padded rows: 1e30 / 0 -> P = 0)
        f32x16_t dq[2] = {zero16, zero16};
        for (int kb = kb_lo; kb < kb_hi; ++kb) {
            const char* kp = bufA + kb * 4096;
            const char* vp = bufB + kb * 4096;
            bf16x8_t ka[4], va[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                ka[s] = a32_lds128(kp + off.R[s]);
                va[s] = a32_lds128(vp + off.R[s]);
            }
            f32x16_t sx = mfma32(ka[0], qf[0], zero16), dpx = mfma32(va[0], dof[0], zero16);
#pragma unroll
            for (int s = 1; s < 4; ++s) {
                sx = mfma32(ka[s], qf[s], sx);
                dpx = mfma32(va[s], dof[s], dpx);
            }
            f32x16_t ds;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[v], SCALE_LOG2E, -my_lse));
                ds[v] = p * (dpx[v] - my_D);
            }
            if (kb * 32 + 32 > Tk) {                    // key boundary (wave-uniform): rows = keys (v & 3) + 8 (v >> 2) + 4 hh
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (kb * 32 + (v & 3) + 8 * (v >> 2) + 4 * hh >= Tk) ds[v] = 0.f;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8_t df = a32_pack(ds, t);
#pragma unroll
                for (int D = 0; D < 2; ++D)
                    dq[D] = mfma32(a32_tr_pair(kp + off.T[D][0] + t * 2048, kp + off.T[D][1] + t * 2048), df, dq[D]);
            }
        }
        if (!tail) {
            unsigned short* qrow = dqbase + (long)q * a.dq_ld;
#pragma unroll
            for (int D = 0; D < 2; ++D)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x2_t pk2 = {pack_bf16x2(dq[D][4 * j] * oscale, dq[D][4 * j + 1] * oscale),
                                         pack_bf16x2(dq[D][4 * j + 2] * oscale, dq[D][4 * j + 3] * oscale)};
                    *reinterpret_cast<u32x2_t*>(qrow + 32 * D + 8 * j + 4 * hh) = pk2;
                }
        } else if (r < remq) {
            float* rq = red2 + r * 64;
#pragma unroll
            for (int D = 0; D < 2; ++D)
#pragma unroll
                for (int v = 0; v < 16; ++v) atomicAdd(rq + 32 * D + (v & 3) + 8 * (v >> 2) + 4 * hh, dq[D][v]);
        }
    };
    for (int qt = w; qt < nfq; qt += A32B_WAVES) phase2(std::false_type{}, qt);
    B32_T(6);
    if (remq) {
        phase2(std::true_type{}, nfq);
        __syncthreads();                 // the tail sums are complete
        if (w == 0)
            for (int rr = 0; rr < remq; ++rr)
                dqbase[(long)(nfq * 32 + rr) * a.dq_ld + lane] = f32_to_bf16(red2[rr * 64 + lane] * oscale);
    }
    B32_T(7);
}
