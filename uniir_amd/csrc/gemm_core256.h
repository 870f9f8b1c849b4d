// LDS-DMA MFMA GEMM main loop for gfx950, templated on the block shape.
//
// Every wave owns a 128x64 sub-tile (8x4 MFMA 16x16x32 tiles = 128 accumulator registers); a workgroup is
// WAVES_M x WAVES_N waves, i.e. a (128*WAVES_M) x (64*WAVES_N) block tile, K step BK (32 or 64).
//   <2,4,64>: 256x256x64, 8 waves, 128 KiB LDS, 1 workgroup / CU: the GEMM fallback loop (glds_mainloop_asm) when a K
//             split has fewer than 3 steps; the production loop of this shape is the ping-pong one in gemm_core_pp.h
//   <2,2,32>: 256x128x32, 4 waves,  48 KiB LDS, 2-3 workgroups / CU: the top-k scan for 65..128 queries (topk.hip)
// Operands go HBM -> LDS directly with global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass), double
// buffered, one barrier per K step.  global_load_lds writes LDS lane-linearly (wave-uniform base + lane*16 B), so
// the bank-conflict swizzles are applied to the per-lane SOURCE address and undone on the fragment reads (the same
// involution on both sides):
//   KMAJ tile [rows][BK]  : 16-B chunk index ^= (row & 7)            (BK = 64, 128-B rows)
//                                            ^= ((row >> 3) & 1) << 1 (BK = 32,  64-B rows)   -> ds_read_b128 frags
//   TMAJ tile [BK][cols]  : 32-B chunk index ^= f(krow), f = (krow & 3) | ((krow >> 3) & 1) << 2
//                                                                     -> ds_read_b64_tr_b16 frags
// Requirements (checked by the host wrapper): K range a multiple of BK; out-of-range M/N rows are clamped to a
// valid row (their products are never stored).
#pragma once
#include "common.h"

DEVINL int tmaj_f(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

template <int BK>
DEVINL int kmaj_swz(int row) {
    return BK == 64 ? (row & 7) : (((row >> 3) & 1) << 1);
}

// issue this thread's global_load_lds for one operand tile; EXT = tile extent along M or N (rows of a KMAJ tile,
// columns of a TMAJ tile); T = threads per workgroup.  Each wave-instruction fills 1 KiB of LDS.
template <bool TMAJ, int EXT, int BK, int T>
DEVINL void glds_stage(const unsigned short* __restrict__ base, long ld, int mn0, int mn_total, int k0, char* lds,
                       int tid, int wave_uniform) {
    constexpr int CHUNKS = EXT * BK / 8;   // 16-B chunks in the tile image
    constexpr int PER = CHUNKS / T;
    static_assert(CHUNKS % T == 0, "tile must split evenly over the workgroup");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = i * T + tid;  // chunk id == LDS position
        long off;
        if (!TMAJ) {
            constexpr int CPR = BK / 8;  // chunks per row
            const int row = c / CPR, slot = c % CPR;
            const int src = slot ^ kmaj_swz<BK>(row);
            const int gm = min(mn0 + row, mn_total - 1);
            off = (long)gm * ld + k0 + src * 8;
        } else {
            constexpr int CPR = EXT / 8;
            const int krow = c / CPR, slot = c % CPR;
            const int src = (((slot >> 1) ^ tmaj_f(krow)) << 1) | (slot & 1);
            const int gm = min(mn0 + src * 8, mn_total - 8);
            off = (long)(k0 + krow) * ld + gm;
        }
        char* dst = lds + (i * T + wave_uniform * 64) * 16;  // wave-uniform LDS base of this instruction
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + off),
                                         (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
    }
}

// fragment: 8 consecutive k (k-step s) of row/col base16 + (lane & 15)
template <bool TMAJ, int EXT, int BK>
DEVINL u32x4_t glds_frag(const char* lds, int base16, int s, int lane) {
    if (!TMAJ) {
        const int rl = base16 + (lane & 15);
        const int kc = s * 4 + (lane >> 4);
        return *reinterpret_cast<const u32x4_t*>(lds + rl * (BK * 2) + ((kc ^ kmaj_swz<BK>(rl)) << 4));
    } else {
        const int t = lane & 15, g = lane >> 4;
        const int krow = s * 32 + 8 * g + (t >> 2);
        const int col = base16 + 4 * (t & 3);  // element column inside the EXT-wide row
        const int c32 = col >> 4, within = (col & 15) * 2;
        const int f = tmaj_f(krow);            // identical for krow and krow + 4
        const s16x4_t lo = lds_read_tr16(lds + krow * (EXT * 2) + ((c32 ^ f) << 5) + within);
        const s16x4_t hi = lds_read_tr16(lds + (krow + 4) * (EXT * 2) + ((c32 ^ f) << 5) + within);
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
        const u32x4_t r = {l2[0], l2[1], h2[0], h2[1]};
        return r;
    }
}

template <int WAVES_M, int WAVES_N, int BK>
struct GldsShape {
    static constexpr int BM = 128 * WAVES_M, BN = 64 * WAVES_N, T = 64 * WAVES_M * WAVES_N;
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
};

// Accumulate the wave's 128x64 sub-tile over k in [kbeg, kend); acc computed swapped (lane owns 4 consecutive N).
template <typename Elem, bool A_TMAJ, bool B_TMAJ, int WAVES_M, int WAVES_N, int BK>
DEVINL void glds_mainloop(const unsigned short* __restrict__ A, long lda, int M, const unsigned short* __restrict__ B,
                          long ldb, int N, int m0, int n0, int kbeg, int kend, char* lds, f32x4_t (&acc)[8][4]) {
    using S = GldsShape<WAVES_M, WAVES_N, BK>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (w / WAVES_N) * 128, wn = (w % WAVES_N) * 64;
    const int nk = (kend - kbeg) / BK;
    if (nk <= 0) return;
    glds_stage<A_TMAJ, S::BM, BK, S::T>(A, lda, m0, M, kbeg, lds, tid, w);
    glds_stage<B_TMAJ, S::BN, BK, S::T>(B, ldb, n0, N, kbeg, lds + S::A_BYTES, tid, w);
    __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) in front of the barrier
    for (int t = 0; t < nk; ++t) {
        const char* cur = lds + (t & 1) * S::STAGE_BYTES;
        char* nxt = lds + ((t + 1) & 1) * S::STAGE_BYTES;
        if (t + 1 < nk) {
            const int k0 = kbeg + (t + 1) * BK;
            glds_stage<A_TMAJ, S::BM, BK, S::T>(A, lda, m0, M, k0, nxt, tid, w);
            glds_stage<B_TMAJ, S::BN, BK, S::T>(B, ldb, n0, N, k0, nxt + S::A_BYTES, tid, w);
        }
        // B fragments of the whole K step up front, A fragments streamed one ahead of the MFMAs that use them:
        // the LDS latency of fragment i+1 hides under the 4 MFMAs of fragment i (pinned with sched_group_barrier).
        constexpr int NS = BK / 32;
        u32x4_t bf[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[s][j] = glds_frag<B_TMAJ, S::BN, BK>(cur + S::A_BYTES, wn + j * 16, s, lane);
        u32x4_t af = glds_frag<A_TMAJ, S::BM, BK>(cur, wm, 0, lane);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u32x4_t an = af;
                if (i + 1 < 8) an = glds_frag<A_TMAJ, S::BM, BK>(cur, wm + (i + 1) * 16, s, lane);
                else if (s + 1 < NS) an = glds_frag<A_TMAJ, S::BM, BK>(cur, wm, s + 1, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Elem::mfma(bf[s][j], af, acc[i][j]);
                af = an;
                if (i + 1 < 8 || s + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, A_TMAJ ? 2 : 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------------------------------
// Hand-pipelined variant of the K step for the 256x256x64 shape: the fragment reads are inline-asm ds_reads that the
// compiler does not track, so they can be issued one fragment AHEAD of the MFMAs that consume them and retired with
// COUNTED s_waitcnt lgkmcnt(N) (LDS returns in order) instead of the lgkmcnt(0) hipcc emits for builtin reads.
// Only LDS-DMA (VMEM) and these asm reads touch LDS inside the loop, so no compiler-generated DS wait interferes.
DEVINL unsigned lds_addr32(const char* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF>
DEVINL u32x4_t asm_ds_read_b128(unsigned addr) {
    u32x4_t r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
    return r;
}
template <int OFF>
DEVINL u32x2_t asm_ds_read_tr16(unsigned addr) {
    u32x2_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
    return r;
}
template <int N>
DEVINL void asm_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N));
    __builtin_amdgcn_sched_barrier(0);
}

// per-lane address state of one operand's fragments for the 256-extent, BK = 64 shape
template <bool TMAJ>
struct FragAddr {
    unsigned a[TMAJ ? 8 : 2];   // KMAJ: one base per k-step s; TMAJ: one base per 16-wide tile (s, +4 rows are immediates)
    DEVINL void init(int base16, int lane, int ntiles) {
        if (!TMAJ) {
            const int rl = base16 + (lane & 15);
#pragma unroll
            for (int s = 0; s < 2; ++s) a[s] = rl * 128 + (((s * 4 + (lane >> 4)) ^ (rl & 7)) << 4);
        } else {
            const int t = lane & 15, g = lane >> 4;
            const int krow = 8 * g + (t >> 2);
            const int f = tmaj_f(krow);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int col = base16 + i * 16 + 4 * (t & 3);
                a[i] = krow * 512 + (((col >> 4) ^ f) << 5) + (col & 15) * 2;
            }
        }
    }
    // fragment of tile I (16 rows/cols), k-step S, relative to stage base address `sb`
    template <int I, int S>
    DEVINL u32x4_t read(unsigned sb) const {
        if (!TMAJ) {
            return asm_ds_read_b128<I * 2048>(sb + a[S]);
        } else {
            const u32x2_t lo = asm_ds_read_tr16<S * 16384>(sb + a[I]);
            const u32x2_t hi = asm_ds_read_tr16<S * 16384 + 2048>(sb + a[I]);
            const u32x4_t r = {lo[0], lo[1], hi[0], hi[1]};
            return r;
        }
    }
};

// fragment IDX (= S*8 + I) of the K step: issue fragment IDX + DIST, wait until fragment IDX has landed (counted:
// the DIST younger fragments stay in flight), then the 4 MFMAs of fragment IDX.
template <typename Elem, bool A_TMAJ, int IDX, int DIST>
DEVINL void kstep_tile(const FragAddr<A_TMAJ>& fa, unsigned sa, const u32x4_t (&bf)[2][4], u32x4_t (&ring)[DIST + 1],
                       f32x4_t (&acc)[8][4]) {
    constexpr int NX = IDX + DIST;
    if (NX < 16) ring[NX % (DIST + 1)] = fa.template read<(NX < 16 ? NX % 8 : 0), (NX < 16 ? NX / 8 : 0)>(sa);
    constexpr int younger = (16 - 1 - IDX) < DIST ? (16 - 1 - IDX) : DIST;   // fragments issued after IDX
    asm_wait_lgkm<younger * (A_TMAJ ? 2 : 1)>();
    constexpr int I = IDX % 8, S = IDX / 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[I][j] = Elem::mfma(bf[S][j], ring[IDX % (DIST + 1)], acc[I][j]);
    __builtin_amdgcn_sched_barrier(0);
}

template <typename Elem, bool A_TMAJ, bool B_TMAJ>
DEVINL void glds_mainloop_asm(const unsigned short* __restrict__ A, long lda, int M, const unsigned short* __restrict__ B,
                              long ldb, int N, int m0, int n0, int kbeg, int kend, char* lds, f32x4_t (&acc)[8][4]) {
    using S = GldsShape<2, 4, 64>;
    constexpr int BK = 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (w / 4) * 128, wn = (w % 4) * 64;
    const int nk = (kend - kbeg) / BK;
    if (nk <= 0) return;
    FragAddr<A_TMAJ> fa;
    FragAddr<B_TMAJ> fb;
    fa.init(wm, lane, 8);
    fb.init(wn, lane, 4);
    const unsigned lbase = lds_addr32(lds);
    glds_stage<A_TMAJ, S::BM, BK, S::T>(A, lda, m0, M, kbeg, lds, tid, w);
    glds_stage<B_TMAJ, S::BN, BK, S::T>(B, ldb, n0, N, kbeg, lds + S::A_BYTES, tid, w);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        char* nxt = lds + ((t + 1) & 1) * S::STAGE_BYTES;
        if (t + 1 < nk) {
            const int k0 = kbeg + (t + 1) * BK;
            glds_stage<A_TMAJ, S::BM, BK, S::T>(A, lda, m0, M, k0, nxt, tid, w);
            glds_stage<B_TMAJ, S::BN, BK, S::T>(B, ldb, n0, N, k0, nxt + S::A_BYTES, tid, w);
        }
        const unsigned sa = lbase + (t & 1) * S::STAGE_BYTES;
        const unsigned sb = sa + S::A_BYTES;
        u32x4_t bf[2][4];
        bf[0][0] = fb.template read<0, 0>(sb); bf[0][1] = fb.template read<1, 0>(sb);
        bf[0][2] = fb.template read<2, 0>(sb); bf[0][3] = fb.template read<3, 0>(sb);
        bf[1][0] = fb.template read<0, 1>(sb); bf[1][1] = fb.template read<1, 1>(sb);
        bf[1][2] = fb.template read<2, 1>(sb); bf[1][3] = fb.template read<3, 1>(sb);
        constexpr int DIST = 1;
        u32x4_t ring[DIST + 1];
        ring[0] = fa.template read<0, 0>(sa);
        if (DIST > 1) ring[1 % (DIST + 1)] = fa.template read<1, 0>(sa);
#define KT(IDX) kstep_tile<Elem, A_TMAJ, IDX, DIST>(fa, sa, bf, ring, acc)
        KT(0); KT(1); KT(2); KT(3); KT(4); KT(5); KT(6); KT(7);
        KT(8); KT(9); KT(10); KT(11); KT(12); KT(13); KT(14); KT(15);
#undef KT
        __syncthreads();
    }
}
