// 256x128x64 LDS-DMA GEMM loop for gfx950, TWO workgroups per CU (round 3).
//
// Why: the 256x256 ping-pong kernel (gemm_core_pp.h) runs one 8-wave workgroup per CU.  Its two wave groups hide each other's
// load segments, but nothing runs beside its pipeline fill (6 half-tiles before the first MFMA) and its epilogue (12 % of a
// K = 1024 tile): the matrix pipes idle there (MFMA issue 76.6 % of a K = 1024 tile, 82 % in steady state,
// profiles/r01_gemm_timeline_v7.txt).  Here a workgroup is ONE group of four waves (one per SIMD) on a 256x128 tile with the same
// 128 accumulator registers per wave, 80 KiB of LDS -- so two workgroups share a CU and each SIMD holds one wave of either: while
// one workgroup fills its pipeline, converts and stores its tile, or sits in a load segment, the other one owns the matrix pipe.
// No barrier couples the two; within a workgroup there is ONE s_barrier per phase.
//
// Tile decomposition.  Per K step three 16-KiB units: A0 = rows [0,128), A1 = rows [128,256) of the A tile, B = all 128 columns
// (the same swizzled 128-extent images as gemm_core_pp.h: K-contiguous operands [128][64] with 16-B chunk ^= row & 7,
// M/N-contiguous operands [64][128] with the tmaj_f swizzle).  Wave (wr = w >> 1, wc = w & 1) owns rows {128 h + 64 wr + [0,64)} x
// cols {64 h' + 32 wc + [0,32)}, h, h' in {0,1}: four 64x32 quadrants of 4x2 MFMA tiles x 2 k-substeps = 16 MFMAs = one phase.
// Quadrant order (A0,B0') (A0,B1') (A1,B1') (A1,B0'): fragment reads 12 / 4 / 8 / 0 per phase like the 8-wave kernel.
//
// LDS ring: 5 slots of 16 KiB, unit u = 3 t + j (j = 0: A0, 1: B, 2: A1) lives in slot u % 5.  Phase (t, q) stages
//   q = 0 -> B(t+1)    q = 2 -> A1(t+1)    q = 3 -> A0(t+2)            (4 LDS-DMA instructions per thread each)
// after the prologue has staged A0(0), B(0), A1(0), A0(1).  Every load segment ends with a counted vmcnt (12 / 8 / 12 / 8 in steady
// state) in front of the phase's barrier.  Hazards, checked for 3..69 K steps by tools/r3/check_pp2_schedule.py:
//   RAW: a unit read in phase p is covered by every wave's vmcnt of a phase < p (the waits precede the barriers);
//   WAR: the slot staged in phase p was last read in a phase <= p - 2 (unit u overwrites u - 5: A1(t+1) -> A0(t), A0(t+2) -> B(t),
//        B(t+1) -> A1(t-1)); the readers' lgkmcnt(0) of that phase precedes their arrival at barrier p - 1.
// In flight per workgroup: 2 - 3 units = 32 - 48 KiB (64 - 96 KiB per CU), for 3 - 4 phases of its own (6 - 8 phase times of the
// shared matrix pipe).  Needs >= 3 K steps.
#pragma once
#include "gemm_core_pp.h"

template <typename Elem, bool A_TMAJ, bool B_TMAJ>
struct PP2State {
    HalfFrag<A_TMAJ> fa;
    HalfFrag<B_TMAJ> fb0, fb1;        // columns 64 h' + 32 wc of the 128-column B image
    __amdgpu_buffer_rsrc_t rA, rB;    // operand bases of this block tile (at k = kbeg)
    unsigned gA[2][4], gB[4];         // per-lane source byte offsets of a unit's four DMA instructions
    unsigned kstepA, kstepB;          // byte offset of one K step
    unsigned lbase;
    char* lds;
    int w;
};

// this thread's four source byte offsets of one 16-KiB unit (128 rows / columns starting at mn0), relative to the block's operand
// base; same source-side swizzles as pp_src, 256 threads
template <bool TMAJ>
DEVINL void pp2_src(long ld, int mn_base, int mn0, int mn_total, int tid, unsigned (&vo)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 256 + tid;
        long off;
        if (!TMAJ) {
            const int row = c >> 3, slot = c & 7;
            const int src = slot ^ (row & 7);
            const int gm = min(mn0 + row, mn_total - 1) - mn_base;
            off = (long)gm * ld + src * 8;
        } else {
            const int krow = c >> 4, slot = c & 15;
            const int src = (((slot >> 1) ^ tmaj_f(krow)) << 1) | (slot & 1);
            const int gm = min(mn0 + src * 8, mn_total - 8) - mn_base;
            off = (long)krow * ld + gm;
        }
        vo[i] = (unsigned)(off * 2);
    }
}

DEVINL void pp2_stage(__amdgpu_buffer_rsrc_t rs, const unsigned (&vo)[4], unsigned koff_bytes, char* slot, int w) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        char* dst = slot + (i * 256 + w * 64) * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3)))*)dst, 16, vo[i], koff_bytes, 0, 0);
    }
}

// ring slots of the current K step's units and where this K step stages (all scalar)
struct PP2Slots {
    int a0, b, a1;
    DEVINL void advance() {      // + 3 mod 5
        a0 = a0 >= 2 ? a0 - 2 : a0 + 3;
        b = b >= 2 ? b - 2 : b + 3;
        a1 = a1 >= 2 ? a1 - 2 : a1 + 3;
    }
};

// one K step (4 phases).  MODE 0: steady state; 1: second to last K step (no A0(t+2)); 2: last (stages nothing)
template <typename Elem, bool A_TMAJ, bool B_TMAJ, int MODE>
DEVINL void pp2_kstep(const PP2State<Elem, A_TMAJ, B_TMAJ>& st, int t, const PP2Slots& sl, f32x4_t (&acc)[8][4]) {
    const unsigned sA0 = st.lbase + sl.a0 * 16384, sB = st.lbase + sl.b * 16384, sA1 = st.lbase + sl.a1 * 16384;
    // staging targets: B(t+1) -> the slot of A1(t-1) = (b + 3) % 5; A1(t+1) -> A0(t)'s slot; A0(t+2) -> B(t)'s slot
    const int sb_next = sl.b >= 2 ? sl.b - 2 : sl.b + 3;
    const unsigned kA1 = (unsigned)(t + 1) * st.kstepA, kB1 = (unsigned)(t + 1) * st.kstepB, kA2 = (unsigned)(t + 2) * st.kstepA;
    u32x4_t af[4][2], b0[2][2], b1[2][2];
    // ---- phase 0: quadrant (A0,B0'); reads B0' then A0; stages B(t+1)
    b0[0][0] = st.fb0.template read<0, 0>(sB); b0[1][0] = st.fb0.template read<1, 0>(sB);
    b0[0][1] = st.fb0.template read<0, 1>(sB); b0[1][1] = st.fb0.template read<1, 1>(sB);
    __builtin_amdgcn_sched_barrier(0);
    af[0][0] = st.fa.template read<0, 0>(sA0); af[1][0] = st.fa.template read<1, 0>(sA0);
    af[2][0] = st.fa.template read<2, 0>(sA0); af[3][0] = st.fa.template read<3, 0>(sA0);
    af[0][1] = st.fa.template read<0, 1>(sA0); af[1][1] = st.fa.template read<1, 1>(sA0);
    af[2][1] = st.fa.template read<2, 1>(sA0); af[3][1] = st.fa.template read<3, 1>(sA0);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE <= 1) pp2_stage(st.rB, st.gB, kB1, st.lds + sb_next * 16384, st.w);
    __builtin_amdgcn_sched_barrier(0);
    asm_wait_vm<MODE <= 1 ? 12 : 4>();
    pp_barrier();
    asm_wait_lgkm<0>();
    pp_mfma16<Elem, 0, 0>(af, b0, acc);
    // ---- phase 1: quadrant (A0,B1'); reads B1'; stages nothing
    b1[0][0] = st.fb1.template read<0, 0>(sB); b1[1][0] = st.fb1.template read<1, 0>(sB);
    b1[0][1] = st.fb1.template read<0, 1>(sB); b1[1][1] = st.fb1.template read<1, 1>(sB);
    __builtin_amdgcn_sched_barrier(0);
    asm_wait_vm<MODE <= 1 ? 8 : 0>();
    pp_barrier();
    asm_wait_lgkm<0>();
    pp_mfma16<Elem, 0, 1>(af, b1, acc);
    // ---- phase 2: quadrant (A1,B1'); reads A1; stages A1(t+1) into A0(t)'s slot
    af[0][0] = st.fa.template read<0, 0>(sA1); af[1][0] = st.fa.template read<1, 0>(sA1);
    af[2][0] = st.fa.template read<2, 0>(sA1); af[3][0] = st.fa.template read<3, 0>(sA1);
    af[0][1] = st.fa.template read<0, 1>(sA1); af[1][1] = st.fa.template read<1, 1>(sA1);
    af[2][1] = st.fa.template read<2, 1>(sA1); af[3][1] = st.fa.template read<3, 1>(sA1);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE <= 1) pp2_stage(st.rA, st.gA[1], kA1, st.lds + sl.a0 * 16384, st.w);
    __builtin_amdgcn_sched_barrier(0);
    asm_wait_vm<MODE <= 1 ? 12 : 0>();
    pp_barrier();
    asm_wait_lgkm<0>();
    pp_mfma16<Elem, 1, 1>(af, b1, acc);
    // ---- phase 3: quadrant (A1,B0'), B0' still in registers; stages A0(t+2) into B(t)'s slot
    if (MODE == 0) pp2_stage(st.rA, st.gA[0], kA2, st.lds + sl.b * 16384, st.w);
    __builtin_amdgcn_sched_barrier(0);
    asm_wait_vm<MODE == 0 ? 8 : (MODE == 1 ? 4 : 0)>();
    pp_barrier();
    pp_mfma16<Elem, 1, 0>(af, b0, acc);
}

// accumulators: acc[4h + i][2h' + j] = MFMA tile at rows 128h + 64wr + 16i, cols 64h' + 32wc + 16j (swapped operands: a lane owns 4
// consecutive N)
template <typename Elem, bool A_TMAJ, bool B_TMAJ>
DEVINL void glds_mainloop_pp2(const unsigned short* __restrict__ A, long lda, int M, const unsigned short* __restrict__ B, long ldb,
                              int N, int m0, int n0, int kbeg, int kend, char* lds, f32x4_t (&acc)[8][4]) {
    PP2State<Elem, A_TMAJ, B_TMAJ> st;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    st.fa.init((w >> 1) * 64, lane);
    st.fb0.init((w & 1) * 32, lane);
    st.fb1.init(64 + (w & 1) * 32, lane);
    pp2_src<A_TMAJ>(lda, m0, m0, M, tid, st.gA[0]);
    pp2_src<A_TMAJ>(lda, m0, m0 + 128, M, tid, st.gA[1]);
    pp2_src<B_TMAJ>(ldb, n0, n0, N, tid, st.gB);
    const unsigned short* baseA = A + (A_TMAJ ? (long)kbeg * lda + m0 : (long)m0 * lda + kbeg);
    const unsigned short* baseB = B + (B_TMAJ ? (long)kbeg * ldb + n0 : (long)n0 * ldb + kbeg);
    st.rA = __builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, -1, 0x00020000);
    st.rB = __builtin_amdgcn_make_buffer_rsrc((void*)baseB, 0, -1, 0x00020000);
    st.kstepA = (unsigned)(A_TMAJ ? 128 * lda : 128);
    st.kstepB = (unsigned)(B_TMAJ ? 128 * ldb : 128);
    st.lbase = lds_addr32(lds);
    st.lds = lds;
    st.w = w;
    const int nk = (kend - kbeg) / 64;
    // prologue: units 0..3 = A0(0) B(0) A1(0) A0(1) into slots 0..3
    pp2_stage(st.rA, st.gA[0], 0, lds + 0 * 16384, w);
    pp2_stage(st.rB, st.gB, 0, lds + 1 * 16384, w);
    pp2_stage(st.rA, st.gA[1], 0, lds + 2 * 16384, w);
    pp2_stage(st.rA, st.gA[0], st.kstepA, lds + 3 * 16384, w);
    __builtin_amdgcn_sched_barrier(0);
    asm_wait_vm<8>();
    pp_barrier();
    PP2Slots sl = {0, 1, 2};
    for (int t = 0; t < nk - 2; ++t) {
        pp2_kstep<Elem, A_TMAJ, B_TMAJ, 0>(st, t, sl, acc);
        sl.advance();
    }
    pp2_kstep<Elem, A_TMAJ, B_TMAJ, 1>(st, nk - 2, sl, acc);
    sl.advance();
    pp2_kstep<Elem, A_TMAJ, B_TMAJ, 2>(st, nk - 1, sl, acc);
    pp_barrier();          // every wave is done reading the ring: the epilogue may reuse it
}
