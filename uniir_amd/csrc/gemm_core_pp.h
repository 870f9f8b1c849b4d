// Ping-pong ("8-phase") main loop of the 256x256x64 LDS-DMA GEMM for gfx950.
//
// 8 waves = two groups of four (group g = waves 4g..4g+3, one wave per SIMD each).  Group 1 runs ONE s_barrier behind
// group 0, and every phase is  [LDS fragment reads + one half-tile of LDS-DMA prefetch] s_barrier [16 MFMAs] s_barrier,
// so in every barrier interval one group feeds the four matrix pipes while the other group issues its LDS / DMA
// traffic: the matrix pipe of a SIMD always has exactly one wave in its MFMA segment.
//
// Tile decomposition.  The 256x64 A image of a K step is staged as two half-tiles A0 = rows [0,128), A1 = rows
// [128,256) (B alike along N), each 16 KiB, in the SAME swizzled layouts as gemm_core256.h with extent 128.  A wave
// (wr = w >> 2 in 0..1, wc = w & 3) owns rows {128 h + 64 wr + [0,64)} x cols {128 h' + 32 wc + [0,32)}, h, h' in {0,1}:
// four 64x32 quadrants (h,h'), each 4x2 MFMA tiles x 2 k-substeps = 16 MFMAs = one phase.  Quadrant order per K step
// (A0,B0) (A0,B1) (A1,B1) (A1,B0): 12 / 4 / 8 / 0 fragment reads; every half-tile is read in exactly one phase
// (q = 0,0,1,2 for A0,B0,B1,A1).
//
// LDS ring: 2 K steps x 4 half-tiles x 16 KiB = 128 KiB, slot(t, j) = ((t & 1) * 4 + j), j = 0:A0 1:B0 2:B1 3:A1.
// Staging runs D = 6 half-tiles ahead in the order A0,B0,B1,A1: phase (t,q) stages half-tile 4t+q+6, i.e.
//   q=0 -> (t+1,B1)   q=1 -> (t+1,A1)   q=2 -> (t+2,A0)   q=3 -> (t+2,B0).
// Hazards (MI355X guide: LDS-DMA is ordered for a ds_read only by the issuing waves' counted vmcnt followed by a
// barrier the reader has passed, one barrier more for the staggered group; a slot may be re-staged >= 2 phases after
// its last ds_read):
//   RAW: every load segment ends with s_waitcnt vmcnt(8) = "all but the 4 youngest half-tiles have landed" BEFORE the
//        phase's first barrier; phase P reads half-tiles <= 4t+{1,2,3} which is <= (P-1)+6-4: retired one phase earlier.
//   WAR: the slot staged in phase (t,q) was last read in phase (t-1,1) / (t-1,2) / (t,0) / (t,0): >= 2 phases before.
// The last two K steps stage nothing / less and use exact smaller counts (MODE 1, 2).  Needs nk >= 3 K steps.
#pragma once
#include "gemm_core256.h"

#ifndef PP_EXP
#define PP_EXP 0   // timing experiments only (tools/build_exp.sh): 1 no staging, 2 no fragment reads, 4 no MFMA, 8 no vmcnt
#endif

template <int N>
DEVINL void asm_wait_vm() {
    if (PP_EXP & 9) return;
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N));
    __builtin_amdgcn_sched_barrier(0);
}
DEVINL void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// per-lane LDS byte offsets of the fragment reads inside one 128-extent half-tile image
template <bool TMAJ>
struct HalfFrag {
    unsigned a[TMAJ ? 4 : 2];
    DEVINL void init(int base16, int lane) {
        if (!TMAJ) {
            const int rl = base16 + (lane & 15);
#pragma unroll
            for (int s = 0; s < 2; ++s) a[s] = rl * 128 + (((s * 4 + (lane >> 4)) ^ (rl & 7)) << 4);
        } else {
            const int t = lane & 15, g = lane >> 4;
            const int krow = 8 * g + (t >> 2);
            const int f = tmaj_f(krow);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = base16 + i * 16 + 4 * (t & 3);
                a[i] = krow * 256 + (((col >> 4) ^ f) << 5) + (col & 15) * 2;
            }
        }
    }
    template <int I, int S>
    DEVINL u32x4_t read(unsigned sb) const {
        if (PP_EXP & 2) return u32x4_t{sb, a[0], sb, a[1]};
        if (!TMAJ) {
            return asm_ds_read_b128<I * 2048>(sb + a[S]);
        } else {
            const u32x2_t lo = asm_ds_read_tr16<S * 8192>(sb + a[I]);
            const u32x2_t hi = asm_ds_read_tr16<S * 8192 + 1024>(sb + a[I]);
            const u32x4_t r = {lo[0], lo[1], hi[0], hi[1]};
            return r;
        }
    }
};

// this thread's two source byte offsets of one half-tile, relative to the block's operand base (row / column mn_base
// of the block tile, K offset kbeg); same source-side swizzles as glds_stage.  The loads are MUBUF LDS-DMA
// (buffer_load_dwordx4 ... offen lds): one SGPR descriptor per operand, a 32-bit per-lane offset, and the K advance as
// the scalar soffset -- no 64-bit per-lane address arithmetic in the loop and half the address traffic of global_load.
template <bool TMAJ>
DEVINL void pp_src(long ld, int mn_base, int mn0, int mn_total, int tid, unsigned (&vo)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = i * 512 + tid;
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

template <int AUX = 0>      // cache policy of this operand's stream: 0 default, 2 = nt (read once: the top-k scan's pool rows)
DEVINL void pp_stage(__amdgpu_buffer_rsrc_t rs, const unsigned (&vo)[2], unsigned koff_bytes, char* slot, int w) {
    if (PP_EXP & 1) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        char* dst = slot + (i * 512 + w * 64) * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3)))*)dst, 16, vo[i], koff_bytes, 0, AUX);
    }
}

template <typename Elem, int H, int HP>
DEVINL void pp_mfma16(const u32x4_t (&af)[4][2], const u32x4_t (&bf)[2][2], f32x4_t (&acc)[8][4]) {
    if (PP_EXP & 4) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[H * 4 + i][HP * 2 + j] = Elem::mfma(bf[j][s], af[i][s], acc[H * 4 + i][HP * 2 + j]);
    __builtin_amdgcn_s_setprio(0);
}
// Row sums of A^T next to the GEMM (ROWSUM kernels): rs += the 8 k-values this lane holds of A row 16 I + (lane & 15), for both
// k-substeps of the fragments in `af` -- for a weight-gradient GEMM dW = dy^T x that is the bias gradient, taken from the dy
// fragments the matrix pipe is fed with anyway (v_dot2c_f32_bf16 against ones) instead of a second pass over dy.
//  * The work is spread evenly: a K step is summed by the workgroup of ONE column panel (t % tiles_n), and there wave (wr, wc)
//    takes fragment I = wc -- 4 dot products per 64 MFMAs on average, the same for every wave (summing in the first column
//    panel's waves wc = 0, 1 only made those workgroups 17 % slower, and the GEMM with them).
//  * It runs in the wave's LOAD segment of the following phase (the A fragments stay in registers for two phases), not between
//    the MFMAs: a second variant of the MFMA sequence makes the compiler move all 128 accumulators between two register sets
//    (+13 % on the GEMM).
//  * An ext-vector element must be copied to a scalar before __builtin_bit_cast: bit_cast(T, v[c]) reads element 0 with this hipcc.
template <int I>
DEVINL void pp_rowsum_frag(const u32x4_t (&af)[4][2], float& rs) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    const bf2_t ones = __builtin_bit_cast(bf2_t, 0x3F803F80u);
    float r1 = 0.f;                              // two chains: the dot products are back-to-back dependent otherwise
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned e0 = af[I][0][c], e1 = af[I][1][c];
        rs = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, e0), ones, rs, false);
        r1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, e1), ones, r1, false);
    }
    rs += r1;
}
DEVINL void pp_rowsum_wc(int wc, const u32x4_t (&af)[4][2], float& rs) {
    switch (wc) {
        case 0: pp_rowsum_frag<0>(af, rs); break;
        case 1: pp_rowsum_frag<1>(af, rs); break;
        case 2: pp_rowsum_frag<2>(af, rs); break;
        default: pp_rowsum_frag<3>(af, rs); break;
    }
}

template <typename Elem, bool A_TMAJ, bool B_TMAJ>
struct PPState {
    HalfFrag<A_TMAJ> fa;
    HalfFrag<B_TMAJ> fb;
    __amdgpu_buffer_rsrc_t rA, rB;    // operand bases of this block tile (at k = kbeg)
    unsigned gA[2][2], gB[2][2];      // [half][load] per-lane byte offsets
    unsigned kstepA, kstepB;          // byte offset of one K step
    unsigned lbase;
    char* lds;
    int w;
    int rs_nt, rs_tiles;              // ROWSUM kernels: this workgroup sums the K steps t with t % rs_tiles == rs_nt (-1: none)
};

// one K step (4 phases).  MODE 0: steady state; 1: second to last K step (stages q=0,1 only); 2: last (stages nothing)
// HALFN (the top-k scan with <= 128 queries): the column half B1 holds no real columns -- its fragment reads and the two MFMA
// phases that use it are dropped (half the matrix work); the staging schedule, and with it every hazard argument above, is unchanged
template <typename Elem, bool A_TMAJ, bool B_TMAJ, int MODE, bool ROWSUM = false, bool HALFN = false, int AUXA = 0>
DEVINL void pp_kstep(const PPState<Elem, A_TMAJ, B_TMAJ>& st, int t, f32x4_t (&acc)[8][4], float (&rs)[2], int& rs_cd) {
    // this workgroup's turn every rs_tiles-th K step (a countdown: t % rs_tiles with a run-time divisor costs ~100 cycles per K step)
    bool rs_turn = false;
    if (ROWSUM && st.rs_nt >= 0) {
        rs_turn = rs_cd == 0;
        rs_cd = rs_turn ? st.rs_tiles - 1 : rs_cd - 1;
    }
    const unsigned sb = st.lbase + (t & 1) * 65536;            // this K step's four slots
    char* const cur = st.lds + (t & 1) * 65536;
    char* const oth = st.lds + ((t + 1) & 1) * 65536;
    const unsigned kA1 = (unsigned)(t + 1) * st.kstepA, kB1 = (unsigned)(t + 1) * st.kstepB;
    const unsigned kA2 = (unsigned)(t + 2) * st.kstepA, kB2 = (unsigned)(t + 2) * st.kstepB;
    u32x4_t af[4][2], b0[2][2], b1[2][2];
    // ---- phase 0: quadrant (A0,B0); reads B0 then A0; stages (t+1, B1)
    b0[0][0] = st.fb.template read<0, 0>(sb + 16384); b0[1][0] = st.fb.template read<1, 0>(sb + 16384);
    b0[0][1] = st.fb.template read<0, 1>(sb + 16384); b0[1][1] = st.fb.template read<1, 1>(sb + 16384);
    __builtin_amdgcn_sched_barrier(0);
    af[0][0] = st.fa.template read<0, 0>(sb); af[1][0] = st.fa.template read<1, 0>(sb);
    af[2][0] = st.fa.template read<2, 0>(sb); af[3][0] = st.fa.template read<3, 0>(sb);
    af[0][1] = st.fa.template read<0, 1>(sb); af[1][1] = st.fa.template read<1, 1>(sb);
    af[2][1] = st.fa.template read<2, 1>(sb); af[3][1] = st.fa.template read<3, 1>(sb);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE <= 1) pp_stage(st.rB, st.gB[1], kB1, oth + 2 * 16384, st.w);
    __builtin_amdgcn_sched_barrier(0);
    asm_wait_vm<MODE <= 1 ? 8 : 2>();
    pp_barrier();
    asm_wait_lgkm<0>();
    pp_mfma16<Elem, 0, 0>(af, b0, acc);
    pp_barrier();
    // ---- phase 1: quadrant (A0,B1); reads B1; stages (t+1, A1)
    if (!HALFN) {
        b1[0][0] = st.fb.template read<0, 0>(sb + 2 * 16384); b1[1][0] = st.fb.template read<1, 0>(sb + 2 * 16384);
        b1[0][1] = st.fb.template read<0, 1>(sb + 2 * 16384); b1[1][1] = st.fb.template read<1, 1>(sb + 2 * 16384);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE <= 1) pp_stage<AUXA>(st.rA, st.gA[1], kA1, oth + 3 * 16384, st.w);
    __builtin_amdgcn_sched_barrier(0);
    if (ROWSUM && rs_turn) pp_rowsum_wc(st.w & 3, af, rs[0]);            // A0 rows (still in registers)
    __builtin_amdgcn_sched_barrier(0);
    asm_wait_vm<MODE <= 1 ? 8 : 0>();
    pp_barrier();
    asm_wait_lgkm<0>();
    if (!HALFN) pp_mfma16<Elem, 0, 1>(af, b1, acc);
    pp_barrier();
    // ---- phase 2: quadrant (A1,B1); reads A1; stages (t+2, A0)
    af[0][0] = st.fa.template read<0, 0>(sb + 3 * 16384); af[1][0] = st.fa.template read<1, 0>(sb + 3 * 16384);
    af[2][0] = st.fa.template read<2, 0>(sb + 3 * 16384); af[3][0] = st.fa.template read<3, 0>(sb + 3 * 16384);
    af[0][1] = st.fa.template read<0, 1>(sb + 3 * 16384); af[1][1] = st.fa.template read<1, 1>(sb + 3 * 16384);
    af[2][1] = st.fa.template read<2, 1>(sb + 3 * 16384); af[3][1] = st.fa.template read<3, 1>(sb + 3 * 16384);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 0) pp_stage<AUXA>(st.rA, st.gA[0], kA2, cur + 0 * 16384, st.w);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 0) asm_wait_vm<8>();
    pp_barrier();
    asm_wait_lgkm<0>();
    if (!HALFN) pp_mfma16<Elem, 1, 1>(af, b1, acc);
    pp_barrier();
    // ---- phase 3: quadrant (A1,B0), B0 still in registers; stages (t+2, B0)
    if (MODE == 0) pp_stage(st.rB, st.gB[0], kB2, cur + 1 * 16384, st.w);
    __builtin_amdgcn_sched_barrier(0);
    if (ROWSUM && rs_turn) pp_rowsum_wc(st.w & 3, af, rs[1]);            // A1 rows (still in registers)
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 0) asm_wait_vm<8>();
    if (MODE == 1) asm_wait_vm<4>();
    pp_barrier();
    pp_mfma16<Elem, 1, 0>(af, b0, acc);
    pp_barrier();
}

// accumulators: acc[4h + i][2h' + j] = MFMA tile at rows 128h + 64wr + 16i, cols 128h' + 32wc + 16j (swapped operands:
// a lane owns 4 consecutive N, like gemm_core256.h)
//
// pp_setup: per-tile state (source offsets + descriptors); pp_prologue: stage half-tiles 0..5 (the ring must not be
// read any more: call after pp_main's final barrier); pp_main: everything from the first wait to the re-aligning barrier.
template <typename Elem, bool A_TMAJ, bool B_TMAJ>
DEVINL void pp_setup(PPState<Elem, A_TMAJ, B_TMAJ>& st, const unsigned short* __restrict__ A, long lda, int M,
                     const unsigned short* __restrict__ B, long ldb, int N, int m0, int n0, int kbeg, char* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    st.fa.init((w >> 2) * 64, lane);
    st.fb.init((w & 3) * 32, lane);
    pp_src<A_TMAJ>(lda, m0, m0, M, tid, st.gA[0]);
    pp_src<A_TMAJ>(lda, m0, m0 + 128, M, tid, st.gA[1]);
    pp_src<B_TMAJ>(ldb, n0, n0, N, tid, st.gB[0]);
    pp_src<B_TMAJ>(ldb, n0, n0 + 128, N, tid, st.gB[1]);
    const unsigned short* baseA = A + (A_TMAJ ? (long)kbeg * lda + m0 : (long)m0 * lda + kbeg);
    const unsigned short* baseB = B + (B_TMAJ ? (long)kbeg * ldb + n0 : (long)n0 * ldb + kbeg);
    st.rA = __builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, -1, 0x00020000);
    st.rB = __builtin_amdgcn_make_buffer_rsrc((void*)baseB, 0, -1, 0x00020000);
    st.kstepA = (unsigned)(A_TMAJ ? 128 * lda : 128);
    st.kstepB = (unsigned)(B_TMAJ ? 128 * ldb : 128);
    st.lbase = lds_addr32(lds);
    st.lds = lds;
    st.w = w;
}

template <typename Elem, bool A_TMAJ, bool B_TMAJ, int AUXA = 0>
DEVINL void pp_prologue(const PPState<Elem, A_TMAJ, B_TMAJ>& st) {
    // half-tiles 0..5 = (0,A0) (0,B0) (0,B1) (0,A1) (1,A0) (1,B0)
    char* lds = st.lds;
    pp_stage<AUXA>(st.rA, st.gA[0], 0, lds + 0 * 16384, st.w);
    pp_stage(st.rB, st.gB[0], 0, lds + 1 * 16384, st.w);
    pp_stage(st.rB, st.gB[1], 0, lds + 2 * 16384, st.w);
    pp_stage<AUXA>(st.rA, st.gA[1], 0, lds + 3 * 16384, st.w);
    pp_stage<AUXA>(st.rA, st.gA[0], st.kstepA, lds + 65536 + 0 * 16384, st.w);
    pp_stage(st.rB, st.gB[0], st.kstepB, lds + 65536 + 1 * 16384, st.w);
    __builtin_amdgcn_sched_barrier(0);
}

template <typename Elem, bool A_TMAJ, bool B_TMAJ, bool ROWSUM = false, bool HALFN = false, int AUXA = 0>
DEVINL void pp_main(const PPState<Elem, A_TMAJ, B_TMAJ>& st, int nk, f32x4_t (&acc)[8][4], float (&rs)[2]) {
    const int wr = st.w >> 2;
    asm_wait_vm<8>();
    pp_barrier();
    if (wr == 1) pp_barrier();          // group 1 runs one barrier behind from here on
    int rs_cd = st.rs_nt;                // first turn at t = rs_nt
    for (int t = 0; t < nk - 2; ++t) pp_kstep<Elem, A_TMAJ, B_TMAJ, 0, ROWSUM, HALFN, AUXA>(st, t, acc, rs, rs_cd);
    pp_kstep<Elem, A_TMAJ, B_TMAJ, 1, ROWSUM, HALFN, AUXA>(st, nk - 2, acc, rs, rs_cd);
    pp_kstep<Elem, A_TMAJ, B_TMAJ, 2, ROWSUM, HALFN, AUXA>(st, nk - 1, acc, rs, rs_cd);
    if (wr == 0) pp_barrier();          // re-align the groups
}

// a_rowsum (ROWSUM kernels, transposed A only): += sum over this block's K range of A^T's rows m0 .. m0 + 255; nt / tiles_n: the
// block's column panel and their number.
template <typename Elem, bool A_TMAJ, bool B_TMAJ, bool ROWSUM = false, bool HALFN = false, int AUXA = 0>
DEVINL void glds_mainloop_pp(const unsigned short* __restrict__ A, long lda, int M, const unsigned short* __restrict__ B,
                             long ldb, int N, int m0, int n0, int kbeg, int kend, char* lds, f32x4_t (&acc)[8][4],
                             float* a_rowsum = nullptr, int nt = 0, int tiles_n = 1, bool rs_store = false) {
    PPState<Elem, A_TMAJ, B_TMAJ> st;
    pp_setup(st, A, lda, M, B, ldb, N, m0, n0, kbeg, lds);
    st.rs_nt = (ROWSUM && a_rowsum) ? nt : -1;
    st.rs_tiles = tiles_n;
    float rs[2] = {0.f, 0.f};
    pp_prologue<Elem, A_TMAJ, B_TMAJ, AUXA>(st);
    pp_main<Elem, A_TMAJ, B_TMAJ, ROWSUM, HALFN, AUXA>(st, (kend - kbeg) / 64, acc, rs);
    if (ROWSUM && st.rs_nt >= 0) {
        // rs[h]: lane (row 128 h + 64 wr + 16 wc + (lane & 15), k group lane >> 4) -> add the four k groups, one atomic per row
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v = rs[h];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int m = m0 + h * 128 + (st.w >> 2) * 64 + 16 * (st.w & 3) + (lane & 15);
            if (lane < 16 && m < M) {
                if (rs_store) a_rowsum[m] = v;          // a_rowsum is this block's partial row [M]; gemm_impl reduces in block order
                else unsafeAtomicAdd(a_rowsum + m, v);
            }
        }
    }
}
