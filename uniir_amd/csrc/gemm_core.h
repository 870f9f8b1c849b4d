// MFMA GEMM main loop for 16-bit operands on gfx950 (wave64, v_mfma_f32_16x16x32_{bf16,f16}).
//
// Block tile 128x128x64, 256 threads = 4 waves (2x2), each wave owns 64x64 = 4x4 MFMA tiles.
// Operands are staged global -> registers -> LDS (double buffered, one barrier per K step):
//   * K-contiguous operand  ("KMAJ"): LDS image [128 rows][64 k] bf16, 128-B rows, the 16-B chunk
//     index XOR-swizzled with (row & 7) so the ds_read_b128 fragment reads are conflict free.
//   * M/N-contiguous operand ("TMAJ", i.e. stored [K][M] or [K][N]): LDS image [64 k][128 mn] with a
//     288-B row pitch; fragments come from ds_read_b64_tr_b16 (hardware transposing read), two reads
//     per 8-element fragment.
// The accumulator is computed "swapped" (mfma(Bfrag, Afrag)), so a lane owns 4 consecutive N
// columns of one M row: acc[i][j][r] = C[m0 + wm*64 + i*16 + (lane&15)][n0 + wn*64 + j*16 + 4*(lane>>4) + r].
#pragma once
#include "common.h"

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_TPITCH 288                 // bytes per k-row of a TMAJ tile (256 + 32 pad)
#define GEMM_OPBYTES (GEMM_BK * GEMM_TPITCH)   // 18432 >= 128*128 (KMAJ image)
#define GEMM_LDS_BYTES (4 * GEMM_OPBYTES)      // 2 stages x 2 operands

struct ElemBF16 {
    static constexpr bool F16 = false;
    static DEVINL f32x4_t mfma(u32x4_t a, u32x4_t b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                       __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
struct ElemF16 {
    static constexpr bool F16 = true;       // 16-bit OUTPUTS of the epilogues are fp16 too (EPI_BF16 / BIAS_ACT / ACT_ONLY)
    static DEVINL f32x4_t mfma(u32x4_t a, u32x4_t b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                      __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

// One operand's global->register->LDS staging state (4 x 16 B per thread per tile).
template <bool TMAJ>
struct OperandStage {
    u32x4_t r[4];
    unsigned okmask;
    // base: element pointer to operand origin; ld: leading dimension in elements.
    // mn0: first row (KMAJ) / column (TMAJ) of this block's panel; mn_total: operand extent in that dim.
    DEVINL void load(const unsigned short* __restrict__ base, long ld, int mn0, int mn_total, int k0,
                     int kend, int tid) {
        // branch-free: always load from a clamped (valid) address, then select zero for out-of-range chunks
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            bool ok;
            long off;
            if (!TMAJ) {
                const int row = c >> 3, kc = c & 7;
                const int gm = mn0 + row, gk = k0 + kc * 8;
                ok = (gm < mn_total) && (gk < kend);
                off = (long)min(gm, mn_total - 1) * ld + min(gk, kend - 8);
            } else {
                const int krow = c >> 4, mc = c & 15;
                const int gk = k0 + krow, gm = mn0 + mc * 8;
                ok = (gk < kend) && (gm < mn_total);
                off = (long)min(gk, kend - 1) * ld + min(gm, mn_total - 8);
            }
            // the zero-select is deferred to store() so the wait for the load lands after the MFMAs
            r[i] = *reinterpret_cast<const u32x4_t*>(base + off);
            okmask = (i == 0 ? 0u : okmask) | ((ok ? 1u : 0u) << i);
        }
    }
    DEVINL void store(char* lds, int tid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            if (!((okmask >> i) & 1u)) r[i] = u32x4_t{0u, 0u, 0u, 0u};
            int off;
            if (!TMAJ) {
                const int row = c >> 3, kc = c & 7;
                off = row * 128 + ((kc ^ (row & 7)) << 4);
            } else {
                const int krow = c >> 4, mc = c & 15;
                off = krow * GEMM_TPITCH + mc * 16;
            }
            *reinterpret_cast<u32x4_t*>(lds + off) = r[i];
        }
    }
};

// Fragment (8 consecutive k of one row/col, for k-step s in {0,1}) of 16-row tile starting at local
// row/col `base16` inside the 128-wide panel.
template <bool TMAJ>
DEVINL u32x4_t read_frag(const char* lds, int base16, int s, int lane) {
    if (!TMAJ) {
        const int rl = base16 + (lane & 15);
        const int kc = s * 4 + (lane >> 4);
        return *reinterpret_cast<const u32x4_t*>(lds + rl * 128 + ((kc ^ (rl & 7)) << 4));
    } else {
        const int t = lane & 15, g = lane >> 4;
        const int krow = s * 32 + 8 * g + (t >> 2);
        const int col = base16 + 4 * (t & 3);
        const char* p = lds + krow * GEMM_TPITCH + col * 2;
        s16x4_t lo = lds_read_tr16(p);
        s16x4_t hi = lds_read_tr16(p + 4 * GEMM_TPITCH);
        u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
        u32x4_t r = {l2[0], l2[1], h2[0], h2[1]};
        return r;
    }
}

// Accumulate C[128x128] tile over k in [kbeg, kend). acc must be zero-initialised by the caller.
template <typename Elem, bool A_TMAJ, bool B_TMAJ>
DEVINL void gemm_mainloop(const unsigned short* __restrict__ A, long lda, int M,
                          const unsigned short* __restrict__ B, long ldb, int N, int m0, int n0,
                          int kbeg, int kend, char* lds, f32x4_t (&acc)[4][4]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    OperandStage<A_TMAJ> sa;
    OperandStage<B_TMAJ> sb;
    const int nk = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;
    if (nk <= 0) return;
    sa.load(A, lda, m0, M, kbeg, kend, tid);
    sb.load(B, ldb, n0, N, kbeg, kend, tid);
    sa.store(lds, tid);
    sb.store(lds + GEMM_OPBYTES, tid);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        char* cur = lds + (t & 1) * 2 * GEMM_OPBYTES;
        char* nxt = lds + ((t + 1) & 1) * 2 * GEMM_OPBYTES;
        const bool more = (t + 1 < nk);
        if (more) {
            const int k0 = kbeg + (t + 1) * GEMM_BK;
            sa.load(A, lda, m0, M, k0, kend, tid);
            sb.load(B, ldb, n0, N, k0, kend, tid);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4_t af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = read_frag<A_TMAJ>(cur, wm + i * 16, s, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = read_frag<B_TMAJ>(cur + GEMM_OPBYTES, wn + j * 16, s, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Elem::mfma(bf[j], af[i], acc[i][j]);
        }
        if (more) {
            sa.store(nxt, tid);
            sb.store(nxt + GEMM_OPBYTES, tid);
        }
        __syncthreads();
    }
}
