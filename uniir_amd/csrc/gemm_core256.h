// 256x256x64 MFMA GEMM main loop for gfx950: 512 threads = 8 waves (2 along M x 4 along N), each wave owns a
// 128x64 sub-tile (8x4 MFMA 16x16x32 tiles, 128 accumulator registers).  Operands go HBM -> LDS directly with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass), double buffered (2 x 64 KiB), one barrier per K step.
//
// global_load_lds writes LDS lane-linearly (wave-uniform base + lane*16 B), so the bank-conflict swizzles are
// applied to the per-lane SOURCE address and undone on the fragment reads (the same involution on both sides):
//   KMAJ tile [256 rows][64 k]   (128-B rows):  16-B chunk index  ^= (row & 7)           -> ds_read_b128 frags
//   TMAJ tile [64 k][256 mn]     (512-B rows):  32-B chunk index  ^= f(krow), f = (krow&3)|((krow>>3)&1)<<2
//                                                                                        -> ds_read_b64_tr_b16 frags
// Requirements (checked by the host wrapper): K range a multiple of 64; out-of-range M/N rows are clamped to a
// valid row (their products are never stored).
#pragma once
#include "common.h"

#define G256_BM 256
#define G256_BN 256
#define G256_BK 64
#define G256_OPBYTES (256 * 64 * 2)            // 32 KiB per operand per stage
#define G256_LDS_BYTES (4 * G256_OPBYTES)      // 128 KiB

DEVINL int tmaj_f(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// issue the 4 global_load_lds of this thread for one operand tile (each wave-instruction fills 1 KiB of LDS)
template <bool TMAJ>
DEVINL void g256_stage(const unsigned short* __restrict__ base, long ld, int mn0, int mn_total, int k0, char* lds,
                       int tid, int wave_uniform) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 512 + tid;  // 16-B chunk id inside the 32-KiB tile image == LDS position
        long off;
        if (!TMAJ) {
            const int row = c >> 3, slot = c & 7;
            const int src = slot ^ (row & 7);
            const int gm = min(mn0 + row, mn_total - 1);
            off = (long)gm * ld + k0 + src * 8;
        } else {
            const int krow = c >> 5, slot = c & 31;
            const int src = (((slot >> 1) ^ tmaj_f(krow)) << 1) | (slot & 1);
            const int gm = min(mn0 + src * 8, mn_total - 8);
            off = (long)(k0 + krow) * ld + gm;
        }
        char* dst = lds + (i * 512 + wave_uniform * 64) * 16;  // wave-uniform LDS base of this instruction
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + off),
                                         (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
    }
}

template <bool TMAJ>
DEVINL u32x4_t g256_frag(const char* lds, int base16, int s, int lane) {
    if (!TMAJ) {
        const int rl = base16 + (lane & 15);
        const int kc = s * 4 + (lane >> 4);
        return *reinterpret_cast<const u32x4_t*>(lds + rl * 128 + ((kc ^ (rl & 7)) << 4));
    } else {
        const int t = lane & 15, g = lane >> 4;
        const int krow = s * 32 + 8 * g + (t >> 2);
        const int col = base16 + 4 * (t & 3);  // element column inside the 256-wide row
        const int c32 = col >> 4, within = (col & 15) * 2;
        const s16x4_t lo = lds_read_tr16(lds + krow * 512 + ((c32 ^ tmaj_f(krow)) << 5) + within);
        const s16x4_t hi = lds_read_tr16(lds + (krow + 4) * 512 + ((c32 ^ tmaj_f(krow + 4)) << 5) + within);
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
        const u32x4_t r = {l2[0], l2[1], h2[0], h2[1]};
        return r;
    }
}

// acc[i][j] (i < 8 m-tiles, j < 4 n-tiles of this wave), computed swapped: lane owns 4 consecutive N of one M row.
template <typename Elem, bool A_TMAJ, bool B_TMAJ>
DEVINL void g256_mainloop(const unsigned short* __restrict__ A, long lda, int M, const unsigned short* __restrict__ B,
                          long ldb, int N, int m0, int n0, int kbeg, int kend, char* lds, f32x4_t (&acc)[8][4]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (w >> 2) * 128, wn = (w & 3) * 64;
    const int nk = (kend - kbeg) / G256_BK;
    if (nk <= 0) return;
    g256_stage<A_TMAJ>(A, lda, m0, M, kbeg, lds, tid, w);
    g256_stage<B_TMAJ>(B, ldb, n0, N, kbeg, lds + G256_OPBYTES, tid, w);
    __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) in front of the barrier
    for (int t = 0; t < nk; ++t) {
        char* cur = lds + (t & 1) * 2 * G256_OPBYTES;
        char* nxt = lds + ((t + 1) & 1) * 2 * G256_OPBYTES;
        if (t + 1 < nk) {
            const int k0 = kbeg + (t + 1) * G256_BK;
            g256_stage<A_TMAJ>(A, lda, m0, M, k0, nxt, tid, w);
            g256_stage<B_TMAJ>(B, ldb, n0, N, k0, nxt + G256_OPBYTES, tid, w);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4_t bf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = g256_frag<B_TMAJ>(cur + G256_OPBYTES, wn + j * 16, s, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u32x4_t af = g256_frag<A_TMAJ>(cur, wm + i * 16, s, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Elem::mfma(bf[j], af, acc[i][j]);
            }
        }
        __syncthreads();
    }
}
