// uniir_gemm: 16-bit MFMA GEMM with fused epilogues (see include/uniir_hip.h, gemm_core.h).
// Replaces the cuBLAS calls behind nn.Linear / nn.MultiheadAttention.in_proj / Conv2d-as-GEMM of the
// CLIP towers (openai/CLIP model.py as called from clip_sf.py:43-47) and their autograd backward.
#include "gemm_core.h"
#include "gemm_core256.h"
#include "gemm_core_pp.h"
#include "../../include/uniir_hip.h"
#include "timing_filter.h"
#include <stdlib.h>
#include <vector>

// per-row epilogue operands (residual stream, stashed pre-activation) are read exactly once: loaded with the nt policy.
// MEASURED (round 3, same box, interleaved twice, ViT-L/14 512-pair step): 634.5 / 635.6 ms with plain loads, 632.9 / 631.9 ms with
// nt (+0.4 %); -DUNIIR_EPI_NT=0 builds the plain loads.
#ifndef UNIIR_EPI_NT
#define UNIIR_EPI_NT 1
#endif
#if UNIIR_EPI_NT
#define EPI_LD(p) __builtin_nontemporal_load(p)
#else
#define EPI_LD(p) (*(p))
#endif

struct GemmKArgs {
    const unsigned short* A;
    const unsigned short* B;
    void* C;
    void* C2;
    const float* bias;
    const float* resid;
    const unsigned short* aux;
    int M, N, K;
    long lda, ldb, ldc, ldaux;
    int epilogue, act, k_splits, tiles_m, tiles_n;
    float alpha;
    float* slab;  // split-K slabs [k_splits][M][N] (plain stores) or nullptr (atomics)
    int asm_loop;   // 1 counted-lgkmcnt double-buffer loop, 2 ping-pong loop
    float* colsum;  // optional [N]: += column sums of the fp32 result (bias gradient), 256-tile staged epilogue only
    int raster_gm, raster_cw;   // 2-D tile rasterisation block (row panels x column panels), 0 = row-major
    int skip_f;                 // EPI_BIAS_ACT without the pre-activation store (UNIIR_EPI_ACT_ONLY: forward-only passes)
    const float* row_scale;     // EPI_RESID_F32: (v + bias) * row_scale[m] + resid (DropPath factor of the row's item), or nullptr
    float* a_rowsum;            // optional [M]: += sum_k A^T[m][k] (transposed-A ping-pong kernel only, see gemm_core_pp.h)
    float* colsum_part;         // deterministic form of colsum: row panel mt STORES its column partial at [mt][N] (gemm_impl reduces)
    float* rowsum_part;         // ... of a_rowsum: block (split, nt) stores its row partial at [split * tiles_n + nt][M]
};

// act_fwd / act_bwd: common.h (shared with the stand-alone activation pass of elementwise.hip)

template <int EPI, int MT, int NT, bool F16 = false>
DEVINL void gemm_epilogue(const GemmKArgs& p, const f32x4_t (&acc)[MT][NT], int m0, int n0, int wm, int wn) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn + j * 16 + 4 * (lane >> 4);
            if (m < p.M && n < p.N) {
                f32x4_t v = acc[i][j] * p.alpha;
                if (p.bias) v += *reinterpret_cast<const f32x4_t*>(p.bias + n);
                const long off = (long)m * p.ldc + n;
                if (EPI == UNIIR_EPI_BF16) {
                    u32x2_t o = {pack16x2<F16>(v[0], v[1]), pack16x2<F16>(v[2], v[3])};
                    *reinterpret_cast<u32x2_t*>((unsigned short*)p.C + off) = o;
                } else if (EPI == UNIIR_EPI_BIAS_ACT) {
                    u32x2_t o = {pack16x2<F16>(v[0], v[1]), pack16x2<F16>(v[2], v[3])};
                    if (!p.skip_f) *reinterpret_cast<u32x2_t*>((unsigned short*)p.C + off) = o;
                    const float g0 = act_fwd(unpack16_lo<F16>(o[0]), p.act);
                    const float g1 = act_fwd(unpack16_hi<F16>(o[0]), p.act);
                    const float g2 = act_fwd(unpack16_lo<F16>(o[1]), p.act);
                    const float g3 = act_fwd(unpack16_hi<F16>(o[1]), p.act);
                    u32x2_t o2 = {pack16x2<F16>(g0, g1), pack16x2<F16>(g2, g3)};
                    *reinterpret_cast<u32x2_t*>((unsigned short*)p.C2 + off) = o2;
                } else if (EPI == UNIIR_EPI_RESID_F32) {
                    if (p.row_scale) v *= p.row_scale[m];
                    if (p.resid) v += *reinterpret_cast<const f32x4_t*>(p.resid + off);
                    *reinterpret_cast<f32x4_t*>((float*)p.C + off) = v;
                    if (p.C2) {
                        u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                        *reinterpret_cast<u32x2_t*>((unsigned short*)p.C2 + off) = o;
                    }
                } else if (EPI == UNIIR_EPI_DACT) {
                    const u32x2_t a = *reinterpret_cast<const u32x2_t*>(p.aux + (long)m * p.ldaux + n);
                    const float f0 = __uint_as_float(a[0] << 16), f1 = __uint_as_float(a[0] & 0xffff0000u);
                    const float f2 = __uint_as_float(a[1] << 16), f3 = __uint_as_float(a[1] & 0xffff0000u);
                    u32x2_t o = {pack_bf16x2(v[0] * act_bwd(f0, p.act), v[1] * act_bwd(f1, p.act)),
                                 pack_bf16x2(v[2] * act_bwd(f2, p.act), v[3] * act_bwd(f3, p.act))};
                    *reinterpret_cast<u32x2_t*>((unsigned short*)p.C + off) = o;
                } else if (EPI == UNIIR_EPI_F32) {
                    *reinterpret_cast<f32x4_t*>((float*)p.C + off) = v;
                } else {  // UNIIR_EPI_ATOMIC_F32
                    float* c = (float*)p.C + off;
                    unsafeAtomicAdd(c + 0, v[0]);
                    unsafeAtomicAdd(c + 1, v[1]);
                    unsafeAtomicAdd(c + 2, v[2]);
                    unsafeAtomicAdd(c + 3, v[3]);
                }
            }
        }
    }
}

template <int MT, int NT, bool F16 = false>
DEVINL void gemm_epilogue_dispatch(const GemmKArgs& p, const f32x4_t (&acc)[MT][NT], int m0, int n0, int wm, int wn) {
    switch (p.epilogue) {
        case UNIIR_EPI_BF16: gemm_epilogue<UNIIR_EPI_BF16, MT, NT, F16>(p, acc, m0, n0, wm, wn); break;
        case UNIIR_EPI_BIAS_ACT: gemm_epilogue<UNIIR_EPI_BIAS_ACT, MT, NT, F16>(p, acc, m0, n0, wm, wn); break;
        case UNIIR_EPI_RESID_F32: gemm_epilogue<UNIIR_EPI_RESID_F32, MT, NT>(p, acc, m0, n0, wm, wn); break;
        case UNIIR_EPI_DACT: gemm_epilogue<UNIIR_EPI_DACT, MT, NT>(p, acc, m0, n0, wm, wn); break;
        case UNIIR_EPI_F32: gemm_epilogue<UNIIR_EPI_F32, MT, NT>(p, acc, m0, n0, wm, wn); break;
        default: gemm_epilogue<UNIIR_EPI_ATOMIC_F32, MT, NT>(p, acc, m0, n0, wm, wn); break;
    }
}

template <typename Elem, bool A_TMAJ, bool B_TMAJ>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmKArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nwg = gridDim.x;
    int id = xcd_remap(blockIdx.x, nwg);
    const int tiles = p.tiles_m * p.tiles_n;
    const int split = id / tiles;
    id -= split * tiles;
    const int mt = id / p.tiles_n, nt = id - mt * p.tiles_n;
    const int m0 = mt * GEMM_BM, n0 = nt * GEMM_BN;
    // split-K range, aligned to BK
    int kbeg = 0, kend = p.K;
    if (p.k_splits > 1) {
        const int ksteps = (p.K + GEMM_BK - 1) / GEMM_BK;
        const int per = (ksteps + p.k_splits - 1) / p.k_splits;
        kbeg = split * per * GEMM_BK;
        kend = min(p.K, (split + 1) * per * GEMM_BK);
        if (kbeg >= kend) return;
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    gemm_mainloop<Elem, A_TMAJ, B_TMAJ>(p.A, p.lda, p.M, p.B, p.ldb, p.N, m0, n0, kbeg, kend, lds, acc);

    const int w = threadIdx.x >> 6;
    if (p.slab) {
        GemmKArgs q = p;
        q.C = p.slab + (long)split * p.M * p.N;
        q.ldc = p.N;
        q.bias = nullptr;
        gemm_epilogue<UNIIR_EPI_F32, 4, 4>(q, acc, m0, n0, (w >> 1) * 64, (w & 1) * 64);
        return;
    }
    gemm_epilogue_dispatch<4, 4, Elem::F16>(p, acc, m0, n0, (w >> 1) * 64, (w & 1) * 64);
}

// Epilogue of the 256x256 (8-wave) tile staged through LDS so that every global access is a full 16-B-per-lane,
// row-contiguous transaction (the MFMA fragment layout alone gives 32-B pieces at a row stride).  The main loop's
// 128 KiB of LDS are free at this point.
//   bf16 outputs (EPI_BF16, EPI_BIAS_ACT): one pass, image [256][256] bf16, 16-B chunk index ^= (row & 15)
//   fp32 math on the way out (EPI_RESID_F32, EPI_DACT, EPI_F32 / split-K slabs): two passes of 128 rows,
//   image [128][256] f32, 16-B chunk index ^= (row & 7)
// PP = accumulator map of gemm_core_pp.h (acc[4h+i][2h'+j] at rows 128h + 64wr + 16i, cols 128h' + 32wc + 16j) instead
// of the contiguous 128x64 wave tile at (wm, wn).
// The epilogue is VALU-issue bound (2 waves per SIMD, ~4 clk per instruction), so it is written for instruction count:
// hardware bf16 packing, LDS / global addresses as one per-thread base plus compile-time immediates, bounds checks
// hoisted to one wave-uniform "full tile" test, alpha skipped when it is 1.
template <bool PP>
DEVINL int epi_row(int i, int w, int wm) {   // first row of accumulator tile i inside the 256-row block tile
    return PP ? (i >> 2) * 128 + (w >> 2) * 64 + (i & 3) * 16 : wm + i * 16;
}
template <bool PP>
DEVINL int epi_col(int j, int w, int wn) {
    return PP ? (j >> 1) * 128 + (w & 3) * 32 + (j & 1) * 16 : wn + j * 16;
}

// copy-out of one 128-row fp32 pass: thread -> rows r0 + 8 it, one 16-B chunk; EPI / ACT are compile-time so that the
// loop body carries no per-element branching
template <int EPI, int ACT, int SRCSTEP = 8192>      // SRCSTEP: LDS bytes between the thread's rows (8 rows of the staged image)
DEVINL void epi_f32_copy(const GemmKArgs& p, const char* src, long off0, long offa, long rstep, long rstep_aux, bool full,
                         bool colok, int rows_left, f32x4_t& csum, int mrow) {
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        if (full || (colok && 8 * it < rows_left)) {
            f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + it * SRCSTEP);
            const long off = off0 + it * rstep;
            if (EPI == UNIIR_EPI_RESID_F32) {
                if (p.row_scale) v *= p.row_scale[mrow + 8 * it];
                if (p.resid) v += EPI_LD(reinterpret_cast<const f32x4_t*>(p.resid + off));
                *reinterpret_cast<f32x4_t*>((float*)p.C + off) = v;
                if (p.C2) {
                    const u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<u32x2_t*>((unsigned short*)p.C2 + off) = o;
                }
            } else if (EPI == UNIIR_EPI_DACT) {
                const long oa = offa + it * rstep_aux;
                const u32x2_t a = EPI_LD(reinterpret_cast<const u32x2_t*>(p.aux + oa));
                const float f0 = __uint_as_float(a[0] << 16), f1 = __uint_as_float(a[0] & 0xffff0000u);
                const float f2 = __uint_as_float(a[1] << 16), f3 = __uint_as_float(a[1] & 0xffff0000u);
                v[0] *= act_bwd(f0, ACT); v[1] *= act_bwd(f1, ACT);
                v[2] *= act_bwd(f2, ACT); v[3] *= act_bwd(f3, ACT);
                const u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *reinterpret_cast<u32x2_t*>((unsigned short*)p.C + off) = o;
                if (p.C2) {  // recomputed activation act(aux) for the wgrad of the next linear
                    const u32x2_t g2 = {pack_bf16x2(act_fwd(f0, ACT), act_fwd(f1, ACT)),
                                        pack_bf16x2(act_fwd(f2, ACT), act_fwd(f3, ACT))};
                    *reinterpret_cast<u32x2_t*>((unsigned short*)p.C2 + oa) = g2;
                }
            } else {  // UNIIR_EPI_F32 (also the split-K slabs)
                *reinterpret_cast<f32x4_t*>((float*)p.C + off) = v;
            }
            csum += v;
        }
    }
}

// buffer addressing for the operand LOADS of the full-tile copy-outs: one SGPR descriptor at the TILE's first element, one 32-bit
// per-lane byte offset shared by every row, the row advance as a scalar offset -- no 64-bit per-lane address per row (16 rows x 3
// tensors of them is what pushed the first version of these loops into scratch).
// Loads only.  Measured (round 5, tools/r5/dact_dbg.py): 16-byte buffer STORES with a scalar row offset returned corrupted dwords
// when the next row's VALU rewrote the data registers right behind them -- the ">64-bit VMEM store, then VALU write of its data"
// hazard, which the compiler's hazard recogniser only pads when soffset is NOT a register -- so the outputs leave through plain
// global stores from a per-thread pointer.
DEVINL __amdgpu_buffer_rsrc_t epi_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, -1, 0x00020000);
}
// EPI_DACT copy-out of one 128-row pass, EIGHT columns per thread and row (round 4): the operand f, the result and the optional
// act(f) output are bf16, so four columns per thread meant 8-byte global loads and stores -- half-width requests for 2 x 2.2 GB (+2.2)
// per c_proj dgrad.  Thread -> rows r0 + 16 it, columns 8 c8 .. 8 c8 + 7: two 16-byte LDS reads (the two chunks of a thread share a
// bank group: a 2-way conflict on reads that are a small part of this loop), one 16-byte load of f, one 16-byte store (+1).
template <int ACT>
DEVINL void epi_dact_copy8(const GemmKArgs& p, const char* src0, const char* src1, long off0, long offa, long rstep, long rstep_aux,
                           bool full, bool colok, int rows_left, f32x4_t& cs0, f32x4_t& cs1) {
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
        if (full || (colok && 16 * it < rows_left)) {
            f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src0 + it * 16384);
            f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(src1 + it * 16384);
            const long oa = offa + it * rstep_aux;
            const u32x4_t a = EPI_LD(reinterpret_cast<const u32x4_t*>(p.aux + oa));
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = __uint_as_float(a[e] << 16);
                f[2 * e + 1] = __uint_as_float(a[e] & 0xffff0000u);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v0[e] *= act_bwd(f[e], ACT);
                v1[e] *= act_bwd(f[4 + e], ACT);
            }
            const u32x4_t o = {pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
            *reinterpret_cast<u32x4_t*>((unsigned short*)p.C + off0 + it * rstep) = o;
            if (p.C2) {  // recomputed activation act(aux) for the wgrad of the next linear
                u32x4_t g2;
#pragma unroll
                for (int e = 0; e < 4; ++e) g2[e] = pack_bf16x2(act_fwd(f[2 * e], ACT), act_fwd(f[2 * e + 1], ACT));
                *reinterpret_cast<u32x4_t*>((unsigned short*)p.C2 + oa) = g2;
            }
            cs0 += v0;
            cs1 += v1;
        }
    }
}

// the same on a full tile with the pass's eight operand pieces already requested (round 5: see epilogue256_resid_full -- inside the
// loop above every load is followed by s_waitcnt vmcnt(0) because the act(f)-output option is a branch between the rows)
template <int ACT, bool HAS_C2>
DEVINL void epi_dact_copy8_full(const char* src0, const char* src1, char* pC, char* pC2, unsigned soC, unsigned soA, unsigned rowC,
                                unsigned rowA, const u32x4_t (&a8)[8], f32x4_t& cs0, f32x4_t& cs1) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src0 + it * 16384);
        f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(src1 + it * 16384);
        const u32x4_t a = a8[it];
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[2 * e] = __uint_as_float(a[e] << 16);
            f[2 * e + 1] = __uint_as_float(a[e] & 0xffff0000u);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v0[e] *= act_bwd(f[e], ACT);
            v1[e] *= act_bwd(f[4 + e], ACT);
        }
        const u32x4_t o = {pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
        *reinterpret_cast<u32x4_t*>(pC + (size_t)(soC + 16 * it * rowC)) = o;
        if (HAS_C2) {
            u32x4_t g2;
#pragma unroll
            for (int e = 0; e < 4; ++e) g2[e] = pack_bf16x2(act_fwd(f[2 * e], ACT), act_fwd(f[2 * e + 1], ACT));
            *reinterpret_cast<u32x4_t*>(pC2 + (size_t)(soA + 16 * it * rowA)) = g2;
        }
        cs0 += v0;
        cs1 += v1;
    }
}
template <bool HAS_C2>
DEVINL void epi_dact_copy8_full_act(int act, const char* src0, const char* src1, char* pC, char* pC2, unsigned soC, unsigned soA,
                                    unsigned rowC, unsigned rowA, const u32x4_t (&a8)[8], f32x4_t& cs0, f32x4_t& cs1) {
    if (act == UNIIR_ACT_QUICKGELU) epi_dact_copy8_full<UNIIR_ACT_QUICKGELU, HAS_C2>(src0, src1, pC, pC2, soC, soA, rowC, rowA, a8, cs0, cs1);
    else if (act == UNIIR_ACT_GELU_ERF) epi_dact_copy8_full<UNIIR_ACT_GELU_ERF, HAS_C2>(src0, src1, pC, pC2, soC, soA, rowC, rowA, a8, cs0, cs1);
    else epi_dact_copy8_full<UNIIR_ACT_RELU, HAS_C2>(src0, src1, pC, pC2, soC, soA, rowC, rowA, a8, cs0, cs1);
}

struct DactFull {
    __amdgpu_buffer_rsrc_t rA;       // operand f: buffer loads (tile base, 32-bit lane offset vA, scalar row offset)
    char *pC, *pC2;                  // this thread's first element of the outputs (see the note on buffer stores above)
    unsigned vA, rowC, rowA;
    const char *src8a, *src8b;
    int act;
    bool has_c2;
};
template <int H>
DEVINL void dact8_request(const DactFull& d, u32x4_t (&fa)[8]) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        fa[it] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(d.rA, d.vA, (unsigned)(128 * H + 16 * it) * d.rowA, UNIIR_EPI_NT ? 2 : 0));
    }
}
template <int H>
DEVINL void dact8_pass_full(const DactFull& d, const f32x4_t (&acc)[8][4], char* const (&sj)[4], f32x4_t alpha4, int w, f32x4_t& cs0,
                            f32x4_t& cs1, const u32x4_t (&fa)[8]) {
    if (H) __syncthreads();
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int rb = ((w >> 2) * 64 + ii * 16) * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(sj[j] + rb) = acc[4 * H + ii][j] * alpha4;
    }
    __syncthreads();
    if (d.has_c2) epi_dact_copy8_full_act<true>(d.act, d.src8a, d.src8b, d.pC, d.pC2, 128u * H * d.rowC, 128u * H * d.rowA, d.rowC, d.rowA, fa, cs0, cs1);
    else epi_dact_copy8_full_act<false>(d.act, d.src8a, d.src8b, d.pC, d.pC2, 128u * H * d.rowC, 128u * H * d.rowA, d.rowC, d.rowA, fa, cs0, cs1);
}

// The epilogue of the DACT-only instantiation of the ping-pong kernel (gemm_glds_kernel<.., 4>: the c_proj dgrad of the towers).  Its
// own kernel because the 256x256 kernel sits at 244-246 registers: the same code inside the shared epilogue256_staged spilled ~30
// registers in EVERY instantiation and slowed the long-K plain dgrads by 7 % (measured, round 4).  Also measured here: four rows
// instead of two per unrolled group (no change), and the first four operand pieces of a pass requested before the pass is staged,
// the copy loop fully unrolled (2.56 -> 5.49 ms: no spills, 209 registers, and a schedule that waits for everything).
DEVINL void epilogue256_dact8(const GemmKArgs& p, const f32x4_t (&acc)[8][4], int m0, int n0, char* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool full = (m0 + 256 <= p.M) && (n0 + 256 <= p.N);
    const f32x4_t alpha4 = {p.alpha, p.alpha, p.alpha, p.alpha};
    char* sj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = epi_col<true>(j, w, 0) + 4 * lg;
        sj[j] = lds + li * 1024 + (((nl >> 2) ^ (li & 7)) << 4);
    }
    const int r8 = tid >> 5, c8 = tid & 31;      // copy-out: rows r8 + 16 it of the pass, 16-B chunks 2 c8 and 2 c8 + 1 of the fp32 image
    const char* src8a = lds + r8 * 1024 + (((2 * c8) ^ (r8 & 7)) << 4);
    const char* src8b = lds + r8 * 1024 + (((2 * c8 + 1) ^ (r8 & 7)) << 4);
    const bool colok8 = n0 + c8 * 8 < p.N;
    const long rstep = 16L * p.ldc, rstep_aux = 16L * p.ldaux;
    f32x4_t cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};
    if (full) {
        // buffer addressing relative to the tile (see epi_rsrc); the pass's operand pieces are requested before the pass is staged
        DactFull d;
        d.rA = epi_rsrc(p.aux + (long)m0 * p.ldaux + n0);
        d.vA = (unsigned)(r8 * (int)p.ldaux + c8 * 8) * 2u;
        d.pC = (char*)((unsigned short*)p.C + (long)(m0 + r8) * p.ldc + n0 + c8 * 8);
        d.pC2 = p.C2 ? (char*)((unsigned short*)p.C2 + (long)(m0 + r8) * p.ldaux + n0 + c8 * 8) : nullptr;
        d.rowC = (unsigned)p.ldc * 2u;
        d.rowA = (unsigned)p.ldaux * 2u;
        d.src8a = src8a; d.src8b = src8b; d.act = p.act; d.has_c2 = p.C2 != nullptr;
        u32x4_t fa0[8], fa1[8];        // (both requests before pass 0 is staged: measured 2.29 vs 2.25 ms, not kept)
        dact8_request<0>(d, fa0);
        __builtin_amdgcn_sched_barrier(0);
        dact8_pass_full<0>(d, acc, sj, alpha4, w, cs0, cs1, fa0);
        __builtin_amdgcn_sched_barrier(0);
        dact8_request<1>(d, fa1);
        __builtin_amdgcn_sched_barrier(0);
        dact8_pass_full<1>(d, acc, sj, alpha4, w, cs0, cs1, fa1);
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h) __syncthreads();
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int rb = ((w >> 2) * 64 + ii * 16) * 1024;
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(sj[j] + rb) = acc[4 * h + ii][j] * alpha4;
            }
            __syncthreads();
            const int mrow8 = m0 + 128 * h + r8;
            const long o8 = (long)mrow8 * p.ldc + n0 + c8 * 8, a8 = (long)mrow8 * p.ldaux + n0 + c8 * 8;
            const int left8 = p.M - mrow8;
            if (p.act == UNIIR_ACT_QUICKGELU) epi_dact_copy8<UNIIR_ACT_QUICKGELU>(p, src8a, src8b, o8, a8, rstep, rstep_aux, full, colok8, left8, cs0, cs1);
            else if (p.act == UNIIR_ACT_GELU_ERF) epi_dact_copy8<UNIIR_ACT_GELU_ERF>(p, src8a, src8b, o8, a8, rstep, rstep_aux, full, colok8, left8, cs0, cs1);
            else epi_dact_copy8<UNIIR_ACT_RELU>(p, src8a, src8b, o8, a8, rstep, rstep_aux, full, colok8, left8, cs0, cs1);
        }
    }
    if (p.colsum) {      // the thread's 8 columns as two chunks, summed over its rows of both passes; 16 threads share them
        __syncthreads();
        f32x4_t* red = reinterpret_cast<f32x4_t*>(lds);
        red[2 * tid] = cs0;
        red[2 * tid + 1] = cs1;
        __syncthreads();
        if (tid < 64) {          // chunk tid = columns 4 tid ..: thread column group tid >> 1, half tid & 1
            f32x4_t s = red[2 * (tid >> 1) + (tid & 1)];
#pragma unroll
            for (int k = 1; k < 16; ++k) s += red[2 * ((tid >> 1) + 32 * k) + (tid & 1)];
            const int n = n0 + tid * 4;
            if (n < p.N) {
                if (p.colsum_part) {          // this row panel's partial; added to colsum in panel order by reduce_partials
                    *reinterpret_cast<f32x4_t*>(p.colsum_part + (long)(m0 >> 8) * p.N + n) = s;
                } else {
                    unsafeAtomicAdd(p.colsum + n + 0, s[0]);
                    unsafeAtomicAdd(p.colsum + n + 1, s[1]);
                    unsafeAtomicAdd(p.colsum + n + 2, s[2]);
                    unsafeAtomicAdd(p.colsum + n + 3, s[3]);
                }
            }
        }
    }
}

// bf16 copy-out with the activation copy (EPI_BIAS_ACT): C <- f, C2 <- act(f)
template <int ACT, bool F16 = false, int SRCSTEP = 8192>
DEVINL void epi_bf16_copy_act(const char* src, unsigned short* c1, unsigned short* c2, long rstep, bool full, bool colok,
                              int rows_left, bool skip_f) {
#pragma unroll 2
    for (int it = 0; it < 16; ++it) {
        if (full || (colok && 16 * it < rows_left)) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + it * SRCSTEP);
            if (!skip_f) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(c1));
            u32x4_t g;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                g[e] = pack16x2<F16>(act_fwd(unpack16_lo<F16>(v[e]), ACT), act_fwd(unpack16_hi<F16>(v[e]), ACT));
            *reinterpret_cast<u32x4_t*>(c2) = g;
        }
        c1 += rstep;
        c2 += rstep;
    }
}

// EPI_RESID_F32 on a full tile with a residual operand (the out_proj / c_proj forward of every tower).  Round 5: the generic copy-out
// above carries its run-time options (row_scale / resid / C2 / partial tiles) as wave-uniform BRANCHES inside the row loop, and with a
// branch between them hipcc issues ONE residual load per row followed by s_waitcnt vmcnt(0) -- which on gfx9 also drains the previous
// row's stores: 32 serialised HBM round trips per tile (28 us per tile at 19 GB/s per CU, profiles/r05_epilogue_stamps.txt).  Here the
// options are compile-time, and a pass's 16 residual pieces are requested up front -- before the pass is even staged -- into 64
// registers that the main loop's fragments no longer need: one round trip per pass, under the staging.
struct ResidFull {
    __amdgpu_buffer_rsrc_t rR;   // residual operand: buffer loads (tile base, 32-bit lane offset voff, scalar row offset)
    char *pC, *pC2;              // this thread's first element of the outputs
    unsigned voff, rowb;
    const char* src;
    const float* row_scale;      // + first row of the wave (wave-uniform)
};
template <int H, bool HAS_SCALE>
DEVINL void resid_request(const ResidFull& d, f32x4_t (&r)[16], float (&sc)[HAS_SCALE ? 16 : 1]) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        r[it] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(d.rR, d.voff, (unsigned)(128 * H + 8 * it) * d.rowb, UNIIR_EPI_NT ? 2 : 0));
        if (HAS_SCALE) sc[it] = d.row_scale[128 * H + 8 * it];
    }
}
template <int H>
DEVINL void resid_stage(const f32x4_t (&acc)[8][4], const f32x4_t (&bv)[4], char* const (&sj)[4], f32x4_t alpha4, int w) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int rb = ((w >> 2) * 64 + ii * 16) * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(sj[j] + rb) = acc[4 * H + ii][j] * alpha4 + bv[j];
    }
}
template <int H, bool HAS_C2, bool HAS_SCALE>
DEVINL void resid_copy_out(const ResidFull& d, const f32x4_t (&r)[16], const float (&sc)[HAS_SCALE ? 16 : 1], f32x4_t& csum) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const unsigned so = (unsigned)(128 * H + 8 * it) * d.rowb;
        f32x4_t v = *reinterpret_cast<const f32x4_t*>(d.src + it * 8192);
        if (HAS_SCALE) v *= sc[it];
        v += r[it];
        *reinterpret_cast<f32x4_t*>(d.pC + (size_t)so) = v;
        if (HAS_C2) {
            const u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *reinterpret_cast<u32x2_t*>(d.pC2 + (size_t)(so >> 1)) = o;
        }
        csum += v;
    }
}
template <bool HAS_C2, bool HAS_SCALE>
DEVINL void epilogue256_resid_full(const GemmKArgs& p, const f32x4_t (&acc)[8][4], const f32x4_t (&bv)[4], char* const (&sj)[4],
                                   int m0, int n0, char* lds, f32x4_t& csum) {
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const f32x4_t alpha4 = {p.alpha, p.alpha, p.alpha, p.alpha};
    const int r0 = tid >> 6, ch = tid & 63;   // copy-out: row r0 + 8 it of the pass, 16-B chunk ch (4 floats)
    const long tile = (long)m0 * p.ldc + n0;
    ResidFull d;
    d.src = lds + r0 * 1024 + ((ch ^ (r0 & 7)) << 4);
    d.rR = epi_rsrc(p.resid + tile);
    d.voff = (unsigned)(r0 * (int)p.ldc + ch * 4) * 4u;
    d.pC = (char*)((float*)p.C + tile) + d.voff;
    d.pC2 = HAS_C2 ? (char*)((unsigned short*)p.C2 + tile) + (d.voff >> 1) : nullptr;
    d.rowb = (unsigned)p.ldc * 4u;
    d.row_scale = HAS_SCALE ? p.row_scale + m0 + w : nullptr;     // r0 == w: the row is wave-uniform -> scalar loads, no VGPRs
    // order: request pass 0's pieces | stage pass 0 | copy pass 0 out | request pass 1's pieces | stage pass 1 | copy pass 1 out.
    // (Measured and dropped: pass 1's request under pass 0's copy-out -- 64 + 64 operand registers next to the 64 accumulators still to
    // be staged spill ~50 registers, and every spill reload is a load in the same vmcnt queue: out forward 0.81 instead of 0.68 ms.)
    f32x4_t ra[16], rb[16];
    float sa[HAS_SCALE ? 16 : 1], sb[HAS_SCALE ? 16 : 1];
    resid_request<0, HAS_SCALE>(d, ra, sa);
    resid_stage<0>(acc, bv, sj, alpha4, w);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    resid_copy_out<0, HAS_C2, HAS_SCALE>(d, ra, sa, csum);
    __builtin_amdgcn_sched_barrier(0);
    resid_request<1, HAS_SCALE>(d, rb, sb);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    resid_stage<1>(acc, bv, sj, alpha4, w);
    __syncthreads();
    resid_copy_out<1, HAS_C2, HAS_SCALE>(d, rb, sb, csum);
}

template <bool PP, bool F16 = false>
DEVINL void epilogue256_staged(const GemmKArgs& p, const f32x4_t (&acc)[8][4], int m0, int n0, int wm, int wn,
                               char* lds, int epi) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool full = (m0 + 256 <= p.M) && (n0 + 256 <= p.N);
    const f32x4_t alpha4 = {p.alpha, p.alpha, p.alpha, p.alpha};
    f32x4_t bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + epi_col<PP>(j, w, wn) + 4 * lg;
        bv[j] = (p.bias && n < p.N) ? *reinterpret_cast<const f32x4_t*>(p.bias + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    if (epi == UNIIR_EPI_BF16 || epi == UNIIR_EPI_BIAS_ACT) {
        // image [256][256] bf16, 512 B per row, 16-B chunk index ^= (row & 15) (== li): rows r and r+8 of a ds_write_b64 lane group land on different banks
        char* sj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = epi_col<PP>(j, w, wn) + 4 * lg;
            sj[j] = lds + li * 512 + (((nl >> 3) ^ li) << 4) + ((nl & 7) << 1);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rb = epi_row<PP>(i, w, wm) * 512;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4_t v = acc[i][j] * alpha4 + bv[j];     // two v_pk_fma_f32 (alpha == 1 is exact)
                const u32x2_t o = {pack16x2<F16>(v[0], v[1]), pack16x2<F16>(v[2], v[3])};
                *reinterpret_cast<u32x2_t*>(sj[j] + rb) = o;
            }
        }
        __syncthreads();
        // thread -> (row r0 + 16 it, 16-B chunk ch): LDS address and global offset advance by constants
        const int r0 = tid >> 5, ch = tid & 31;
        const char* src = lds + r0 * 512 + ((ch ^ r0) << 4);
        const long off0 = (long)(m0 + r0) * p.ldc + n0 + ch * 8;
        unsigned short* c1 = (unsigned short*)p.C + off0;
        unsigned short* c2 = (unsigned short*)p.C2 + off0;
        const long rstep = 16L * p.ldc;
        const bool colok = n0 + ch * 8 < p.N;
        const int rows_left = p.M - m0 - r0;
        const bool act = epi == UNIIR_EPI_BIAS_ACT;
        if (full && !act) {          // the common case: straight copy, one pointer bump per row group
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + it * 8192);
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(c1));
                c1 += rstep;
            }
            return;
        }
        if (!act) {
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                if (colok && 16 * it < rows_left) {
                    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + it * 8192);
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(c1));
                }
                c1 += rstep;
            }
        } else if (p.act == UNIIR_ACT_QUICKGELU) {
            epi_bf16_copy_act<UNIIR_ACT_QUICKGELU, F16>(src, c1, c2, rstep, full, colok, rows_left, p.skip_f != 0);
        } else if (p.act == UNIIR_ACT_GELU_ERF) {
            epi_bf16_copy_act<UNIIR_ACT_GELU_ERF, F16>(src, c1, c2, rstep, full, colok, rows_left, p.skip_f != 0);
        } else {
            epi_bf16_copy_act<UNIIR_ACT_RELU, F16>(src, c1, c2, rstep, full, colok, rows_left, p.skip_f != 0);
        }
        return;
    }
    // fp32 math on the way out: two passes of 128 rows, image [128][256] f32 (1 KiB per row), 16-B chunk ^= (row & 7).
    // !PP: the waves with wm == 128 h stage all their 8 tiles in pass h; PP: every wave stages tiles 4h..4h+3.
    char* sj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = epi_col<PP>(j, w, wn) + 4 * lg;
        sj[j] = lds + li * 1024 + (((nl >> 2) ^ (li & 7)) << 4);
    }
    const int r0 = tid >> 6, ch = tid & 63;   // copy-out: row r0 + 8 it of the pass, 16-B chunk ch (4 floats)
    const char* src = lds + r0 * 1024 + ((ch ^ (r0 & 7)) << 4);
    const bool colok = n0 + ch * 4 < p.N;
    const long rstep = 8L * p.ldc, rstep_aux = 8L * p.ldaux;
    f32x4_t csum = {0.f, 0.f, 0.f, 0.f};
    const bool resid_fast = PP && epi == UNIIR_EPI_RESID_F32 && full && p.resid != nullptr;
    if (resid_fast) {
        if (p.C2) {
            if (p.row_scale) epilogue256_resid_full<true, true>(p, acc, bv, sj, m0, n0, lds, csum);
            else epilogue256_resid_full<true, false>(p, acc, bv, sj, m0, n0, lds, csum);
        } else {
            if (p.row_scale) epilogue256_resid_full<false, true>(p, acc, bv, sj, m0, n0, lds, csum);
            else epilogue256_resid_full<false, false>(p, acc, bv, sj, m0, n0, lds, csum);
        }
    }
    // (EPI_DACT's operand loads sit in the copy-out loop, four rows at a time: their round trip is the larger half of this
    // epilogue's cost -- c_proj dgrad at ViT-L/14 x 1024 items 2.99 ms against 1.91 plain for 2.2 GB more.  Requesting a pass's 16
    // pieces before it is staged was tried in round 4, as plain loads, staggered over the two passes, on full tiles only, and as asm
    // loads behind one explicit wait: hipcc answered with 8 .. 140 spilled registers in every instantiation of the kernel, none kept.)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (resid_fast) break;
        if (h) __syncthreads();
        if (PP || wm == 128 * h) {
#pragma unroll
            for (int ii = 0; ii < (PP ? 4 : 8); ++ii) {
                const int i = PP ? 4 * h + ii : ii;
                const int rb = (PP ? (w >> 2) * 64 + ii * 16 : ii * 16) * 1024;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4_t*>(sj[j] + rb) = acc[i][j] * alpha4 + bv[j];
            }
        }
        __syncthreads();
        const int mrow = m0 + 128 * h + r0;
        const long off0 = (long)mrow * p.ldc + n0 + ch * 4;
        const long offa = (long)mrow * p.ldaux + n0 + ch * 4;
        const int rows_left = p.M - mrow;
        if (epi == UNIIR_EPI_RESID_F32)
            epi_f32_copy<UNIIR_EPI_RESID_F32, 0>(p, src, off0, offa, rstep, rstep_aux, full, colok, rows_left, csum, mrow);
        else if (epi == UNIIR_EPI_F32)
            epi_f32_copy<UNIIR_EPI_F32, 0>(p, src, off0, offa, rstep, rstep_aux, full, colok, rows_left, csum, mrow);
        else if (p.act == UNIIR_ACT_QUICKGELU)
            epi_f32_copy<UNIIR_EPI_DACT, UNIIR_ACT_QUICKGELU>(p, src, off0, offa, rstep, rstep_aux, full, colok, rows_left, csum, mrow);
        else if (p.act == UNIIR_ACT_GELU_ERF)
            epi_f32_copy<UNIIR_EPI_DACT, UNIIR_ACT_GELU_ERF>(p, src, off0, offa, rstep, rstep_aux, full, colok, rows_left, csum, mrow);
        else
            epi_f32_copy<UNIIR_EPI_DACT, UNIIR_ACT_RELU>(p, src, off0, offa, rstep, rstep_aux, full, colok, rows_left, csum, mrow);
    }
    if (p.colsum) {
        // this thread's 4 columns (ch = tid & 63) summed over its rows of both passes; 8 threads share a column group
        __syncthreads();
        f32x4_t* red = reinterpret_cast<f32x4_t*>(lds);
        red[tid] = csum;
        __syncthreads();
        if (tid < 64) {
            f32x4_t s = red[tid];
#pragma unroll
            for (int k = 1; k < 8; ++k) s += red[tid + 64 * k];
            const int n = n0 + tid * 4;
            if (n < p.N) {
                if (p.colsum_part) {          // this row panel's partial; added to colsum in panel order by reduce_partials
                    *reinterpret_cast<f32x4_t*>(p.colsum_part + (long)(m0 >> 8) * p.N + n) = s;
                } else {
                    unsafeAtomicAdd(p.colsum + n + 0, s[0]);
                    unsafeAtomicAdd(p.colsum + n + 1, s[1]);
                    unsafeAtomicAdd(p.colsum + n + 2, s[2]);
                    unsafeAtomicAdd(p.colsum + n + 3, s[3]);
                }
            }
        }
    }
}


// LDS-DMA GEMM (gemm_core256.h): block tile (128*WM) x (64*WN), K step BK.  Used when K % BK == 0.
// split-K / wgrad accumulate epilogue for the ping-pong accumulator map
DEVINL void epilogue_atomic_pp(const GemmKArgs& p, const f32x4_t (&acc)[8][4], int m0, int n0) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + (i >> 2) * 128 + (w >> 2) * 64 + (i & 3) * 16 + li;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + (j >> 1) * 128 + (w & 3) * 32 + (j & 1) * 16 + 4 * lg;
            if (m < p.M && n < p.N) {
                float* c = (float*)p.C + (long)m * p.ldc + n;
                const f32x4_t v = acc[i][j] * p.alpha;
                unsafeAtomicAdd(c + 0, v[0]);
                unsafeAtomicAdd(c + 1, v[1]);
                unsafeAtomicAdd(c + 2, v[2]);
                unsafeAtomicAdd(c + 3, v[3]);
            }
        }
    }
}

#if defined(UNIIR_EXP_BUILD) && defined(PP_TS)   // per-workgroup s_memtime stamps (100 MHz): start, main loop done, end
__device__ unsigned long long g_pp_ts[3 * 65536];
extern "C" int uniir_debug_read_ts(void* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_ts), (size_t)n * 8) == hipSuccess ? 0 : 1;
}
#if PP_TS == 2     // wall-clock stamps (100 MHz)
#define PP_STAMP(i) if (threadIdx.x == 0 && blockIdx.x < 65536) g_pp_ts[3 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime()
#else
#define PP_STAMP(i) if (threadIdx.x == 0 && blockIdx.x < 65536) g_pp_ts[3 * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime()
#endif
#else
#define PP_STAMP(i)
#endif
// LOOP: 0 = compiler-scheduled loop, 1 = counted-lgkmcnt asm loop, 2 = ping-pong 8-phase loop (gemm_core_pp.h), 3 = the same
// with the row sums of A^T (a_rowsum)
template <typename Elem, bool A_TMAJ, bool B_TMAJ, int WM, int WN, int BK, int LOOP>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_glds_kernel(GemmKArgs p) {
    using S = GldsShape<WM, WN, BK>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.tiles_m * p.tiles_n;
    const int split = id / tiles;
    id -= split * tiles;
    // 2-D rasterisation for wide outputs: the ~32 tiles an XCD works on at a time form a block of 8 row panels x 4 column
    // panels (column index fastest inside the block) instead of ~2 row panels x all columns -- 6 MB of distinct operand
    // panels per XCD-L2 working set instead of 9 MB at N = 4096 (fc forward / proj dgrad +4 ... 5 %).  With <= 4 column
    // panels the plain row-major order already is that block.
    int mt, nt;
    if (p.raster_gm > 0 && p.tiles_n > p.raster_cw) {
        const int GM = p.raster_gm, CW = p.raster_cw;
        const int gwidth = GM * p.tiles_n;
        const int gfirst = (id / gwidth) * GM;
        const int gsize = min(p.tiles_m - gfirst, GM);
        const int idg = id % gwidth;                     // position inside the group of gsize row panels
        const int chunk = idg / (gsize * CW);            // column chunk of (up to) CW panels
        const int cw = min(CW, p.tiles_n - chunk * CW);
        const int r = idg - chunk * gsize * CW;
        mt = gfirst + r / cw;
        nt = chunk * CW + r % cw;
    } else {
        mt = id / p.tiles_n;
        nt = id - mt * p.tiles_n;
    }
    const int m0 = mt * S::BM, n0 = nt * S::BN;
    int kbeg = 0, kend = p.K;
    if (p.k_splits > 1) {
        const int ksteps = p.K / BK;
        const int per = (ksteps + p.k_splits - 1) / p.k_splits;
        kbeg = split * per * BK;
        kend = min(p.K, (split + 1) * per * BK);
        if (kbeg >= kend) return;
    }
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    PP_STAMP(0);
    constexpr bool PP = (WM == 2 && WN == 4 && BK == 64 && (LOOP == 2 || LOOP == 3 || LOOP == 4));
    if (PP)
        glds_mainloop_pp<Elem, A_TMAJ, B_TMAJ, LOOP == 3>(p.A, p.lda, p.M, p.B, p.ldb, p.N, m0, n0, kbeg, kend, lds, acc,
                                                          (LOOP == 3 && p.rowsum_part) ? p.rowsum_part + (long)(split * p.tiles_n + nt) * p.M
                                                                                       : p.a_rowsum,
                                                          nt, p.tiles_n, LOOP == 3 && p.rowsum_part != nullptr);
    else if (WM == 2 && WN == 4 && BK == 64 && LOOP == 1)
        glds_mainloop_asm<Elem, A_TMAJ, B_TMAJ>(p.A, p.lda, p.M, p.B, p.ldb, p.N, m0, n0, kbeg, kend, lds, acc);
    else
        glds_mainloop<Elem, A_TMAJ, B_TMAJ, WM, WN, BK>(p.A, p.lda, p.M, p.B, p.ldb, p.N, m0, n0, kbeg, kend, lds, acc);
    PP_STAMP(1);
    const int w = threadIdx.x >> 6;
    const int wm = (w / WN) * 128, wn = (w % WN) * 64;
    if constexpr (LOOP == 4) {      // the DACT-only instantiation (launch_glds_dact): bias-free, no split-K
        epilogue256_dact8(p, acc, m0, n0, lds);
        PP_STAMP(2);
        return;
    }
    if (WM * WN == 8) {   // 256x256 tile: LDS-staged, fully coalesced epilogue
        if (p.slab) {
            GemmKArgs q = p;
            q.C = p.slab + (long)split * p.M * p.N;
            q.ldc = p.N;
            q.bias = nullptr;
            epilogue256_staged<PP, Elem::F16>(q, acc, m0, n0, wm, wn, lds, UNIIR_EPI_F32);
        } else if (p.epilogue == UNIIR_EPI_ATOMIC_F32) {
            if (PP) epilogue_atomic_pp(p, acc, m0, n0);
            else gemm_epilogue<UNIIR_EPI_ATOMIC_F32, 8, 4>(p, acc, m0, n0, wm, wn);
        } else {
            epilogue256_staged<PP, Elem::F16>(p, acc, m0, n0, wm, wn, lds, p.epilogue);
        }
        PP_STAMP(2);
        return;
    }
    if (p.slab) {
        GemmKArgs q = p;
        q.C = p.slab + (long)split * p.M * p.N;
        q.ldc = p.N;
        q.bias = nullptr;
        gemm_epilogue<UNIIR_EPI_F32, 8, 4>(q, acc, m0, n0, wm, wn);
        return;
    }
    gemm_epilogue_dispatch<8, 4, Elem::F16>(p, acc, m0, n0, wm, wn);
}

template <typename Elem, int WM, int WN, int BK, int ASM>
static int launch_glds(GemmKArgs a, int a_tmaj, int b_tmaj, hipStream_t st) {
    using S = GldsShape<WM, WN, BK>;
    a.tiles_m = (a.M + S::BM - 1) / S::BM;
    a.tiles_n = (a.N + S::BN - 1) / S::BN;
    a.raster_gm = 8;          // 2-D tile rasterisation: 8 row panels x 4 column panels per XCD working set (round-3 sweep, tools/r3)
    a.raster_cw = 4;
    const int grid = a.tiles_m * a.tiles_n * a.k_splits;
    dim3 g(grid), b(S::T);
    const size_t sm = S::LDS_BYTES;
#define LAUNCHG(AT, BT)                                                                           \
    do {                                                                                          \
        static PerDeviceOnce attr_set;                                                            \
        if (attr_set.first()) {                                                                   \
            (void)hipFuncSetAttribute((const void*)gemm_glds_kernel<Elem, AT, BT, WM, WN, BK, ASM>,    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);       \
        }                                                                                         \
        hipLaunchKernelGGL((gemm_glds_kernel<Elem, AT, BT, WM, WN, BK, ASM>), g, b, sm, st, a);        \
    } while (0)
#ifdef UNIIR_EXP_BUILD   // fast experimental build (tools/build_exp.sh): NT ping-pong kernel only
    if (!a_tmaj && !b_tmaj && ASM == 2) LAUNCHG(false, false);
    else return UNIIR_EUNSUPPORTED;
#else
    if (!a_tmaj && !b_tmaj) LAUNCHG(false, false);
    else if (!a_tmaj && b_tmaj) LAUNCHG(false, true);
    else if (a_tmaj && !b_tmaj) LAUNCHG(true, false);
    else LAUNCHG(true, true);
#endif
#undef LAUNCHG
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// shape choice: 0 = general 128x128 register-staged kernel, 1 = 256x256x64 LDS-DMA kernel (1 workgroup / CU)
static int gemm_shape(const GemmKArgs& a, int a_tmaj, int b_tmaj) {
    if (a.K % 64) return 0;
    if (a.M < 256 || a.N < 128) return 0;    // small problems: the 128-tile kernel fills the chip better
    return 1;
}

// does this problem run the ping-pong loop?
static bool pp_eligible(const GemmKArgs& a, int a_tmaj, int b_tmaj) {
    if (gemm_shape(a, a_tmaj, b_tmaj) != 1 || a.asm_loop != 2) return false;
    // the ping-pong loop needs >= 3 K steps in every split
    const int ksteps = a.K / 64, per = (ksteps + a.k_splits - 1) / a.k_splits;
    const int last = ksteps - per * (a.k_splits - 1);
    // ... and addresses the K advance of a T-major operand as a 32-bit byte offset
    const bool k32 = (!a_tmaj || (uint64_t)a.K * a.lda * 2 < (1ull << 32)) &&
                     (!b_tmaj || (uint64_t)a.K * a.ldb * 2 < (1ull << 32));
    return last >= 3 && k32;
}
// the transposed-A / transposed-B ping-pong kernel with the row sums of A^T (bf16 only: v_dot2c_f32_bf16)
static int launch_glds_rowsum(GemmKArgs a, hipStream_t st) {
    using S = GldsShape<2, 4, 64>;
    a.tiles_m = (a.M + S::BM - 1) / S::BM;
    a.tiles_n = (a.N + S::BN - 1) / S::BN;
    a.raster_gm = 0;
    a.raster_cw = 4;
    static PerDeviceOnce attr_set;
    if (attr_set.first())
        (void)hipFuncSetAttribute((const void*)gemm_glds_kernel<ElemBF16, true, true, 2, 4, 64, 3>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES);
    hipLaunchKernelGGL((gemm_glds_kernel<ElemBF16, true, true, 2, 4, 64, 3>), dim3(a.tiles_m * a.tiles_n * a.k_splits), dim3(S::T),
                       S::LDS_BYTES, st, a);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// dx = (dy @ w) * act'(f) [act(f) out] [column sums] on the ping-pong kernel with the 8-column copy-out (bf16, dy K-contiguous, w
// K-major: the towers' c_proj dgrad)
static int launch_glds_dact(GemmKArgs a, hipStream_t st) {
    using S = GldsShape<2, 4, 64>;
    a.tiles_m = (a.M + S::BM - 1) / S::BM;
    a.tiles_n = (a.N + S::BN - 1) / S::BN;
    a.raster_gm = 8;
    a.raster_cw = 4;
    static PerDeviceOnce attr_set;
    if (attr_set.first())
        (void)hipFuncSetAttribute((const void*)gemm_glds_kernel<ElemBF16, false, true, 2, 4, 64, 4>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES);
    hipLaunchKernelGGL((gemm_glds_kernel<ElemBF16, false, true, 2, 4, 64, 4>), dim3(a.tiles_m * a.tiles_n), dim3(S::T), S::LDS_BYTES, st, a);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

template <typename Elem>
static int launch_gemm(const GemmKArgs& a, int a_tmaj, int b_tmaj, hipStream_t st) {
    const int shape = gemm_shape(a, a_tmaj, b_tmaj);
    if (shape == 1) {
        if (pp_eligible(a, a_tmaj, b_tmaj)) {
#ifndef UNIIR_EXP_BUILD
            if (a.a_rowsum && std::is_same<Elem, ElemBF16>::value) return launch_glds_rowsum(a, st);
            if (a.epilogue == UNIIR_EPI_DACT && std::is_same<Elem, ElemBF16>::value && !a_tmaj && b_tmaj && a.k_splits == 1 &&
                !a.slab && !a.bias && a.N % 8 == 0 && a.ldc % 8 == 0 && a.ldaux % 8 == 0 && !((uintptr_t)a.aux & 15) &&
                (!a.C2 || !((uintptr_t)a.C2 & 15)))
                return launch_glds_dact(a, st);
#endif
            return launch_glds<Elem, 2, 4, 64, 2>(a, a_tmaj, b_tmaj, st);
        }
        return launch_glds<Elem, 2, 4, 64, 1>(a, a_tmaj, b_tmaj, st);
    }
    const int grid = a.tiles_m * a.tiles_n * a.k_splits;
    dim3 g(grid), b(256);
    const size_t sm = GEMM_LDS_BYTES;
#define LAUNCH(AT, BT)                                                                            \
    do {                                                                                          \
        static PerDeviceOnce attr_set;                                                            \
        if (attr_set.first()) {                                                                   \
            (void)hipFuncSetAttribute((const void*)gemm_kernel<Elem, AT, BT>,                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);             \
        }                                                                                         \
        hipLaunchKernelGGL((gemm_kernel<Elem, AT, BT>), g, b, sm, st, a);                         \
    } while (0)
#ifdef UNIIR_EXP_BUILD
    return UNIIR_EUNSUPPORTED;
#else
    if (!a_tmaj && !b_tmaj) LAUNCH(false, false);
    else if (!a_tmaj && b_tmaj) LAUNCH(false, true);
    else if (a_tmaj && !b_tmaj) LAUNCH(true, false);
    else LAUNCH(true, true);
#endif
#undef LAUNCH
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// C[m][n] (+)= sum_s slab[s][m][n]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int splits, long mn, int N,
                                                            float* __restrict__ C, long ldc) {
    const long nv = mn >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        f32x4_t s = *reinterpret_cast<const f32x4_t*>(slab + 4 * i);
        for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4_t*>(slab + (long)k * mn + 4 * i);
        const long e = 4 * i, m = e / N, n = e - m * N;
        f32x4_t* dst = reinterpret_cast<f32x4_t*>(C + m * ldc + n);
        *dst = *dst + s;
    }
}

// Measurement hook (bench.py roofline): every `stride`-th uniir_gemm call is bracketed by a pair of HIP events on the stream
// it is launched on, whoever the caller is (the C towers of tower.hip or a host-language loop).  Host-side bookkeeping only;
// not thread-safe (one measuring thread), off by default.
#define GT_MAX 4096
#define GT_FMAX 8192
static struct {
    int stride = 0;
    long counter = 0;
    int n = 0;
    bool created = false;
    bool filter = false;          // sample the launches on `only` alone (uniir_gemm_timing_on)
    bool fcreated = false;
    hipStream_t only = nullptr;
    int nf = 0;                   // uniir_gemm calls seen on OTHER streams while filtering: the windows in which the device was shared
    hipEvent_t ev[2 * GT_MAX];
    double flop[GT_MAX];
    hipEvent_t base;
    hipEvent_t fev[2 * GT_FMAX];
} g_gt;
extern "C" int uniir_gemm_timing(int32_t stride) {
    if (stride < 0) return UNIIR_EINVAL;
    if (stride > 0 && !g_gt.created) {
        for (int i = 0; i < 2 * GT_MAX; ++i)
            if (hipEventCreate(&g_gt.ev[i]) != hipSuccess) return UNIIR_ELAUNCH;
        g_gt.created = true;
    }
    g_gt.stride = stride;
    g_gt.counter = 0;
    g_gt.n = 0;
    g_gt.nf = 0;
    g_gt.filter = false;
    return UNIIR_OK;
}
// The same, counting and sampling the launches on ONE stream only.  An event pair measures the time between two points of its
// stream, which is the kernel's own duration only while no other stream shares the device.  With the two towers on two streams
// (clip_model.CLIP.side_leg) the measuring run follows the stream that carries the image tower, and every uniir_gemm call on another
// stream is bracketed too -- not as a sample but as a record of WHEN the device was shared: uniir_gemm_timing_read leaves out the
// samples that intersect those windows (windows closer than two mean sample durations are one window: the LayerNorm / attention
// kernels between two GEMMs of the other tower share the device just the same; the rule is host arithmetic in timing_filter.h) and
// uniir_gemm_timing_read_ex also returns how many it left out -- or that it kept them all because too few would have been left.
extern "C" int uniir_gemm_timing_on(int32_t stride, void* stream) {
    const int rc = uniir_gemm_timing(stride);
    if (rc != UNIIR_OK || stride == 0) return rc;
    if (!g_gt.fcreated) {
        if (hipEventCreate(&g_gt.base) != hipSuccess) return UNIIR_ELAUNCH;
        for (int i = 0; i < 2 * GT_FMAX; ++i)
            if (hipEventCreate(&g_gt.fev[i]) != hipSuccess) return UNIIR_ELAUNCH;
        g_gt.fcreated = true;
    }
    g_gt.filter = true;
    g_gt.only = (hipStream_t)stream;
    if (hipEventRecord(g_gt.base, g_gt.only) != hipSuccess) return UNIIR_ELAUNCH;
    return UNIIR_OK;
}
// sums over the sampled launches (call after synchronising the device): algorithmic 2 M N K, elapsed milliseconds, count; shared
// (may be NULL): samples left out because another stream's GEMMs shared the device with them (uniir_gemm_timing_on); fallback (may be
// NULL): 1 when the rule would have left fewer than GT_MIN_KEEP samples and the sums are over ALL samples instead (shared then counts
// what the rule wanted to drop).  The rule itself is host arithmetic in timing_filter.h.
extern "C" int uniir_gemm_timing_read_ex(double* flop, double* ms, int32_t* launches, int32_t* shared, int32_t* fallback) {
    if (!flop || !ms || !launches) return UNIIR_EINVAL;
    const int n = g_gt.n, nwin = g_gt.filter ? g_gt.nf : 0;
    std::vector<float> dur(n), samples(2 * (size_t)n), windows(2 * (size_t)nwin);
    std::vector<uint8_t> keep(n ? n : 1, 1);
    for (int i = 0; i < n; ++i)
        if (hipEventElapsedTime(&dur[i], g_gt.ev[2 * i], g_gt.ev[2 * i + 1]) != hipSuccess) return UNIIR_ELAUNCH;
    int fb = 0, left_out = 0;
    if (nwin > 0 && n > 0) {          // everything on the time axis of `base`
        for (int i = 0; i < nwin; ++i)
            if (hipEventElapsedTime(&windows[2 * i], g_gt.base, g_gt.fev[2 * i]) != hipSuccess ||
                hipEventElapsedTime(&windows[2 * i + 1], g_gt.base, g_gt.fev[2 * i + 1]) != hipSuccess) return UNIIR_ELAUNCH;
        for (int i = 0; i < n; ++i) {
            if (hipEventElapsedTime(&samples[2 * i], g_gt.base, g_gt.ev[2 * i]) != hipSuccess) return UNIIR_ELAUNCH;
            samples[2 * i + 1] = dur[i];
        }
        gt_filter_samples(windows.data(), nwin, samples.data(), n, -1.f, keep.data(), &fb, &left_out);
    }
    double f = 0.0, t = 0.0;
    int kept = 0;
    for (int i = 0; i < n; ++i)
        if (keep[i]) { f += g_gt.flop[i]; t += dur[i]; ++kept; }
    *flop = f; *ms = t; *launches = kept;
    if (shared) *shared = left_out;
    if (fallback) *fallback = fb;
    return UNIIR_OK;
}
extern "C" int uniir_gemm_timing_read(double* flop, double* ms, int32_t* launches) {
    return uniir_gemm_timing_read_ex(flop, ms, launches, nullptr, nullptr);
}
// the rule alone, on caller-supplied times (no device needed): tests, and a host-language caller that keeps its own event log
extern "C" int uniir_gemm_timing_filter(const float* windows, int32_t nwin, const float* samples, int32_t n, float merge_ms,
                                        uint8_t* keep, int32_t* fallback) {
    if (nwin < 0 || n < 0 || (nwin && !windows) || (n && (!samples || !keep))) return UNIIR_EINVAL;
    int fb = 0;
    const int kept = gt_filter_samples(windows, nwin, samples, n, merge_ms, keep, &fb, nullptr);
    if (fallback) *fallback = fb;
    return kept;
}

static int gemm_impl(const uniir_gemm_desc* d, void* stream);
extern "C" int uniir_gemm(const uniir_gemm_desc* d, void* stream) {
    if (g_gt.stride > 0 && d && g_gt.filter && g_gt.only != (hipStream_t)stream) {          // another stream while one is measured
        if (g_gt.nf >= GT_FMAX) return gemm_impl(d, stream);
        const int i = g_gt.nf++;
        (void)hipEventRecord(g_gt.fev[2 * i], (hipStream_t)stream);
        const int rc = gemm_impl(d, stream);
        (void)hipEventRecord(g_gt.fev[2 * i + 1], (hipStream_t)stream);
        return rc;
    }
    const bool sample = g_gt.stride > 0 && d && (++g_gt.counter % g_gt.stride) == 0 && g_gt.n < GT_MAX;
    if (!sample) return gemm_impl(d, stream);
    const int i = g_gt.n;
    (void)hipEventRecord(g_gt.ev[2 * i], (hipStream_t)stream);
    const int rc = gemm_impl(d, stream);
    (void)hipEventRecord(g_gt.ev[2 * i + 1], (hipStream_t)stream);
    if (rc == UNIIR_OK) {
        g_gt.flop[i] = 2.0 * d->M * d->N * d->K;
        g_gt.n = i + 1;
    }
    return rc;
}

static int gemm_impl(const uniir_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->B || !d->C) return UNIIR_EINVAL;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->k_splits < 1) return UNIIR_EINVAL;
    if (d->epilogue < 0 || d->epilogue > UNIIR_EPI_ACT_ONLY) return UNIIR_EINVAL;
    if (d->k_splits > 1 && d->epilogue != UNIIR_EPI_ATOMIC_F32) return UNIIR_EINVAL;
    if (d->epilogue == UNIIR_EPI_ACT_ONLY && d->C2) return UNIIR_EINVAL;
    if (d->row_scale && d->epilogue != UNIIR_EPI_RESID_F32) return UNIIR_EINVAL;
    if (d->epilogue == UNIIR_EPI_BIAS_ACT && !d->C2) return UNIIR_EINVAL;
    if (d->epilogue == UNIIR_EPI_DACT && !d->aux) return UNIIR_EINVAL;
    // column sums of the result ride the fp32 copy-out passes only (DACT / RESID_F32 / F32 epilogues)
    if (d->colsum && d->epilogue != UNIIR_EPI_DACT && d->epilogue != UNIIR_EPI_RESID_F32 && d->epilogue != UNIIR_EPI_F32)
        return UNIIR_EUNSUPPORTED;
    // fp16 operands (the forward-only embedding towers): the 16-bit outputs of EPI_BF16 / BIAS_ACT / ACT_ONLY are fp16 as well; the
    // backward-side epilogues (activation gradient, the bf16 copy of a residual output) exist for bf16 only
    if (d->dtype == UNIIR_DT_F16 && (d->epilogue == UNIIR_EPI_DACT || (d->epilogue == UNIIR_EPI_RESID_F32 && d->C2))) return UNIIR_EUNSUPPORTED;
    if (d->N % 8) return UNIIR_ESHAPE;
    if (!d->a_tmaj && (d->K % 8)) return UNIIR_ESHAPE;
    if (!d->b_tmaj && (d->K % 8)) return UNIIR_ESHAPE;
    if (d->a_tmaj && (d->M % 8)) return UNIIR_ESHAPE;
    if ((d->lda % 8) || (d->ldb % 8) || (d->ldc % 4)) return UNIIR_EALIGN;
    if (((uintptr_t)d->A & 15) || ((uintptr_t)d->B & 15) || ((uintptr_t)d->C & 15)) return UNIIR_EALIGN;
    if (d->aux && ((d->ldaux % 4) || ((uintptr_t)d->aux & 7))) return UNIIR_EALIGN;
    if (d->bias && ((uintptr_t)d->bias & 15)) return UNIIR_EALIGN;
    // A weight gradient (C += A^T B, both operands K-major) whose reduction length is not a multiple of the 64-row K step -- the
    // packed text tower: K = the live rows of the batch -- would fall to the general 128-tile kernel for the WHOLE product (measured,
    // round 4: 188 us instead of ~45 us per text-tower weight gradient, 9 ms per train step).  The last K % 64 rows are a second,
    // tiny accumulating product instead; the multiple-of-64 part keeps the LDS-DMA kernel.  Every output element still receives its
    // additions in one fixed order (slab sum, then one add from the tail's only K step): deterministic.
    if (d->a_tmaj && d->b_tmaj && d->epilogue == UNIIR_EPI_ATOMIC_F32 && (d->K % 64) && d->K >= 64 * 16 && d->M >= 256 && d->N >= 128) {
        uniir_gemm_desc head = *d, tail = *d;
        const int km = d->K / 64 * 64;
        head.K = km;
        tail.K = d->K - km;
        tail.A = (const char*)d->A + (int64_t)km * d->lda * 2;
        tail.B = (const char*)d->B + (int64_t)km * d->ldb * 2;
        tail.k_splits = 1;
        tail.splitk_ws = nullptr;
        tail.splitk_ws_bytes = 0;
        const int rc = gemm_impl(&head, stream);
        return rc ? rc : gemm_impl(&tail, stream);
    }
    GemmKArgs a;
    a.A = (const unsigned short*)d->A;
    a.B = (const unsigned short*)d->B;
    a.C = d->C;
    a.C2 = d->C2;
    a.bias = d->bias;
    a.resid = d->resid;
    a.aux = (const unsigned short*)d->aux;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.ldaux = d->ldaux;
    a.epilogue = d->epilogue; a.act = d->act; a.k_splits = d->k_splits;
    a.skip_f = 0;
    a.row_scale = d->epilogue == UNIIR_EPI_RESID_F32 ? d->row_scale : nullptr;
    if (d->epilogue == UNIIR_EPI_ACT_ONLY) {      // the BIAS_ACT epilogue without its first output: C receives act(v + bias)
        a.epilogue = UNIIR_EPI_BIAS_ACT;
        a.C2 = d->C;
        a.skip_f = 1;
    }

    a.tiles_m = (d->M + GEMM_BM - 1) / GEMM_BM;
    a.tiles_n = (d->N + GEMM_BN - 1) / GEMM_BN;
    a.alpha = d->alpha;
    a.slab = nullptr;
    a.colsum = d->colsum;
    a.a_rowsum = nullptr;
    a.colsum_part = nullptr;
    a.rowsum_part = nullptr;
    a.asm_loop = 2;
    hipStream_t st = (hipStream_t)stream;
    if (d->k_splits > 1) {
        // never leave a split empty (an empty split would leave its slab unwritten)
        const int bk = GEMM_BK;  // 64: every kernel shape splits K on multiples of 64
        const int ksteps = (d->K + bk - 1) / bk;
        int splits = d->k_splits > ksteps ? ksteps : d->k_splits;
        const int per = (ksteps + splits - 1) / splits;
        splits = (ksteps + per - 1) / per;
        a.k_splits = splits;
        if (d->splitk_ws && !((uintptr_t)d->splitk_ws & 15) && (d->N % 4 == 0) &&
            d->splitk_ws_bytes >= (int64_t)splits * d->M * d->N * 4)
            a.slab = (float*)d->splitk_ws;
    }
    // row sums of A^T (the bias gradient of a weight-gradient GEMM): inside the transposed ping-pong kernel when the problem
    // runs it, otherwise as a separate pass over A
    bool rowsum_fused = false;
    if (d->a_rowsum) {
        if (!d->a_tmaj || d->dtype != UNIIR_DT_BF16) return UNIIR_EUNSUPPORTED;     // (the separate pass reads bf16 too)
        rowsum_fused = d->dtype == UNIIR_DT_BF16 && d->b_tmaj && pp_eligible(a, d->a_tmaj, d->b_tmaj);
        if (rowsum_fused) {
            a.a_rowsum = d->a_rowsum;
            // one partial per (split, column panel) block, reduced in block order below (atomics in arrival order without a scratch)
            // (the 256 x 256 kernel's own panel counts: launch_glds_rowsum)
            a.rowsum_part = reduce_scratch(st, (int64_t)a.k_splits * ((d->N + 255) / 256) * d->M * 4);
        }
    }
    int rc;
    // colsum and the DACT epilogue's second output act(aux) exist in the LDS-staged epilogue of the 256-tile kernel only; when the
    // problem runs the general 128-tile kernel they are produced HERE by separate passes (bf16 column sums of the stored result,
    // act(aux) elementwise), so that callers need not mirror gemm_shape()
    const bool staged = gemm_shape(a, d->a_tmaj, d->b_tmaj) == 1;
    bool colsum_after = false;
    if (!staged) {
        if (d->colsum) {
            if (d->epilogue != UNIIR_EPI_DACT) return UNIIR_EUNSUPPORTED;      // fp32 outputs: no separate column-sum pass exists
            a.colsum = nullptr;
            colsum_after = true;
        }
        if (d->epilogue == UNIIR_EPI_DACT && d->C2) {
            if (d->ldaux != d->N) return UNIIR_EUNSUPPORTED;
            rc = uniir_act_fwd(d->aux, d->C2, (int64_t)d->M * d->N, d->act, stream);
            if (rc) return rc;
            a.C2 = nullptr;
        }
    }
    if (staged && a.colsum && (d->N % 4 == 0))       // the 256-row panels' column partials, reduced in panel order below
        a.colsum_part = reduce_scratch(st, (int64_t)((d->M + 255) / 256) * d->N * 4);
    if (d->dtype == UNIIR_DT_BF16) rc = launch_gemm<ElemBF16>(a, d->a_tmaj, d->b_tmaj, st);
#ifndef UNIIR_EXP_BUILD
    else if (d->dtype == UNIIR_DT_F16) rc = launch_gemm<ElemF16>(a, d->a_tmaj, d->b_tmaj, st);
#endif
    else return UNIIR_EINVAL;
    if (rc) return rc;
    if (a.colsum_part) {
        rc = reduce_partials(a.colsum_part, (d->M + 255) / 256, d->N, d->N, d->colsum, nullptr, nullptr, 0, st);
        if (rc) return rc;
    }
    if (a.rowsum_part) {
        rc = reduce_partials(a.rowsum_part, a.k_splits * ((d->N + 255) / 256), d->M, d->M, d->a_rowsum, nullptr, nullptr, 0, st);
        if (rc) return rc;
    }
    if (colsum_after) {
        rc = uniir_colsum_bf16(d->C, d->ldc, d->colsum, d->M, d->N, stream);
        if (rc) return rc;
    }
    if (d->a_rowsum && !rowsum_fused) {
        rc = uniir_colsum_bf16(d->A, d->lda, d->a_rowsum, d->K, d->M, stream);
        if (rc) return rc;
    }
    if (a.slab) {
        const long mn = (long)d->M * d->N;
        long g = (mn / 4 + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, a.slab, a.k_splits, mn, d->N,
                           (float*)d->C, (long)d->ldc);
        HIP_LAUNCH_CHECK();
    }
    return UNIIR_OK;
}
