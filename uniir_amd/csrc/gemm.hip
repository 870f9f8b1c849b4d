// uniir_gemm: 16-bit MFMA GEMM with fused epilogues (see include/uniir_hip.h, gemm_core.h).
// Replaces the cuBLAS calls behind nn.Linear / nn.MultiheadAttention.in_proj / Conv2d-as-GEMM of the
// CLIP towers (openai/CLIP model.py as called from clip_sf.py:43-47) and their autograd backward.
#include "gemm_core.h"
#include "gemm_core256.h"
#include "../../include/uniir_hip.h"
#include <stdlib.h>

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
};

DEVINL float act_fwd(float x, int act) {
    if (act == UNIIR_ACT_QUICKGELU) return x / (1.0f + __expf(-1.702f * x));
    if (act == UNIIR_ACT_GELU_ERF) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    return fmaxf(x, 0.0f);
}
DEVINL float act_bwd(float x, int act) {
    if (act == UNIIR_ACT_QUICKGELU) {
        const float s = 1.0f / (1.0f + __expf(-1.702f * x));
        return s * (1.0f + 1.702f * x * (1.0f - s));
    }
    if (act == UNIIR_ACT_GELU_ERF) {
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
        return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
    }
    return x > 0.0f ? 1.0f : 0.0f;
}

template <int EPI, int MT, int NT>
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
                    u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<u32x2_t*>((unsigned short*)p.C + off) = o;
                } else if (EPI == UNIIR_EPI_BIAS_ACT) {
                    u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<u32x2_t*>((unsigned short*)p.C + off) = o;
                    const float g0 = act_fwd(__uint_as_float(o[0] << 16), p.act);
                    const float g1 = act_fwd(__uint_as_float(o[0] & 0xffff0000u), p.act);
                    const float g2 = act_fwd(__uint_as_float(o[1] << 16), p.act);
                    const float g3 = act_fwd(__uint_as_float(o[1] & 0xffff0000u), p.act);
                    u32x2_t o2 = {pack_bf16x2(g0, g1), pack_bf16x2(g2, g3)};
                    *reinterpret_cast<u32x2_t*>((unsigned short*)p.C2 + off) = o2;
                } else if (EPI == UNIIR_EPI_RESID_F32) {
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

template <int MT, int NT>
DEVINL void gemm_epilogue_dispatch(const GemmKArgs& p, const f32x4_t (&acc)[MT][NT], int m0, int n0, int wm, int wn) {
    switch (p.epilogue) {
        case UNIIR_EPI_BF16: gemm_epilogue<UNIIR_EPI_BF16, MT, NT>(p, acc, m0, n0, wm, wn); break;
        case UNIIR_EPI_BIAS_ACT: gemm_epilogue<UNIIR_EPI_BIAS_ACT, MT, NT>(p, acc, m0, n0, wm, wn); break;
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
    gemm_epilogue_dispatch<4, 4>(p, acc, m0, n0, (w >> 1) * 64, (w & 1) * 64);
}

// 256x256x64 tile, 8 waves, LDS-DMA staging (gemm_core256.h).  Used when K % 64 == 0.
template <typename Elem, bool A_TMAJ, bool B_TMAJ>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmKArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.tiles_m * p.tiles_n;
    const int split = id / tiles;
    id -= split * tiles;
    const int mt = id / p.tiles_n, nt = id - mt * p.tiles_n;
    const int m0 = mt * G256_BM, n0 = nt * G256_BN;
    int kbeg = 0, kend = p.K;
    if (p.k_splits > 1) {
        const int ksteps = p.K / G256_BK;
        const int per = (ksteps + p.k_splits - 1) / p.k_splits;
        kbeg = split * per * G256_BK;
        kend = min(p.K, (split + 1) * per * G256_BK);
        if (kbeg >= kend) return;
    }
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    g256_mainloop<Elem, A_TMAJ, B_TMAJ>(p.A, p.lda, p.M, p.B, p.ldb, p.N, m0, n0, kbeg, kend, lds, acc);
    const int w = threadIdx.x >> 6;
    if (p.slab) {
        GemmKArgs q = p;
        q.C = p.slab + (long)split * p.M * p.N;
        q.ldc = p.N;
        q.bias = nullptr;
        gemm_epilogue<UNIIR_EPI_F32, 8, 4>(q, acc, m0, n0, (w >> 2) * 128, (w & 3) * 64);
        return;
    }
    gemm_epilogue_dispatch<8, 4>(p, acc, m0, n0, (w >> 2) * 128, (w & 3) * 64);
}

template <typename Elem>
static int launch_gemm256(GemmKArgs a, int a_tmaj, int b_tmaj, hipStream_t st) {
    a.tiles_m = (a.M + G256_BM - 1) / G256_BM;
    a.tiles_n = (a.N + G256_BN - 1) / G256_BN;
    const int grid = a.tiles_m * a.tiles_n * a.k_splits;
    dim3 g(grid), b(512);
    const size_t sm = G256_LDS_BYTES;
#define LAUNCH256(AT, BT)                                                                         \
    do {                                                                                          \
        static bool attr_set = false;                                                             \
        if (!attr_set) {                                                                          \
            (void)hipFuncSetAttribute((const void*)gemm256_kernel<Elem, AT, BT>,                  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);       \
            attr_set = true;                                                                      \
        }                                                                                         \
        hipLaunchKernelGGL((gemm256_kernel<Elem, AT, BT>), g, b, sm, st, a);                      \
    } while (0)
    if (!a_tmaj && !b_tmaj) LAUNCH256(false, false);
    else if (!a_tmaj && b_tmaj) LAUNCH256(false, true);
    else if (a_tmaj && !b_tmaj) LAUNCH256(true, false);
    else LAUNCH256(true, true);
#undef LAUNCH256
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

static bool use_256(const GemmKArgs& a, int a_tmaj, int b_tmaj) {
    static const char* force = getenv("UNIIR_GEMM_TILE");
    if (force && force[0] == '1') return false;             // UNIIR_GEMM_TILE=128 forces the general kernel
    if (a.K % G256_BK) return false;
    if (a.M < 256 || a.N < 128) return false;               // small problems: the 128-tile kernel fills the chip better
    if (a_tmaj && a.M < 8) return false;
    if (b_tmaj && a.N < 8) return false;
    return true;
}

template <typename Elem>
static int launch_gemm(const GemmKArgs& a, int a_tmaj, int b_tmaj, hipStream_t st) {
    if (use_256(a, a_tmaj, b_tmaj)) return launch_gemm256<Elem>(a, a_tmaj, b_tmaj, st);
    const int grid = a.tiles_m * a.tiles_n * a.k_splits;
    dim3 g(grid), b(256);
    const size_t sm = GEMM_LDS_BYTES;
#define LAUNCH(AT, BT)                                                                            \
    do {                                                                                          \
        static bool attr_set = false;                                                             \
        if (!attr_set) {                                                                          \
            (void)hipFuncSetAttribute((const void*)gemm_kernel<Elem, AT, BT>,                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);             \
            attr_set = true;                                                                      \
        }                                                                                         \
        hipLaunchKernelGGL((gemm_kernel<Elem, AT, BT>), g, b, sm, st, a);                         \
    } while (0)
    if (!a_tmaj && !b_tmaj) LAUNCH(false, false);
    else if (!a_tmaj && b_tmaj) LAUNCH(false, true);
    else if (a_tmaj && !b_tmaj) LAUNCH(true, false);
    else LAUNCH(true, true);
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

extern "C" int uniir_gemm(const uniir_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->B || !d->C) return UNIIR_EINVAL;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->k_splits < 1) return UNIIR_EINVAL;
    if (d->epilogue < 0 || d->epilogue > UNIIR_EPI_ATOMIC_F32) return UNIIR_EINVAL;
    if (d->k_splits > 1 && d->epilogue != UNIIR_EPI_ATOMIC_F32) return UNIIR_EINVAL;
    if (d->epilogue == UNIIR_EPI_BIAS_ACT && !d->C2) return UNIIR_EINVAL;
    if (d->epilogue == UNIIR_EPI_DACT && !d->aux) return UNIIR_EINVAL;
    if (d->N % 8) return UNIIR_ESHAPE;
    if (!d->a_tmaj && (d->K % 8)) return UNIIR_ESHAPE;
    if (!d->b_tmaj && (d->K % 8)) return UNIIR_ESHAPE;
    if (d->a_tmaj && (d->M % 8)) return UNIIR_ESHAPE;
    if ((d->lda % 8) || (d->ldb % 8) || (d->ldc % 4)) return UNIIR_EALIGN;
    if (((uintptr_t)d->A & 15) || ((uintptr_t)d->B & 15) || ((uintptr_t)d->C & 15)) return UNIIR_EALIGN;
    if (d->aux && ((d->ldaux % 4) || ((uintptr_t)d->aux & 7))) return UNIIR_EALIGN;
    if (d->bias && ((uintptr_t)d->bias & 15)) return UNIIR_EALIGN;
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
    a.tiles_m = (d->M + GEMM_BM - 1) / GEMM_BM;
    a.tiles_n = (d->N + GEMM_BN - 1) / GEMM_BN;
    a.alpha = d->alpha;
    a.slab = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (d->k_splits > 1) {
        // never leave a split empty (an empty split would leave its slab unwritten)
        const int bk = use_256(a, d->a_tmaj, d->b_tmaj) ? G256_BK : GEMM_BK;
        const int ksteps = (d->K + bk - 1) / bk;
        int splits = d->k_splits > ksteps ? ksteps : d->k_splits;
        const int per = (ksteps + splits - 1) / splits;
        splits = (ksteps + per - 1) / per;
        a.k_splits = splits;
        if (d->splitk_ws && !((uintptr_t)d->splitk_ws & 15) && (d->N % 4 == 0) &&
            d->splitk_ws_bytes >= (int64_t)splits * d->M * d->N * 4)
            a.slab = (float*)d->splitk_ws;
    }
    int rc;
    if (d->dtype == UNIIR_DT_BF16) rc = launch_gemm<ElemBF16>(a, d->a_tmaj, d->b_tmaj, st);
    else if (d->dtype == UNIIR_DT_F16) rc = launch_gemm<ElemF16>(a, d->a_tmaj, d->b_tmaj, st);
    else return UNIIR_EINVAL;
    if (rc) return rc;
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
