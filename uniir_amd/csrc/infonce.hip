// In-batch InfoNCE forward / backward in fp32 (clip_sf.py:133-144 of the UniIR tree):
//   score = q @ all_p^T * exp(logit_scale); loss = CrossEntropy(score, rank*b + arange(b));
//   accuracy = mean(argmax(score, 1) == target).
// The similarity GEMMs run on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): every dot product is a k-ordered
// fmaf chain, so the logits are reproducible bit-for-bit by oracle/infonce_oracle.c.
#include "common.h"
#include "../../include/uniir_hip.h"
#include <stdlib.h>

#define SG_BM 64
#define SG_BN 64
#define SG_BK 16
#define SG_PITCH 68  // floats per k-row in LDS (64 + 4 pad)

// C[m][n] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]; alpha = alpha_host * (alpha_dev ? *alpha_dev : 1)
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, long sam, long sak,
                                                    const float* __restrict__ B, long sbk, long sbn,
                                                    float* __restrict__ C, long ldc, int M, int N, int K,
                                                    float alpha_host, const float* __restrict__ alpha_dev,
                                                    int accumulate) {
    __shared__ float As[SG_BK * SG_PITCH];
    __shared__ float Bs[SG_BK * SG_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.y * SG_BM, n0 = blockIdx.x * SG_BN;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool a_kfast = (sak == 1), b_kfast = (sbk == 1);
    for (int k0 = 0; k0 < K; k0 += SG_BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            int kk, mm;
            if (a_kfast) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }
            const int gm = m0 + mm, gk = k0 + kk;
            As[kk * SG_PITCH + mm] = (gm < M && gk < K) ? A[(long)gm * sam + (long)gk * sak] : 0.f;
            int kb, nn;
            if (b_kfast) { kb = e & 15; nn = e >> 4; } else { nn = e & 63; kb = e >> 6; }
            const int gn = n0 + nn, gk2 = k0 + kb;
            Bs[kb * SG_PITCH + nn] = (gn < N && gk2 < K) ? B[(long)gk2 * sbk + (long)gn * sbn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < SG_BK / 4; ++ks) {
            const int kr = ks * 4 + (lane >> 4);
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[kr * SG_PITCH + wm + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bs[kr * SG_PITCH + wn + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const float alpha = alpha_host * (alpha_dev ? *alpha_dev : 1.0f);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + 4 * (lane >> 4) + r;
                const int n = n0 + wn + j * 16 + (lane & 15);
                if (m < M && n < N) {
                    float* c = C + (long)m * ldc + n;
                    *c = accumulate ? *c + acc[i][j][r] * alpha : acc[i][j][r] * alpha;
                }
            }
}

// The same GEMM for the big shapes (BLIP's [256 x 768] x [768 x 57 344] queue logits and their backward; the InfoNCE logits at
// global batch 4096): 128 x 128 tile, 4 waves of 64 x 64 (4 x 4 MFMA tiles), K step 16, 16-byte global loads along whichever
// operand dimension is contiguous, the next K step's operands fetched into registers while the current one computes.  Every
// output element is still ONE accumulator fed by v_mfma_f32_16x16x4_f32 in ascending k: bit-identical to sgemm_kernel and to
// the CPU oracle's fmaf chain.  The 64 x 64 kernel above measured 15.7 TFLOP/s on the queue logits (10 % of the fp32 MFMA peak:
// scalar loads, one tile per wave).
#define SGL_BM 128
#define SGL_BN 128
#define SGL_PITCH 132     // floats per k-row in LDS (128 + 4 pad)
// loads one operand tile [128 mn][16 k] as 2 float4 per thread; CONTIG_K: k is the contiguous dimension, else mn is
template <bool CONTIG_K>
DEVINL void sgl_fetch(const float* __restrict__ P, long s_mn, long s_k, int mn0, int mn_total, int k0, int tid, f32x4_t (&v)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;                      // 512 float4 per tile
        if (CONTIG_K) {
            const int mn = e >> 2, k4 = (e & 3) * 4;
            const int g = min(mn0 + mn, mn_total - 1);    // clamped rows: their products are never stored
            v[i] = *reinterpret_cast<const f32x4_t*>(P + (long)g * s_mn + (long)(k0 + k4));
        } else {
            const int k = e >> 5, mn4 = (e & 31) * 4;
            const int g = min(mn0 + mn4, mn_total - 4);   // mn_total % 4 == 0 on this path
            v[i] = *reinterpret_cast<const f32x4_t*>(P + (long)(k0 + k) * s_k + (long)g);
        }
    }
}
template <bool CONTIG_K>
DEVINL void sgl_park(float* lds, int tid, const f32x4_t (&v)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;
        if (CONTIG_K) {
            const int mn = e >> 2, k4 = (e & 3) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[(k4 + r) * SGL_PITCH + mn] = v[i][r];
        } else {
            const int k = e >> 5, mn4 = (e & 31) * 4;
            *reinterpret_cast<f32x4_t*>(lds + k * SGL_PITCH + mn4) = v[i];
        }
    }
}
template <bool A_CK, bool B_CK>
__global__ __launch_bounds__(256) void sgemm128_kernel(const float* __restrict__ A, long sam, long sak,
                                                       const float* __restrict__ B, long sbk, long sbn,
                                                       float* __restrict__ C, long ldc, int M, int N, int K, float alpha_host,
                                                       const float* __restrict__ alpha_dev, int accumulate,
                                                       float* __restrict__ slab = nullptr, int kslice = 0) {
    __shared__ __attribute__((aligned(16))) float As[SG_BK * SGL_PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[SG_BK * SGL_PITCH];
    if (slab) {       // split-K: slice blockIdx.z of the reduction -> its own [M][N] slab, un-scaled (sgemm_splitk_reduce_kernel adds them)
        const int kbeg = blockIdx.z * kslice;
        A += (long)kbeg * sak;
        B += (long)kbeg * sbk;
        K = min(K - kbeg, kslice);
        C = slab + (long)blockIdx.z * M * N;
        ldc = N;
        accumulate = 0;
        alpha_host = 1.0f;
        alpha_dev = nullptr;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.y * SGL_BM, n0 = blockIdx.x * SGL_BN;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t va[2], vb[2];
    sgl_fetch<A_CK>(A, sam, sak, m0, M, 0, tid, va);
    sgl_fetch<B_CK>(B, sbn, sbk, n0, N, 0, tid, vb);
    for (int k0 = 0; k0 < K; k0 += SG_BK) {
        sgl_park<A_CK>(As, tid, va);
        sgl_park<B_CK>(Bs, tid, vb);
        __syncthreads();
        if (k0 + SG_BK < K) {
            sgl_fetch<A_CK>(A, sam, sak, m0, M, k0 + SG_BK, tid, va);
            sgl_fetch<B_CK>(B, sbn, sbk, n0, N, k0 + SG_BK, tid, vb);
        }
#pragma unroll
        for (int ks = 0; ks < SG_BK / 4; ++ks) {
            const int kr = ks * 4 + (lane >> 4);
            float af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = As[kr * SGL_PITCH + wm + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = Bs[kr * SGL_PITCH + wn + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const float alpha = alpha_host * (alpha_dev ? *alpha_dev : 1.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + 4 * (lane >> 4) + r;
                const int n = n0 + wn + j * 16 + (lane & 15);
                if (m < M && n < N) {
                    float* c = C + (long)m * ldc + n;
                    *c = accumulate ? *c + acc[i][j][r] * alpha : acc[i][j][r] * alpha;
                }
            }
}

static int launch_sgemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C,
                        long ldc, int M, int N, int K, float alpha, const float* alpha_dev, hipStream_t st,
                        int accumulate = 0) {
    // the 128-tile kernel: both operands with one unit stride, 16-byte aligned vectors, whole K steps, enough tiles to matter
    const bool a_ck = sak == 1, a_cm = sam == 1, b_ck = sbk == 1, b_cn = sbn == 1;
    auto vec_ok = [](const float* p, long ld, bool contig_k, int ext) {
        return (((uintptr_t)p) & 15) == 0 && ld % 4 == 0 && (contig_k || ext % 4 == 0);
    };
    const bool big = K % SG_BK == 0 && M >= 64 && N >= 64 && (long)M * N >= 128L * 128 * 8 &&
                     (a_ck || a_cm) && (b_ck || b_cn) && vec_ok(A, a_ck ? sam : sak, a_ck, M) && vec_ok(B, b_ck ? sbn : sbk, b_ck, N);
    if (big) {
        dim3 grid((N + SGL_BN - 1) / SGL_BN, (M + SGL_BM - 1) / SGL_BM);
#define SGL(AK, BK_) hipLaunchKernelGGL((sgemm128_kernel<AK, BK_>), grid, dim3(256), 0, st, A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha, alpha_dev, accumulate)
        if (a_ck && b_ck) SGL(true, true);
        else if (a_ck) SGL(true, false);
        else if (b_ck) SGL(false, true);
        else SGL(false, false);
#undef SGL
        HIP_LAUNCH_CHECK();
        return UNIIR_OK;
    }
    dim3 grid((N + SG_BN - 1) / SG_BN, (M + SG_BM - 1) / SG_BM);
    hipLaunchKernelGGL(sgemm_kernel, grid, dim3(256), 0, st, A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha,
                       alpha_dev, accumulate);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

extern "C" int uniir_sgemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                           float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, float alpha, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return UNIIR_EINVAL;
    return launch_sgemm(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha, nullptr, (hipStream_t)stream);
}
extern "C" int uniir_sgemm_acc(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                               float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, float alpha, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return UNIIR_EINVAL;
    return launch_sgemm(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha, nullptr, (hipStream_t)stream, 1);
}

// ------------------------------------------------------------------------------------------------------------
// Long-reduction form (BLIP: d feat[b][E] = dsim[b][K] x queue[K][E] with K = 57 344 and only 2 x 6 output tiles -- 12 workgroups on
// 256 CUs, 4.3 ms per call): deterministic split-K.  The reduction is cut into `splits` slices of whole K steps, slice s goes to
// slab s of a caller-owned workspace, and one pass adds the slabs IN SLICE ORDER (so the result is reproducible run to run; it
// differs from the un-split k-ordered chain in the last bits, which is why this is a separate entry point and the logits -- whose
// bit-exactness against the oracle is tested -- never take it).
__global__ __launch_bounds__(256) void sgemm_splitk_reduce_kernel(const float* __restrict__ slab, int splits, long mn, int N,
                                                                  float* __restrict__ C, long ldc, float alpha, int accumulate) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= mn) return;
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(slab + i);
    for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4_t*>(slab + (long)s * mn + i);
    const long m = i / N, n = i - m * N;              // N % 4 == 0: the four elements share a row
    float* c = C + m * ldc + n;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = accumulate ? c[r] + v[r] * alpha : v[r] * alpha;
}
static int sgemm_splits(int M, int N, int K) {
    const long tiles = (long)((M + SGL_BM - 1) / SGL_BM) * ((N + SGL_BN - 1) / SGL_BN);
    long s = (1024 + tiles - 1) / tiles;               // ~4 workgroups per CU
    const long max_s = K / (SG_BK * 8);                // >= 8 K steps per slice
    if (s > max_s) s = max_s;
    return s < 1 ? 1 : (int)s;
}
extern "C" int64_t uniir_sgemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (int64_t)sgemm_splits(M, N, K) * M * N * 4;
}
extern "C" int uniir_sgemm_splitk(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C,
                                  int64_t ldc, int32_t M, int32_t N, int32_t K, float alpha, int32_t accumulate, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    if (!A || !B || !C || !workspace || M <= 0 || N <= 0 || K <= 0) return UNIIR_EINVAL;
    const bool a_ck = sak == 1, a_cm = sam == 1, b_ck = sbk == 1, b_cn = sbn == 1;
    auto vec_ok = [](const float* p, long ld, bool contig_k, int ext) {
        return (((uintptr_t)p) & 15) == 0 && ld % 4 == 0 && (contig_k || ext % 4 == 0);
    };
    const int splits = sgemm_splits(M, N, K);
    // shapes the 128-tile kernel does not take, or nothing to split: the plain kernel
    if (splits < 2 || K % SG_BK || N % 4 || !(a_ck || a_cm) || !(b_ck || b_cn) || !vec_ok(A, a_ck ? sam : sak, a_ck, M) ||
        !vec_ok(B, b_ck ? sbn : sbk, b_ck, N))
        return launch_sgemm(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha, nullptr, (hipStream_t)stream, accumulate);
    if (((uintptr_t)workspace & 15) || workspace_bytes < uniir_sgemm_splitk_workspace_bytes(M, N, K)) return UNIIR_EINVAL;
    const int ksteps = K / SG_BK;
    const int kslice = ((ksteps + splits - 1) / splits) * SG_BK;
    const int used = (K + kslice - 1) / kslice;        // slices that hold at least one K step
    hipStream_t st = (hipStream_t)stream;
    float* slab = (float*)workspace;
    dim3 grid((N + SGL_BN - 1) / SGL_BN, (M + SGL_BM - 1) / SGL_BM, used);
#define SGL(AK, BK_) hipLaunchKernelGGL((sgemm128_kernel<AK, BK_>), grid, dim3(256), 0, st, A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, 1.0f, nullptr, 0, slab, kslice)
    if (a_ck && b_ck) SGL(true, true);
    else if (a_ck) SGL(true, false);
    else if (b_ck) SGL(false, true);
    else SGL(false, false);
#undef SGL
    const long mn = (long)M * N;
    hipLaunchKernelGGL(sgemm_splitk_reduce_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, st, slab, used, mn, N, C,
                       (long)ldc, alpha, accumulate);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// one block per row: lse, first-argmax, per-row loss and hit.  stats = [lse(b) | loss_i(b) | hit_i(b)]
__global__ __launch_bounds__(256) void nce_rowstats_kernel(const float* __restrict__ score, int b, int B,
                                                           int toff, float* __restrict__ stats) {
    __shared__ float smax[4];
    __shared__ int sarg[4];
    __shared__ float ssum[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* row = score + (long)i * B;
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int j = tid; j < B; j += 256) {
        const float v = row[j];
        if (v > mx) { mx = v; am = j; }  // strict: keeps the first index within this thread's stride
    }
    // wave arg-max with first-index tie break
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (ov > mx || (ov == mx && oa < am)) { mx = ov; am = oa; }
    }
    if (lane == 0) { smax[w] = mx; sarg[w] = am; }
    __syncthreads();
    mx = smax[0]; am = sarg[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (smax[k] > mx || (smax[k] == mx && sarg[k] < am)) { mx = smax[k]; am = sarg[k]; }
    float s = 0.f;
    for (int j = tid; j < B; j += 256) s += expf(row[j] - mx);
    s = wave_sum(s);
    if (lane == 0) ssum[w] = s;
    __syncthreads();
    if (tid == 0) {
        const float tot = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        const float lse = mx + logf(tot);
        const int t = toff + i;
        stats[i] = lse;
        stats[b + i] = lse - row[t];
        stats[2 * b + i] = (am == t) ? 1.0f : 0.0f;
    }
}
// deterministic single-block mean of n values: out[0] = mean(v[0:n]) * mul
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int n, float mul,
                                                   float* __restrict__ out) {
    __shared__ float part[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0] * mul / (float)n;
}

extern "C" int uniir_infonce_fwd(const float* q, const float* all_p, const float* scale, int32_t b, int32_t B,
                                 int32_t dim, int32_t target_offset, float* score, float* row_lse, float* loss,
                                 float* acc, void* stream) {
    if (!q || !all_p || !scale || !score || !row_lse || !loss || !acc) return UNIIR_EINVAL;
    if (b <= 0 || B <= 0 || dim <= 0 || target_offset < 0 || target_offset + b > B) return UNIIR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_sgemm(q, dim, 1, all_p, 1, dim, score, B, b, B, dim, 1.0f, scale, st);
    if (rc) return rc;
    hipLaunchKernelGGL(nce_rowstats_kernel, dim3(b), dim3(256), 0, st, score, b, B, target_offset, row_lse);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_lse + b, b, 1.0f, loss);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_lse + 2 * b, b, 1.0f, acc);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// G[i][j] = (exp(score - lse_i) - [j == t_i]) * dloss / b ;  rowpart[i] = sum_j G[i][j] * score[i][j]
__global__ __launch_bounds__(256) void nce_grad_kernel(const float* __restrict__ score,
                                                       const float* __restrict__ lse,
                                                       const float* __restrict__ dloss, int b, int B, int toff,
                                                       float* __restrict__ G, float* __restrict__ rowpart) {
    __shared__ float ssum[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float gmul = dloss[0] / (float)b, l = lse[i];
    const int t = toff + i;
    float s = 0.f;
    for (int j = tid; j < B; j += 256) {
        const float sc = score[(long)i * B + j];
        const float g = (expf(sc - l) - (j == t ? 1.0f : 0.0f)) * gmul;
        G[(long)i * B + j] = g;
        s += g * sc;
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) ssum[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) rowpart[i] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
}
__global__ __launch_bounds__(256) void nce_dscale_kernel(const float* __restrict__ rowpart, int b,
                                                         const float* __restrict__ scale,
                                                         float* __restrict__ dscale) {
    __shared__ float part[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < b; i += 256) s += rowpart[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dscale[0] = part[0] / scale[0];
}

extern "C" int uniir_infonce_bwd(const float* q, const float* all_p, const float* scale, const float* score,
                                 const float* row_lse, const float* dloss, int32_t b, int32_t B, int32_t dim,
                                 int32_t target_offset, float* gbuf, float* dq, float* d_all_p, float* dscale,
                                 void* stream) {
    if (!q || !all_p || !scale || !score || !row_lse || !dloss || !gbuf || !dq || !d_all_p || !dscale)
        return UNIIR_EINVAL;
    if (b <= 0 || B <= 0 || dim <= 0 || target_offset < 0 || target_offset + b > B) return UNIIR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    float* rowpart = gbuf + (long)b * B;  // gbuf holds b*B + b floats
    hipLaunchKernelGGL(nce_grad_kernel, dim3(b), dim3(256), 0, st, score, row_lse, dloss, b, B, target_offset,
                       gbuf, rowpart);
    hipLaunchKernelGGL(nce_dscale_kernel, dim3(1), dim3(256), 0, st, rowpart, b, scale, dscale);
    // dq[b][dim] = scale * G[b][B] . all_p[B][dim]
    int rc = launch_sgemm(gbuf, B, 1, all_p, dim, 1, dq, dim, b, dim, B, 1.0f, scale, st);
    if (rc) return rc;
    // d_all_p[B][dim] = scale * G^T[B][b] . q[b][dim]
    rc = launch_sgemm(gbuf, 1, B, q, dim, 1, d_all_p, dim, B, dim, b, 1.0f, scale, st);
    if (rc) return rc;
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// -------------------------------------------------------------------------------------------------------------------
// Hard-negative branch of the CLIP_SF loss (clip_sf.py:105-131): for query i the logit row is
//   [ <q_i,p_i>, <q_i,n_{i,0}> .. <q_i,n_{i,N-1}>, then I "in-batch negatives" ] * scale,  I = min(b-1, in_batch_neg_num).
// The reference builds the in-batch negatives as p.unsqueeze(1).expand(-1, bs, -1)[eye == 0].reshape(bs, bs-1, -1)[:, :I]:
// element [i, j] of the expanded tensor is p_i (the expansion runs along dim 1), so every one of the I entries of row i
// is the query's OWN positive p_i.  That is reproduced as is (parity with the reference, golden G3 case n2);
// loss = mean_i -log_softmax(row_i)[0]; accuracy = mean_i [first arg-max of row_i == 0].  All fp32, one block per query.
// -------------------------------------------------------------------------------------------------------------------
DEVINL const float* hn_vec(const float* p, const float* n, int i, int c, int N, int dim) {
    if (c == 0) return p + (long)i * dim;
    if (c <= N) return n + ((long)i * N + (c - 1)) * dim;
    return p + (long)i * dim;      // see the note above: the reference's in-batch entries are p_i itself
}

__global__ __launch_bounds__(256) void hardneg_fwd_kernel(const float* __restrict__ q, const float* __restrict__ p,
                                                          const float* __restrict__ n, const float* __restrict__ scale,
                                                          int N, int I, int dim, float* __restrict__ logits,
                                                          float* __restrict__ row_lse, float* __restrict__ row_loss,
                                                          float* __restrict__ row_hit) {
    __shared__ float red[4];
    __shared__ int redi[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int C = 1 + N + I;
    const float s = *scale;
    const float* qi = q + (long)i * dim;
    float* lrow = logits + (long)i * C;
    for (int c = w; c < C; c += 4) {               // one wave per logit
        const float* v = hn_vec(p, n, i, c, N, dim);
        float a = 0.f;
        for (int e = lane; e < dim; e += 64) a = fmaf(qi[e], v[e], a);
        a = wave_sum(a);
        if (lane == 0) lrow[c] = a * s;
    }
    __syncthreads();
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int c = tid; c < C; c += 256) {
        const float v = lrow[c];
        if (v > mx) { mx = v; am = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (ov > mx || (ov == mx && oa < am)) { mx = ov; am = oa; }
    }
    if (lane == 0) { red[w] = mx; redi[w] = am; }
    __syncthreads();
    mx = red[0]; am = redi[0];
    for (int k = 1; k < 4; ++k) if (red[k] > mx || (red[k] == mx && redi[k] < am)) { mx = red[k]; am = redi[k]; }
    __syncthreads();
    float se = 0.f;
    for (int c = tid; c < C; c += 256) se += expf(lrow[c] - mx);
    se = wave_sum(se);
    if (lane == 0) red[w] = se;
    __syncthreads();
    if (tid == 0) {
        const float lse = mx + logf((red[0] + red[1]) + (red[2] + red[3]));
        row_lse[i] = lse;
        row_loss[i] = lse - lrow[0];
        row_hit[i] = am == 0 ? 1.0f : 0.0f;
    }
}

// dq [b][dim] (written), dn [b*N][dim] (written), dp [b][dim] and dscale (ACCUMULATED with atomics: zero them first)
__global__ __launch_bounds__(256) void hardneg_bwd_kernel(const float* __restrict__ q, const float* __restrict__ p,
                                                          const float* __restrict__ n, const float* __restrict__ scale,
                                                          const float* __restrict__ logits, const float* __restrict__ row_lse,
                                                          const float* __restrict__ dloss, int b, int N, int I, int dim,
                                                          float* __restrict__ dq, float* __restrict__ dp,
                                                          float* __restrict__ dn, float* __restrict__ dscale) {
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int C = 1 + N + I;
    const float s = *scale, g = *dloss / (float)b, lse = row_lse[i];
    const float* qi = q + (long)i * dim;
    const float* lrow = logits + (long)i * C;
    float ds = 0.f;
    for (int c = tid; c < C; c += 256) ds += (expf(lrow[c] - lse) - (c == 0 ? 1.f : 0.f)) * g * lrow[c];
    ds = wave_sum(ds);
    if (lane == 0) red[w] = ds;
    __syncthreads();
    if (tid == 0) atomicAdd(dscale, ((red[0] + red[1]) + (red[2] + red[3])) / s);
    for (int e = tid; e < dim; e += 256) {
        const float qe = qi[e];
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            const float dl = (expf(lrow[c] - lse) - (c == 0 ? 1.f : 0.f)) * g * s;   // d loss / d <q, v_c>
            const float* v = hn_vec(p, n, i, c, N, dim);
            acc = fmaf(dl, v[e], acc);
            if (c >= 1 && c <= N) dn[((long)i * N + (c - 1)) * dim + e] = dl * qe;
            else atomicAdd(dp + (v - p) + e, dl * qe);
        }
        dq[(long)i * dim + e] = acc;
    }
}

extern "C" int uniir_hardneg_fwd(const float* q, const float* p, const float* n, const float* scale, int32_t b, int32_t N,
                                 int32_t I, int32_t dim, float* logits, float* row_lse, float* row_loss, float* row_hit,
                                 void* stream) {
    if (!q || !p || (N > 0 && !n) || !scale || !logits || !row_lse || !row_loss || !row_hit) return UNIIR_EINVAL;
    if (b <= 0 || N < 0 || I < 0 || I > b - 1 || dim <= 0) return UNIIR_EINVAL;
    hipLaunchKernelGGL(hardneg_fwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, q, p, n, scale, N, I, dim, logits,
                       row_lse, row_loss, row_hit);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

extern "C" int uniir_hardneg_bwd(const float* q, const float* p, const float* n, const float* scale, const float* logits,
                                 const float* row_lse, const float* dloss, int32_t b, int32_t N, int32_t I, int32_t dim,
                                 float* dq, float* dp, float* dn, float* dscale, void* stream) {
    if (!q || !p || (N > 0 && (!n || !dn)) || !scale || !logits || !row_lse || !dloss || !dq || !dp || !dscale)
        return UNIIR_EINVAL;
    if (b <= 0 || N < 0 || I < 0 || I > b - 1 || dim <= 0) return UNIIR_EINVAL;
    hipLaunchKernelGGL(hardneg_bwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, q, p, n, scale, logits, row_lse, dloss,
                       b, N, I, dim, dq, dp, dn, dscale);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
