// In-batch InfoNCE forward / backward in fp32 (clip_sf.py:133-144 of the UniIR tree):
//   score = q @ all_p^T * exp(logit_scale); loss = CrossEntropy(score, rank*b + arange(b));
//   accuracy = mean(argmax(score, 1) == target).
// The similarity GEMMs run on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): every dot product is a k-ordered
// fmaf chain, so the logits are reproducible bit-for-bit by oracle/infonce_oracle.c.
#include "common.h"
#include "../../include/uniir_hip.h"

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

static int launch_sgemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C,
                        long ldc, int M, int N, int K, float alpha, const float* alpha_dev, hipStream_t st,
                        int accumulate = 0) {
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
