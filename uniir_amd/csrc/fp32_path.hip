// fp32 forward path of the encoders ("model.float()" semantics of the reference, clip_sf.py:25-26 loads fp32 weights and
// mbeir_embedder.py runs the forward under autocast only when use_fp16 is set): every GEMM is the exact-fp32 MFMA
// uniir_sgemm (v_mfma_f32_16x16x4_f32, infonce.hip), LayerNorm is the fp32 kernel of norm.hip, and the three pieces below
// supply what the 16-bit kernels fuse into their epilogues.  157 TFLOP/s peak instead of 2.5 PFLOP/s: this path exists for
// reference-precision embedding extraction and for the north-star parity bar (fp32 logits within 1e-3, identical top-k ids
// against the fp32 reference), not for throughput.  Forward only.
#include "common.h"
#include "../../include/uniir_hip.h"

DEVINL float act_f32(float x, int act) {
    if (act == UNIIR_ACT_QUICKGELU) return x / (1.0f + expf(-1.702f * x));          // x * sigmoid(1.702 x)
    if (act == UNIIR_ACT_GELU_ERF) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    if (act == UNIIR_ACT_RELU) return fmaxf(x, 0.0f);
    return x;
}

// y[r][c] = (resid ? resid[r][c] : 0) + act(y[r][c] + bias[c])      (act < 0: identity)
__global__ __launch_bounds__(256) void bias_act_f32_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                           const float* __restrict__ resid, long rows, int cols, int act) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cols);
        float v = y[i] + (bias ? bias[c] : 0.f);
        v = act_f32(v, act);
        if (resid) v += resid[i];
        y[i] = v;
    }
}

extern "C" int uniir_bias_act_f32(float* y, const float* bias, const float* resid, int64_t rows, int32_t cols, int32_t act,
                                  void* stream) {
    if (!y || rows < 0 || cols <= 0) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    long blocks = (rows * cols + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(bias_act_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, bias, resid,
                       (long)rows, cols, act);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// x[n][T][w] = (t == 0 ? class_emb : patch_out[n][t-1]) + pos[t], patch_out fp32 (uniir_vit_assemble takes bf16)
__global__ __launch_bounds__(256) void vit_assemble_f32_kernel(const float* __restrict__ po, const float* __restrict__ cls,
                                                               const float* __restrict__ pos, float* __restrict__ x, int n,
                                                               int T, int w) {
    const long total = (long)n * T * w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % w);
        const long row = i / w;
        const int t = (int)(row % T);
        const long im = row / T;
        const float v = t == 0 ? cls[c] : po[(im * (T - 1) + (t - 1)) * (long)w + c];
        x[i] = v + pos[(long)t * w + c];
    }
}

extern "C" int uniir_vit_assemble_f32(const float* patch_out, const float* class_emb, const float* pos_emb, float* x,
                                      int32_t n, int32_t tokens, int32_t width, void* stream) {
    if (!patch_out || !class_emb || !pos_emb || !x || n < 0 || tokens < 2 || width <= 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    long blocks = ((long)n * tokens * width + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(vit_assemble_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, patch_out,
                       class_emb, pos_emb, x, n, tokens, width);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// fp32 attention, head_dim 64: one workgroup per (item, head), one thread per query row (256 rows per pass), keys and
// values streamed through LDS 32 rows at a time, online softmax (running max / sum / output in registers), all fp32 VALU.
#define A32_KB 32
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ q, long q_ld,
                                                            const float* __restrict__ k, const float* __restrict__ v,
                                                            long kv_ld, float* __restrict__ out, long out_ld,
                                                            const int* __restrict__ klen, int Tq, int Tk, int H, int causal,
                                                            float scale) {
    __shared__ float ks[A32_KB][64];
    __shared__ float vs[A32_KB][64];
    const int m = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
    const int kvalid = klen ? min(Tk, klen[m]) : Tk;
    const float* kbase = k + (long)m * Tk * kv_ld + h * 64;
    const float* vbase = v + (long)m * Tk * kv_ld + h * 64;
    for (int q0 = 0; q0 < Tq; q0 += 256) {
        const int qi = q0 + tid;
        const bool live = qi < Tq;
        float qr[64], acc[64];
#pragma unroll
        for (int d = 0; d < 64; ++d) {
            qr[d] = live ? q[((long)m * Tq + qi) * q_ld + h * 64 + d] * scale : 0.f;
            acc[d] = 0.f;
        }
        float mx = -INFINITY, l = 0.f;
        for (int k0 = 0; k0 < kvalid; k0 += A32_KB) {
            __syncthreads();
            for (int e = tid; e < A32_KB * 64; e += 256) {
                const int r = e >> 6, d = e & 63;
                const bool ok = k0 + r < kvalid;
                ks[r][d] = ok ? kbase[(long)(k0 + r) * kv_ld + d] : 0.f;
                vs[r][d] = ok ? vbase[(long)(k0 + r) * kv_ld + d] : 0.f;
            }
            __syncthreads();
            const int nk = min(A32_KB, kvalid - k0);
            for (int r = 0; r < nk; ++r) {
                if (causal && k0 + r > qi) break;
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < 64; ++d) s = fmaf(qr[d], ks[r][d], s);
                const float mn = fmaxf(mx, s);
                const float alpha = expf(mx - mn), p = expf(s - mn);      // exp(-inf) = 0 on the first key
                l = l * alpha + p;
#pragma unroll
                for (int d = 0; d < 64; ++d) acc[d] = fmaf(p, vs[r][d], acc[d] * alpha);
                mx = mn;
            }
        }
        if (live) {
            const float inv = 1.0f / l;
            float* o = out + ((long)m * Tq + qi) * out_ld + h * 64;
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] = acc[d] * inv;
        }
    }
}

extern "C" int uniir_attention_f32_fwd(const float* q, int64_t q_ld, const float* k, const float* v, int64_t kv_ld,
                                       float* out, int64_t out_ld, const int32_t* key_len, int32_t batch, int32_t tq,
                                       int32_t tk, int32_t heads, int32_t causal, float scale, void* stream) {
    if (!q || !k || !v || !out || batch < 0 || heads <= 0 || tq < 1 || tk < 1) return UNIIR_EINVAL;
    if (batch == 0) return UNIIR_OK;
    if (causal && tq != tk) return UNIIR_ESHAPE;
    hipLaunchKernelGGL(attention_f32_kernel, dim3(batch * heads), dim3(256), 0, (hipStream_t)stream, q, (long)q_ld, k, v,
                       (long)kv_ld, out, (long)out_ld, key_len, tq, tk, heads, causal, scale);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
