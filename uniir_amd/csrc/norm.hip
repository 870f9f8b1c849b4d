// LayerNorm forward / backward for the CLIP towers (openai/CLIP model.py LayerNorm subclass: computes in
// fp32, eps 1e-5; used as ln_pre/ln_1/ln_2/ln_post/ln_final, reached from clip_sf.py:43-47).
// HBM-bound: one wave per row, float4 loads, the whole row lives in registers (width <= 2048).
#include "common.h"
#include "../../include/uniir_hip.h"

#define LN_MAXC 8  // float4 chunks per lane -> width <= 64*4*8 = 2048 (kernels are instantiated for NC = 2,3,4,8)
// the row operands are read exactly once per launch: nt policy (-DUNIIR_LN_NT=0 for the A/B build)
#ifndef UNIIR_LN_NT
#define UNIIR_LN_NT 0
#endif
#if UNIIR_LN_NT
#define LN_LD(p) __builtin_nontemporal_load(p)
#else
#define LN_LD(p) (*(p))
#endif

template <int NC, bool EXACT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long x_stride,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     unsigned short* __restrict__ y_bf16,
                                                     float* __restrict__ y_f32, int rows, int width,
                                                     float eps, int flags) {
    // flags bit 0: T5 / RMS norm (y = gamma * x * rsqrt(mean(x^2) + eps)): no mean subtraction, no beta;  bit 1: the 16-bit output is
    // fp16 instead of bf16 (the fp16 forward of the embedder, uniir_clip_tower.dtype16)
    const int rms = flags & 1;
    const bool f16 = (flags & 2) != 0;
    const int lane = threadIdx.x & 63;
    const int nchunk = width >> 2;
    const float inv_w = 1.0f / (float)width;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
        const float* xr = x + row * x_stride;
        f32x4_t v[NC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (EXACT || c < nchunk) {
                v[i] = LN_LD(reinterpret_cast<const f32x4_t*>(xr + 4 * c));
                s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            }
        }
        const float mean = rms ? 0.f : wave_sum(s) * inv_w;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (EXACT || c < nchunk) {
                const f32x4_t d = v[i] - mean;
                q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_w + eps);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (EXACT || c < nchunk) {
                const f32x4_t g = *reinterpret_cast<const f32x4_t*>(gamma + 4 * c);
                const f32x4_t b = rms ? f32x4_t{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4_t*>(beta + 4 * c);
                const f32x4_t o = (v[i] - mean) * rstd * g + b;
                if (y_bf16) {
                    u32x2_t pk = {pack16x2(o[0], o[1], f16), pack16x2(o[2], o[3], f16)};
                    *reinterpret_cast<u32x2_t*>(y_bf16 + row * width + 4 * c) = pk;
                }
                if (y_f32) *reinterpret_cast<f32x4_t*>(y_f32 + row * width + 4 * c) = o;
            }
        }
    }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy * gamma;  dgamma += dy*xhat; dbeta += dy
// (Round 4, measured and not kept: lanes owning PAIRS of neighbouring chunks, so that the bf16 operand / result are 16-byte instead
// of 8-byte accesses -- the change that took 15 % off the GEMM's activation-gradient epilogue: forward 0.295 -> 0.279 ms, backward
// 0.499 -> 0.494 ms at 263 168 x 1024, inside the box-to-box noise, for 8 spilled registers at the backward's 168-register budget.
// These kernels are bound by HBM, not by the number of requests.)
// EXACT: width == 256 * NC (every production width: 512 / 768 / 1024) -- no chunk predication, which is worth 50 VGPRs
// (202 -> 152 at NC = 4: 3 waves / SIMD instead of 2).  The residual-gradient row is loaded with x and dy at the top of the
// row (one exposed memory latency per row instead of two).
template <int NC, bool DY_F32, bool EXACT>
__global__ __launch_bounds__(256, (EXACT && NC <= 4) ? 3 : 1) void ln_bwd_kernel(const float* __restrict__ x, long x_stride,
                                                     const float* __restrict__ gamma,
                                                     const void* __restrict__ dy_,
                                                     const float* __restrict__ dres,
                                                     float* __restrict__ dx, long dx_stride,
                                                     unsigned short* __restrict__ dx_bf16,
                                                     float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta,
                                                     float* __restrict__ dx_colsum, int rows, int width,
                                                     float eps, int rms, const float* __restrict__ bscale,
                                                     float* __restrict__ part) {
    __shared__ float red[4][64 * 4 * NC];  // per wave staging for the column reduction
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nchunk = width >> 2;
    const float inv_w = 1.0f / (float)width;
    f32x4_t ag[NC], ab[NC], ac[NC];   // column partials of dgamma, dbeta and (optional) of dx itself
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        ag[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        ab[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        ac[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    for (long row = (long)blockIdx.x * 4 + w; row < rows; row += (long)gridDim.x * 4) {
        const float* xr = x + row * x_stride;
        f32x4_t v[NC], d[NC], rs[NC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (EXACT || c < nchunk) {
                v[i] = LN_LD(reinterpret_cast<const f32x4_t*>(xr + 4 * c));
                if (dres) rs[i] = LN_LD(reinterpret_cast<const f32x4_t*>(dres + row * dx_stride + 4 * c));
                s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
                if (DY_F32) {
                    d[i] = LN_LD(reinterpret_cast<const f32x4_t*>((const float*)dy_ + row * width + 4 * c));
                } else {
                    const u32x2_t pk =
                        LN_LD(reinterpret_cast<const u32x2_t*>((const unsigned short*)dy_ + row * width + 4 * c));
                    d[i] = f32x4_t{__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xffff0000u),
                                   __uint_as_float(pk[1] << 16), __uint_as_float(pk[1] & 0xffff0000u)};
                }
            }
        }
        const float mean = rms ? 0.f : wave_sum(s) * inv_w;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (EXACT || c < nchunk) {
                v[i] = v[i] - mean;
                q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_w + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (EXACT || c < nchunk) {
                v[i] = v[i] * rstd;  // xhat
                const f32x4_t g = d[i] * *reinterpret_cast<const f32x4_t*>(gamma + 4 * c);
                ag[i] += d[i] * v[i];
                ab[i] += d[i];
                d[i] = g;
                s1 += (g[0] + g[1]) + (g[2] + g[3]);
                s2 += (g[0] * v[i][0] + g[1] * v[i][1]) + (g[2] * v[i][2] + g[3] * v[i][3]);
            }
        }
        const float c1 = rms ? 0.f : wave_sum(s1) * inv_w, c2 = wave_sum(s2) * inv_w;
        // bscale: the row's DropPath factor of the residual BRANCH this gradient enters next (its last linear layer's bias
        // gradient and 16-bit operand see bscale * dx; the fp32 residual gradient itself stays unscaled)
        const float bs = bscale ? bscale[row] : 1.0f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (EXACT || c < nchunk) {
                f32x4_t o = (d[i] - c1 - v[i] * c2) * rstd;
                if (dres) o += rs[i];
                *reinterpret_cast<f32x4_t*>(dx + row * dx_stride + 4 * c) = o;
                if (bscale) o = o * bs;
                if (dx_colsum) ac[i] += o;
                if (dx_bf16) {
                    u32x2_t pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    *reinterpret_cast<u32x2_t*>(dx_bf16 + row * width + 4 * c) = pk;
                }
            }
        }
    }
    // block reduction of the per-wave column partials, then one atomic per column per block
    float* mine = &red[w][0];
    for (int pass = 0; pass < (dx_colsum ? 3 : 2); ++pass) {
        if (pass == 1 && rms) continue;      // no beta
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (EXACT || c < nchunk) *reinterpret_cast<f32x4_t*>(mine + 4 * c) = pass == 0 ? ag[i] : (pass == 1 ? ab[i] : ac[i]);
        }
        __syncthreads();
        float* dst = pass == 0 ? dgamma : (pass == 1 ? dbeta : dx_colsum);
        for (int col = threadIdx.x; col < width; col += 256) {
            const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
            // part: this workgroup's three column partials [block][pass][width]; a fixed-order reduction follows (common.h)
            if (part) part[((long)blockIdx.x * 3 + pass) * width + col] = t;
            else unsafeAtomicAdd(dst + col, t);
        }
        __syncthreads();
    }
}

template <int NC>
static void launch_ln_fwd(const float* x, long x_stride, const float* gamma, const float* beta, unsigned short* yb,
                          float* yf, int rows, int width, float eps, hipStream_t st, int rms = 0) {
    static int resident[2] = {0, 0};
    const bool exact = width == 256 * NC;
    if (!resident[exact]) {
        int per_cu = 0;
        const void* fn = exact ? (const void*)ln_fwd_kernel<NC, true> : (const void*)ln_fwd_kernel<NC, false>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        resident[exact] = per_cu * 256;
    }
    int g = (rows + 3) / 4;
    if (g > resident[exact]) g = resident[exact];
    if (exact)
        hipLaunchKernelGGL((ln_fwd_kernel<NC, true>), dim3(g), dim3(256), 0, st, x, x_stride, gamma, beta, yb, yf, rows, width,
                           eps, rms);
    else
        hipLaunchKernelGGL((ln_fwd_kernel<NC, false>), dim3(g), dim3(256), 0, st, x, x_stride, gamma, beta, yb, yf, rows, width,
                           eps, rms);
}
template <int NC, bool F32>
static void launch_ln_bwd(const float* x, long x_stride, const float* gamma, const void* dy, const float* dres,
                          float* dx, long dx_stride, unsigned short* dxb, float* dgamma, float* dbeta, float* dxsum,
                          int rows, int width, float eps, hipStream_t st, int rms = 0, const float* bscale = nullptr) {
    // persistent rows: one resident wave of workgroups (occupancy x 256 CUs), no tail wave
    static int resident[2] = {0, 0};
    const bool exact = width == 256 * NC;
    if (!resident[exact]) {
        int per_cu = 0;
        const void* fn = exact ? (const void*)ln_bwd_kernel<NC, F32, true> : (const void*)ln_bwd_kernel<NC, F32, false>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        resident[exact] = per_cu * 256;
    }
    int g = (rows + 3) / 4;
    if (g > resident[exact]) g = resident[exact];
    // more than one workgroup: the per-workgroup column partials (d gamma, d beta, column sums of dx) are stored and reduced in a
    // fixed order when the stream has a scratch buffer (atomics in arrival order otherwise)
    float* part = g > 1 ? reduce_scratch(st, (int64_t)g * 3 * width * 4) : nullptr;
    if (exact)
        hipLaunchKernelGGL((ln_bwd_kernel<NC, F32, true>), dim3(g), dim3(256), 0, st, x, x_stride, gamma, dy, dres, dx,
                           dx_stride, dxb, dgamma, dbeta, dxsum, rows, width, eps, rms, bscale, part);
    else
        hipLaunchKernelGGL((ln_bwd_kernel<NC, F32, false>), dim3(g), dim3(256), 0, st, x, x_stride, gamma, dy, dres, dx,
                           dx_stride, dxb, dgamma, dbeta, dxsum, rows, width, eps, rms, bscale, part);
    if (part) (void)reduce_partials(part, g, 3L * width, width, dgamma, rms ? nullptr : dbeta, dxsum, width, st);
}
static inline int ln_nc(int width) {
    const int c = (width / 4 + 63) / 64;
    return c <= 2 ? 2 : (c == 3 ? 3 : (c == 4 ? 4 : 8));
}

// f16: the 16-bit output as fp16 (tower.hip's fp16 forward; the extern "C" entry point writes bf16)
int layernorm_fwd_impl(const float* x, int64_t x_stride, const float* gamma, const float* beta, void* y_16, float* y_f32,
                       int32_t rows, int32_t width, float eps, int f16, void* stream) {
    if (!x || !gamma || !beta || (!y_16 && !y_f32) || rows < 0) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    if (width % 4 || width > 64 * 4 * LN_MAXC || width <= 0 || x_stride % 4) return UNIIR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    unsigned short* yb = (unsigned short*)y_16;
    const int flags = f16 ? 2 : 0;
    switch (ln_nc(width)) {
        case 2: launch_ln_fwd<2>(x, x_stride, gamma, beta, yb, y_f32, rows, width, eps, st, flags); break;
        case 3: launch_ln_fwd<3>(x, x_stride, gamma, beta, yb, y_f32, rows, width, eps, st, flags); break;
        case 4: launch_ln_fwd<4>(x, x_stride, gamma, beta, yb, y_f32, rows, width, eps, st, flags); break;
        default: launch_ln_fwd<8>(x, x_stride, gamma, beta, yb, y_f32, rows, width, eps, st, flags); break;
    }
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
extern "C" int uniir_layernorm_fwd(const float* x, int64_t x_stride, const float* gamma, const float* beta,
                                   void* y_bf16, float* y_f32, int32_t rows, int32_t width, float eps,
                                   void* stream) {
    return layernorm_fwd_impl(x, x_stride, gamma, beta, y_bf16, y_f32, rows, width, eps, 0, stream);
}

static int layernorm_bwd_impl(const float* x, int64_t x_stride, const float* gamma, const void* dy, int32_t dy_is_f32,
                              const float* dres, float* dx_f32, int64_t dx_stride, void* dx_bf16, float* dgamma, float* dbeta,
                              float* dx_colsum, const float* branch_scale, int32_t rows, int32_t width, float eps, void* stream) {
    if (!x || !gamma || !dy || !dx_f32 || !dgamma || !dbeta || rows < 0) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    if (width % 4 || width > 64 * 4 * LN_MAXC || width <= 0 || x_stride % 4 || dx_stride % 4) return UNIIR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    unsigned short* dxb = (unsigned short*)dx_bf16;
#define LNB(NC)                                                                                                   \
    do {                                                                                                          \
        if (dy_is_f32) launch_ln_bwd<NC, true>(x, x_stride, gamma, dy, dres, dx_f32, dx_stride, dxb, dgamma, dbeta, dx_colsum, rows, width, eps, st, 0, branch_scale); \
        else launch_ln_bwd<NC, false>(x, x_stride, gamma, dy, dres, dx_f32, dx_stride, dxb, dgamma, dbeta, dx_colsum, rows, width, eps, st, 0, branch_scale);          \
    } while (0)
    switch (ln_nc(width)) {
        case 2: LNB(2); break;
        case 3: LNB(3); break;
        case 4: LNB(4); break;
        default: LNB(8); break;
    }
#undef LNB
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
extern "C" int uniir_layernorm_bwd(const float* x, int64_t x_stride, const float* gamma, const void* dy,
                                   int32_t dy_is_f32, const float* dres, float* dx_f32, int64_t dx_stride,
                                   void* dx_bf16, float* dgamma, float* dbeta, float* dx_colsum, int32_t rows,
                                   int32_t width, float eps, void* stream) {
    return layernorm_bwd_impl(x, x_stride, gamma, dy, dy_is_f32, dres, dx_f32, dx_stride, dx_bf16, dgamma, dbeta, dx_colsum,
                              nullptr, rows, width, eps, stream);
}
// the same with a per-row factor for what leaves towards the next residual BRANCH: dx_bf16 = bf16(branch_scale[row] * dx) and
// dx_colsum += branch_scale[row] * dx, while dx_f32 (the residual-stream gradient) stays unscaled -- DropPath in backward
extern "C" int uniir_layernorm_bwd_ex(const float* x, int64_t x_stride, const float* gamma, const void* dy,
                                      int32_t dy_is_f32, const float* dres, float* dx_f32, int64_t dx_stride,
                                      void* dx_bf16, float* dgamma, float* dbeta, float* dx_colsum,
                                      const float* branch_scale, int32_t rows, int32_t width, float eps, void* stream) {
    return layernorm_bwd_impl(x, x_stride, gamma, dy, dy_is_f32, dres, dx_f32, dx_stride, dx_bf16, dgamma, dbeta, dx_colsum,
                              branch_scale, rows, width, eps, stream);
}

// RMS norm (transformers T5LayerNorm, used by the CLIP_FF fusion stack): same kernels, rms = 1
extern "C" int uniir_rmsnorm_fwd(const float* x, int64_t x_stride, const float* gamma, void* y_bf16, float* y_f32,
                                 int32_t rows, int32_t width, float eps, void* stream) {
    if (!x || !gamma || (!y_bf16 && !y_f32) || rows < 0) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    if (width % 4 || width > 64 * 4 * LN_MAXC || width <= 0 || x_stride % 4) return UNIIR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    unsigned short* yb = (unsigned short*)y_bf16;
    switch (ln_nc(width)) {
        case 2: launch_ln_fwd<2>(x, x_stride, gamma, nullptr, yb, y_f32, rows, width, eps, st, 1); break;
        case 3: launch_ln_fwd<3>(x, x_stride, gamma, nullptr, yb, y_f32, rows, width, eps, st, 1); break;
        case 4: launch_ln_fwd<4>(x, x_stride, gamma, nullptr, yb, y_f32, rows, width, eps, st, 1); break;
        default: launch_ln_fwd<8>(x, x_stride, gamma, nullptr, yb, y_f32, rows, width, eps, st, 1); break;
    }
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

extern "C" int uniir_rmsnorm_bwd(const float* x, int64_t x_stride, const float* gamma, const void* dy, int32_t dy_is_f32,
                                 const float* dres, float* dx_f32, int64_t dx_stride, void* dx_bf16, float* dgamma,
                                 int32_t rows, int32_t width, float eps, void* stream) {
    if (!x || !gamma || !dy || !dx_f32 || !dgamma || rows < 0) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    if (width % 4 || width > 64 * 4 * LN_MAXC || width <= 0 || x_stride % 4 || dx_stride % 4) return UNIIR_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    unsigned short* dxb = (unsigned short*)dx_bf16;
#define RMSB(NC)                                                                                                       \
    do {                                                                                                               \
        if (dy_is_f32) launch_ln_bwd<NC, true>(x, x_stride, gamma, dy, dres, dx_f32, dx_stride, dxb, dgamma, nullptr, nullptr, rows, width, eps, st, 1); \
        else launch_ln_bwd<NC, false>(x, x_stride, gamma, dy, dres, dx_f32, dx_stride, dxb, dgamma, nullptr, nullptr, rows, width, eps, st, 1);          \
    } while (0)
    switch (ln_nc(width)) {
        case 2: RMSB(2); break;
        case 3: RMSB(3); break;
        case 4: RMSB(4); break;
        default: RMSB(8); break;
    }
#undef RMSB
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
