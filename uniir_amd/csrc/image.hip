// Device-side image transform of the encoders' input pipeline: what upstream
// clip._transform (reached from clip_sf.py:25-26, applied per item in src/data/mbeir_dataset.py:92-100) and BLIP's eval
// transform (backbone/transform/blip_transform.py:41-48) do on CPU workers with Pillow + torchvision:
//   Image.resize((ow, oh), BICUBIC) -> centre crop n x n -> x / 255 -> (v - mean) / std -> fp32 [3][n][n]
// on a decoded RGB uint8 image that already sits in HBM.  Integer stage bit-exact with Pillow's 8-bit resample
// (22-bit fixed-point weights, horizontal pass into a uint8 intermediate, then vertical), float stage the same two IEEE
// divisions.  HBM-bound byte work: three small launches per image, only the rows / columns the crop keeps are computed.
#include "common.h"
#include "../../include/uniir_hip.h"

#define IMG_BITS 22

DEVINL double keys_cubic(double x) {     // Keys cubic, a = -0.5 (Pillow's BICUBIC)
    const double a = -0.5;
    x = x < 0 ? -x : x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
static inline int img_ksize(int in_size, int out_size) {
    const double scale = (double)in_size / out_size;
    return (int)ceil(2.0 * (scale < 1.0 ? 1.0 : scale)) * 2 + 1;
}
// one thread per kept output coordinate o = first + t of one axis: window bounds + fixed-point weights (double arithmetic,
// no contraction in this translation unit, so the table equals the CPU one bit for bit)
__global__ void img_coeffs_kernel(int in_size, int out_size, int first, int count, int ksize, int* __restrict__ bounds,
                                  int* __restrict__ kk) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int o = first + t;
    const double scale = (double)in_size / out_size;
    const double fscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * fscale;
    const double center = (o + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    const int n = xmax - xmin;
    const double inv = 1.0 / fscale;
    double ww = 0.0;
    for (int i = 0; i < n; ++i) ww += keys_cubic((i + xmin - center + 0.5) * inv);
    int* k = kk + (long)t * ksize;
    for (int i = 0; i < ksize; ++i) {
        double v = 0.0;
        if (i < n) {
            v = keys_cubic((i + xmin - center + 0.5) * inv);
            if (ww != 0.0) v = v / ww;
        }
        k[i] = v < 0 ? (int)(-0.5 + v * (1 << IMG_BITS)) : (int)(0.5 + v * (1 << IMG_BITS));
    }
    bounds[2 * t] = xmin;
    bounds[2 * t + 1] = n;
}
DEVINL unsigned char img_clip8(int acc) {
    const int v = acc >> IMG_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
// horizontal pass for the kept columns: tmp[y][t][c], y over all source rows the vertical windows touch
__global__ void img_horizontal_kernel(const unsigned char* __restrict__ src, int w, int y0, int rows, int n, int ksize,
                                      const int* __restrict__ bounds, const int* __restrict__ kk,
                                      unsigned char* __restrict__ tmp) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows * n) return;
    const int t = (int)(idx % n), y = (int)(idx / n);
    const int xmin = bounds[2 * t], cnt = bounds[2 * t + 1];
    const int* k = kk + (long)t * ksize;
    const unsigned char* row = src + ((long)(y0 + y) * w + xmin) * 3;
    int a0 = 1 << (IMG_BITS - 1), a1 = a0, a2 = a0;
    for (int i = 0; i < cnt; ++i) {
        const int kv = k[i];
        a0 += row[3 * i] * kv;
        a1 += row[3 * i + 1] * kv;
        a2 += row[3 * i + 2] * kv;
    }
    unsigned char* o = tmp + ((long)y * n + t) * 3;
    o[0] = img_clip8(a0); o[1] = img_clip8(a1); o[2] = img_clip8(a2);
}
// vertical pass + ToTensor + Normalize: out[c][ty][tx] fp32.  tmp rows are relative to y0; `direct` = no vertical resize
__global__ void img_vertical_kernel(const unsigned char* __restrict__ tmp, int y0, int n, int ksize,
                                    const int* __restrict__ bounds, const int* __restrict__ kk, int direct, int top,
                                    float m0, float m1, float m2, float s0, float s1, float s2, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const int tx = idx % n, ty = idx / n;
    unsigned char p0, p1, p2;
    if (direct) {
        const unsigned char* px = tmp + ((long)(top + ty - y0) * n + tx) * 3;
        p0 = px[0]; p1 = px[1]; p2 = px[2];
    } else {
        const int ymin = bounds[2 * ty], cnt = bounds[2 * ty + 1];
        const int* k = kk + (long)ty * ksize;
        int a0 = 1 << (IMG_BITS - 1), a1 = a0, a2 = a0;
        for (int i = 0; i < cnt; ++i) {
            const unsigned char* px = tmp + ((long)(ymin + i - y0) * n + tx) * 3;
            const int kv = k[i];
            a0 += px[0] * kv;
            a1 += px[1] * kv;
            a2 += px[2] * kv;
        }
        p0 = img_clip8(a0); p1 = img_clip8(a1); p2 = img_clip8(a2);
    }
    const long plane = (long)n * n;
    out[idx] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p0, 255.0f), m0), s0);
    out[plane + idx] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p1, 255.0f), m1), s1);
    out[2 * plane + idx] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p2, 255.0f), m2), s2);
}
// same pixel gather when the horizontal size is unchanged: tmp[y][t][c] = src[y0 + y][left + t][c]
__global__ void img_copy_cols_kernel(const unsigned char* __restrict__ src, int w, int y0, int rows, int left, int n,
                                     unsigned char* __restrict__ tmp) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows * n * 3) return;
    const int c = (int)(idx % 3);
    const long p = idx / 3;
    const int t = (int)(p % n), y = (int)(p / n);
    tmp[idx] = src[((long)(y0 + y) * w + left + t) * 3 + c];
}

static inline int64_t img_align(int64_t x) { return (x + 255) & ~(int64_t)255; }

extern "C" int64_t uniir_image_workspace_bytes(int32_t h, int32_t w, int32_t oh, int32_t ow, int32_t n) {
    if (h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || n <= 0) return 0;
    const int kx = img_ksize(w, ow), ky = img_ksize(h, oh);
    return img_align(8L * n) * 2 + img_align(4L * n * kx) + img_align(4L * n * ky) + img_align(3L * h * n);
}

extern "C" int uniir_image_preprocess(const void* rgb_u8, int32_t h, int32_t w, int32_t oh, int32_t ow, int32_t top,
                                      int32_t left, int32_t n, const float* mean3, const float* std3, float* out,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
    if (!rgb_u8 || !mean3 || !std3 || !out || !workspace) return UNIIR_EINVAL;
    if (h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || n <= 0) return UNIIR_EINVAL;
    if (top < 0 || left < 0 || top + n > oh || left + n > ow) return UNIIR_ESHAPE;
    if (workspace_bytes < uniir_image_workspace_bytes(h, w, oh, ow, n) || ((uintptr_t)workspace & 15)) return UNIIR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int kx = img_ksize(w, ow), ky = img_ksize(h, oh);
    char* ws = (char*)workspace;
    int* bx = (int*)ws;                 ws += img_align(8L * n);
    int* by = (int*)ws;                 ws += img_align(8L * n);
    int* cx = (int*)ws;                 ws += img_align(4L * n * kx);
    int* cy = (int*)ws;                 ws += img_align(4L * n * ky);
    unsigned char* tmp = (unsigned char*)ws;
    const bool hres = ow != w, vres = oh != h;
    // source rows the vertical windows of the kept output rows can touch (host copy of the bound arithmetic)
    int y0 = top, y1 = top + n;
    if (vres) {
        const double scale = (double)h / oh, fscale = scale < 1.0 ? 1.0 : scale, support = 2.0 * fscale;
        y0 = (int)((top + 0.5) * scale - support + 0.5);
        if (y0 < 0) y0 = 0;
        y1 = (int)((top + n - 1 + 0.5) * scale + support + 0.5);
        if (y1 > h) y1 = h;
    }
    const int rows = y1 - y0;
    if (hres) hipLaunchKernelGGL(img_coeffs_kernel, dim3((n + 63) / 64), dim3(64), 0, st, w, ow, left, n, kx, bx, cx);
    if (vres) hipLaunchKernelGGL(img_coeffs_kernel, dim3((n + 63) / 64), dim3(64), 0, st, h, oh, top, n, ky, by, cy);
    if (hres)
        hipLaunchKernelGGL(img_horizontal_kernel, dim3((unsigned)(((long)rows * n + 255) / 256)), dim3(256), 0, st,
                           (const unsigned char*)rgb_u8, w, y0, rows, n, kx, bx, cx, tmp);
    else
        hipLaunchKernelGGL(img_copy_cols_kernel, dim3((unsigned)(((long)rows * n * 3 + 255) / 256)), dim3(256), 0, st,
                           (const unsigned char*)rgb_u8, w, y0, rows, left, n, tmp);
    hipLaunchKernelGGL(img_vertical_kernel, dim3((n * n + 255) / 256), dim3(256), 0, st, tmp, y0, n, ky, by, cy,
                       vres ? 0 : 1, top, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
