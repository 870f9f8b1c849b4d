// HBM-bound glue kernels of the CLIP towers, the fuse/normalise step and the optimizer.
// Reference semantics (UniIR tree): clip_sf.py:53-63 (mask fuse), :88-97 (select + F.normalize),
// clip_scorefusion/train.py:195-199 (AdamW), openai/CLIP model.py (conv1 patch embed, class/positional
// embeddings, token embedding, EOT pooling) as called from clip_sf.py:43-47.
#include "common.h"
#include "../../include/uniir_hip.h"

static inline int grid_for(long work, int per_block, int cap = 8192) {
    long g = (work + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---------------------------------------------------------------------------------------------------
// deterministic column sums: the scratch registry and the fixed-order reduction of per-workgroup partials (common.h)
#define RED_SLOTS 64
static struct { hipStream_t st; int dev; float* buf; int64_t bytes; bool used; unsigned long stamp; } g_red_scratch[RED_SLOTS];
static unsigned long g_red_clock = 0;
static int red_device() {          // the NULL stream is every device's default stream: entries are per (device, stream)
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
}
extern "C" int uniir_reduce_scratch(void* buf, int64_t bytes, void* stream) {
    if ((buf && bytes <= 0) || ((uintptr_t)buf & 255)) return UNIIR_EINVAL;
    int slot = -1, oldest = 0;
    const int dev = red_device();
    for (int i = 0; i < RED_SLOTS; ++i) {
        if (g_red_scratch[i].used && g_red_scratch[i].st == (hipStream_t)stream && g_red_scratch[i].dev == dev) {
            if (!buf) { g_red_scratch[i].used = false; return UNIIR_OK; }
            g_red_scratch[i].buf = (float*)buf; g_red_scratch[i].bytes = bytes; g_red_scratch[i].stamp = ++g_red_clock;
            return UNIIR_OK;
        }
        if (!g_red_scratch[i].used && slot < 0) slot = i;
        if (g_red_scratch[i].stamp < g_red_scratch[oldest].stamp) oldest = i;
    }
    if (!buf) return UNIIR_OK;
    if (slot < 0) slot = oldest;          // table full: the entry used longest ago goes (a stream that has not launched a reduction
                                          // for 63 registrations is gone, or falls back to atomics until it registers again)
    g_red_scratch[slot].st = (hipStream_t)stream; g_red_scratch[slot].dev = dev; g_red_scratch[slot].buf = (float*)buf;
    g_red_scratch[slot].bytes = bytes; g_red_scratch[slot].used = true; g_red_scratch[slot].stamp = ++g_red_clock;
    return UNIIR_OK;
}
float* reduce_scratch(hipStream_t st, int64_t bytes) {
    int dev = -1;
    for (int i = 0; i < RED_SLOTS; ++i)
        if (g_red_scratch[i].used && g_red_scratch[i].st == st) {
            if (dev < 0) dev = red_device();
            if (g_red_scratch[i].dev == dev) {
                g_red_scratch[i].stamp = ++g_red_clock;
                return g_red_scratch[i].bytes >= bytes ? g_red_scratch[i].buf : nullptr;
            }
        }
    return nullptr;
}
// block = 8 groups of 4 columns x 32 chunks of the parts; chunk q adds its contiguous range of parts in order (16-byte loads, all
// independent), the 32 chunk sums are added in order: one fixed summation tree per column, whatever order the producers finished in.
// cols and stride are multiples of 4 at every call site.
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ part, int nparts, long stride, int cols,
                                                             float* __restrict__ d0, float* __restrict__ d1, float* __restrict__ d2,
                                                             long plane) {
    __shared__ f32x4_t red[32][8];
    float* dst = blockIdx.y == 0 ? d0 : (blockIdx.y == 1 ? d1 : d2);
    if (!dst) return;
    const int cg = threadIdx.x & 7, q = threadIdx.x >> 3, c = (blockIdx.x * 8 + cg) * 4;
    const float* p = part + (long)blockIdx.y * plane;
    const int per = (nparts + 31) / 32, j0 = q * per, j1 = min(nparts, j0 + per);
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
        // eight loads in flight per thread (a dependent load-add chain ran at one HBM latency per part: 52 us per call), added in
        // the order of the parts
        int j = j0;
        for (; j + 8 <= j1; j += 8) {
            f32x4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4_t*>(p + (long)(j + u) * stride + c);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; j < j1; ++j) s += *reinterpret_cast<const f32x4_t*>(p + (long)j * stride + c);
    }
    red[q][cg] = s;
    __syncthreads();
    if (q == 0 && c < cols) {
        f32x4_t t = red[0][cg];
#pragma unroll
        for (int k = 1; k < 32; ++k) t += red[k][cg];
        f32x4_t* o = reinterpret_cast<f32x4_t*>(dst + c);
        *o = *o + t;
    }
}
// the same tree, one column per thread (a destination that is not 16-byte aligned)
__global__ __launch_bounds__(256) void partial_reduce1_kernel(const float* __restrict__ part, int nparts, long stride, int cols,
                                                              float* __restrict__ d0, float* __restrict__ d1, float* __restrict__ d2,
                                                              long plane) {
    __shared__ float red[32][8];
    float* dst = blockIdx.y == 0 ? d0 : (blockIdx.y == 1 ? d1 : d2);
    if (!dst) return;
    const int cg = threadIdx.x & 7, q = threadIdx.x >> 3;
    const float* p = part + (long)blockIdx.y * plane;
    const int per = (nparts + 31) / 32, j0 = q * per, j1 = min(nparts, j0 + per);
    for (int e = 0; e < 4; ++e) {          // the four columns of the group one after the other, the float4 kernel's order per column
        const int c = (blockIdx.x * 8 + cg) * 4 + e;
        float s = 0.f;
        if (c < cols)
            for (int j = j0; j < j1; ++j) s += p[(long)j * stride + c];
        red[q][cg] = s;
        __syncthreads();
        if (q == 0 && c < cols) {
            float t = red[0][cg];
            for (int k = 1; k < 32; ++k) t += red[k][cg];
            dst[c] += t;
        }
        __syncthreads();
    }
}
int reduce_partials(const float* part, int nparts, long stride, int cols, float* d0, float* d1, float* d2, long plane, hipStream_t st) {
    if (nparts <= 0 || cols <= 0) return UNIIR_OK;
    const bool vec = !((cols | stride | plane) & 3) && !(((uintptr_t)part | (uintptr_t)d0 | (uintptr_t)d1 | (uintptr_t)d2) & 15);
    if (vec)
        hipLaunchKernelGGL(partial_reduce_kernel, dim3((cols + 31) / 32, 3), dim3(256), 0, st, part, nparts, stride, cols, d0, d1, d2, plane);
    else
        hipLaunchKernelGGL(partial_reduce1_kernel, dim3((cols + 31) / 32, 3), dim3(256), 0, st, part, nparts, stride, cols, d0, d1, d2, plane);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// ---------------------------------------------------------------------------------------------------
// patchify: images f32 [n][3][res][res] -> bf16 [n*g*g][kpad], k = c*P*P + py*P + px
// one thread produces 8 consecutive k (16 B store); reads are P-contiguous runs of a pixel row.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img,
                                                       unsigned short* __restrict__ out, int n, int res,
                                                       int P, int kpad, int f16) {
    const int g = res / P, kreal = 3 * P * P, chunks = kpad >> 3;
    const long total = (long)n * g * g * chunks;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int ch = (int)(t % chunks);
        const long row = t / chunks;
        const int gx = (int)(row % g), gy = (int)((row / g) % g);
        const long im = row / ((long)g * g);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ch * 8 + e;
            float val = 0.f;
            if (k < kreal) {
                const int c = k / (P * P), rem = k - c * P * P, py = rem / P, px = rem - py * P;
                val = img[((im * 3 + c) * res + (gy * P + py)) * (long)res + gx * P + px];
            }
            v[e] = val;
        }
        u32x4_t o = {pack16x2(v[0], v[1], f16 != 0), pack16x2(v[2], v[3], f16 != 0), pack16x2(v[4], v[5], f16 != 0),
                     pack16x2(v[6], v[7], f16 != 0)};
        *reinterpret_cast<u32x4_t*>(out + row * kpad + ch * 8) = o;
    }
}

int patchify_impl(const float* images, void* patches, int32_t n, int32_t res, int32_t patch, int32_t kpad, int f16, void* stream);
extern "C" int uniir_patchify(const float* images, void* patches, int32_t n, int32_t res, int32_t patch,
                              int32_t kpad, void* stream) {
    return patchify_impl(images, patches, n, res, patch, kpad, 0, stream);
}
int patchify_impl(const float* images, void* patches, int32_t n, int32_t res, int32_t patch, int32_t kpad, int f16, void* stream) {
    if (!images || !patches || n < 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    if (patch <= 0 || res % patch || kpad % 8 || kpad < 3 * patch * patch) return UNIIR_ESHAPE;
    const int g = res / patch;
    const long total = (long)n * g * g * (kpad / 8);
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       images, (unsigned short*)patches, n, res, patch, kpad, f16);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// ---------------------------------------------------------------------------------------------------
// vit assemble: x[n][T][w] = (t == 0 ? class_emb : patch_out[n][t-1]) + pos[t]
__global__ __launch_bounds__(256) void vit_assemble_kernel(const unsigned short* __restrict__ po,
                                                           const float* __restrict__ cls,
                                                           const float* __restrict__ pos,
                                                           float* __restrict__ x, int n, int T, int w, int f16) {
    const int wc = w >> 2;
    const long total = (long)n * T * wc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % wc);
        const long row = i / wc;
        const int t = (int)(row % T);
        const long im = row / T;
        f32x4_t v;
        if (t == 0) {
            v = *reinterpret_cast<const f32x4_t*>(cls + 4 * c);
        } else {
            const u32x2_t pk =
                *reinterpret_cast<const u32x2_t*>(po + (im * (T - 1) + (t - 1)) * (long)w + 4 * c);
            v = f32x4_t{unpack16_lo(pk[0], f16 != 0), unpack16_hi(pk[0], f16 != 0), unpack16_lo(pk[1], f16 != 0),
                        unpack16_hi(pk[1], f16 != 0)};
        }
        v += *reinterpret_cast<const f32x4_t*>(pos + (long)t * w + 4 * c);
        *reinterpret_cast<f32x4_t*>(x + row * w + 4 * c) = v;
    }
}

int vit_assemble_impl(const void* patch_out, const float* class_emb, const float* pos_emb, float* x, int32_t n, int32_t tokens,
                      int32_t width, int f16, void* stream);
extern "C" int uniir_vit_assemble(const void* patch_out, const float* class_emb, const float* pos_emb,
                                  float* x, int32_t n, int32_t tokens, int32_t width, void* stream) {
    return vit_assemble_impl(patch_out, class_emb, pos_emb, x, n, tokens, width, 0, stream);
}
int vit_assemble_impl(const void* patch_out, const float* class_emb, const float* pos_emb, float* x, int32_t n, int32_t tokens,
                      int32_t width, int f16, void* stream) {
    if (!patch_out || !class_emb || !pos_emb || !x || n < 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    if (width % 4 || tokens < 2) return UNIIR_ESHAPE;
    const long total = (long)n * tokens * (width / 4);
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned short*)patch_out, class_emb, pos_emb, x, n, tokens,
                       width, f16);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// backward: dpatch = bf16(dx[:,1:,:]); dpos[t] += sum_n dx[n][t]; dclass += sum_n dx[n][0]
// grid = (T, w/256-ish): each block owns one token position t and reduces over n.
__global__ __launch_bounds__(256) void vit_assemble_bwd_kernel(const float* __restrict__ dx,
                                                               unsigned short* __restrict__ dpo,
                                                               float* __restrict__ dcls,
                                                               float* __restrict__ dpos, int n, int T,
                                                               int w) {
    const int t = blockIdx.x;
    for (int col = blockIdx.y * 256 + threadIdx.x; col < w; col += gridDim.y * 256) {
        float s = 0.f;
        for (int im = 0; im < n; ++im) {
            const float v = dx[((long)im * T + t) * w + col];
            s += v;
            if (t > 0) dpo[((long)im * (T - 1) + (t - 1)) * w + col] = f32_to_bf16(v);
        }
        dpos[(long)t * w + col] += s;
        if (t == 0) dcls[col] += s;
    }
}

extern "C" int uniir_vit_assemble_bwd(const float* dx, void* dpatch_out, float* dclass, float* dpos,
                                      int32_t n, int32_t tokens, int32_t width, void* stream) {
    if (!dx || !dpatch_out || !dclass || !dpos || n < 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    hipLaunchKernelGGL(vit_assemble_bwd_kernel, dim3(tokens, (width + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, dx, (unsigned short*)dpatch_out, dclass, dpos, n, tokens, width);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// ---------------------------------------------------------------------------------------------------
// text embed: x[n][t][:] = tok[text[n][t]] + pos[t]; eot[n] = first argmax_t text[n][t]
__global__ __launch_bounds__(256) void text_embed_kernel(const int* __restrict__ text,
                                                         const float* __restrict__ tok,
                                                         const float* __restrict__ pos,
                                                         float* __restrict__ x, int n, int ctx, int w,
                                                         int vocab) {
    const int wc = w >> 2;
    const long total = (long)n * ctx * wc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % wc);
        const long row = i / wc;
        const int t = (int)(row % ctx);
        int id = text[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        f32x4_t v = *reinterpret_cast<const f32x4_t*>(tok + (long)id * w + 4 * c);
        v += *reinterpret_cast<const f32x4_t*>(pos + (long)t * w + 4 * c);
        *reinterpret_cast<f32x4_t*>(x + row * w + 4 * c) = v;
    }
}
__global__ void text_eot_kernel(const int* __restrict__ text, int* __restrict__ eot, int n, int ctx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int best = text[(long)i * ctx], bi = 0;
    for (int t = 1; t < ctx; ++t) {
        const int v = text[(long)i * ctx + t];
        if (v > best) { best = v; bi = t; }
    }
    eot[i] = bi;
}

extern "C" int uniir_text_embed(const int32_t* text, const float* token_emb, const float* pos_emb, float* x,
                                int32_t* eot, int32_t n, int32_t ctx, int32_t width, int32_t vocab,
                                void* stream) {
    if (!text || !token_emb || !pos_emb || !x || n < 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    if (width % 4) return UNIIR_ESHAPE;
    const long total = (long)n * ctx * (width / 4);
    hipLaunchKernelGGL(text_embed_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       text, token_emb, pos_emb, x, n, ctx, width, vocab);
    if (eot)
        hipLaunchKernelGGL(text_eot_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, text, eot,
                           n, ctx);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// ---- deterministic token-embedding gradient (round 6) -------------------------------------------------------------------------
// dtok[id] += sum of the dx rows whose token is id.  Token ids repeat (every caption has the SOT / EOT or [CLS] / [SEP] ids, common
// words collide), so the plain kernels below add with fp32 atomics in arrival order.  With a scratch buffer on the stream the rows are
// bucketed by id instead (counting sort: integer atomics only, whose results do not depend on order), every bucket is put in ascending
// row order, and one workgroup per id adds its rows in that order: the same bits every run.  A bucket longer than TOKD_SORT rows
// (only the padding id of an UNPACKED batch -- rows behind the EOT / the valid length, whose gradients are exact zeros -- or the
// special tokens of a batch of more than 4096 captions) is added in bucket order as filled.
#define TOKD_SORT 4096
DEVINL bool tokd_row(const int* __restrict__ text, const int* __restrict__ row_off, long row, int ctx, int vocab, int* id, long* dxrow) {
    const int t = (int)(row % ctx), m = (int)(row / ctx);
    if (row_off) {
        const int r0 = row_off[m];
        if (t >= row_off[m + 1] - r0) return false;
        *dxrow = r0 + t;
    } else {
        *dxrow = row;
    }
    const int v = text[row];
    *id = v < 0 ? 0 : (v >= vocab ? vocab - 1 : v);
    return true;
}
__global__ __launch_bounds__(256) void tokd_count_kernel(const int* __restrict__ text, const int* __restrict__ row_off, long rows, int ctx,
                                                         int vocab, int* __restrict__ counts) {
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        int id; long dxr;
        if (tokd_row(text, row_off, r, ctx, vocab, &id, &dxr)) atomicAdd(counts + id, 1);
    }
}
// one workgroup: starts = exclusive prefix sums of counts ([vocab + 1]), cursor = a copy the fill pass advances.  A thread owns
// <= TOKD_PER consecutive ids, read with all loads in flight (vocab <= 1024 * TOKD_PER)
#define TOKD_PER 64
__global__ __launch_bounds__(1024) void tokd_scan_kernel(const int* __restrict__ counts, int* __restrict__ starts, int* __restrict__ cursor,
                                                         int vocab) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, per = (vocab + 1023) / 1024, lo = tid * per;
    int c[TOKD_PER];
#pragma unroll
    for (int i = 0; i < TOKD_PER; ++i) c[i] = (i < per && lo + i < vocab) ? counts[lo + i] : 0;
    int s = 0;
#pragma unroll
    for (int i = 0; i < TOKD_PER; ++i) s += c[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
#pragma unroll
    for (int i = 0; i < TOKD_PER; ++i) {
        if (i < per && lo + i < vocab) {
            starts[lo + i] = run;
            cursor[lo + i] = run;
            run += c[i];
        }
    }
    if (tid == 1023) starts[vocab] = part[1023];
}
__global__ __launch_bounds__(256) void tokd_fill_kernel(const int* __restrict__ text, const int* __restrict__ row_off, long rows, int ctx,
                                                        int vocab, int* __restrict__ cursor, int* __restrict__ bucket) {
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        int id; long dxr;
        if (tokd_row(text, row_off, r, ctx, vocab, &id, &dxr)) bucket[atomicAdd(cursor + id, 1)] = (int)dxr;
    }
}
// one workgroup per (token id, 256-column slab): the id's bucket in ascending row order (bitonic sort in LDS), then its rows added in
// that order -- four contiguous quarters of the bucket in parallel (eight 16-byte loads in flight per thread), the four quarter sums
// added in order: one fixed summation tree per element
__global__ __launch_bounds__(256) void tokd_accum_kernel(const int* __restrict__ starts, const int* __restrict__ bucket,
                                                         const float* __restrict__ dx, float* __restrict__ dtok, int w) {
    __shared__ int srt[TOKD_SORT];
    __shared__ f32x4_t red[4][64];
    const int id = blockIdx.x, tid = threadIdx.x;
    const int b0 = starts[id], cnt = starts[id + 1] - b0;
    if (cnt == 0) return;
    const int* rows = bucket + b0;
    const bool sorted = cnt <= TOKD_SORT;
    if (sorted && cnt > 1) {
        int np = 2;
        while (np < cnt) np <<= 1;
        for (int i = tid; i < np; i += 256) srt[i] = i < cnt ? rows[i] : 0x7fffffff;
        __syncthreads();
        for (int k = 2; k <= np; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < np; i += 256) {
                    const int l = i ^ j;
                    if (l > i) {
                        const int a = srt[i], b = srt[l];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { srt[i] = b; srt[l] = a; }
                    }
                }
                __syncthreads();
            }
        rows = nullptr;
    }
    auto row_at = [&](int i) { return (sorted && cnt > 1) ? srt[i] : bucket[b0 + i]; };
    const int cg = tid & 63, q = tid >> 6, col = blockIdx.y * 256 + cg * 4;
    const int per = (cnt + 3) / 4, i0 = q * per, i1 = min(cnt, i0 + per);
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    if (col < w) {
        int i = i0;
        for (; i + 8 <= i1; i += 8) {
            f32x4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4_t*>(dx + (long)row_at(i + u) * w + col);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; i < i1; ++i) s += *reinterpret_cast<const f32x4_t*>(dx + (long)row_at(i) * w + col);
    }
    red[q][cg] = s;
    __syncthreads();
    if (q == 0 && col < w) {
        const f32x4_t t = ((red[0][cg] + red[1][cg]) + red[2][cg]) + red[3][cg];
        f32x4_t* o = reinterpret_cast<f32x4_t*>(dtok + (long)id * w + col);
        *o = *o + t;
    }
}
// the bucketed form when the stream has a scratch buffer: 1 = done, 0 = not applicable (the caller runs the atomic kernel), < 0 = a
// launch failed (nothing may be added a second time)
static int tokd_run(const int* text, const float* dx, const int* row_off, float* dtok, int n, int ctx, int w, int vocab, hipStream_t st) {
    const long rows = (long)n * ctx;
    if (rows >= 0x7fffffffL || vocab > 1024 * TOKD_PER || (w & 3) || (((uintptr_t)dx | (uintptr_t)dtok) & 15)) return 0;
    int* ws = (int*)reduce_scratch(st, (int64_t)(3L * vocab + 4 + rows) * 4);
    if (!ws) return 0;
    int *counts = ws, *starts = ws + vocab + 1, *cursor = starts + vocab + 1, *bucket = cursor + vocab + 1;
    if (hipMemsetAsync(counts, 0, (size_t)(vocab + 1) * 4, st) != hipSuccess) return UNIIR_ELAUNCH;
    const int g = grid_for(rows, 256, 4096);
    hipLaunchKernelGGL(tokd_count_kernel, dim3(g), dim3(256), 0, st, text, row_off, rows, ctx, vocab, counts);
    hipLaunchKernelGGL(tokd_scan_kernel, dim3(1), dim3(1024), 0, st, counts, starts, cursor, vocab);
    hipLaunchKernelGGL(tokd_fill_kernel, dim3(g), dim3(256), 0, st, text, row_off, rows, ctx, vocab, cursor, bucket);
    hipLaunchKernelGGL(tokd_accum_kernel, dim3(vocab, (w + 255) / 256), dim3(256), 0, st, starts, bucket, dx, dtok, w);
    return hipGetLastError() == hipSuccess ? 1 : UNIIR_ELAUNCH;
}

// backward: dtok[text[n][t]] += dx[n][t] (atomics: ids repeat); dpos[t] += sum_n dx[n][t]
__global__ __launch_bounds__(256) void text_embed_bwd_tok_kernel(const int* __restrict__ text,
                                                                 const float* __restrict__ dx,
                                                                 float* __restrict__ dtok, int n, int ctx,
                                                                 int w, int vocab) {
    const long total = (long)n * ctx * w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int col = (int)(i % w);
        const long row = i / w;
        int id = text[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        unsafeAtomicAdd(dtok + (long)id * w + col, dx[i]);
    }
}
__global__ __launch_bounds__(256) void text_embed_bwd_pos_kernel(const float* __restrict__ dx,
                                                                 float* __restrict__ dpos, int n, int ctx,
                                                                 int w) {
    const int t = blockIdx.x;
    for (int col = blockIdx.y * 256 + threadIdx.x; col < w; col += gridDim.y * 256) {
        float s = 0.f;
        for (int im = 0; im < n; ++im) s += dx[((long)im * ctx + t) * w + col];
        dpos[(long)t * w + col] += s;
    }
}

extern "C" int uniir_text_embed_bwd(const int32_t* text, const float* dx, float* dtoken_emb, float* dpos,
                                    int32_t n, int32_t ctx, int32_t width, int32_t vocab, void* stream) {
    if (!text || !dx || !dtoken_emb || !dpos || n < 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    const long total = (long)n * ctx * width;
    const int det = tokd_run(text, dx, nullptr, dtoken_emb, n, ctx, width, vocab, (hipStream_t)stream);
    if (det < 0) return det;
    if (!det)
        hipLaunchKernelGGL(text_embed_bwd_tok_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0,
                           (hipStream_t)stream, text, dx, dtoken_emb, n, ctx, width, vocab);
    hipLaunchKernelGGL(text_embed_bwd_pos_kernel, dim3(ctx, (width + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, dx, dpos, n, ctx, width);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// packed text rows: item n's tokens 0 .. len - 1 (len = row_off[n + 1] - row_off[n]: up to and including its EOT) are the rows
// row_off[n] .. row_off[n + 1] - 1 of x; last[n] = its EOT row (for uniir_gather_rows / uniir_scatter_rows with seq = 0)
__global__ __launch_bounds__(256) void text_embed_packed_kernel(const int* __restrict__ text, const float* __restrict__ tok,
                                                                const float* __restrict__ pos, const int* __restrict__ row_off,
                                                                float* __restrict__ x, int* __restrict__ last, int n, int ctx,
                                                                int w, int vocab) {
    const int wc = w >> 2;
    const long total = (long)n * ctx * wc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % wc);
        const long row = i / wc;
        const int t = (int)(row % ctx), m = (int)(row / ctx);
        const int r0 = row_off[m], len = row_off[m + 1] - r0;
        if (t == 0 && c == 0 && last) last[m] = r0 + len - 1;
        if (t >= len) continue;
        int id = text[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        f32x4_t v = *reinterpret_cast<const f32x4_t*>(tok + (long)id * w + 4 * c);
        v += *reinterpret_cast<const f32x4_t*>(pos + (long)t * w + 4 * c);
        *reinterpret_cast<f32x4_t*>(x + (long)(r0 + t) * w + 4 * c) = v;
    }
}
extern "C" int uniir_text_embed_packed(const int32_t* text, const float* token_emb, const float* pos_emb, const int32_t* row_off,
                                       float* x, int32_t* last_row, int32_t n, int32_t ctx, int32_t width, int32_t vocab,
                                       void* stream) {
    if (!text || !token_emb || !pos_emb || !row_off || !x || n < 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    if (width % 4) return UNIIR_ESHAPE;
    const long total = (long)n * ctx * (width / 4);
    hipLaunchKernelGGL(text_embed_packed_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, text,
                       token_emb, pos_emb, row_off, x, last_row, n, ctx, width, vocab);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
__global__ __launch_bounds__(256) void text_embed_bwd_tok_packed_kernel(const int* __restrict__ text, const float* __restrict__ dx,
                                                                        const int* __restrict__ row_off, float* __restrict__ dtok,
                                                                        int n, int ctx, int w, int vocab) {
    const long total = (long)n * ctx * w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int col = (int)(i % w);
        const long row = i / w;
        const int t = (int)(row % ctx), m = (int)(row / ctx);
        const int r0 = row_off[m];
        if (t >= row_off[m + 1] - r0) continue;
        int id = text[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        unsafeAtomicAdd(dtok + (long)id * w + col, dx[(long)(r0 + t) * w + col]);
    }
}
__global__ __launch_bounds__(256) void text_embed_bwd_pos_packed_kernel(const float* __restrict__ dx, const int* __restrict__ row_off,
                                                                        float* __restrict__ dpos, int n, int ctx, int w) {
    const int t = blockIdx.x;
    for (int col = blockIdx.y * 256 + threadIdx.x; col < w; col += gridDim.y * 256) {
        float s = 0.f;
        for (int im = 0; im < n; ++im) {           // the dense kernel's order over the items; items shorter than t + 1 add nothing
            const int r0 = row_off[im];
            if (t < row_off[im + 1] - r0) s += dx[(long)(r0 + t) * w + col];
        }
        dpos[(long)t * w + col] += s;
    }
}
extern "C" int uniir_text_embed_bwd_packed(const int32_t* text, const float* dx, const int32_t* row_off, float* dtoken_emb,
                                           float* dpos, int32_t n, int32_t ctx, int32_t width, int32_t vocab, void* stream) {
    if (!text || !dx || !row_off || !dtoken_emb || !dpos || n < 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    const long total = (long)n * ctx * width;
    const int det = tokd_run(text, dx, row_off, dtoken_emb, n, ctx, width, vocab, (hipStream_t)stream);
    if (det < 0) return det;
    if (!det)
        hipLaunchKernelGGL(text_embed_bwd_tok_packed_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, text,
                           dx, row_off, dtoken_emb, n, ctx, width, vocab);
    hipLaunchKernelGGL(text_embed_bwd_pos_packed_kernel, dim3(ctx, (width + 255) / 256), dim3(256), 0, (hipStream_t)stream, dx,
                       row_off, dpos, n, ctx, width);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x,
                                                          const int* __restrict__ idx,
                                                          float* __restrict__ out, int n, int seq, int w,
                                                          int scatter_add) {
    const int wc = w >> 2;
    const long total = (long)n * wc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % wc);
        const long r = i / wc;
        const long src = r * seq + (idx ? idx[r] : 0);
        if (!scatter_add) {
            *reinterpret_cast<f32x4_t*>(out + r * w + 4 * c) = *reinterpret_cast<const f32x4_t*>(x + src * w + 4 * c);
        } else {  // here "x" is dout [n][w] and "out" is dx [n*seq][w]
            f32x4_t* d = reinterpret_cast<f32x4_t*>(out + src * w + 4 * c);
            *d = *d + *reinterpret_cast<const f32x4_t*>(x + r * w + 4 * c);
        }
    }
}
extern "C" int uniir_gather_rows(const float* x, const int32_t* idx, float* out, int32_t n, int32_t seq,
                                 int32_t width, void* stream) {
    if (!x || !out || n < 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    if (width % 4) return UNIIR_ESHAPE;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)n * (width / 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, x, idx, out, n, seq, width, 0);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
// one row per item between a [n * seq]-row tensor and an [n]-row tensor, as raw 16-byte pieces (any element type): gather (dir 0:
// small[i] = big[i * seq + idx[i]]) or scatter-overwrite (dir 1: big[i * seq + idx[i]] = small[i]); idx NULL = 0.  Row pitches in
// bytes, row_bytes a multiple of 16.  (tower.hip: the pooled rows of the last block.)
__global__ __launch_bounds__(256) void rows_copy_kernel(const char* __restrict__ src, const int* __restrict__ idx, char* __restrict__ dst,
                                                        int n, int seq, int pieces, long big_pitch, long small_pitch, int dir) {
    const long total = (long)n * pieces;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % pieces);
        const long r = i / pieces;
        const long big = (r * seq + (idx ? idx[r] : 0)) * big_pitch + 16L * c, small = r * small_pitch + 16L * c;
        if (!dir) *reinterpret_cast<u32x4_t*>(dst + small) = *reinterpret_cast<const u32x4_t*>(src + big);
        else *reinterpret_cast<u32x4_t*>(dst + big) = *reinterpret_cast<const u32x4_t*>(src + small);
    }
}
int rows_copy_impl(const void* src, const int32_t* idx, void* dst, int32_t n, int32_t seq, int32_t row_bytes, int64_t big_pitch,
                   int64_t small_pitch, int32_t scatter, void* stream) {
    if (!src || !dst || n < 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    if (row_bytes <= 0 || (row_bytes % 16) || (big_pitch % 16) || (small_pitch % 16)) return UNIIR_ESHAPE;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return UNIIR_EALIGN;
    hipLaunchKernelGGL(rows_copy_kernel, dim3(grid_for((long)n * (row_bytes / 16), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, idx, (char*)dst, n, seq, row_bytes / 16, (long)big_pitch, (long)small_pitch, scatter);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
extern "C" int uniir_scatter_rows(const float* dout, const int32_t* idx, float* dx, int32_t n, int32_t seq,
                                  int32_t width, void* stream) {
    if (!dout || !dx || n < 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    if (width % 4) return UNIIR_ESHAPE;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)n * (width / 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, dout, idx, dx, n, seq, width, 1);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// ---------------------------------------------------------------------------------------------------
DEVINL float act_fwd_e(float x, int act) { return act_fwd(x, act); }      // common.h: the GEMM epilogues' function, bit for bit
__global__ __launch_bounds__(256) void act_fwd_kernel(const u32x4_t* __restrict__ f, u32x4_t* __restrict__ g,
                                                      long nvec, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const u32x4_t a = f[i];
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = act_fwd_e(__uint_as_float(a[e] << 16), act);
            const float hi = act_fwd_e(__uint_as_float(a[e] & 0xffff0000u), act);
            o[e] = pack_bf16x2(lo, hi);
        }
        g[i] = o;
    }
}
extern "C" int uniir_act_fwd(const void* f_bf16, void* g_bf16, int64_t count, int32_t act, void* stream) {
    if (!f_bf16 || !g_bf16 || count < 0) return UNIIR_EINVAL;
    if (count == 0) return UNIIR_OK;
    if (count % 8) return UNIIR_ESHAPE;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(count / 8, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const u32x4_t*)f_bf16, (u32x4_t*)g_bf16, (long)(count / 8), act);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// colsum: out[c] += sum_r x[r][c].  Block = 256 threads covers 64 column-chunks of 8 (512 cols) x 4 row lanes.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const unsigned short* __restrict__ x, long ld,
                                                          float* __restrict__ out, int rows, int cols,
                                                          int rows_per_block, float* __restrict__ part) {
    __shared__ float red[4][512];
    const int cchunk = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + cchunk * 8;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < cols) {
        for (int r = r0 + rl; r < r1; r += 4) {
            const u32x4_t a = *reinterpret_cast<const u32x4_t*>(x + (long)r * ld + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += __uint_as_float(a[e] << 16);
                acc[2 * e + 1] += __uint_as_float(a[e] & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cchunk * 8 + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int gc = blockIdx.x * 512 + c;
        if (gc < cols) {
            const float t = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
            if (part) part[(long)blockIdx.y * cols + gc] = t;          // fixed-order reduction follows (reduce_partials)
            else unsafeAtomicAdd(out + gc, t);
        }
    }
}
extern "C" int uniir_colsum_bf16(const void* x, int64_t ld, float* out, int32_t rows, int32_t cols,
                                 void* stream) {
    if (!x || !out || rows < 0 || cols <= 0) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    if (cols % 8 || ld % 8) return UNIIR_ESHAPE;
    const int gx = (cols + 511) / 512;
    int gy = 2048 / gx;
    if (gy < 1) gy = 1;
    int rpb = (rows + gy - 1) / gy;
    if (rpb < 64) rpb = 64;
    gy = (rows + rpb - 1) / rpb;
    // several row blocks per column: their partial sums go through the stream's scratch and are added in a fixed order
    float* part = gy > 1 ? reduce_scratch((hipStream_t)stream, (int64_t)gy * cols * 4) : nullptr;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x, (long)ld, out, rows, cols, rpb, part);
    HIP_LAUNCH_CHECK();
    if (part) return reduce_partials(part, gy, cols, cols, out, nullptr, nullptr, 0, (hipStream_t)stream);
    return UNIIR_OK;
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ s,
                                                            unsigned short* __restrict__ d, long n, int f16 = 0) {
    const long nv = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(s + 4 * i);
        u32x2_t o = {pack16x2(v[0], v[1], f16 != 0), pack16x2(v[2], v[3], f16 != 0)};
        *reinterpret_cast<u32x2_t*>(d + 4 * i) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float x = s[(nv << 2) + threadIdx.x];
        d[(nv << 2) + threadIdx.x] = f16 ? f32_to_f16(x) : f32_to_bf16(x);
    }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const unsigned short* __restrict__ s,
                                                            float* __restrict__ d, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = bf16_to_f32(s[i]);
}
extern "C" int uniir_cast_f32_to_bf16(const float* src, void* dst, int64_t count, void* stream) {
    if (!src || !dst || count < 0) return UNIIR_EINVAL;
    if (count == 0) return UNIIR_OK;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return UNIIR_EALIGN;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(count / 4 + 1, 256, 16384)), dim3(256), 0,
                       (hipStream_t)stream, src, (unsigned short*)dst, (long)count, 0);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
// fp16 shadow of the master weights for the fp16 forward (uniir_clip_tower.dtype16 = 1)
extern "C" int uniir_cast_f32_to_f16(const float* src, void* dst, int64_t count, void* stream) {
    if (!src || !dst || count < 0) return UNIIR_EINVAL;
    if (count == 0) return UNIIR_OK;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return UNIIR_EALIGN;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(count / 4 + 1, 256, 16384)), dim3(256), 0,
                       (hipStream_t)stream, src, (unsigned short*)dst, (long)count, 1);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
extern "C" int uniir_cast_bf16_to_f32(const void* src, float* dst, int64_t count, void* stream) {
    if (!src || !dst || count < 0) return UNIIR_EINVAL;
    if (count == 0) return UNIIR_OK;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(count, 256, 16384)), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned short*)src, dst, (long)count);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
__global__ __launch_bounds__(256) void cast_pad_rows_kernel(const float* __restrict__ s,
                                                            unsigned short* __restrict__ d, int rows, int cols,
                                                            int ld, int f16) {
    const long total = (long)rows * ld;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % ld);
        const long r = i / ld;
        const float x = c < cols ? s[r * cols + c] : 0.0f;
        d[i] = f16 ? f32_to_f16(x) : f32_to_bf16(x);
    }
}
static int cast_pad_rows_impl(const float* src, void* dst, int32_t rows, int32_t cols, int32_t ld_dst, int f16, void* stream) {
    if (!src || !dst || rows < 0 || cols <= 0 || ld_dst < cols) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    hipLaunchKernelGGL(cast_pad_rows_kernel, dim3(grid_for((long)rows * ld_dst, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, (unsigned short*)dst, rows, cols, ld_dst, f16);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
extern "C" int uniir_cast_pad_rows(const float* src, void* dst, int32_t rows, int32_t cols, int32_t ld_dst,
                                   void* stream) {
    return cast_pad_rows_impl(src, dst, rows, cols, ld_dst, 0, stream);
}
extern "C" int uniir_cast_pad_rows_f16(const float* src, void* dst, int32_t rows, int32_t cols, int32_t ld_dst,
                                       void* stream) {
    return cast_pad_rows_impl(src, dst, rows, cols, ld_dst, 1, stream);
}
__global__ __launch_bounds__(256) void unpad_add_kernel(const float* __restrict__ s, float* __restrict__ d,
                                                        int rows, int cols, int ld) {
    const long total = (long)rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cols);
        const long r = i / cols;
        d[i] += s[r * ld + c];
    }
}
extern "C" int uniir_unpad_add(const float* src, float* dst, int32_t rows, int32_t cols, int32_t ld_src,
                               void* stream) {
    if (!src || !dst || rows < 0 || cols <= 0 || ld_src < cols) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    hipLaunchKernelGGL(unpad_add_kernel, dim3(grid_for((long)rows * cols, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, dst, rows, cols, ld_src);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// ---------------------------------------------------------------------------------------------------
// fuse: emb = txt * tmask + img * imask   (clip_sf.py:61-63)
__global__ __launch_bounds__(256) void fuse_kernel(const float* __restrict__ t, const float* __restrict__ im,
                                                   const long long* __restrict__ tm,
                                                   const long long* __restrict__ imk, float* __restrict__ e,
                                                   int n, int dim) {
    const long total = (long)n * dim;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / dim;
        // same association as the reference: fuse_embeddings(txt_emb, img_emb) = img_emb' + txt_emb'
        // with img_emb' := txt*mask (first arg) -> (txt*tm) + (img*im); fp32 add is commutative.
        e[i] = t[i] * (float)tm[r] + im[i] * (float)imk[r];
    }
}
extern "C" int uniir_fuse_embeddings(const float* txt_emb, const float* img_emb, const int64_t* txt_mask,
                                     const int64_t* img_mask, float* emb, int32_t n, int32_t dim,
                                     void* stream) {
    if (!txt_emb || !img_emb || !txt_mask || !img_mask || !emb || n < 0 || dim <= 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    hipLaunchKernelGGL(fuse_kernel, dim3(grid_for((long)n * dim, 256)), dim3(256), 0, (hipStream_t)stream,
                       txt_emb, img_emb, (const long long*)txt_mask, (const long long*)img_mask, emb, n, dim);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
__global__ __launch_bounds__(256) void fuse_bwd_kernel(const float* __restrict__ de,
                                                       const long long* __restrict__ tm,
                                                       const long long* __restrict__ imk,
                                                       float* __restrict__ dt, float* __restrict__ di, int n,
                                                       int dim) {
    const long total = (long)n * dim;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / dim;
        const float g = de[i];
        dt[i] = g * (float)tm[r];
        di[i] = g * (float)imk[r];
    }
}
extern "C" int uniir_fuse_embeddings_bwd(const float* demb, const int64_t* txt_mask, const int64_t* img_mask,
                                         float* dtxt, float* dimg, int32_t n, int32_t dim, void* stream) {
    if (!demb || !txt_mask || !img_mask || !dtxt || !dimg || n < 0 || dim <= 0) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    hipLaunchKernelGGL(fuse_bwd_kernel, dim3(grid_for((long)n * dim, 256)), dim3(256), 0, (hipStream_t)stream,
                       demb, (const long long*)txt_mask, (const long long*)img_mask, dtxt, dimg, n, dim);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// select + normalize: out[i] = emb[idx[i]] / max(||emb[idx[i]]||_2, 1e-12)   (F.normalize, clip_sf.py:96-97)
// one wave per row.
__global__ __launch_bounds__(256) void select_norm_kernel(const float* __restrict__ emb,
                                                          const int* __restrict__ idx,
                                                          float* __restrict__ out, float* __restrict__ inv,
                                                          int rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* src = emb + (long)(idx ? idx[r] : r) * dim;
    float s = 0.f;
    for (int c = lane; c < dim; c += 64) { const float v = src[c]; s += v * v; }
    s = wave_sum(s);
    const float nrm = fmaxf(sqrtf(s), 1e-12f);
    const float iv = 1.0f / nrm;
    for (int c = lane; c < dim; c += 64) out[(long)r * dim + c] = src[c] / nrm;
    if (lane == 0 && inv) inv[r] = iv;
}
extern "C" int uniir_select_normalize(const float* emb, const int32_t* idx, float* out, float* inv_norm,
                                      int32_t rows, int32_t dim, void* stream) {
    if (!emb || !out || rows < 0 || dim <= 0) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    hipLaunchKernelGGL(select_norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, emb, idx,
                       out, inv_norm, rows, dim);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
// bwd of y = x/||x||: dx = (dy - y <y,dy>) / ||x||; scattered (+=) to demb[idx[i]] with atomics (an item may
// be selected more than once, e.g. as hard negative of several queries).
__global__ __launch_bounds__(256) void select_norm_bwd_kernel(const float* __restrict__ y,
                                                              const float* __restrict__ inv,
                                                              const float* __restrict__ dy,
                                                              const int* __restrict__ idx,
                                                              float* __restrict__ demb, int rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* yr = y + (long)r * dim;
    const float* dr = dy + (long)r * dim;
    float s = 0.f;
    for (int c = lane; c < dim; c += 64) s += yr[c] * dr[c];
    s = wave_sum(s);
    const float iv = inv[r];
    float* dst = demb + (long)(idx ? idx[r] : r) * dim;
    for (int c = lane; c < dim; c += 64) unsafeAtomicAdd(dst + c, (dr[c] - yr[c] * s) * iv);
}
extern "C" int uniir_select_normalize_bwd(const float* out, const float* inv_norm, const float* dout,
                                          const int32_t* idx, float* demb, int32_t rows, int32_t dim,
                                          void* stream) {
    if (!out || !inv_norm || !dout || !demb || rows < 0 || dim <= 0) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    hipLaunchKernelGGL(select_norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, out,
                       inv_norm, dout, idx, demb, rows, dim);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// ---------------------------------------------------------------------------------------------------
// fused AdamW (torch.optim.AdamW single-tensor semantics) + bf16 shadow refresh. 16-24 B/param of traffic.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    unsigned short* __restrict__ pb, long n, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float bc2s,
                                                    float gscale) {
    const long nv = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        f32x4_t pp = *reinterpret_cast<f32x4_t*>(p + 4 * i);
        const f32x4_t gg = *reinterpret_cast<const f32x4_t*>(g + 4 * i) * gscale;
        f32x4_t mm = *reinterpret_cast<f32x4_t*>(m + 4 * i);
        f32x4_t vv = *reinterpret_cast<f32x4_t*>(v + 4 * i);
        pp = pp * (1.0f - lr * wd);
        mm = mm * b1 + gg * (1.0f - b1);
        vv = vv * b2 + gg * gg * (1.0f - b2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float denom = sqrtf(vv[e]) / bc2s + eps;
            pp[e] -= (lr / bc1) * (mm[e] / denom);
        }
        *reinterpret_cast<f32x4_t*>(p + 4 * i) = pp;
        *reinterpret_cast<f32x4_t*>(m + 4 * i) = mm;
        *reinterpret_cast<f32x4_t*>(v + 4 * i) = vv;
        if (pb) {
            u32x2_t o = {pack_bf16x2(pp[0], pp[1]), pack_bf16x2(pp[2], pp[3])};
            *reinterpret_cast<u32x2_t*>(pb + 4 * i) = o;
        }
    }
    // tail (count % 4)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (nv << 2) + threadIdx.x;
        float pp = p[i] * (1.0f - lr * wd);
        const float gg = g[i] * gscale;
        const float mm = m[i] * b1 + gg * (1.0f - b1);
        const float vv = v[i] * b2 + gg * gg * (1.0f - b2);
        pp -= (lr / bc1) * (mm / (sqrtf(vv) / bc2s + eps));
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (pb) pb[i] = f32_to_bf16(pp);
    }
}
extern "C" int uniir_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                void* param_bf16, int64_t count, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int32_t step, float grad_scale, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || count < 0 || step < 1) return UNIIR_EINVAL;
    if (count == 0) return UNIIR_OK;
    if (((uintptr_t)param & 15) || ((uintptr_t)grad & 15) || ((uintptr_t)exp_avg & 15) ||
        ((uintptr_t)exp_avg_sq & 15) || ((uintptr_t)param_bf16 & 7))
        return UNIIR_EALIGN;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(count / 4 + 1, 256, 16384)), dim3(256), 0,
                       (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (unsigned short*)param_bf16,
                       (long)count, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// ---------------------------------------------------------------------------------------------------
// BLIP pieces (uniir_blip): tanh pooler (med.py:499-511), momentum EMA (blip_ff.py:288-292) and the soft-target
// contrastive loss over [in-batch | queue] similarities (blip_ff.py:219-231).
__global__ __launch_bounds__(256) void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = tanhf(x[i]);
}
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                       float* __restrict__ dx, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float t = y[i];
        dx[i] = dy[i] * (1.0f - t * t);
    }
}
extern "C" int uniir_tanh_fwd(const float* x, float* y, int64_t count, void* stream) {
    if (!x || !y || count < 0) return UNIIR_EINVAL;
    if (count == 0) return UNIIR_OK;
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(grid_for(count, 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long)count);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
extern "C" int uniir_tanh_bwd(const float* y, const float* dy, float* dx, int64_t count, void* stream) {
    if (!y || !dy || !dx || count < 0) return UNIIR_EINVAL;
    if (count == 0) return UNIIR_OK;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(count, 256)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, (long)count);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// p_m = m * p_m + (1 - m) * p over a flat buffer; refreshes the momentum model's bf16 shadow in the same pass
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ pm, const float* __restrict__ p,
                                                  unsigned short* __restrict__ pm16, long n, float m) {
    const long nv = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        f32x4_t a = *reinterpret_cast<f32x4_t*>(pm + 4 * i);
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(p + 4 * i);
        a = a * m + b * (1.0f - m);
        *reinterpret_cast<f32x4_t*>(pm + 4 * i) = a;
        if (pm16) {
            u32x2_t o = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])};
            *reinterpret_cast<u32x2_t*>(pm16 + 4 * i) = o;
        }
    }
}
extern "C" int uniir_ema_update(float* param_m, const float* param, void* param_m_bf16, int64_t count, float momentum,
                                void* stream) {
    if (!param_m || !param || count < 0) return UNIIR_EINVAL;
    if (count == 0) return UNIIR_OK;
    if (count % 4 || ((uintptr_t)param_m & 15) || ((uintptr_t)param & 15)) return UNIIR_EALIGN;
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(count / 4, 256, 16384)), dim3(256), 0, (hipStream_t)stream, param_m,
                       param, (unsigned short*)param_m_bf16, (long)count, momentum);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// soft-target contrastive loss, one block per row i of sim [b][n]:
//   pos_j = (ids_all[j] == ids_row[i]);  target = alpha * softmax(sim_m[i]) + (1 - alpha) * pos / sum(pos)
//   loss_i = -sum_j log_softmax(sim[i])_j * target_j ;  dsim[i][j] = (softmax(sim[i])_j - target_j) * gscale
// (sum_j target_j = 1).  hit_i = pos[argmax_j sim[i][j]] (first max), used for the accuracy of blip_ff.py:250-252.
__global__ __launch_bounds__(256) void softce_kernel(const float* __restrict__ sim, const float* __restrict__ sim_m,
                                                     const float* __restrict__ temp,
                                                     const long long* __restrict__ ids_row,
                                                     const long long* __restrict__ ids_all, int n, float alpha,
                                                     float gscale, const float* __restrict__ dloss,
                                                     float* __restrict__ row_loss, float* __restrict__ row_hit,
                                                     float* __restrict__ dsim, float* __restrict__ row_dtemp) {
    __shared__ float red[4];
    __shared__ int redi[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* s = sim + (long)i * n;
    const float* sm = sim_m + (long)i * n;
    const float T = temp ? *temp : 1.0f;   // logits are dot / temp (a division, like blip_ff.py:219-223)
    const long long my = ids_row[i];
    auto block_max = [&](float v) {
        v = wave_max(v);
        if (lane == 0) red[w] = v;
        __syncthreads();
        const float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        return r;
    };
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        if (lane == 0) red[w] = v;
        __syncthreads();
        const float r = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
        return r;
    };
    float mx = -INFINITY, mxm = -INFINITY, npos = 0.f;
    int am = 0x7fffffff;
    for (int j = tid; j < n; j += 256) {
        const float v = s[j] / T;
        if (v > mx) { mx = v; am = j; }
        mxm = fmaxf(mxm, sm[j] / T);
        npos += (ids_all[j] == my) ? 1.0f : 0.0f;
    }
    // first-index argmax over the block
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
    mxm = block_max(mxm);
    npos = block_sum(npos);
    float se = 0.f, sem = 0.f;
    for (int j = tid; j < n; j += 256) { se += expf(s[j] / T - mx); sem += expf(sm[j] / T - mxm); }
    se = block_sum(se);
    sem = block_sum(sem);
    const float lse = mx + logf(se), inv_sem = 1.0f / sem, inv_pos = npos > 0.f ? 1.0f / npos : 0.f;
    const float gs = gscale * (dloss ? *dloss : 1.0f);
    float loss = 0.f, dT = 0.f;
    for (int j = tid; j < n; j += 256) {
        const float sj = s[j] / T;
        const float lsm = sj - lse;
        const float tgt = alpha * (expf(sm[j] / T - mxm) * inv_sem) + (1.0f - alpha) * ((ids_all[j] == my) ? inv_pos : 0.f);
        loss -= lsm * tgt;
        const float g = (expf(lsm) - tgt) * gs;        // d loss / d logit_j
        if (dsim) dsim[(long)i * n + j] = g / T;       // d loss / d dot_j
        dT -= g * sj;                                  // d logit_j / d temp = -logit_j / temp
    }
    loss = block_sum(loss);
    dT = block_sum(dT);
    if (tid == 0) {
        row_loss[i] = loss;
        row_hit[i] = (ids_all[am] == my) ? 1.0f : 0.0f;
        if (row_dtemp) row_dtemp[i] = dT / T;
    }
}
extern "C" int uniir_softce(const float* sim, const float* sim_m, const float* temp, const int64_t* ids_row,
                            const int64_t* ids_all, int32_t b, int32_t n, float alpha, float gscale,
                            const float* dloss, float* row_loss, float* row_hit, float* dsim, float* row_dtemp,
                            void* stream) {
    if (!sim || !sim_m || !ids_row || !ids_all || !row_loss || !row_hit || b <= 0 || n <= 0) return UNIIR_EINVAL;
    hipLaunchKernelGGL(softce_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, sim, sim_m, temp,
                       (const long long*)ids_row, (const long long*)ids_all, n, alpha, gscale, dloss, row_loss, row_hit,
                       dsim, row_dtemp);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// ------------------------------------------------------------------------------------------------------------
// [CLIP_FF] mean over the tokens of each item (clip_ff.py:186-191) and its backward.
//   fwd: out[n][w] = (1/T) sum_t x[n][t][w];  bwd: dx[n][t][w] = dout[n][w] / T (fp32)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void meanpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int W) {
    const int n = blockIdx.x;
    const float inv = 1.0f / (float)T;
    for (int c = threadIdx.x; c < W / 4; c += 256) {
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
        const float* base = x + (long)n * T * W + 4 * c;
        for (int t = 0; t < T; ++t) s += *reinterpret_cast<const f32x4_t*>(base + (long)t * W);
        *reinterpret_cast<f32x4_t*>(out + (long)n * W + 4 * c) = s * inv;
    }
}
__global__ __launch_bounds__(256) void meanpool_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int T, int W) {
    const int n = blockIdx.x;
    const float inv = 1.0f / (float)T;
    for (int c = threadIdx.x; c < W / 4; c += 256) {
        const f32x4_t g = *reinterpret_cast<const f32x4_t*>(dout + (long)n * W + 4 * c) * inv;
        float* base = dx + (long)n * T * W + 4 * c;
        for (int t = blockIdx.y; t < T; t += gridDim.y) *reinterpret_cast<f32x4_t*>(base + (long)t * W) = g;
    }
}
extern "C" int uniir_meanpool_fwd(const float* x, float* out, int32_t n, int32_t tokens, int32_t width, void* stream) {
    if (!x || !out || n < 0 || tokens <= 0 || width <= 0 || width % 4) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    hipLaunchKernelGGL(meanpool_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, x, out, tokens, width);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
extern "C" int uniir_meanpool_bwd(const float* dout, float* dx, int32_t n, int32_t tokens, int32_t width, void* stream) {
    if (!dout || !dx || n < 0 || tokens <= 0 || width <= 0 || width % 4) return UNIIR_EINVAL;
    if (n == 0) return UNIIR_OK;
    hipLaunchKernelGGL(meanpool_bwd_kernel, dim3(n, 8), dim3(256), 0, (hipStream_t)stream, dout, dx, tokens, width);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

// ------------------------------------------------------------------------------------------------------------
// [DROPOUT] train-mode dropout of the BLIP MED BERT (hidden_dropout_prob / attention_probs_dropout_prob 0.1,
// backbone/configs/med_config.json), the T5 fusion stack (dropout_rate 0.1) and DropPath of BLIP's ViT-large
// (backbone/blip.py:245-254).  Element masks are regenerated from (seed, element index) -- see common.h.
//   dropout_f32 : y = (resid ? resid : 0) + x * mask(idx) [* rowscale[row / div]]   -> fp32 and / or bf16 copies
//   dropout_bf16: y = x * mask(idx) [* rowscale[row / div]]  (bf16 -> bf16; the masked gradient fed to wgrad / dgrad)
//   dropout_mask: out[idx] = mask value (tests)
// idx = row * cols + col of the logical [rows][cols] tensor (ld = row pitch of x / y in elements).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* __restrict__ x, const float* __restrict__ resid,
                                                          float* __restrict__ y32, unsigned short* __restrict__ y16,
                                                          long rows, int cols, float p, unsigned seed,
                                                          const float* __restrict__ rowscale, int div,
                                                          const int* __restrict__ row_map) {
    const unsigned th = drop_threshold(p);
    const float ks = 1.0f / (1.0f - p);
    const long nv = rows * (cols / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const long row = i / (cols / 4);
        const int c = (int)(i - row * (cols / 4)) * 4;
        const long e = row * cols + c;
        const long me = row_map ? (long)row_map[row] * cols + c : e;      // the mask's element index: the LOGICAL (dense) row
        f32x4_t v = *reinterpret_cast<const f32x4_t*>(x + e);
        const float rs = rowscale ? rowscale[row / div] : 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= (p > 0.f ? drop_scale((unsigned)(me + k), seed, th, ks) : 1.0f) * rs;
        if (resid) v += *reinterpret_cast<const f32x4_t*>(resid + e);
        if (y32) *reinterpret_cast<f32x4_t*>(y32 + e) = v;
        if (y16) {
            const u32x2_t pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *reinterpret_cast<u32x2_t*>(y16 + e) = pk;
        }
    }
}
__global__ __launch_bounds__(256) void dropout_bf16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                           long rows, int cols, long ld, float p, unsigned seed,
                                                           const float* __restrict__ rowscale, int div,
                                                           const int* __restrict__ row_map) {
    const unsigned th = drop_threshold(p);
    const float ks = 1.0f / (1.0f - p);
    const long nv = rows * (cols / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const long row = i / (cols / 4);
        const int c = (int)(i - row * (cols / 4)) * 4;
        const long e = (row_map ? (long)row_map[row] : row) * cols + c;
        const u32x2_t a = *reinterpret_cast<const u32x2_t*>(x + row * ld + c);
        float v[4] = {__uint_as_float(a[0] << 16), __uint_as_float(a[0] & 0xffff0000u), __uint_as_float(a[1] << 16),
                      __uint_as_float(a[1] & 0xffff0000u)};
        const float rs = rowscale ? rowscale[row / div] : 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= (p > 0.f ? drop_scale((unsigned)(e + k), seed, th, ks) : 1.0f) * rs;
        const u32x2_t pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2_t*>(y + row * ld + c) = pk;
    }
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, long count, float p, unsigned seed) {
    const unsigned th = drop_threshold(p);
    const float ks = 1.0f / (1.0f - p);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256)
        out[i] = drop_scale((unsigned)i, seed, th, ks);
}
extern "C" int uniir_dropout_f32(const float* x, const float* resid, float* y_f32, void* y_bf16, int64_t rows, int32_t cols,
                                 float p, uint32_t seed, const float* rowscale, int32_t rows_per_scale, void* stream) {
    if (!x || (!y_f32 && !y_bf16) || rows < 0 || cols <= 0 || cols % 4 || p < 0.f || p >= 1.f) return UNIIR_EINVAL;
    if (rowscale && rows_per_scale <= 0) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    hipLaunchKernelGGL(dropout_f32_kernel, dim3(grid_for(rows * (cols / 4), 256, 16384)), dim3(256), 0, (hipStream_t)stream, x,
                       resid, y_f32, (unsigned short*)y_bf16, (long)rows, cols, p, seed, rowscale, rows_per_scale > 0 ? rows_per_scale : 1,
                       (const int*)nullptr);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
// the same on PACKED rows: row r of x / resid / y is row row_map[r] of the logical (dense) tensor the mask is defined on, so a packed
// BERT draws exactly the mask elements its live rows have in the padded batch
extern "C" int uniir_dropout_f32_rows(const float* x, const float* resid, float* y_f32, void* y_bf16, int64_t rows, int32_t cols,
                                      float p, uint32_t seed, const int32_t* row_map, void* stream) {
    if (!x || (!y_f32 && !y_bf16) || !row_map || rows < 0 || cols <= 0 || cols % 4 || p < 0.f || p >= 1.f) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    hipLaunchKernelGGL(dropout_f32_kernel, dim3(grid_for(rows * (cols / 4), 256, 16384)), dim3(256), 0, (hipStream_t)stream, x,
                       resid, y_f32, (unsigned short*)y_bf16, (long)rows, cols, p, seed, (const float*)nullptr, 1, row_map);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
extern "C" int uniir_dropout_bf16_rows(const void* x, void* y, int64_t rows, int32_t cols, int64_t ld, float p, uint32_t seed,
                                       const int32_t* row_map, void* stream) {
    if (!x || !y || !row_map || rows < 0 || cols <= 0 || cols % 4 || ld % 4 || p < 0.f || p >= 1.f) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    hipLaunchKernelGGL(dropout_bf16_kernel, dim3(grid_for(rows * (cols / 4), 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x, (unsigned short*)y, (long)rows, cols, (long)ld, p, seed, (const float*)nullptr, 1,
                       row_map);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
extern "C" int uniir_dropout_bf16(const void* x, void* y, int64_t rows, int32_t cols, int64_t ld, float p, uint32_t seed,
                                  const float* rowscale, int32_t rows_per_scale, void* stream) {
    if (!x || !y || rows < 0 || cols <= 0 || cols % 4 || ld % 4 || p < 0.f || p >= 1.f) return UNIIR_EINVAL;
    if (rowscale && rows_per_scale <= 0) return UNIIR_EINVAL;
    if (rows == 0) return UNIIR_OK;
    hipLaunchKernelGGL(dropout_bf16_kernel, dim3(grid_for(rows * (cols / 4), 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x, (unsigned short*)y, (long)rows, cols, (long)ld, p, seed, rowscale,
                       rows_per_scale > 0 ? rows_per_scale : 1, (const int*)nullptr);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
extern "C" int uniir_dropout_mask(float* out, int64_t count, float p, uint32_t seed, void* stream) {
    if (!out || count < 0 || p < 0.f || p >= 1.f) return UNIIR_EINVAL;
    if (count == 0) return UNIIR_OK;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(count, 256, 16384)), dim3(256), 0, (hipStream_t)stream, out, (long)count, p, seed);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}
