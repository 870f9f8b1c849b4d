// Shared device helpers for the gfx950 kernels of uniir_amd (wave64, MFMA 16x16x32, LDS tr-reads).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef unsigned short bf16_t;   // raw bf16 bits
typedef unsigned short f16_t;    // raw fp16 bits (only moved around / fed to MFMA)

#define UNIIR_OK 0
#define UNIIR_EINVAL (-1)
#define UNIIR_ESHAPE (-2)
#define UNIIR_EALIGN (-3)
#define UNIIR_ELAUNCH (-4)
#define UNIIR_EUNSUPPORTED (-5)

#define DEVINL __device__ __forceinline__

DEVINL float bf16_to_f32(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even in hardware (gfx950 v_cvt_pk_bf16_f32: one instruction per pair)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 hwbf16x2_t;
DEVINL bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
DEVINL unsigned pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hwbf16x2_t));
}
DEVINL float f16_to_f32(f16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
DEVINL f16_t f32_to_f16(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
// The 16-bit storage type of a forward pass: bf16 (training and the default extraction) or fp16 (the reference embedder's
// autocast(fp16), mbeir_embedder.py:52-56: 3 more mantissa bits for the forward-only towers).  Round-to-nearest-even both ways.
DEVINL unsigned pack_f16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
    const h2_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned, v);
}
template <bool F16>
DEVINL unsigned pack16x2(float lo, float hi) { return F16 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi); }
DEVINL unsigned pack16x2(float lo, float hi, bool f16) { return f16 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi); }
template <bool F16>
DEVINL float unpack16_lo(unsigned u) { return F16 ? f16_to_f32((unsigned short)(u & 0xffffu)) : __uint_as_float(u << 16); }
template <bool F16>
DEVINL float unpack16_hi(unsigned u) { return F16 ? f16_to_f32((unsigned short)(u >> 16)) : __uint_as_float(u & 0xffff0000u); }
DEVINL float unpack16_lo(unsigned u, bool f16) { return f16 ? f16_to_f32((unsigned short)(u & 0xffffu)) : __uint_as_float(u << 16); }
DEVINL float unpack16_hi(unsigned u, bool f16) { return f16 ? f16_to_f32((unsigned short)(u >> 16)) : __uint_as_float(u & 0xffff0000u); }

DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVINL float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// LDS transposing read: the 16 lanes of a group each give the address of 4 contiguous
// 16-bit elements (8 B, 8-B aligned); together a 4x16 block M (lane t -> row t>>2, cols 4*(t&3)..+3).
// Lane t receives column t: {M[0][t], M[1][t], M[2][t], M[3][t]}.
DEVINL s16x4_t lds_read_tr16(const void* lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4_t __attribute__((address_space(3)))*)(lds_addr));
}

DEVINL int xcd_remap(int bid, int nwg) {
    // bijective XCD-aware remap (8 XCDs): blocks that land on one XCD get a contiguous chunk of tiles
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// hipFuncSetAttribute (dynamic LDS opt-in) is a PER-DEVICE setting: one process may launch on every visible GPU
// (mbeir_retriever's single-process shards), so the "done" memo of a call site is kept per device
struct PerDeviceOnce {
    bool done[64] = {};
    bool first() {
        int d = 0;
        (void)hipGetDevice(&d);
        d &= 63;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

// ---- deterministic cross-workgroup column sums (round 6) -------------------------------------------------------------------------
// Bias / LayerNorm-weight / embedding gradients are sums over all rows of a tensor, taken by many workgroups.  Their fp32 atomic adds
// land in arrival order, so two runs of one step differed in the last bits.  With a caller-owned scratch buffer registered for the
// launch stream (uniir_reduce_scratch) every workgroup STORES its partial instead and one small kernel adds the partials in a fixed
// order (reduce_partials); without one the kernels keep their atomics.  Host-side table, one measuring/training thread per process.
float* reduce_scratch(hipStream_t st, int64_t bytes);       // the stream's scratch if it holds `bytes`, else nullptr
// dst_k[c] += sum_j part[j * stride + k * plane + c]  (j < nparts, c < cols, up to three destinations k; NULL ones are skipped)
int reduce_partials(const float* part, int nparts, long stride, int cols, float* d0, float* d1, float* d2, long plane, hipStream_t st);

#define HIP_LAUNCH_CHECK()                                         \
    do {                                                           \
        hipError_t e__ = hipGetLastError();                        \
        if (e__ != hipSuccess) return UNIIR_ELAUNCH;               \
    } while (0)

// max over the 16 lanes of a DPP row (lanes sharing lane >> 4), result in every lane: four DPP steps, no LDS crossbar
DEVINL float row16_max(float x) {
    auto step = [](float v, auto ctrl) {
        const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true);
        return fmaxf(v, __builtin_bit_cast(float, o));
    };
    x = step(x, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    x = step(x, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    x = step(x, std::integral_constant<int, 0x141>{});   // row_half_mirror
    x = step(x, std::integral_constant<int, 0x140>{});   // row_mirror
    return x;
}

// Reductions across the 4 rows of a wave (lanes sharing lane & 15) with the gfx950 row-swap instructions (VALU, no LDS
// crossbar round trip like __shfl_xor(.., 16 / 32)): v_permlane32_swap(x, x) leaves {x[lane % 32], x[lane % 32 + 32]} in the
// two results, v_permlane16_swap(x, x) the two rows of each 32-lane half.  The result is in every lane.
DEVINL float group_max(float v) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
DEVINL float group_sum(float v) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ------------------------------------------------------------------------------------------------------------
// MLP activations of the towers and their derivatives (UNIIR_ACT_*: 0 QuickGELU x sigma(1.702 x) -- openai/CLIP model.py; 1 the exact
// erf-GELU of BLIP's ViT / MED BERT -- vit.py:24-42, med.py:298; 2 ReLU -- T5), evaluated on bf16-rounded inputs, results rounded to
// bf16 by the callers.
//  * Contraction is switched off and the one fused multiply-add per formula is written out: the same source then gives the same bits in
//    every epilogue instantiation it is inlined into (the DACT copy-out exists with and without the act(f) output as two template
//    instances, and "dx is bitwise the same either way" is a tested property).
//  * erf-GELU (round 5): Phi(x) = 0.5 erfc(-x / sqrt 2) from Abramowitz-Stegun 7.1.26 on |x| (|error| < 1.5e-7 on erf, i.e. 2^-23 of
//    the cdf's range: far below the bf16 rounding that follows) -- one v_rcp, one v_exp shared with the density term of the
//    derivative, seven fma, no branch.  libm's erff is two branchy polynomial ranges + its own exp: the c_proj dgrad epilogue of
//    BLIP's MLPs took 3.05 ms with it against 2.24 ms for QuickGELU at 263 k x 4096 (tools/r5/epi_forms.py).  The fp32 reference path
//    (fp32_path.hip) keeps erff.
// ------------------------------------------------------------------------------------------------------------
DEVINL float gelu_cdf(float x, float& E) {          // Phi(x); E = exp(-x^2 / 2)
#pragma clang fp contract(off)
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.23164189f, 1.0f));        // 0.3275911 / sqrt 2
    const float h = -0.5f * x;
    E = __expf(h * x);
    float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    poly = __builtin_fmaf(t, poly, 1.421413741f);
    poly = __builtin_fmaf(t, poly, -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.254829592f);
    const float q = (0.5f * t) * (poly * E);          // 0.5 erfc(|x| / sqrt 2)
    return x >= 0.0f ? 1.0f - q : q;
}
DEVINL float act_fwd(float x, int act) {
#pragma clang fp contract(off)
    // __builtin_amdgcn_rcpf: 1 ulp, one instruction (an IEEE division is ~10); the results are rounded to bf16
    if (act == 0) {
        const float t = 1.702f * x;
        return x * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
    }
    if (act == 1) {
        float E;
        return x * gelu_cdf(x, E);
    }
    return fmaxf(x, 0.0f);
}
DEVINL float act_bwd(float x, int act) {
#pragma clang fp contract(off)
    if (act == 0) {
        const float t = 1.702f * x;
        const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-t));
        return s * __builtin_fmaf(t, 1.0f - s, 1.0f);
    }
    if (act == 1) {
        float E;
        const float cdf = gelu_cdf(x, E);
        return __builtin_fmaf(x * 0.3989422804014327f, E, cdf);
    }
    return x > 0.0f ? 1.0f : 0.0f;
}

// ------------------------------------------------------------------------------------------------------------
// Dropout masks: counter-based (no state): element idx of a call with seed s is kept iff fmix32(idx * golden ^ s) >= p * 2^32.
// The same (seed, idx) regenerates the mask in backward.  keep_scale = 1 / (1 - p) for kept elements, 0 otherwise.
// ------------------------------------------------------------------------------------------------------------
DEVINL unsigned drop_hash(unsigned idx, unsigned seed) {
    unsigned h = idx * 0x9E3779B1u ^ seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
DEVINL unsigned drop_threshold(float p) { return (unsigned)fminf(p * 4294967296.0f, 4294967040.0f); }
DEVINL float drop_scale(unsigned idx, unsigned seed, unsigned thresh, float keep_scale) {
    return drop_hash(idx, seed) >= thresh ? keep_scale : 0.0f;
}
