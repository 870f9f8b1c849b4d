// What do the LDS fragment reads cost a power-limited MFMA loop?  (dev tool, round 3)
// Every CU busy with 8 waves (2 per SIMD) that do nothing but 16x16x32 bf16 MFMAs on operands RE-READ FROM LDS at a given rate:
// R ds_read_b128 per 32 MFMAs -- R = 12 is the shipped GEMM (128x64 wave tile), R = 8 a 128x128 wave tile, R = 0 registers only,
// R = 24 twice the shipped rate.  Random bf16 data in LDS (the board's power cap is what limits the rate), zero data as the control.
// No global traffic inside the loop.  hipcc --offload-arch=gfx950 -O2 -o probe_lds_energy probe_lds_energy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int R>
__global__ __launch_bounds__(512) void k(const u32x4_t* __restrict__ src, float* out, int iters) {
    __shared__ u32x4_t lds[8192];                  // 128 KiB like the GEMM's ring
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = src[i];
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 16;
    const unsigned wbase = base + (threadIdx.x >> 6) * 16384;
    u32x4_t f[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) f[i] = lds[(threadIdx.x * 24 + i) & 8191];
    f32x4_t c[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        // R fresh fragments per 32 MFMAs (conflict-free 1-KiB reads: lane * 16 bytes), at a rotating offset
        const unsigned a = wbase + ((it * 1024) & 8191);
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[r]) : "v"(a), "i"((r & 7) * 1024));
        if (R) asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, f[i + ((j & 1) ? 12 : 0)]),
                                                                  __builtin_bit_cast(bf16x8_t, f[8 + j + ((i & 1) ? 12 : 0)]), c[i][j], 0, 0, 0);
    }
    float acc_out = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc_out += c[i][j][0] + c[i][j][3];
    if (acc_out == 123.456f) out[0] = acc_out;
}
template <int R>
double run(const u32x4_t* src, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 0);
    hipLaunchKernelGGL(k<R>, dim3(256), dim3(512), 0, 0, src, out, iters / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<R>, dim3(256), dim3(512), 0, 0, src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * 8 * iters * 32.0 * 16384.0;
    return flop / (ms * 1e-3) / 1e12;
}
int main() {
    std::vector<unsigned> h(8192 * 4);
    u32x4_t* src; float* out;
    hipMalloc(&src, h.size() * 4); hipMalloc(&out, 64);
    const int iters = 60000;
    for (int mode = 0; mode < 2; ++mode) {
        for (auto& x : h) {
            unsigned lo = (rand() & 0x807F) | ((120 + rand() % 8) << 7), hi = (rand() & 0x807F) | ((120 + rand() % 8) << 7);
            x = mode ? (lo | (hi << 16)) : 0u;
        }
        hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep)
            printf("%s operands: reads per 32 MFMAs 0 / 8 / 12 / 24 -> %.0f / %.0f / %.0f / %.0f TF/s\n", mode ? "random" : "zero  ",
                   run<0>(src, out, iters), run<8>(src, out, iters), run<12>(src, out, iters), run<24>(src, out, iters));
    }
    return 0;
}
