// VALU issue-cost probe (dev tool): cycles per wave64 instruction for the softmax instruction mix, one wave alone on a SIMD and two
// waves sharing one.  hipcc --offload-arch=gfx950 -O2 -o probe_valu probe_valu.hip && ./probe_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define ITER 256
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, float x) {
    float a0 = x + threadIdx.x, a1 = a0 * 0.5f, a2 = a0 * 0.25f, a3 = a0 * 0.125f, a4 = a0 + 1, a5 = a0 + 2, a6 = a0 + 3, a7 = a0 + 4;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (MODE == 0) {        // 8 independent v_exp_f32
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 1) { // 8 independent v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 2) { // 4 v_pk_fma_f32 + 4 v_pk_mul_f32
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            } else if (MODE == 3) { // alternate exp / fma (4 + 4)
                asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %4, %4\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %5, %5\n v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %6, %6\n v_exp_f32 %3, %3\n v_fma_f32 %7, %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 4) { // 1 exp : 3 fma (2 + 6)
                asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 5) { // cvt_pk_bf16 x4 + max3 x4
                asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_max3_f32 %1, %1, %3, %5\n v_max3_f32 %3, %3, %5, %7\n v_max3_f32 %5, %5, %7, %1\n v_max3_f32 %7, %7, %1, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 6) { // dependent chain of exps (latency)
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0"
                             : "+v"(a0));
            } else if (MODE == 7) { // dependent chain of fma (latency)
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0"
                             : "+v"(a0));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
    for (int threads : {64, 256, 512, 1024}) {      // 1 wave; 1 / SIMD; 2 / SIMD; 4 / SIMD (one workgroup on one CU)
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.001f);
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("%-28s waves/CU %2d: %6.2f counter ticks per instruction per wave\n", name, threads / 64, (double)mx / (ITER * REP));
    }
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64 * 8);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
    printf("clock rate %d kHz, wall clock rate %d kHz (s_memtime ticks)\n", clk, wc);
    run<1>("v_fma_f32", out, cyc);
    run<0>("v_exp_f32", out, cyc);
    run<2>("v_pk_fma/mul_f32", out, cyc);
    run<3>("exp:fma 1:1", out, cyc);
    run<4>("exp:fma 1:3", out, cyc);
    run<5>("cvt_pk_bf16 + max3", out, cyc);
    run<6>("exp dependent chain", out, cyc);
    run<7>("fma dependent chain", out, cyc);
    return 0;
}
