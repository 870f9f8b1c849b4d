// Probe of ds_read_b64_tr_b16 and MFMA 16x16x32 bf16 lane layouts on the actual GPU (dev tool, not shipped path).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k_tr(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    // each lane passes address of element 4*l (8 B per lane, contiguous)
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + 4 * l));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = t[j];
}
__global__ void k_mfma(const unsigned short* a, const unsigned short* b, float* c) {
    // a: 16x32 row-major bf16 (A[i][k]), b: 32x16 row-major (B[k][j]); c: 16x16
    int l = threadIdx.x;
    unsigned short av[8], bv[8];
    for (int j = 0; j < 8; ++j) { av[j] = a[(l & 15) * 32 + 8 * (l >> 4) + j]; bv[j] = b[(8 * (l >> 4) + j) * 16 + (l & 15)]; }
    bf16x8 A = *(bf16x8*)av, B = *(bf16x8*)bv;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) c[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
int main() {
    short* d; hipMalloc(&d, 512);
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("TR16 probe: lane -> 4 values (lds[i]=i, lane addr = 4*lane)\n");
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
        for (int j = 0; j < 4; ++j) if (h[l*4+j] != (l & 15) + j * 16 + (l >> 4) * 64) ok = 0;
    }
    printf("TR16 expected-semantic %s\n", ok ? "MATCH" : "MISMATCH");
    unsigned short ha[512], hb[512]; float ref[256] = {0}, hc[256];
    for (int i = 0; i < 512; ++i) { ha[i] = f2bf((float)((i * 7) % 13 - 6)); hb[i] = f2bf((float)((i * 5) % 11 - 5)); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += (float)((( (i*32+k) * 7) % 13) - 6) * (float)((((k*16+j) * 5) % 11) - 5); ref[i*16+j] = s; }
    unsigned short *da, *db; float* dc; hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dc, 1024);
    hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dc);
    hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) if (hc[i] != ref[i]) ++bad;
    printf("MFMA 16x16x32 layout %s (bad=%d)\n", bad ? "MISMATCH" : "MATCH", bad);
    return 0;
}
