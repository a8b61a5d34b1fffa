// Do matrix instructions of one wave run beside vector instructions of the other waves of the same SIMD?
// Each workgroup has 8 waves (2 per SIMD).  Mode 0: every wave runs a chain of packed f32 FMAs.  Mode 1/2: every
// wave runs f32 / bf16 matrix instructions.  Mode 3/4: even waves the vector chain, odd waves f32 / bf16 matrix
// instructions.  Overlap shows as time(3) ~ max(time(0), time(1)) / ... rather than their sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters)
{
    const int wave = threadIdx.x >> 6;
    // waves are dealt to SIMDs round-robin, so waves w and w + 4 share one: the split is by bit 2
    const bool matrix = MODE == 1 || MODE == 2 || ((MODE == 3 || MODE == 4) && (wave & 4));
    const bool vector = MODE == 0 || ((MODE == 3 || MODE == 4) && !(wave & 4));
    float r = 0;
    if (vector) {
        f32x2 a = { 1.0f + threadIdx.x * 1e-7f, 1.0f }, b = { 0.5f, 0.25f };
        f32x2 c0 = a, c1 = b, c2 = a + b, c3 = a - b;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                c0 = __builtin_elementwise_fma(c0, a, b);
                c1 = __builtin_elementwise_fma(c1, a, b);
                c2 = __builtin_elementwise_fma(c2, a, b);
                c3 = __builtin_elementwise_fma(c3, a, b);
            }
        }
        r = c0.x + c0.y + c1.x + c1.y + c2.x + c2.y + c3.x + c3.y;
    }
    if (matrix) {
        f32x16 d0 = { 0 }, d1 = { 0 };
        if (MODE == 1 || MODE == 3) {
            const float a = 1.0f + threadIdx.x, b = 0.5f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 2; ++u) { // 4 x 64 cycles
                    d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
                }
            }
        } else {
            bf16x8 a, b;
            for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(1.0f + q); b[q] = (__bf16)0.5f; }
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { // 8 x 32 cycles
                    d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
                }
            }
        }
        for (int q = 0; q < 16; ++q) r += d0[q] + d1[q];
    }
    if (MODE == 5 || MODE == 6) { // every wave: matrix instructions with independent vector instructions between them
        f32x2 a = { 1.0f + threadIdx.x * 1e-7f, 1.0f }, b = { 0.5f, 0.25f };
        f32x2 c0 = a, c1 = b, c2 = a + b, c3 = a - b;
        f32x16 d0 = { 0 }, d1 = { 0 };
        const float fa = 1.0f + threadIdx.x, fb = 0.5f;
        bf16x8 ha, hb;
        for (int q = 0; q < 8; ++q) { ha[q] = (__bf16)(1.0f + q); hb[q] = (__bf16)0.5f; }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MODE == 5) {
                    if (u & 1) d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, d1, 0, 0, 0);
                    else d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, d0, 0, 0, 0);
                } else {
                    d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, d1, 0, 0, 0);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    c0 = __builtin_elementwise_fma(c0, a, b);
                    c1 = __builtin_elementwise_fma(c1, a, b);
                    c2 = __builtin_elementwise_fma(c2, a, b);
                    c3 = __builtin_elementwise_fma(c3, a, b);
                }
            }
        }
        r = c0.x + c0.y + c1.x + c1.y + c2.x + c2.y + c3.x + c3.y;
        for (int q = 0; q < 16; ++q) r += d0[q] + d1[q];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE> float run(float *d, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    // per iteration: vector wave 64 packed FMAs (256 cycles alone on its SIMD); matrix wave 256 cycles of matrix pipe
    printf("vector only (8 waves):        %.2f ms\n", run<0>(d, iters));
    printf("f32 matrix only (8 waves):    %.2f ms\n", run<1>(d, iters));
    printf("bf16 matrix only (8 waves):   %.2f ms\n", run<2>(d, iters));
    printf("4 vector + 4 f32 matrix:      %.2f ms\n", run<3>(d, iters));
    printf("4 vector + 4 bf16 matrix:     %.2f ms\n", run<4>(d, iters));
    printf("8 waves, f32 matrix + vector interleaved in each:  %.2f ms\n", run<5>(d, iters));
    printf("8 waves, bf16 matrix + vector interleaved in each: %.2f ms\n", run<6>(d, iters));
    printf("expected if alone: one wave per SIMD, 256 cycles per iteration: %.2f ms at 2.4 GHz\n", iters * 256 / 2.4e6);
    return 0;
}
