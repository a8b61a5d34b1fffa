// Probe of v_mfma_f32_32x32x2_f32 on gfx950: operand/result layout and the rounding of its two multiply-adds,
// and of v_permlane32_swap.  Prints what it finds; no part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float *A, const float *B, const float *C, float *D, unsigned *swp)
{
    const int l = threadIdx.x;
    // A is 32x2 (row i, column k), B is 2x32, C/D are 32x32
    const float a = A[(l % 32) * 2 + l / 32];
    const float b = B[(l / 32) * 32 + l % 32];
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + l % 32];
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + l % 32] = d[r];
    auto s = __builtin_amdgcn_permlane32_swap(1000u + l, 2000u + l, false, false);
    swp[l] = s[0];
    swp[64 + l] = s[1];
}

int main()
{
    const int trials = 2000;
    long bad_layout = 0, m01 = 0, m10 = 0, munf = 0, other = 0;
    double worst = 0;
    float *dA, *dB, *dC, *dD; unsigned *dS;
    hipMalloc(&dA, 64 * 4); hipMalloc(&dB, 64 * 4); hipMalloc(&dC, 1024 * 4); hipMalloc(&dD, 1024 * 4); hipMalloc(&dS, 128 * 4);
    std::vector<float> A(64), B(64), C(1024), D(1024);
    unsigned S[128];
    srand(1);
    auto rnd = [](float scale) { return scale * (float)((rand() / (double)RAND_MAX) * 2 - 1); };
    for (int t = 0; t < trials; ++t) {
        const float sc = (t % 3 == 0) ? 1.0f : (t % 3 == 1) ? 1e-3f : 100.0f;
        for (auto &x : A) x = rnd(1.0f);
        for (auto &x : B) x = rnd(sc);
        for (auto &x : C) x = (t % 2) ? rnd(sc) : 0.0f;
        hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, dS);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        hipMemcpy(S, dS, 512, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                const float a0 = A[i * 2], a1 = A[i * 2 + 1], b0 = B[j], b1 = B[32 + j], c = C[i * 32 + j], d = D[i * 32 + j];
                const float f01 = fmaf(a1, b1, fmaf(a0, b0, c)), f10 = fmaf(a0, b0, fmaf(a1, b1, c));
                const float unf = (a0 * b0 + c) + a1 * b1;
                const double ex = (double)a0 * b0 + (double)a1 * b1 + c;
                const double mag = fabs((double)a0 * b0) + fabs((double)a1 * b1) + fabs(c);
                if (fabs(d - ex) > 1e-5 * mag + 1e-30) ++bad_layout;
                if (d == f01) ++m01; else if (d == f10) ++m10; else if (d == unf) ++munf; else ++other;
                if (mag > 0) worst = fmax(worst, fabs(d - ex) / mag);
            }
    }
    printf("values far from the exact sum (layout wrong): %ld\n", bad_layout);
    printf("equal to fma(a1,b1,fma(a0,b0,c)): %ld, else equal to fma(a0,b0,fma(a1,b1,c)): %ld, else unfused: %ld, other: %ld\n", m01, m10, munf, other);
    printf("worst |d - exact| / (|a0 b0| + |a1 b1| + |c|) = %.3g (2^-24 = %.3g)\n", worst, ldexp(1.0, -24));
    printf("permlane32_swap(x = 1000 + l, y = 2000 + l): r0[0]=%u r0[31]=%u r0[32]=%u r0[63]=%u | r1[0]=%u r1[31]=%u r1[32]=%u r1[63]=%u\n",
           S[0], S[31], S[32], S[63], S[64], S[95], S[96], S[127]);
    return 0;
}
