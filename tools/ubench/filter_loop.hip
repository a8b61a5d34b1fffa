// Micro-benchmark: the phase-1 circle filter loop in isolation (scalar-operand primitives),
// cycles per (wave, circle) at several occupancies; plus f32 VALU ops with an SGPR operand.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

struct Circle32 { float cx, cy, r2q, pad; };

template <int UNROLL>
__global__ __launch_bounds__(256) void filt(const Circle32 *__restrict__ c32, int n, int reps, float eb, unsigned *out)
{
    const float oxf = threadIdx.x * 1.5f + blockIdx.x, oyf = threadIdx.x * 0.7f;
    const float dxq = 0.6f, dyq = 0.8f + threadIdx.x * 1e-3f;
    unsigned total = 0;
    for (int r = 0; r < reps; ++r) {
        for (int lo = 0; lo < n; lo += 32) {
            unsigned acc = 0;
#pragma unroll UNROLL
            for (int s = lo; s < lo + 32; ++s) {
                const Circle32 c = c32[s];
                const float ocx = oxf - c.cx, ocy = oyf - c.cy;
                const float bq = __builtin_fmaf(dxq, ocx, dyq * ocy);
                const float l2 = __builtin_fmaf(ocx, ocx, ocy * ocy);
                const float inner = c.r2q - l2;
                const float g = __builtin_fmaf(bq, bq, inner);
                const float fb = eb - bq;
                const unsigned keep = ~__float_as_uint(g) & (~__float_as_uint(fb) | ~__float_as_uint(inner));
                acc = __builtin_amdgcn_alignbit(acc, keep, 31);
            }
            total += __popc(acc);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = total;
}

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int OP> __global__ void k(float *out, float sa, float sb)
{
    float fa[16];
    for (int i = 0; i < 16; ++i) fa[i] = (float)threadIdx.x + i;
    unsigned ia[16];
    for (int i = 0; i < 16; ++i) ia[i] = threadIdx.x + i;
    float fb = sa + threadIdx.x;
    for (int it = 0; it < 4096; ++it) {
        if constexpr (OP == 0) {
#define X(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(fa[i]) : "s"(sa));
            REP16(X)
#undef X
        } else if constexpr (OP == 1) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa[i]) : "v"(fb), "s"(sa));
            REP16(X)
#undef X
        } else if constexpr (OP == 2) {
#define X(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x13" : "+v"(ia[i]) : "v"(ia[(i+1)&15]), "v"(ia[(i+2)&15]));
            REP16(X)
#undef X
        } else if constexpr (OP == 3) {
#define X(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(fa[i]) : "v"(fb));
            REP16(X)
#undef X
        } else if constexpr (OP == 4) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(fa[i]) : "v"(fb));
            REP16(X)
#undef X
        } else if constexpr (OP == 5) {
#define X(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(ia[i]) : "v"(ia[(i+1)&15]));
            REP16(X)
#undef X
        } else if constexpr (OP == 6) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&fa[(i&7)*2]) : "s"(*(double*)&sa));
            REP16(X)
#undef X
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += fa[i] + ia[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> void runop(const char *name, int w, float *d)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<OP><<<256 * w, 256>>>(d, 1.5f, 0.5f); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 8; ++i) k<OP><<<256 * w, 256>>>(d, 1.5f, 0.5f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
    printf("%-22s waves/SIMD %d: %.2f cycles/wave-instr @2.4GHz\n", name, w, ms * 1e-3 * 2.4e9 / (4096.0 * 16 * w));
}

int main()
{
    const int n = 224;
    std::vector<Circle32> h(n + 32);
    for (int i = 0; i < n + 32; ++i) h[i] = Circle32{ (float)(i * 37 % 800), (float)(i * 91 % 600), 25.4f, 0.f };
    Circle32 *d; CHECK(hipMalloc(&d, sizeof(Circle32) * h.size()));
    CHECK(hipMemcpy(d, h.data(), sizeof(Circle32) * h.size(), hipMemcpyHostToDevice));
    unsigned *out; CHECK(hipMalloc(&out, 4 * 256 * 256 * 8));
    const int reps = 200;
    for (int w : {1, 2, 4, 6, 8}) {
        for (int u : {2, 4, 8}) {
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            auto launch = [&]() {
                if (u == 2) filt<2><<<256 * w, 256>>>(d, n, reps, 1e-3f, out);
                else if (u == 4) filt<4><<<256 * w, 256>>>(d, n, reps, 1e-3f, out);
                else filt<8><<<256 * w, 256>>>(d, n, reps, 1e-3f, out);
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0)); launch(); launch(); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            const double tests_per_simd = (double)reps * n * w;
            printf("filter loop unroll %d, waves/SIMD %d: %.3f ms -> %.1f cycles per (wave, circle) @2.4GHz\n", u, w, ms,
                   ms * 1e-3 * 2.4e9 / tests_per_simd);
        }
    }
    float *fo = (float *)out;
    for (int w : {2, 4}) {
        runop<3>("v_sub_f32 vv", w, fo); runop<0>("v_sub_f32 sv", w, fo); runop<4>("v_fmac_f32 vv", w, fo);
        runop<1>("v_fma_f32 vvs", w, fo); runop<2>("v_bitop3_b32", w, fo); runop<5>("v_lshl_or_b32", w, fo);
        runop<6>("v_pk_add_f32 vs", w, fo);
    }
    return 0;
}
