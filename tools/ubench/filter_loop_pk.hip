// Micro-benchmark: the packed-f32 phase-1 circle filter loop in isolation (SoA, scalar-operand pairs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int UNROLL>
__global__ __launch_bounds__(256) void filt(const float *__restrict__ cxs, const float *__restrict__ cys, const float *__restrict__ rs,
                                            int n, int reps, float eb, unsigned *out)
{
    const float oxf = threadIdx.x * 1.5f + blockIdx.x, oyf = threadIdx.x * 0.7f;
    const float dxq = 0.6f, dyq = 0.8f + threadIdx.x * 1e-3f;
    float ebv;
    asm("v_mov_b32 %0, %1" : "=v"(ebv) : "s"(eb));
    const f32x2 ox2 = { oxf, oxf }, oy2 = { oyf, oyf }, dx2 = { dxq, dxq }, dy2 = { dyq, dyq }, eb2 = { ebv, ebv };
    unsigned total = 0;
    for (int r = 0; r < reps; ++r) {
        for (int lo = 0; lo < n; lo += 32) {
            unsigned acc = 0;
#pragma unroll UNROLL
            for (int s = lo; s < lo + 32; s += 2) {
                const f32x2 cx = *reinterpret_cast<const f32x2 *>(cxs + s);
                const f32x2 cy = *reinterpret_cast<const f32x2 *>(cys + s);
                const f32x2 r2q = *reinterpret_cast<const f32x2 *>(rs + s);
                const f32x2 ocx = ox2 - cx, ocy = oy2 - cy;
                const f32x2 bq = __builtin_elementwise_fma(dx2, ocx, dy2 * ocy);
                const f32x2 l2 = __builtin_elementwise_fma(ocx, ocx, ocy * ocy);
                const f32x2 inner = r2q - l2;
                const f32x2 g = __builtin_elementwise_fma(bq, bq, inner);
                const f32x2 fb = eb2 - bq;
                const unsigned keep0 = ~__float_as_uint(g.x) & (~__float_as_uint(fb.x) | ~__float_as_uint(inner.x));
                const unsigned keep1 = ~__float_as_uint(g.y) & (~__float_as_uint(fb.y) | ~__float_as_uint(inner.y));
                acc = __builtin_amdgcn_alignbit(acc, keep0, 31);
                acc = __builtin_amdgcn_alignbit(acc, keep1, 31);
            }
            total += __popc(acc);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = total;
}

struct Pair { f32x2 cx, cy, r2q, pad; };
template <int UNROLL>
__global__ __launch_bounds__(256) void filt_il(const Pair *__restrict__ pr, int n, int reps, float eb, unsigned *out)
{
    const float oxf = threadIdx.x * 1.5f + blockIdx.x, oyf = threadIdx.x * 0.7f;
    const float dxq = 0.6f, dyq = 0.8f + threadIdx.x * 1e-3f;
    float ebv;
    asm("v_mov_b32 %0, %1" : "=v"(ebv) : "s"(eb));
    const f32x2 ox2 = { oxf, oxf }, oy2 = { oyf, oyf }, dx2 = { dxq, dxq }, dy2 = { dyq, dyq }, eb2 = { ebv, ebv };
    unsigned total = 0;
    for (int r = 0; r < reps; ++r) {
        for (int lo = 0; lo < n / 2; lo += 16) {
            unsigned acc = 0;
#pragma unroll UNROLL
            for (int s = lo; s < lo + 16; ++s) {
                const Pair c = pr[s];
                const f32x2 ocx = ox2 - c.cx, ocy = oy2 - c.cy;
                const f32x2 bq = __builtin_elementwise_fma(dx2, ocx, dy2 * ocy);
                const f32x2 l2 = __builtin_elementwise_fma(ocx, ocx, ocy * ocy);
                const f32x2 inner = c.r2q - l2;
                const f32x2 g = __builtin_elementwise_fma(bq, bq, inner);
                const f32x2 fb = eb2 - bq;
                const unsigned keep0 = ~__float_as_uint(g.x) & (~__float_as_uint(fb.x) | ~__float_as_uint(inner.x));
                const unsigned keep1 = ~__float_as_uint(g.y) & (~__float_as_uint(fb.y) | ~__float_as_uint(inner.y));
                acc = __builtin_amdgcn_alignbit(acc, keep0, 31);
                acc = __builtin_amdgcn_alignbit(acc, keep1, 31);
            }
            total += __popc(acc);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = total;
}

int main()
{
    const int n = 224;
    std::vector<float> hx(n + 32), hy(n + 32), hr(n + 32);
    for (int i = 0; i < n + 32; ++i) { hx[i] = (float)(i * 37 % 800); hy[i] = (float)(i * 91 % 600); hr[i] = 25.4f; }
    float *dx, *dy, *dr;
    CHECK(hipMalloc(&dx, 4 * hx.size())); CHECK(hipMalloc(&dy, 4 * hx.size())); CHECK(hipMalloc(&dr, 4 * hx.size()));
    CHECK(hipMemcpy(dx, hx.data(), 4 * hx.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dy, hy.data(), 4 * hx.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dr, hr.data(), 4 * hx.size(), hipMemcpyHostToDevice));
    unsigned *out; CHECK(hipMalloc(&out, 4 * 256 * 256 * 8));
    const int reps = 200;
    for (int w : {1, 2, 4, 5, 6, 8}) {
        for (int u : {1, 2, 4}) {
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            auto launch = [&]() {
                if (u == 1) filt<1><<<256 * w, 256>>>(dx, dy, dr, n, reps, 1e-3f, out);
                else if (u == 2) filt<2><<<256 * w, 256>>>(dx, dy, dr, n, reps, 1e-3f, out);
                else filt<4><<<256 * w, 256>>>(dx, dy, dr, n, reps, 1e-3f, out);
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0)); launch(); launch(); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            printf("packed filter loop unroll %d pairs, waves/SIMD %d: %.3f ms -> %.1f cycles per (wave, circle) @2.4GHz\n", u, w, ms,
                   ms * 1e-3 * 2.4e9 / ((double)reps * n * w));
        }
    }
    std::vector<Pair> hp(n / 2 + 16);
    for (size_t i = 0; i < hp.size(); ++i) hp[i] = Pair{ { hx[2 * i % n], hx[(2 * i + 1) % n] }, { hy[2 * i % n], hy[(2 * i + 1) % n] }, { 25.4f, 25.4f }, { 0, 0 } };
    Pair *dp; CHECK(hipMalloc(&dp, sizeof(Pair) * hp.size()));
    CHECK(hipMemcpy(dp, hp.data(), sizeof(Pair) * hp.size(), hipMemcpyHostToDevice));
    for (int w : {4, 5, 6, 8}) {
        for (int u : {1, 2, 4}) {
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            auto launch = [&]() {
                if (u == 1) filt_il<1><<<256 * w, 256>>>(dp, n, reps, 1e-3f, out);
                else if (u == 2) filt_il<2><<<256 * w, 256>>>(dp, n, reps, 1e-3f, out);
                else filt_il<4><<<256 * w, 256>>>(dp, n, reps, 1e-3f, out);
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0)); launch(); launch(); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            printf("interleaved pairs, unroll %d pairs, waves/SIMD %d: %.3f ms -> %.1f cycles per (wave, circle) @2.4GHz\n", u, w, ms,
                   ms * 1e-3 * 2.4e9 / ((double)reps * n * w));
        }
    }
    return 0;
}
