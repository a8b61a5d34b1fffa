// Micro-benchmark: issue rate of the VALU instructions the tracer's inner loops are made of.
// Each kernel runs ITER x 16 independent instructions per wave; all CUs loaded with WAVES waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

constexpr int ITER = 4096;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP> __global__ void k(double *out, double sa, double sb)
{
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = (double)threadIdx.x * 1e-3 + i;
    double b = sa + threadIdx.x, c = sb;
    unsigned ia[16];
    for (int i = 0; i < 16; ++i) ia[i] = threadIdx.x + i;
    float fa[16];
    for (int i = 0; i < 16; ++i) fa[i] = (float)threadIdx.x + i;
    float fb = (float)sa, fc = (float)sb;
    for (int it = 0; it < ITER; ++it) {
        if constexpr (OP == 0) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(X)
#undef X
        } else if constexpr (OP == 1) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(X)
#undef X
        } else if constexpr (OP == 2) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X)
#undef X
        } else if constexpr (OP == 3) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "s"(sa));
            REP16(X)
#undef X
        } else if constexpr (OP == 4) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "s"(sa));
            REP16(X)
#undef X
        } else if constexpr (OP == 5) {
#define X(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if constexpr (OP == 6) {
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(ia[i]) : "v"(ia[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if constexpr (OP == 7) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa[i]) : "v"(fb), "v"(fc));
            REP16(X)
#undef X
        } else if constexpr (OP == 8) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(fa[i]) : "v"(fb));
            REP16(X)
#undef X
        } else if constexpr (OP == 9) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(ia[i]), "v"(ia[(i + 1) & 15]) : "vcc");
            REP16(X)
#undef X
        } else if constexpr (OP == 10) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if constexpr (OP == 11) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if constexpr (OP == 12) {
#define X(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (OP == 13) {
#define X(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (OP == 14) {
            // alternating f64 mul and 32-bit op (does the int op hide behind f64?)
#define X(i) asm volatile("v_mul_f64 %0, %0, %2\n\tv_or_b32 %1, %1, %1" : "+v"(a[i]), "+v"(ia[i]) : "v"(b));
            REP16(X)
#undef X
        } else if constexpr (OP == 15) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            REP16(X)
#undef X
        } else if constexpr (OP == 16) {
#define X(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
            REP16(X)
#undef X
        } else if constexpr (OP == 17) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(fa[i]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (OP == 18) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(fa[i]));
            REP16(X)
#undef X
        } else if constexpr (OP == 19) {
#define X(i) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if constexpr (OP == 20) {
#define X(i) asm volatile("v_pk_lshrrev_b16 %0, 15, %0" : "+v"(ia[i]));
            REP16(X)
#undef X
        } else if constexpr (OP == 21) {
#define X(i) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ia[(i + 1) & 15]), "v"(ia[(i + 2) & 15]));
            REP16(X)
#undef X
        } else if constexpr (OP == 22) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ia[(i + 1) & 15]), "v"(ia[(i + 2) & 15]));
            REP16(X)
#undef X
        } else if constexpr (OP == 23) {
#define X(i) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ia[(i + 1) & 15]), "v"(ia[(i + 2) & 15]));
            REP16(X)
#undef X
        } else if constexpr (OP == 24) {
#define X(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if constexpr (OP == 25) {
#define X(i) asm volatile("v_lshrrev_b32 %0, 31, %0" : "+v"(ia[i]));
            REP16(X)
#undef X
        } else if constexpr (OP == 26) {
#define X(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(ia[i]) : "v"(ia[(i + 1) & 15]), "v"(ia[(i + 2) & 15]));
            REP16(X)
#undef X
        } else if constexpr (OP == 27) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(ia[i]) : "v"(fa[i]), "v"(fb) : "vcc");
            REP16(X)
#undef X
        } else if constexpr (OP == 28) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(X)
#undef X
        } else if constexpr (OP == 29) {
#define X(i) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(b));
            REP16(X)
#undef X
        } else if constexpr (OP == 30) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[i]) : "v"(ia[(i + 1) & 15]) : "vcc");
            REP16(X)
#undef X
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + ia[i] + fa[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> void run(const char *name, int waves_per_simd, double *d_out)
{
    const int blocks = 256 * waves_per_simd; // 256-thread blocks: 4 waves = 1 per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<OP><<<blocks, 256>>>(d_out, 1.0000001, 0.5);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<OP><<<blocks, 256>>>(d_out, 1.0000001, 0.5);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts_per_simd = (double)ITER * 16 * waves_per_simd;
    const double cyc = ms * 1e-3 * 2.4e9 / insts_per_simd;
    printf("%-28s waves/SIMD %d : %8.3f ms  -> %.2f cycles/wave-instr @2.4GHz\n", name, waves_per_simd, ms, cyc);
}

int main()
{
    double *d_out; CHECK(hipMalloc(&d_out, sizeof(double) * 256 * 256 * 8));
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f64 vv", w, d_out);
        run<1>("v_mul_f64 vv", w, d_out);
        run<2>("v_fma_f64 vvv", w, d_out);
        run<3>("v_add_f64 vs", w, d_out);
        run<4>("v_mul_f64 vs", w, d_out);
        run<5>("v_or_b32", w, d_out);
        run<6>("v_alignbit_b32", w, d_out);
        run<7>("v_fma_f32", w, d_out);
        run<8>("v_mul_f32", w, d_out);
        run<9>("v_mad_u64_u32", w, d_out);
        run<10>("v_mul_hi_u32", w, d_out);
        run<11>("v_mul_lo_u32", w, d_out);
        run<12>("v_rsq_f64", w, d_out);
        run<13>("v_rcp_f64", w, d_out);
        run<14>("v_mul_f64 + v_or_b32 pair", w, d_out);
        run<15>("v_pk_fma_f32", w, d_out);
        run<16>("v_cmp_lt_f64", w, d_out);
        run<17>("v_cvt_f32_f64", w, d_out);
        run<18>("v_sqrt_f32", w, d_out);
        run<19>("v_cvt_pk_i16_i32", w, d_out);
        run<20>("v_pk_lshrrev_b16", w, d_out);
        run<21>("v_pk_mad_u16", w, d_out);
        run<22>("v_perm_b32", w, d_out);
        run<23>("v_bfi_b32", w, d_out);
        run<24>("v_lshl_or_b32", w, d_out);
        run<25>("v_lshrrev_b32", w, d_out);
        run<26>("v_bitop3_b32", w, d_out);
        run<27>("v_cmp_lt_f32 + v_addc_co_u32", w, d_out);
        run<28>("v_pk_mul_f32", w, d_out);
        run<29>("v_mov_b64", w, d_out);
        run<30>("v_cndmask_b32", w, d_out);
    }
    return 0;
}
