// valubench.hip — issue cost of the VALU instructions the bit-exact sincos is made of, on one MI355X.
// Development tool (not part of the library):  hipcc --offload-arch=gfx950 -O2 tools/valubench.hip -o tools/bin/valubench
//
// Every test: 1024 workgroups x 256 lanes (4 waves per SIMD on every CU), each wave runs ITER iterations of a body of 16
// independent instructions of one kind.  Reported: cycles per wave-instruction per SIMD, relative to v_fma_f32 = the
// guide's 2 cycles (MI355X_MICROARCH.md, per-instruction constants).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 2000;

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

// ---- one kernel per instruction kind; d[] / f[] / u[] are eight independent chains
#define KERNEL_D_D(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(256) void NAME(double *out, double seed)                             \
    {                                                                                                 \
        double d[8];                                                                                  \
        for (int i = 0; i < 8; ++i) d[i] = seed + threadIdx.x * 1e-3 + i;                             \
        const double k = seed * 0.999;                                                                \
        for (int it = 0; it < ITER; ++it) {                                                           \
            for (int r = 0; r < 2; ++r) {                                                             \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(d[i]) : "v"(k)); \
            }                                                                                         \
        }                                                                                             \
        double s = 0;                                                                                 \
        for (int i = 0; i < 8; ++i) s += d[i];                                                        \
        if (s == 12345.678) out[0] = s;                                                               \
    }

KERNEL_D_D(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
KERNEL_D_D(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_D_D(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL_D_D(k_trunc_f64, "v_trunc_f64 %0, %0")
KERNEL_D_D(k_floor_f64, "v_floor_f64 %0, %0")
KERNEL_D_D(k_rndne_f64, "v_rndne_f64 %0, %0")
KERNEL_D_D(k_fract_f64, "v_fract_f64 %0, %0")
KERNEL_D_D(k_ldexp_f64, "v_ldexp_f64 %0, %0, 3")

#define KERNEL_F_F(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(256) void NAME(double *out, double seed)                             \
    {                                                                                                 \
        float f[8];                                                                                   \
        for (int i = 0; i < 8; ++i) f[i] = (float)seed + threadIdx.x * 1e-3f + i;                     \
        const float k = (float)seed * 0.999f;                                                         \
        for (int it = 0; it < ITER; ++it) {                                                           \
            for (int r = 0; r < 2; ++r) {                                                             \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(f[i]) : "v"(k)); \
            }                                                                                         \
        }                                                                                             \
        float s = 0;                                                                                  \
        for (int i = 0; i < 8; ++i) s += f[i];                                                        \
        if (s == 12345.678f) out[0] = s;                                                              \
    }

KERNEL_F_F(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
KERNEL_F_F(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL_F_F(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
KERNEL_F_F(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL_F_F(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_F_F(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL_F_F(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL_F_F(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_F_F(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL_F_F(k_and_or_b32, "v_and_or_b32 %0, %0, %1, %1")
KERNEL_F_F(k_sin_f32, "v_sin_f32 %0, %0")
KERNEL_F_F(k_rcp_f32, "v_rcp_f32 %0, %0")

// f32 <-> f64 and int <-> f64: source and destination of different width
#define KERNEL_CVT(NAME, ASM, DT, ST)                                                                 \
    __global__ __launch_bounds__(256) void NAME(double *out, double seed)                             \
    {                                                                                                 \
        DT d[8];                                                                                      \
        ST s[8];                                                                                      \
        for (int i = 0; i < 8; ++i) { s[i] = (ST)(seed + threadIdx.x + i); d[i] = 0; }                \
        for (int it = 0; it < ITER; ++it) {                                                           \
            for (int r = 0; r < 2; ++r) {                                                             \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "=v"(d[i]) : "v"(s[i])); \
            }                                                                                         \
        }                                                                                             \
        double t = 0;                                                                                 \
        for (int i = 0; i < 8; ++i) t += (double)d[i];                                                \
        if (t == 12345.678) out[0] = t;                                                               \
    }

KERNEL_CVT(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1", double, float)
KERNEL_CVT(k_cvt_f32_f64, "v_cvt_f32_f64 %0, %1", float, double)
KERNEL_CVT(k_cvt_i32_f64, "v_cvt_i32_f64 %0, %1", int, double)
KERNEL_CVT(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %1", double, int)
KERNEL_CVT(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %1", double, unsigned)

__global__ __launch_bounds__(256) void k_pk_fma_f32(double *out, double seed)
{
    f32x2 f[8];
    for (int i = 0; i < 8; ++i) f[i] = f32x2{(float)seed + threadIdx.x * 1e-3f + i, (float)seed};
    const f32x2 k = f32x2{(float)seed * 0.999f, 0.5f};
    for (int it = 0; it < ITER; ++it) {
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(k));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i].x + f[i].y;
    if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void k_mad_u64_u32(double *out, double seed)
{
    unsigned long long d[8];
    unsigned a = (unsigned)seed + threadIdx.x, b = (unsigned)seed * 3u + 1u;
    for (int i = 0; i < 8; ++i) d[i] = i;
    for (int it = 0; it < ITER; ++it) {
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(a), "v"(b) : "vcc");
        }
    }
    unsigned long long s = 0;
    for (int i = 0; i < 8; ++i) s += d[i];
    if (s == 12345) out[0] = (double)s;
}

struct Test { const char *name; void (*fn)(double *, double); };

int main()
{
    double *d_out;
    CHECK(hipMalloc(&d_out, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const std::vector<Test> tests = {
        {"v_fma_f32", k_fma_f32}, {"v_mul_f32", k_mul_f32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_fma_f64", k_fma_f64},
        {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_trunc_f64", k_trunc_f64}, {"v_floor_f64", k_floor_f64},
        {"v_rndne_f64", k_rndne_f64}, {"v_fract_f64", k_fract_f64}, {"v_ldexp_f64", k_ldexp_f64},
        {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_i32_f64", k_cvt_i32_f64},
        {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_f32_u32", k_cvt_f32_u32},
        {"v_cvt_i32_f32", k_cvt_i32_f32}, {"v_cndmask_b32", k_cndmask}, {"v_mul_lo_u32", k_mul_lo_u32},
        {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_add_u32", k_add_u32},
        {"v_lshl_add_u32", k_lshl_add_u32}, {"v_and_or_b32", k_and_or_b32}, {"v_sin_f32", k_sin_f32}, {"v_rcp_f32", k_rcp_f32},
    };
    std::vector<double> ms(tests.size());
    for (int round = 0; round < 3; ++round)
        for (size_t t = 0; t < tests.size(); ++t) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(tests[t].fn, dim3(1024), dim3(256), 0, 0, d_out, 1.5);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float m;
            CHECK(hipEventElapsedTime(&m, e0, e1));
            if (round == 0 || m < ms[t]) ms[t] = m;
        }
    const double base = ms[0];
    printf("%-16s %10s %s\n", "instruction", "ms", "cycles per wave-instruction per SIMD (v_fma_f32 = 2)");
    for (size_t t = 0; t < tests.size(); ++t) printf("%-16s %10.4f %6.2f\n", tests[t].name, ms[t], 2.0 * ms[t] / base);
    return 0;
}
