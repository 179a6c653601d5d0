// mixbench.hip — design-space probe for the fused i16->i16 shift kernel (development tool).
// All variants do the real unpack / unfused complex multiply / pack arithmetic on a 1 GiB stream
// with period P = 1024 (the headline configuration); they differ in where the correctors come from.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mixbench tools/mixbench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void unpack_i16(uint32_t w, float &re, float &im)
{
    re = (float)(int16_t)(w & 0xffffu) * 0x1p-15f;
    im = (float)(int16_t)(w >> 16) * 0x1p-15f;
}
__device__ __forceinline__ int f32_as_i16(float x)
{
    x = (x != x) ? 0.0f : x;
    x = fminf(fmaxf(x, -32768.0f), 32767.0f);
    return (int)x;
}
__device__ __forceinline__ uint32_t mixpack(uint32_t w, float c, float s)
{
    float a, b;
    unpack_i16(w, a, b);
    const float re = __fsub_rn(__fmul_rn(a, c), __fmul_rn(b, s));
    const float im = __fadd_rn(__fmul_rn(a, s), __fmul_rn(b, c));
    const int i = f32_as_i16(__fmul_rn(re, 32767.0f));
    const int q = f32_as_i16(__fmul_rn(im, 32767.0f));
    return ((uint32_t)i & 0xffffu) | ((uint32_t)q << 16);
}

__global__ void fill(uint32_t *p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = x;
    }
}
__global__ void fill_lut(float2 *t, uint32_t n, uint32_t P)
{
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const float th = -6.2831855f * (0.0048828125f * (float)((e % P) + 1));
        t[e] = make_float2(cosf(th), sinf(th));
    }
}

// K0: one-shot, no table: corrector constant (upper bound: copy + ALU)
template <int BLOCK, int V>
__global__ __launch_bounds__(BLOCK) void k_const(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, const float2 *__restrict__ lut, uint32_t P)
{
    const uint64_t t0 = (uint64_t)blockIdx.x * BLOCK * V;
    u32x4 q[V];
#pragma unroll
    for (int v = 0; v < V; ++v) q[v] = __builtin_nontemporal_load(in + t0 + v * BLOCK + threadIdx.x);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = mixpack(q[v][k], 0.6f, 0.8f);
        __builtin_nontemporal_store(o, out + t0 + v * BLOCK + threadIdx.x);
    }
}

// K1: one-shot, table read straight from global (32 B per lane per vector), table extended past P
template <int BLOCK, int V, bool POW2>
__global__ __launch_bounds__(BLOCK) void k_glut(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, const float2 *__restrict__ lut, uint32_t P)
{
    const uint64_t t0 = (uint64_t)blockIdx.x * BLOCK * V;
    u32x4 q[V];
#pragma unroll
    for (int v = 0; v < V; ++v) q[v] = __builtin_nontemporal_load(in + t0 + v * BLOCK + threadIdx.x);
    const uint32_t tmod = (BLOCK * V * 4) % P;
    uint32_t ph;
    if (POW2) ph = (blockIdx.x * tmod) & (P - 1);
    else ph = ((blockIdx.x % P) * tmod) % P;
    const float2 *tab = lut + ph;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const uint32_t e = (v * BLOCK + threadIdx.x) * 4;
        float2 cs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cs[k] = tab[e + k];
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = mixpack(q[v][k], cs[k].x, cs[k].y);
        __builtin_nontemporal_store(o, out + t0 + v * BLOCK + threadIdx.x);
    }
}

// K2: one-shot, period-strided rows: the lane's R vectors sit P samples apart, so they share
// the same 4 correctors (table bytes per sample / R).  Requires BLOCK*4 == P here (P = 1024, BLOCK = 256).
template <int BLOCK, int R>
__global__ __launch_bounds__(BLOCK) void k_rows(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, const float2 *__restrict__ lut, uint32_t P)
{
    const uint64_t t0 = (uint64_t)blockIdx.x * BLOCK * R;      // R consecutive periods
    u32x4 q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) q[r] = __builtin_nontemporal_load(in + t0 + r * BLOCK + threadIdx.x);
    float2 cs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cs[k] = lut[threadIdx.x * 4 + k];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = mixpack(q[r][k], cs[k].x, cs[k].y);
        __builtin_nontemporal_store(o, out + t0 + r * BLOCK + threadIdx.x);
    }
}

// K2b: period-strided rows with a SMALL workgroup: BLOCK lanes cover one column slice of R
// consecutive periods; P/4/BLOCK consecutive workgroups cover the R periods completely.
template <int BLOCK, int R>
__global__ __launch_bounds__(BLOCK) void k_rows2(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, const float2 *__restrict__ lut, uint32_t P)
{
    const uint32_t vpr = P / 4;                  // vectors per row (period)
    const uint32_t cb = vpr / BLOCK;             // column slices per row
    const uint32_t col = blockIdx.x % cb, rg = blockIdx.x / cb;
    const uint64_t base = (uint64_t)rg * R * vpr + col * BLOCK + threadIdx.x;
    u32x4 q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) q[r] = __builtin_nontemporal_load(in + base + (uint64_t)r * vpr);
    float2 cs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cs[k] = lut[(col * BLOCK + threadIdx.x) * 4 + k];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = mixpack(q[r][k], cs[k].x, cs[k].y);
        __builtin_nontemporal_store(o, out + base + (uint64_t)r * vpr);
    }
}

// K1b: as K1 but the table is requested before the samples
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_glut_first(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, const float2 *__restrict__ lut, uint32_t P)
{
    const uint64_t t0 = (uint64_t)blockIdx.x * BLOCK;
    const uint32_t tmod = (BLOCK * 4) % P;
    const uint32_t ph = (blockIdx.x * tmod) & (P - 1);
    const u32x4 *tab = reinterpret_cast<const u32x4 *>(lut + ph + threadIdx.x * 4);
    const u32x4 c01 = tab[0], c23 = tab[1];
    const u32x4 q = __builtin_nontemporal_load(in + t0 + threadIdx.x);
    u32x4 o;
    o[0] = mixpack(q[0], __uint_as_float(c01[0]), __uint_as_float(c01[1]));
    o[1] = mixpack(q[1], __uint_as_float(c01[2]), __uint_as_float(c01[3]));
    o[2] = mixpack(q[2], __uint_as_float(c23[0]), __uint_as_float(c23[1]));
    o[3] = mixpack(q[3], __uint_as_float(c23[2]), __uint_as_float(c23[3]));
    __builtin_nontemporal_store(o, out + t0 + threadIdx.x);
}

// K3: persistent, table in LDS (the first design): block-cyclic tiles of BLOCK*U vectors
template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void k_lds(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, const float2 *__restrict__ lut, uint32_t P, uint64_t n_vec)
{
    extern __shared__ float2 tab[];
    for (uint32_t e = threadIdx.x; e < P; e += BLOCK) tab[e] = lut[e];
    __syncthreads();
    const uint64_t tile = (uint64_t)BLOCK * U;
    for (uint64_t t0 = (uint64_t)blockIdx.x * tile; t0 < n_vec; t0 += (uint64_t)gridDim.x * tile) {
        u32x4 q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) q[u] = __builtin_nontemporal_load(in + t0 + u * BLOCK + threadIdx.x);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t ph = (uint32_t)(((t0 + u * BLOCK + threadIdx.x) * 4) & (P - 1));
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float2 cs = tab[ph + k]; o[k] = mixpack(q[u][k], cs.x, cs.y); }
            __builtin_nontemporal_store(o, out + t0 + u * BLOCK + threadIdx.x);
        }
    }
}

struct Var { const char *name; int kind; void *fn; int block; int per_block_vec; int lds; int grid; std::vector<double> ms; };

int main(int argc, char **argv)
{
    const uint64_t bytes = 1ull << 30, n_vec = bytes / 16;
    const uint32_t P = 1024;
    u32x4 *in, *out; float2 *lut;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&lut, (P + 8192) * 8));
    fill<<<4096, 256>>>((uint32_t *)in, bytes / 4);
    fill_lut<<<64, 256>>>(lut, P + 8192, P);
    CK(hipDeviceSynchronize());
    typedef void (*k4)(const u32x4 *, u32x4 *, const float2 *, uint32_t);
    typedef void (*k5)(const u32x4 *, u32x4 *, const float2 *, uint32_t, uint64_t);
    std::vector<Var> vs;
#define ADD4(NAME, FN, B, PBV) vs.push_back({NAME, 4, (void *)(k4)FN, B, PBV, 0, (int)(n_vec / (PBV)), {}})
    ADD4("const<128,1>", (k_const<128, 1>), 128, 128);
    ADD4("const<128,2>", (k_const<128, 2>), 128, 256);
    ADD4("const<256,1>", (k_const<256, 1>), 256, 256);
    ADD4("const<256,2>", (k_const<256, 2>), 256, 512);
    ADD4("glut<128,1,pow2>", (k_glut<128, 1, true>), 128, 128);
    ADD4("glut<128,2,pow2>", (k_glut<128, 2, true>), 128, 256);
    ADD4("glut<256,1,pow2>", (k_glut<256, 1, true>), 256, 256);
    ADD4("glut<256,2,pow2>", (k_glut<256, 2, true>), 256, 512);
    ADD4("glut<128,2,mod>", (k_glut<128, 2, false>), 128, 256);
    ADD4("glut<256,1,mod>", (k_glut<256, 1, false>), 256, 256);
    ADD4("rows<256,1>", (k_rows<256, 1>), 256, 256);
    ADD4("rows<256,2>", (k_rows<256, 2>), 256, 512);
    ADD4("rows<256,4>", (k_rows<256, 4>), 256, 1024);
    ADD4("rows<256,8>", (k_rows<256, 8>), 256, 2048);
    ADD4("glut<64,1,pow2>", (k_glut<64, 1, true>), 64, 64);
    ADD4("glut<64,2,pow2>", (k_glut<64, 2, true>), 64, 128);
    ADD4("glut_first<128>", (k_glut_first<128>), 128, 128);
    ADD4("glut_first<256>", (k_glut_first<256>), 256, 256);
    ADD4("rows2<64,2>", (k_rows2<64, 2>), 64, 128);
    ADD4("rows2<64,4>", (k_rows2<64, 4>), 64, 256);
    ADD4("rows2<128,2>", (k_rows2<128, 2>), 128, 256);
    ADD4("rows2<128,4>", (k_rows2<128, 4>), 128, 512);
    ADD4("rows2<256,2>", (k_rows2<256, 2>), 256, 512);
    ADD4("rows2<64,8>", (k_rows2<64, 8>), 64, 512);
    vs.push_back({"lds<256,4> grid2048", 5, (void *)(k5)k_lds<256, 4>, 256, 0, (int)(P * 8), 2048, {}});
    vs.push_back({"lds<256,4> grid4096", 5, (void *)(k5)k_lds<256, 4>, 256, 0, (int)(P * 8), 4096, {}});
    vs.push_back({"lds<256,1> grid4096", 5, (void *)(k5)k_lds<256, 1>, 256, 0, (int)(P * 8), 4096, {}});
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](Var &v, int iters) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < iters; ++i) {
            if (v.kind == 4) hipLaunchKernelGGL((k4)v.fn, dim3(v.grid), dim3(v.block), 0, 0, in, out, lut, P);
            else hipLaunchKernelGGL((k5)v.fn, dim3(v.grid), dim3(v.block), v.lds, 0, in, out, lut, P, n_vec);
        }
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        return (double)ms / iters;
    };
    for (int w = 0; w < 3; ++w) for (auto &v : vs) run(v, 5);
    for (int r = 0; r < 15; ++r) for (auto &v : vs) v.ms.push_back(run(v, 10));
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const double med = v.ms[v.ms.size() / 2];
        printf("%-24s grid=%7d  med %.4f ms %7.1f GB/s   best %7.1f  worst %7.1f\n", v.name, v.grid, med,
               2.0 * bytes / med / 1e6, 2.0 * bytes / v.ms[0] / 1e6, 2.0 * bytes / v.ms.back() / 1e6);
    }
    return 0;
}
