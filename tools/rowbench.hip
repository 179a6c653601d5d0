// rowbench.hip — what does the memory system give a wavefront that takes R rows of a matrix, D bytes apart?
// Development tool (round 3): the access pattern of the walk / span kernels without their arithmetic, so that the
// ceiling of a (rows per wavefront, row distance, workgroup shape) choice is known before the kernel is written.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/rowbench tools/rowbench.hip && tools/bin/rowbench
// A stream of N bytes in, N bytes out (i16 -> i16: 4 B per sample on both sides) is viewed as rows of Lb bytes; a
// chunk is R consecutive rows; a one-wavefront workgroup takes one 1 KiB window of the R rows of a chunk (16 bytes per
// lane per row, non-temporal), optionally spends VALU work the way the real kernel does (WORK_FIX dependent f64 fmas
// once per wavefront: the slice of correctors; WORK_ROW f32 fmas per row on the loaded data: unpack / mix / pack),
// stores, and exits.  Windows are consecutive workgroups (blockIdx.x), chunks are blockIdx.y.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void fill(uint32_t *p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = x;
    }
}

__device__ __forceinline__ double fix_work(int n, double x)
{
    // n dependent f64 fmas (the serial part of a sincos evaluation)
    double a = x;
    for (int i = 0; i < n; ++i) a = __builtin_fma(a, 0.999999, 1e-9);
    return a;
}

__device__ __forceinline__ u32x4 row_work(int n, u32x4 v, float k)
{
    // n f32 fmas on the loaded data
    float a = __uint_as_float(v[0] & 0x3fffffffu), b = __uint_as_float(v[1] & 0x3fffffffu);
    for (int i = 0; i < n; i += 2) { a = __builtin_fmaf(a, k, b); b = __builtin_fmaf(b, k, a); }
    v[0] ^= __float_as_uint(a) & 1u;
    v[1] ^= __float_as_uint(b) & 1u;
    return v;
}

// one wavefront per workgroup, R rows
template <int R>
__global__ __launch_bounds__(64) void tall(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint64_t Lb,
                                           int work_fix, int work_row, float k)
{
    const uint64_t base = (uint64_t)blockIdx.y * R * Lb + (uint64_t)blockIdx.x * 1024u + threadIdx.x * 16u;
    u32x4 q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) q[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in + base + (uint64_t)r * Lb));
    uint32_t x = 0;
    if (work_fix) x = (uint32_t)(fix_work(work_fix, (double)threadIdx.x) > 1e300);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        u32x4 v = q[r];
        if (work_row) v = row_work(work_row, v, k);
        v[2] ^= x;
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out + base + (uint64_t)r * Lb));
    }
}

// WAVES wavefronts x U rows per workgroup with a barrier between the loads and the stores (today's walk kernel);
// rows_in_chunk <= WAVES * U: wavefronts past it leave at once
template <int WAVES, int U, bool BAR = true>
__global__ __launch_bounds__(WAVES * 64) void wgk(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint64_t Lb,
                                                  int rows_in_chunk, int work_fix, int work_row, float k)
{
    __shared__ double sl[WAVES * 64 + 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int r0 = (int)wave * U;
    if (r0 >= rows_in_chunk) return;
    const uint64_t base = ((uint64_t)blockIdx.y * rows_in_chunk + r0) * Lb + (uint64_t)blockIdx.x * 1024u + lane * 16u;
    u32x4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool valid = r0 + u < rows_in_chunk;
        q[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in + (valid ? base + (uint64_t)u * Lb : (uint64_t)lane * 16u)));
    }
    uint32_t x = 0;
    if constexpr (BAR) {
        if (work_fix) sl[threadIdx.x] = fix_work(work_fix, (double)threadIdx.x);
        __syncthreads();
        if (work_fix) x = (uint32_t)(sl[threadIdx.x ^ 1u] > 1e300);
    } else {      // the same workgroup without its barrier (round 4: is it the grouping or the barrier that costs?)
        if (work_fix) x = (uint32_t)(fix_work(work_fix, (double)threadIdx.x) > 1e300);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (r0 + u >= rows_in_chunk) break;
        u32x4 v = q[u];
        if (work_row) v = row_work(work_row, v, k);
        v[2] ^= x;
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out + base + (uint64_t)u * Lb));
    }
}

struct Case {
    std::string name;
    uint64_t Lb;
    int R;            // rows per chunk
    int waves, U;     // 0,0: tall<R>
    int work_fix, work_row;
    std::vector<double> ms;
    uint64_t bytes;
};

static uint8_t *g_in, *g_out;
static uint64_t g_n;

template <int R> static void launch_tall(const Case &c, dim3 grid) { tall<R><<<grid, 64>>>(g_in, g_out, c.Lb, c.work_fix, c.work_row, 0.999f); }
static bool g_nobar = false;
template <int W, int U> static void launch_wg(const Case &c, dim3 grid)
{
    if (g_nobar) wgk<W, U, false><<<grid, W * 64>>>(g_in, g_out, c.Lb, c.R, c.work_fix, c.work_row, 0.999f);
    else         wgk<W, U, true><<<grid, W * 64>>>(g_in, g_out, c.Lb, c.R, c.work_fix, c.work_row, 0.999f);
}

static bool launch(Case &c)
{
    const uint64_t rows = g_n / c.Lb, chunks = rows / c.R;
    if (chunks == 0 || chunks > 65535) { printf("%s: %llu chunks do not fit a 2-D grid\n", c.name.c_str(), (unsigned long long)chunks); return false; }
    const dim3 grid((uint32_t)(c.Lb / 1024), (uint32_t)chunks);
    c.bytes = 2ull * chunks * c.R * (c.Lb / 1024) * 1024;
    if (c.waves == 0) {
        switch (c.R) {
#define T(RR) case RR: launch_tall<RR>(c, grid); break;
            T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(8) T(9) T(10) T(12) T(14) T(16)
#undef T
            default: return false;
        }
    } else {
        g_nobar = c.name.find("nobar") != std::string::npos;
#define W(WW, UU) if (c.waves == WW && c.U == UU) { launch_wg<WW, UU>(c, grid); return true; }
        W(4, 2) W(5, 2) W(8, 2) W(3, 3) W(4, 3) W(2, 4) W(2, 5) W(2, 6) W(2, 8) W(3, 4) W(4, 4)
#undef W
        return false;
    }
    return true;
}

static double burst(Case &c, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) if (!launch(c)) return -1;
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    g_n = 1ull << 30;
    const char *set = argc > 1 ? argv[1] : "all";
    CK(hipMalloc(&g_in, g_n + (64ull << 20)));
    CK(hipMalloc(&g_out, g_n + (64ull << 20)));
    fill<<<4096, 256>>>((uint32_t *)g_in, g_n / 4);
    CK(hipMemset(g_out, 0, g_n));
    CK(hipDeviceSynchronize());

    std::vector<Case> cs;
    auto add = [&](const char *tag, uint64_t Lb, int R, int waves, int U, int wf, int wr) {
        char nm[160];
        if (waves) snprintf(nm, sizeof nm, "%s wg %dx%d rows=%d Lb=%llu fix=%d row=%d", tag, waves, U, R, (unsigned long long)Lb, wf, wr);
        else snprintf(nm, sizeof nm, "%s tall R=%d Lb=%llu fix=%d row=%d", tag, R, (unsigned long long)Lb, wf, wr);
        cs.push_back({nm, Lb, R, waves, U, wf, wr, {}, 0});
    };
    const uint64_t K = 1024;
    const bool all = !strcmp(set, "all");
    if (all || !strcmp(set, "dist")) {
        // A: the distance between the two rows of a wavefront, nothing else
        for (uint64_t Lb : {16 * K, 24 * K, 32 * K, 40 * K, 48 * K, 64 * K, 96 * K, 128 * K, 256 * K, 441 * K, 1000 * K, 4000 * K}) add("A", Lb, 2, 0, 0, 0, 0);
        for (uint64_t Lb : {32 * K, 64 * K, 441 * K}) add("A", Lb, 1, 0, 0, 0, 0);
    }
    if (all || !strcmp(set, "tall")) {
        // B: rows per wavefront at the row distance of a track-mode second (P = 113 027 -> 441.5 KiB)
        for (int R : {1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 16}) add("B", 441 * K, R, 0, 0, 0, 0);
        for (int R : {2, 4, 8, 9, 16}) add("B", 32 * K, R, 0, 0, 0, 0);
        for (int R : {2, 4, 9}) add("B", 4000 * K, R, 0, 0, 0, 0);
    }
    if (all || !strcmp(set, "work")) {
        // C: with the arithmetic: 288 sincos per slice = 4.5 per lane x ~45 instructions for a single wavefront
        // (1.125 per thread of a 4-wavefront workgroup), 48 instructions per lane per row
        for (int R : {2, 4, 5, 6, 8, 9, 12, 16}) add("C", 441 * K, R, 0, 0, 200, 48);
        add("C", 441 * K, 8, 4, 2, 50, 48);
        add("C", 441 * K, 9, 5, 2, 40, 48);
        add("C", 441 * K, 5, 4, 2, 50, 48);
        add("C", 441 * K, 4, 4, 2, 50, 48);
        add("C", 441 * K, 9, 3, 3, 67, 48);
        add("C", 441 * K, 9, 4, 3, 50, 48);
        add("C", 441 * K, 10, 5, 2, 40, 48);
        add("C", 441 * K, 16, 8, 2, 25, 48);
        add("C", 441 * K, 8, 2, 4, 100, 48);
        add("C", 441 * K, 9, 2, 5, 100, 48);
        add("C", 441 * K, 12, 2, 6, 100, 48);
        add("C", 441 * K, 16, 2, 8, 100, 48);
        add("C", 441 * K, 12, 3, 4, 67, 48);
        add("C", 441 * K, 16, 4, 4, 50, 48);
        // no arithmetic, workgroup shapes only
        add("C0", 441 * K, 8, 4, 2, 0, 0);
        add("C0", 441 * K, 9, 5, 2, 0, 0);
        add("C0", 441 * K, 9, 3, 3, 0, 0);
        add("C0", 441 * K, 8, 2, 4, 0, 0);
        add("C0", 441 * K, 16, 2, 8, 0, 0);
    }
    if (all || !strcmp(set, "align")) {
        // D: rows that start on 128-byte lines but not on 1 KiB boundaries (an odd period: 113 027 samples = 452 108 bytes,
        // rows at the 128-byte boundary below), against rows of whole KiB
        for (uint64_t Lb : {441 * K, 441 * K + 128, 441 * K + 512, 441 * K + 640, (uint64_t)452096}) {
            add("D", Lb, 2, 0, 0, 0, 0);
            add("D", Lb, 8, 4, 2, 50, 48);
            add("D", Lb, 9, 5, 2, 40, 48);
            add("D", Lb, 5, 4, 2, 50, 48);
        }
    }
    if (!strcmp(set, "group")) {
        // H (round 4): what costs a 4-wavefront workgroup 4 points against four one-wavefront workgroups — the grouping or
        // the barrier?  Same windows, same rows (441 KiB and 32 KiB apart), with and without the barrier.
        for (uint64_t Lb : {441 * K, 32 * K}) {
            add("H", Lb, 2, 0, 0, 0, 0);
            add("H", Lb, 8, 0, 0, 0, 0);
            add("H", Lb, 8, 4, 2, 0, 0);
            add("H nobar", Lb, 8, 4, 2, 0, 0);
            add("H", Lb, 4, 2, 4, 0, 0);         // (2 wavefronts: rows_in_chunk 4 = 2 x 2)
            add("H", Lb, 16, 8, 2, 0, 0);
            add("H nobar", Lb, 16, 8, 2, 0, 0);
            add("H", Lb, 8, 4, 2, 50, 48);
            add("H nobar", Lb, 8, 4, 2, 50, 48);
            add("H", Lb, 2, 0, 0, 200, 48);
        }
    }
    if (!strcmp(set, "pitch")) {
        // E: the row pitch in fine steps (4 KiB from 1 MiB to 1.5 MiB, then coarser), 4 wavefronts x 2 rows: is there a
        // structure in the pitch (channel / bank interleave) that a planner could aim for?
        for (uint64_t Lb = 1024 * K; Lb < 1536 * K; Lb += 4 * K) add("E", Lb, 8, 4, 2, 0, 0);
        for (uint64_t Lb = 256 * K; Lb < 1024 * K; Lb += 32 * K) add("E", Lb, 8, 4, 2, 0, 0);
        for (uint64_t Lb = 1536 * K; Lb <= 4096 * K; Lb += 64 * K) add("E", Lb, 8, 4, 2, 0, 0);
    }
    if (!strcmp(set, "pitch3")) {       // the whole range a planner can choose from, 1 KiB steps
        for (uint64_t Lb = 128 * K; Lb <= 2304 * K; Lb += 1 * K) add("G", Lb, 8, 4, 2, 0, 0);
    }
    if (!strcmp(set, "pitch2")) {
        for (uint64_t Lb = 1024 * K; Lb < 1152 * K; Lb += 1 * K) add("F", Lb, 8, 4, 2, 0, 0);
    }
    const int rounds = cs.size() > 500 ? 3 : 7, reps = cs.size() > 500 ? 5 : 8;
    for (int w = 0; w < 2; ++w) for (Case &c : cs) burst(c, 3);
    for (int r = 0; r < rounds; ++r) for (Case &c : cs) c.ms.push_back(burst(c, reps));
    for (Case &c : cs) {
        std::sort(c.ms.begin(), c.ms.end());
        const double med = c.ms[c.ms.size() / 2];
        printf("%-52s med %.4f ms %7.1f GB/s %5.1f %%   best %5.1f worst %5.1f\n", c.name.c_str(), med, c.bytes / med / 1e6,
               c.bytes / med / 1e6 / 80.0, c.bytes / c.ms[0] / 1e6 / 80.0, c.bytes / c.ms.back() / 1e6 / 80.0);
    }
    return 0;
}
