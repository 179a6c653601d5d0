// membench.hip — what does MI355X's memory system give a 1 GiB -> 1 GiB stream?
// Development tool (not part of the library): sweeps copy-kernel shapes so the
// fused kernel's access pattern can be chosen from measurements.
//   hipcc --offload-arch=gfx950 -O3 -o membench tools/membench.hip && ./membench [bytes]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

enum { LD_PLAIN = 0, LD_NT = 1 };
enum { ST_PLAIN = 0, ST_NT = 1 };

template <int LD> __device__ __forceinline__ u32x4 ld(const u32x4 *p)
{
    if constexpr (LD == LD_NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int ST> __device__ __forceinline__ void st(u32x4 *p, u32x4 v)
{
    if constexpr (ST == ST_NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

__global__ void fill(uint32_t *p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = x;
    }
}

// block-cyclic tiles: tile t -> block t % grid.  U vectors per lane per tile.
template <int BLOCK, int U, int LD, int ST>
__global__ __launch_bounds__(BLOCK) void copy_cyclic(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n_vec)
{
    const uint64_t tile = (uint64_t)BLOCK * U;
    for (uint64_t t0 = (uint64_t)blockIdx.x * tile; t0 < n_vec; t0 += (uint64_t)gridDim.x * tile) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<LD>(in + t0 + (uint64_t)u * BLOCK + threadIdx.x);
#pragma unroll
        for (int u = 0; u < U; ++u) st<ST>(out + t0 + (uint64_t)u * BLOCK + threadIdx.x, v[u]);
    }
}

// contiguous chunk per block: block b owns [b*chunk, (b+1)*chunk)
template <int BLOCK, int U, int LD, int ST>
__global__ __launch_bounds__(BLOCK) void copy_chunk(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n_vec)
{
    const uint64_t tile = (uint64_t)BLOCK * U;
    const uint64_t tiles = n_vec / tile;
    const uint64_t per = (tiles + gridDim.x - 1) / gridDim.x;
    const uint64_t b0 = (uint64_t)blockIdx.x * per, b1 = (b0 + per < tiles) ? b0 + per : tiles;
    for (uint64_t t = b0; t < b1; ++t) {
        const uint64_t t0 = t * tile;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<LD>(in + t0 + (uint64_t)u * BLOCK + threadIdx.x);
#pragma unroll
        for (int u = 0; u < U; ++u) st<ST>(out + t0 + (uint64_t)u * BLOCK + threadIdx.x, v[u]);
    }
}

// XCD-aware: blocks of one XCD (b % 8) sweep one contiguous eighth of the buffer cyclically
template <int BLOCK, int U, int LD, int ST>
__global__ __launch_bounds__(BLOCK) void copy_xcd(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n_vec)
{
    const uint64_t tile = (uint64_t)BLOCK * U;
    const uint64_t tiles = n_vec / tile;
    const uint32_t xcd = blockIdx.x & 7u, lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
    const uint64_t per = tiles / 8;
    for (uint64_t t = lb; t < per; t += nlb) {
        const uint64_t t0 = ((uint64_t)xcd * per + t) * tile;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<LD>(in + t0 + (uint64_t)u * BLOCK + threadIdx.x);
#pragma unroll
        for (int u = 0; u < U; ++u) st<ST>(out + t0 + (uint64_t)u * BLOCK + threadIdx.x, v[u]);
    }
}

// one-shot: every block copies exactly one tile (grid = tiles)
template <int BLOCK, int U, int LD, int ST>
__global__ __launch_bounds__(BLOCK) void copy_oneshot(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n_vec)
{
    const uint64_t t0 = (uint64_t)blockIdx.x * BLOCK * U;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<LD>(in + t0 + (uint64_t)u * BLOCK + threadIdx.x);
#pragma unroll
    for (int u = 0; u < U; ++u) st<ST>(out + t0 + (uint64_t)u * BLOCK + threadIdx.x, v[u]);
}

// read-only and write-only streams (what each direction can do alone)
template <int BLOCK, int U, int LD>
__global__ __launch_bounds__(BLOCK) void read_only(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n_vec)
{
    const uint64_t tile = (uint64_t)BLOCK * U;
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t t0 = (uint64_t)blockIdx.x * tile; t0 < n_vec; t0 += (uint64_t)gridDim.x * tile) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= ld<LD>(in + t0 + (uint64_t)u * BLOCK + threadIdx.x);
    }
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) out[threadIdx.x] = acc;
}
template <int BLOCK, int U, int ST>
__global__ __launch_bounds__(BLOCK) void write_only(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n_vec)
{
    const uint64_t tile = (uint64_t)BLOCK * U;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (uint64_t t0 = (uint64_t)blockIdx.x * tile; t0 < n_vec; t0 += (uint64_t)gridDim.x * tile) {
#pragma unroll
        for (int u = 0; u < U; ++u) st<ST>(out + t0 + (uint64_t)u * BLOCK + threadIdx.x, v);
    }
}

// store flavours by cache-policy bits (gfx950: sc0, sc1, nt)
#define ASM_STORE(NAME, BITS)                                                                  \
    __device__ __forceinline__ void NAME(u32x4 *p, u32x4 v)                                     \
    {                                                                                           \
        asm volatile("global_store_dwordx4 %0, %1, off " BITS : : "v"(p), "v"(v) : "memory"); \
    }
ASM_STORE(st_sc0, "sc0")
ASM_STORE(st_sc1, "sc1")
ASM_STORE(st_sc0sc1, "sc0 sc1")
ASM_STORE(st_sc1nt, "sc1 nt")
ASM_STORE(st_sc0nt, "sc0 nt")
ASM_STORE(st_all, "sc0 sc1 nt")

template <int F> __device__ __forceinline__ void stf(u32x4 *p, u32x4 v)
{
    if constexpr (F == 0) *p = v;
    else if constexpr (F == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (F == 2) st_sc0(p, v);
    else if constexpr (F == 3) st_sc1(p, v);
    else if constexpr (F == 4) st_sc0sc1(p, v);
    else if constexpr (F == 5) st_sc1nt(p, v);
    else if constexpr (F == 6) st_sc0nt(p, v);
    else st_all(p, v);
}

template <int BLOCK, int U, int F>
__global__ __launch_bounds__(BLOCK) void write_flavour(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n_vec)
{
    const uint64_t tile = (uint64_t)BLOCK * U;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (uint64_t t0 = (uint64_t)blockIdx.x * tile; t0 < n_vec; t0 += (uint64_t)gridDim.x * tile) {
#pragma unroll
        for (int u = 0; u < U; ++u) stf<F>(out + t0 + (uint64_t)u * BLOCK + threadIdx.x, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int BLOCK, int U, int F>
__global__ __launch_bounds__(BLOCK) void copy_flavour(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n_vec)
{
    const uint64_t tile = (uint64_t)BLOCK * U;
    for (uint64_t t0 = (uint64_t)blockIdx.x * tile; t0 < n_vec; t0 += (uint64_t)gridDim.x * tile) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(in + t0 + (uint64_t)u * BLOCK + threadIdx.x);
#pragma unroll
        for (int u = 0; u < U; ++u) stf<F>(out + t0 + (uint64_t)u * BLOCK + threadIdx.x, v[u]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// software-pipelined persistent copy: loads of tile i+1 are issued before the stores of tile i
template <int BLOCK, int U, int LD, int ST>
__global__ __launch_bounds__(BLOCK) void copy_prefetch(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n_vec)
{
    const uint64_t tile = (uint64_t)BLOCK * U;
    const uint64_t stride = (uint64_t)gridDim.x * tile;
    uint64_t t0 = (uint64_t)blockIdx.x * tile;
    if (t0 >= n_vec) return;
    u32x4 cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = ld<LD>(in + t0 + (uint64_t)u * BLOCK + threadIdx.x);
    for (;;) {
        const uint64_t t1 = t0 + stride;
        const bool more = t1 < n_vec;
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = ld<LD>(in + t1 + (uint64_t)u * BLOCK + threadIdx.x);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) st<ST>(out + t0 + (uint64_t)u * BLOCK + threadIdx.x, cur[u]);
        if (!more) break;
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        t0 = t1;
    }
}

template <int BLOCK, int U, int ST>
__global__ __launch_bounds__(BLOCK) void write_oneshot(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n_vec)
{
    const uint64_t t0 = (uint64_t)blockIdx.x * BLOCK * U;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
#pragma unroll
    for (int u = 0; u < U; ++u) st<ST>(out + t0 + (uint64_t)u * BLOCK + threadIdx.x, v);
}

// persistent, XCD-interleaved: the stream is cut into units of BLOCK*16 bytes; unit q is handled by a
// workgroup whose (blockIdx + SHIFT) % 8 == q % 8, i.e. (with block b on XCD b % 8 and SHIFT = 0) every
// XCD touches exactly the units a one-shot launch would give it.
template <int BLOCK, int U, int SHIFT, bool PRE>
__global__ __launch_bounds__(BLOCK) void copy_xcdunit(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n_vec)
{
    const uint32_t xcd = (blockIdx.x + SHIFT) & 7u, lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
    const uint64_t units = n_vec / BLOCK;          // units of BLOCK vectors
    const uint64_t per_xcd = units / 8;            // unit q = 8*k + xcd, k in [0, per_xcd)
    u32x4 cur[U], nxt[U];
    uint64_t k = (uint64_t)lb * U;
    if (k >= per_xcd) return;
    if constexpr (!PRE) {
        for (; k < per_xcd; k += (uint64_t)nlb * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = __builtin_nontemporal_load(in + (8 * (k + u) + xcd) * BLOCK + threadIdx.x);
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_nontemporal_store(cur[u], out + (8 * (k + u) + xcd) * BLOCK + threadIdx.x);
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = __builtin_nontemporal_load(in + (8 * (k + u) + xcd) * BLOCK + threadIdx.x);
        for (;;) {
            const uint64_t k1 = k + (uint64_t)nlb * U;
            const bool more = k1 < per_xcd;
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) nxt[u] = __builtin_nontemporal_load(in + (8 * (k1 + u) + xcd) * BLOCK + threadIdx.x);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_nontemporal_store(cur[u], out + (8 * (k + u) + xcd) * BLOCK + threadIdx.x);
            if (!more) break;
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            k = k1;
        }
    }
}

typedef void (*kern_t)(const u32x4 *, u32x4 *, uint64_t);

struct Variant { const char *name; kern_t k; int block; int unroll; int grid_mode; /*0 = CUs*bpc, 1 = tiles*/ double bytes_factor; };

static double run(const Variant &v, const u32x4 *in, u32x4 *out, uint64_t n_vec, int grid, int iters)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(v.k, dim3(grid), dim3(v.block), 0, 0, in, out, n_vec);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(v.k, dim3(grid), dim3(v.block), 0, 0, in, out, n_vec);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        ts.push_back(ms / iters);
    }
    std::sort(ts.begin(), ts.end());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ts[ts.size() / 2];
}

#define V4(K, B, U) \
    {#K "<" #B "," #U ",plain,plain>", K<B, U, LD_PLAIN, ST_PLAIN>, B, U, 0, 2.0}, \
    {#K "<" #B "," #U ",nt,plain>", K<B, U, LD_NT, ST_PLAIN>, B, U, 0, 2.0}, \
    {#K "<" #B "," #U ",plain,nt>", K<B, U, LD_PLAIN, ST_NT>, B, U, 0, 2.0}, \
    {#K "<" #B "," #U ",nt,nt>", K<B, U, LD_NT, ST_NT>, B, U, 0, 2.0}

int main(int argc, char **argv)
{
    uint64_t bytes = argc > 1 ? strtoull(argv[1], 0, 0) : (1ull << 30);
    const uint64_t n_vec = bytes / 16;
    u32x4 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    fill<<<4096, 256>>>((uint32_t *)in, bytes / 4);
    CK(hipDeviceSynchronize());
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, buffer %llu MiB each way\n", prop.gcnArchName, cus, (unsigned long long)(bytes >> 20));

    if (argc > 2) {   // vendor baselines
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(a, 0));
            for (int i = 0; i < 10; ++i) CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0));
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("%-40s %.4f ms  %8.1f GB/s\n", "hipMemcpyDtoD", ms / 10, 2.0 * bytes / (ms / 10) / 1e6);
            CK(hipEventRecord(a, 0));
            for (int i = 0; i < 10; ++i) CK(hipMemsetAsync(out, 0x5a, bytes, 0));
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
            printf("%-40s %.4f ms  %8.1f GB/s\n", "hipMemset", ms / 10, 1.0 * bytes / (ms / 10) / 1e6);
        }
    }
#define WF(F) {"write_flavour<256,4," #F ">", write_flavour<256, 4, F>, 256, 4, 0, 1.0}
#define CF(F) {"copy_flavour<256,4," #F ">", copy_flavour<256, 4, F>, 256, 4, 0, 2.0}
    std::vector<Variant> vs2 = {
        WF(0), WF(1), WF(2), WF(3), WF(4), WF(5), WF(6), WF(7),
        CF(0), CF(1), CF(2), CF(3), CF(4), CF(5), CF(6), CF(7),
        {"write_flavour<256,8,0>", write_flavour<256, 8, 0>, 256, 8, 0, 1.0},
        {"write_flavour<256,1,0>", write_flavour<256, 1, 0>, 256, 1, 0, 1.0},
        {"write_flavour<1024,4,0>", write_flavour<1024, 4, 0>, 1024, 4, 0, 1.0},
        {"write_oneshot<256,4,plain>", write_oneshot<256, 4, ST_PLAIN>, 256, 4, 1, 1.0},
        {"write_oneshot<256,8,nt>", write_oneshot<256, 8, ST_NT>, 256, 8, 1, 1.0},
        {"write_oneshot<256,16,plain>", write_oneshot<256, 16, ST_PLAIN>, 256, 16, 1, 1.0},
        {"copy_prefetch<256,4,nt,nt>", copy_prefetch<256, 4, LD_NT, ST_NT>, 256, 4, 0, 2.0},
        {"copy_prefetch<256,2,nt,nt>", copy_prefetch<256, 2, LD_NT, ST_NT>, 256, 2, 0, 2.0},
        {"copy_prefetch<256,8,nt,nt>", copy_prefetch<256, 8, LD_NT, ST_NT>, 256, 8, 0, 2.0},
        {"copy_prefetch<256,4,nt,plain>", copy_prefetch<256, 4, LD_NT, ST_PLAIN>, 256, 4, 0, 2.0},
        {"copy_prefetch<512,4,nt,nt>", copy_prefetch<512, 4, LD_NT, ST_NT>, 512, 4, 0, 2.0},
        {"copy_oneshot<256,16,nt,nt>", copy_oneshot<256, 16, LD_NT, ST_NT>, 256, 16, 1, 2.0},
        {"copy_oneshot<512,8,nt,nt>", copy_oneshot<512, 8, LD_NT, ST_NT>, 512, 8, 1, 2.0},
        {"copy_oneshot<256,8,nt,plain>", copy_oneshot<256, 8, LD_NT, ST_PLAIN>, 256, 8, 1, 2.0},
        {"copy_oneshot<256,2,nt,nt>", copy_oneshot<256, 2, LD_NT, ST_NT>, 256, 2, 1, 2.0},
        {"copy_oneshot<64,8,nt,nt>", copy_oneshot<64, 8, LD_NT, ST_NT>, 64, 8, 1, 2.0},
        {"copy_oneshot<128,8,nt,nt>", copy_oneshot<128, 8, LD_NT, ST_NT>, 128, 8, 1, 2.0},
    };
    std::vector<Variant> vs = {
        V4(copy_cyclic, 256, 1), V4(copy_cyclic, 256, 2), V4(copy_cyclic, 256, 4), V4(copy_cyclic, 256, 8),
        V4(copy_cyclic, 512, 4), V4(copy_cyclic, 1024, 2), V4(copy_cyclic, 1024, 4),
        V4(copy_chunk, 256, 4), V4(copy_xcd, 256, 4), V4(copy_xcd, 256, 8),
        {"copy_oneshot<256,4,nt,nt>", copy_oneshot<256, 4, LD_NT, ST_NT>, 256, 4, 1, 2.0},
        {"copy_oneshot<256,8,nt,nt>", copy_oneshot<256, 8, LD_NT, ST_NT>, 256, 8, 1, 2.0},
        {"copy_oneshot<256,4,plain,plain>", copy_oneshot<256, 4, LD_PLAIN, ST_PLAIN>, 256, 4, 1, 2.0},
        {"copy_oneshot<1024,4,nt,nt>", copy_oneshot<1024, 4, LD_NT, ST_NT>, 1024, 4, 1, 2.0},
        {"read_only<256,4,plain>", read_only<256, 4, LD_PLAIN>, 256, 4, 0, 1.0},
        {"read_only<256,4,nt>", read_only<256, 4, LD_NT>, 256, 4, 0, 1.0},
        {"read_only<256,8,nt>", read_only<256, 8, LD_NT>, 256, 8, 0, 1.0},
        {"write_only<256,4,plain>", write_only<256, 4, ST_PLAIN>, 256, 4, 0, 1.0},
        {"write_only<256,4,nt>", write_only<256, 4, ST_NT>, 256, 4, 0, 1.0},
    };
#define OS(B, U) {"copy_oneshot<" #B "," #U ",nt,nt>", copy_oneshot<B, U, LD_NT, ST_NT>, B, U, 1, 2.0}
#define PF(B, U) {"copy_prefetch<" #B "," #U ",nt,nt>", copy_prefetch<B, U, LD_NT, ST_NT>, B, U, 2, 2.0}
#define CY(B, U) {"copy_cyclic<" #B "," #U ",nt,nt>", copy_cyclic<B, U, LD_NT, ST_NT>, B, U, 2, 2.0}
    std::vector<Variant> vs3 = {
        OS(64, 4), OS(64, 8), OS(64, 16), OS(128, 2), OS(128, 4), OS(128, 8), OS(128, 16), OS(256, 2), OS(256, 4), OS(256, 8),
        PF(64, 4), PF(64, 8), PF(128, 2), PF(128, 4), PF(128, 8), PF(256, 1), PF(256, 2), PF(256, 4), PF(512, 2), PF(512, 4),
        CY(128, 4), CY(128, 8), CY(256, 4),
    };
    if (argc > 2 && atoi(argv[2]) == 4) {
        // finalists, round-robin interleaved so that clock / thermal drift hits all of them equally
        struct Fin { Variant v; int grid; std::vector<double> ms; };
        auto tiles = [&](int b, int u) { return (int)(n_vec / ((uint64_t)b * u)); };
#define WO(B, U) {"write_oneshot<" #B "," #U ",plain>", write_oneshot<B, U, ST_PLAIN>, B, U, 1, 1.0}
        std::vector<Fin> fs = {
            {OS(128, 2), tiles(128, 2), {}}, {OS(64, 4), tiles(64, 4), {}}, {OS(128, 4), tiles(128, 4), {}},
            {OS(128, 8), tiles(128, 8), {}}, {OS(256, 2), tiles(256, 2), {}}, {OS(256, 4), tiles(256, 4), {}},
            {OS(256, 8), tiles(256, 8), {}}, {OS(64, 2), tiles(64, 2), {}}, {OS(128, 1), tiles(128, 1), {}},
            {OS(256, 1), tiles(256, 1), {}},
            {CY(256, 4), 4096, {}}, {CY(256, 4), 2048, {}}, {CY(256, 4), 8192, {}}, {CY(128, 4), 8192, {}},
            {PF(256, 4), 512, {}}, {PF(256, 4), 2048, {}}, {PF(512, 4), 256, {}}, {PF(256, 2), 2048, {}},
            {WO(256, 4), tiles(256, 4), {}}, {WO(256, 1), tiles(256, 1), {}}, {WO(128, 2), tiles(128, 2), {}},
            {WO(64, 4), tiles(64, 4), {}}, {WO(1024, 1), tiles(1024, 1), {}}, {WO(256, 2), tiles(256, 2), {}},
        };
#define XU(B, U, S, P) {"copy_xcdunit<" #B "," #U ",shift" #S ",pre" #P ">", copy_xcdunit<B, U, S, P>, B, U, 0, 2.0}
        if (argc > 3) {
            fs = {
                {OS(128, 1), tiles(128, 1), {}}, {OS(256, 1), tiles(256, 1), {}},
                {XU(256, 1, 0, false), 2048, {}}, {XU(256, 1, 0, false), 4096, {}}, {XU(256, 1, 0, false), 8192, {}},
                {XU(256, 1, 1, false), 4096, {}}, {XU(256, 1, 4, false), 4096, {}},
                {XU(256, 2, 0, false), 2048, {}}, {XU(256, 2, 0, false), 4096, {}},
                {XU(256, 4, 0, false), 2048, {}}, {XU(256, 4, 0, false), 4096, {}}, {XU(256, 4, 1, false), 2048, {}},
                {XU(256, 1, 0, true), 2048, {}}, {XU(256, 1, 0, true), 4096, {}},
                {XU(256, 2, 0, true), 2048, {}}, {XU(256, 2, 0, true), 1024, {}},
                {XU(256, 4, 0, true), 512, {}}, {XU(256, 4, 0, true), 1024, {}}, {XU(256, 4, 0, true), 2048, {}},
                {XU(128, 1, 0, false), 4096, {}}, {XU(128, 1, 0, false), 8192, {}}, {XU(128, 2, 0, false), 8192, {}},
                {XU(128, 2, 0, true), 4096, {}}, {XU(128, 4, 0, true), 4096, {}}, {XU(128, 4, 0, false), 4096, {}},
                {XU(512, 1, 0, false), 2048, {}}, {XU(512, 2, 0, true), 1024, {}},
            };
        }
        for (int w = 0; w < 3; ++w) for (auto &f : fs) run(f.v, in, out, n_vec, f.grid, 5);   // warm up ~1 s
        for (int r = 0; r < 15; ++r) for (auto &f : fs) f.ms.push_back(run(f.v, in, out, n_vec, f.grid, 10));
        for (auto &f : fs) {
            std::sort(f.ms.begin(), f.ms.end());
            const double med = f.ms[f.ms.size() / 2], mn = f.ms[0], mx = f.ms.back();
            printf("%-34s grid=%7d  med %.4f ms %7.1f GB/s   best %7.1f  worst %7.1f\n", f.v.name, f.grid, med,
                   f.v.bytes_factor * bytes / med / 1e6, f.v.bytes_factor * bytes / mn / 1e6, f.v.bytes_factor * bytes / mx / 1e6);
        }
        return 0;
    }
    const int bpcs[] = {2, 4, 8, 16, 32};
    if (argc > 2) vs = atoi(argv[2]) == 3 ? vs3 : vs2;
    for (const Variant &v : vs) {
        if (v.grid_mode == 1) {
            const int grid = (int)(n_vec / ((uint64_t)v.block * v.unroll));
            const double ms = run(v, in, out, n_vec, grid, 10);
            printf("%-40s grid=%8d  %.4f ms  %8.1f GB/s\n", v.name, grid, ms, v.bytes_factor * bytes / ms / 1e6);
            continue;
        }
        if (v.grid_mode == 2) {   // waves-per-CU sweep: grid = CUs * wpc * 64 / block
            const int wpcs[] = {4, 8, 12, 16, 24, 32, 64};
            for (int wpc : wpcs) {
                const int grid = cus * wpc * 64 / v.block;
                if (grid < cus) continue;
                double best = 1e9, sum = 0;
                for (int r = 0; r < 3; ++r) { const double ms = run(v, in, out, n_vec, grid, 10); best = ms < best ? ms : best; sum += ms; }
                printf("%-40s grid=%8d  %.4f ms  %8.1f GB/s  (best %8.1f)\n", v.name, grid, sum / 3, v.bytes_factor * bytes / (sum / 3) / 1e6, v.bytes_factor * bytes / best / 1e6);
            }
            fflush(stdout);
            continue;
        }
        for (int bpc : bpcs) {
            if ((int64_t)bpc * v.block > 2048 * 2) continue;   // more than the CU can hold twice over: skip
            const int grid = cus * bpc;
            const double ms = run(v, in, out, n_vec, grid, 10);
            printf("%-40s grid=%8d  %.4f ms  %8.1f GB/s\n", v.name, grid, ms, v.bytes_factor * bytes / ms / 1e6);
        }
        fflush(stdout);
    }
    return 0;
}
