// pcie_probe — what each way of crossing PCIe gives on this box, alone and against each other (profiles/r06_ring.md).
//   hipcc --offload-arch=gfx950 -O2 tools/pcie_probe.hip -o /tmp/pcie_probe && /tmp/pcie_probe [MiB per transfer] [repeats]
// Legs: copy engine H2D / D2H (hipMemcpyAsync, pinned), a copy kernel loading from / storing to host-mapped memory, and the
// pairs that can run against each other on two streams.  Every figure is bytes of ONE direction per second.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void copy16(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        u32x4 v = NT_LD ? __builtin_nontemporal_load(in + i) : in[i];
        if (NT_ST) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

// persistent form: `groups` workgroups, each striding over the buffer in 4 KiB steps (bounds the requests in flight)
__global__ __launch_bounds__(256) void copy16_loop(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Leg {
    const char *name;
    int kind;        // 0 engine H2D, 1 engine D2H, 2 kernel host->HBM, 3 kernel HBM->host, 4 kernel host->host
    int variant;     // kernels: 0 nt/nt one-shot, 1 plain/plain one-shot, 2 loop with 1024 groups, 3 loop with 256 groups
};

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 256;
    const int reps = argc > 2 ? atoi(argv[2]) : 16;
    const unsigned hflags = argc > 3 ? (unsigned)strtoul(argv[3], nullptr, 0) : 0;
    const size_t bytes = mib << 20, n = bytes / 16;
    CK(hipSetDevice(0));
    char *h[4], *d[4];
    void *m[4];
    for (int i = 0; i < 4; ++i) {
        CK(hipHostMalloc((void **)&h[i], bytes, hipHostMallocPortable | hipHostMallocMapped | hflags));
        memset(h[i], i + 1, bytes);
        CK(hipHostGetDevicePointer(&m[i], h[i], 0));
        CK(hipMalloc((void **)&d[i], bytes));
        CK(hipMemset(d[i], 0, bytes));
    }
    hipStream_t s[4];
    for (int i = 0; i < 4; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    CK(hipDeviceSynchronize());

    auto issue = [&](const Leg &l, int slot, hipStream_t st) {
        switch (l.kind) {
        case 0: CK(hipMemcpyAsync(d[slot], h[slot], bytes, hipMemcpyHostToDevice, st)); break;
        case 1: CK(hipMemcpyAsync(h[slot], d[slot], bytes, hipMemcpyDeviceToHost, st)); break;
        default: {
            const u32x4 *src = (const u32x4 *)(l.kind == 3 ? (void *)d[slot] : m[slot]);
            u32x4 *dst = (u32x4 *)(l.kind == 2 ? (void *)d[slot] : l.kind == 3 ? m[slot] : m[(slot + 1) % 4]);
            const unsigned grid = (unsigned)((n + 255) / 256);
            if (l.variant == 0) copy16<true, true><<<grid, 256, 0, st>>>(src, dst, n);
            else if (l.variant == 1) copy16<false, false><<<grid, 256, 0, st>>>(src, dst, n);
            else copy16_loop<<<l.variant == 2 ? 1024 : 256, 256, 0, st>>>(src, dst, n);
            CK(hipGetLastError());
        }
        }
    };
    auto run = [&](std::vector<Leg> legs) {
        // leg k runs on stream k, slot k (slot 2k / 2k+1 for host->host); all legs issue `reps` transfers back to back
        for (int w = 0; w < 2; ++w) {
            const int r = w == 0 ? 2 : reps;
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int i = 0; i < r; ++i)
                for (size_t k = 0; k < legs.size(); ++k) issue(legs[k], legs[k].kind == 4 ? (int)(2 * k) % 4 : (int)k, s[k]);
            std::vector<double> done(legs.size());
            for (size_t k = 0; k < legs.size(); ++k) { CK(hipStreamSynchronize(s[k])); done[k] = now() - t0; }
            if (w == 1) {
                for (size_t k = 0; k < legs.size(); ++k)
                    printf("%s%-34s %7.2f GB/s", k ? "   ||   " : "", legs[k].name, (double)bytes * reps / done[k] / 1e9);
                printf("\n");
            }
        }
    };
    const Leg eh2d = {"engine H2D", 0, 0}, ed2h = {"engine D2H", 1, 0};
    const Leg kin = {"kernel host->HBM (nt)", 2, 0}, kin_p = {"kernel host->HBM (plain)", 2, 1}, kin_l = {"kernel host->HBM (1024 groups)", 2, 2},
              kin_s = {"kernel host->HBM (256 groups)", 2, 3};
    const Leg kout = {"kernel HBM->host (nt)", 3, 0}, kout_p = {"kernel HBM->host (plain)", 3, 1}, kout_l = {"kernel HBM->host (1024 groups)", 3, 2};
    const Leg kboth = {"kernel host->host (nt)", 4, 0}, kboth_l = {"kernel host->host (1024 groups)", 4, 2};
    printf("%zu MiB per transfer, %d transfers per leg, extra hipHostMalloc flags 0x%x\n", mib, reps, hflags);
    run({eh2d});
    run({ed2h});
    run({eh2d, ed2h});
    run({eh2d, eh2d});
    run({ed2h, ed2h});
    run({eh2d, eh2d, ed2h, ed2h});
    run({kin});
    run({kin_p});
    run({kin_l});
    run({kin_s});
    run({kout});
    run({kout_p});
    run({kout_l});
    run({kboth});
    run({kboth_l});
    run({eh2d, kout});
    run({eh2d, kout_l});
    run({eh2d, eh2d, kout});
    run({kin, ed2h});
    run({kin, kout});
    run({kin_l, kout_l});
    return 0;
}
