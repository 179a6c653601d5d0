// bar_probe.hip — can the host write device memory directly (large BAR), and what does a doorbell in VRAM cost against one
// in host memory?  hipcc --offload-arch=gfx950 -O2 -o tools/bin/bar_probe tools/bar_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <signal.h>
#include <setjmp.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
// echo kernel: polls `door` (wherever it lives), copies 8 KiB from `in` to `out` (host memory), writes `done` (host memory)
__global__ void echo(volatile uint32_t *door, const u4 *in, u4 *out, volatile uint32_t *done, int rounds)
{
    __shared__ uint32_t s;
    uint32_t last = 0;
    for (int r = 0; r < rounds;) {
        if (threadIdx.x == 0) s = __hip_atomic_load((uint32_t *)door, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        const uint32_t d = s;
        __syncthreads();
        if (d == 0xffffffffu) return;
        if (d == last) continue;
        u4 a = __builtin_nontemporal_load(in + threadIdx.x), b = __builtin_nontemporal_load(in + threadIdx.x + 256);
        a.x ^= d; b.x ^= d;
        __builtin_nontemporal_store(a, out + threadIdx.x);
        __builtin_nontemporal_store(b, out + threadIdx.x + 256);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store((uint32_t *)done, d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        last = d;
        ++r;
    }
}
int main()
{
    CK(hipSetDevice(0));
    char *hbuf, *hdev;
    CK(hipHostMalloc((void **)&hbuf, 1 << 20, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&hdev, hbuf, 0));
    memset(hbuf, 0, 1 << 20);
    char *vram = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&vram, 1 << 20, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    bool host_can_write = false;
    if (e == hipSuccess) {
        CK(hipMemset(vram, 0, 1 << 20));
        signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
        if (sigsetjmp(jb, 1) == 0) {
            volatile uint32_t *p = (volatile uint32_t *)vram;
            p[0] = 0x12345678u;
            __sync_synchronize();
            uint32_t back = 0;
            CK(hipMemcpy(&back, vram, 4, hipMemcpyDeviceToHost));
            printf("host store to device memory: read back through hipMemcpy %08x, through the mapping %08x\n", back, p[0]);
            host_can_write = back == 0x12345678u;
        } else {
            printf("host store to device memory: fault (no host mapping of device memory here)\n");
        }
        signal(SIGSEGV, SIG_DFL); signal(SIGBUS, SIG_DFL);
    }
    const int R = 20000;
    for (int mode = 0; mode < (host_can_write ? 2 : 1); ++mode) {
        // mode 0: doorbell and input in host memory (the kernel polls over PCIe); mode 1: both in device memory (the host pushes)
        volatile uint32_t *door_h = mode ? (volatile uint32_t *)vram : (volatile uint32_t *)hbuf;
        uint32_t *door_d = mode ? (uint32_t *)vram : (uint32_t *)hdev;
        char *in_h = mode ? vram + 4096 : hbuf + 4096;
        const u4 *in_d = (const u4 *)(mode ? vram + 4096 : hdev + 4096);
        u4 *out_d = (u4 *)(hdev + 65536);
        volatile uint32_t *done_h = (volatile uint32_t *)(hbuf + 32768);
        uint32_t *done_d = (uint32_t *)(hdev + 32768);
        *door_h = 0; *done_h = 0;
        __sync_synchronize();
        hipStream_t st;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        echo<<<1, 256, 0, st>>>(door_d, in_d, out_d, done_d, R + 100);
        static char payload[8192];
        for (int i = 0; i < 8192; ++i) payload[i] = (char)i;
        double t = 0;
        for (int r = 1; r <= R + 100; ++r) {
            if (r == 101) t = now();
            memcpy(in_h, payload, 8192);
            __atomic_store_n((uint32_t *)door_h, (uint32_t)r, __ATOMIC_RELEASE);
            while (__atomic_load_n((uint32_t *)done_h, __ATOMIC_ACQUIRE) != (uint32_t)r) __builtin_ia32_pause();
        }
        const double dt = now() - t;
        CK(hipStreamSynchronize(st));
        printf("%s: %.2f us per 8 KiB round trip (ring -> kernel reads 8 KiB -> writes 8 KiB to host -> completion word)\n",
               mode ? "doorbell + payload pushed into device memory" : "doorbell + payload in host memory, polled over PCIe", dt / R * 1e6);
        CK(hipStreamDestroy(st));
    }
    return 0;
}
