// dispatchbench.hip — how fast can MI355X launch workgroups?  (development tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int B> __global__ __launch_bounds__(B) void empty_k(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
template <int B> __global__ __launch_bounds__(B) void touch_k(const int *p, int *q)
{   // one 4-byte load + store per lane (L2-resident 1 MiB window): launch + minimal memory work
    const unsigned i = (blockIdx.x * B + threadIdx.x) & 0x3ffff;
    q[i] = p[i] + 1;
}
template <typename F> double timeit(F f, int iters)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}
int main()
{
    int *p, *q; CK(hipMalloc(&p, 4 << 20)); CK(hipMalloc(&q, 4 << 20)); CK(hipMemset(p, 0, 4 << 20));
    const unsigned grids[] = {65536, 262144, 524288, 1048576, 2097152};
    for (unsigned g : grids) {
        double t64 = timeit([&] { empty_k<64><<<g, 64>>>(nullptr); }, 10);
        double t128 = timeit([&] { empty_k<128><<<g, 128>>>(nullptr); }, 10);
        double t256 = timeit([&] { empty_k<256><<<g, 256>>>(nullptr); }, 10);
        double m64 = timeit([&] { touch_k<64><<<g, 64>>>(p, q); }, 10);
        printf("grid %8u: empty 64/128/256 lanes: %.4f / %.4f / %.4f ms  (%.2f / %.2f / %.2f WG per ns);  touch<64>: %.4f ms\n",
               g, t64, t128, t256, g / t64 / 1e6, g / t128 / 1e6, g / t256 / 1e6, m64);
    }
    return 0;
}
