/* block_async_bench — the reference's 8 KiB block through the C ABI: dpx_shift_block (one call, one wait) against
 * dpx_shift_block_async / dpx_wait with 1, 2 and 4 blocks in flight, and dpx_shift_blocks with 64 blocks per call; each with the
 * resident block kernel (a doorbell per block: the default) and without it (a launch per block: round 3's path).
 *   gcc -O2 -Iinclude tools/block_async_bench.c -Ldoppler_amd/lib -ldoppler_hip -Wl,-rpath,$PWD/doppler_amd/lib -o tools/bin/block_async_bench */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "doppler_hip.h"
#include "doppler_hip_debug.h"

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

int main(void)
{
    dpx_ctx *ctx;
    if (dpx_ctx_create(0, &ctx) != DPX_OK) { fprintf(stderr, "%s\n", dpx_last_error()); return 1; }
    enum { NB = 20000, BLK = 8192 };
    static int16_t in[BLK / 2];
    static char out[4][BLK];
    for (int i = 0; i < BLK / 2; ++i) in[i] = (int16_t)((i * 7919) % 20000 - 10000);
    uint32_t sn = 0;
    size_t n;
    double t, dt;
    for (int resident = 1; resident >= 0; --resident) {
    dpx_set_resident(ctx, resident);
    printf("== %s\n", resident ? "resident block kernel (doorbell per block)" : "a launch per block (DPX_RESIDENT=0)");
    sn = 0;
    for (int i = 0; i < 200; ++i) dpx_shift_block(ctx, in, BLK, DPX_FMT_I16, out[0], BLK, DPX_FMT_I16, &sn, 5000.0f, 1024000, &n);
    t = now();
    for (int i = 0; i < NB; ++i) dpx_shift_block(ctx, in, BLK, DPX_FMT_I16, out[0], BLK, DPX_FMT_I16, &sn, 5000.0f, 1024000, &n);
    dt = now() - t;
    printf("dpx_shift_block, 8 KiB per call:           %6.2f us per block = %7.1f Msamples/s\n", dt / NB * 1e6, NB * 2048.0 / dt / 1e6);
    const uint32_t sn_sync = sn;
    for (int depth = 1; depth <= 4; depth *= 2) {
        sn = 0;
        for (int i = 0; i < 200; ++i) dpx_shift_block(ctx, in, BLK, DPX_FMT_I16, out[0], BLK, DPX_FMT_I16, &sn, 5000.0f, 1024000, &n);
        dpx_ticket tk[4];
        int head = 0, inflight = 0;
        t = now();
        for (int i = 0; i < NB; ++i) {
            if (inflight == depth) {
                if (dpx_wait(ctx, tk[head % 4], out[head % 4], BLK, &n) != DPX_OK) { fprintf(stderr, "%s\n", dpx_last_error()); return 1; }
                ++head;
                --inflight;
            }
            if (dpx_shift_block_async(ctx, in, BLK, DPX_FMT_I16, DPX_FMT_I16, &sn, 5000.0f, 1024000, &tk[(head + inflight) % 4]) != DPX_OK) {
                fprintf(stderr, "%s\n", dpx_last_error());
                return 1;
            }
            ++inflight;
        }
        while (inflight) { dpx_wait(ctx, tk[head % 4], out[head % 4], BLK, &n); ++head; --inflight; }
        dt = now() - t;
        printf("dpx_shift_block_async, %d blocks in flight: %6.2f us per block = %7.1f Msamples/s  (counter %s)\n", depth, dt / NB * 1e6,
               NB * 2048.0 / dt / 1e6, sn == sn_sync ? "as the synchronous loop's" : "DIFFERS");
    }
    uint64_t launches = 0, blocks = 0;
    dpx_resident_stats(ctx, &launches, &blocks);
    printf("resident kernel so far: %llu launches, %llu blocks through doorbells\n", (unsigned long long)launches, (unsigned long long)blocks);
    }
    {
        enum { K = 64 };
        static int16_t big[K * BLK / 2];
        static char bout[K * BLK];
        float hz[K];
        for (int i = 0; i < K; ++i) { hz[i] = 5000.0f; memcpy(big + i * (BLK / 2), in, BLK); }
        sn = 0;
        for (int i = 0; i < 20; ++i) dpx_shift_blocks(ctx, big, K * BLK, DPX_FMT_I16, bout, K * BLK, DPX_FMT_I16, &sn, hz, K, 1024000, &n);
        t = now();
        for (int i = 0; i < NB / K; ++i) dpx_shift_blocks(ctx, big, K * BLK, DPX_FMT_I16, bout, K * BLK, DPX_FMT_I16, &sn, hz, K, 1024000, &n);
        dt = now() - t;
        printf("dpx_shift_blocks, 64 blocks per call:      %6.2f us per block = %7.1f Msamples/s\n", dt / (NB / K * K) * 1e6, (NB / K * K) * 2048.0 / dt / 1e6);
    }
    dpx_ctx_destroy(ctx);
    return 0;
}
