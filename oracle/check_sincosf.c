/*
 * TEST INFRASTRUCTURE — pins oracle/sincosf_glibc.c against this host's libm.
 *
 * Compares orc_sincosf_glibc235 (both contraction variants) with libm's
 * sincosf bit-for-bit over a range of float bit patterns, multi-threaded.
 *   check_sincosf                exhaustive: all 2^32 patterns
 *   check_sincosf START COUNT    patterns START .. START+COUNT-1
 *   check_sincosf --stride K     every K-th pattern (quick mode for pytest)
 * Also checks libm cexpf(0+i*theta) == orc_cexpf_imag_glibc235 on the same
 * patterns when --cexp is given.  NaN results compare equal to any NaN.
 * --expf additionally checks orc_expf_glibc235 against libm expf on the same patterns.
 * Exit status 0 iff the variant reported by orc_detect_libm_variant() has
 * zero mismatches.
 */
#define _GNU_SOURCE /* sincosf */
#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sincosf_glibc.h"

typedef struct {
    uint64_t start, count, stride;
    int do_cexp, do_expf;
    uint64_t mism_expf[2];
    uint64_t mism[2];
    uint64_t mism_cexp[2];
    uint32_t first_bad[2];
} job_t;

static inline uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline int same(float a, float b) { return bits(a) == bits(b) || (a != a && b != b); }

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    for (uint64_t k = 0; k < j->count; k += j->stride) {
        uint32_t u = (uint32_t)(j->start + k);
        float y, ls, lc;
        memcpy(&y, &u, 4);
        if (j->do_expf) {
            const float le = expf(y);
            for (int v = 0; v < 2; ++v)
                if (!same(orc_expf_glibc235(y, v), le)) j->mism_expf[v]++;
        }
        sincosf(y, &ls, &lc);
        float complex z = 0;
        if (j->do_cexp) z = cexpf(0.0f + y * I);
        for (int v = 0; v < 2; ++v) {
            float s, c;
            orc_sincosf_glibc235(y, &s, &c, v);
            if (!same(s, ls) || !same(c, lc)) {
                if (!j->mism[v]) j->first_bad[v] = u;
                j->mism[v]++;
            }
            if (j->do_cexp) {
                float re, im;
                orc_cexpf_imag_glibc235(y, &re, &im, v);
                if (!same(re, crealf(z)) || !same(im, cimagf(z))) j->mism_cexp[v]++;
            }
        }
    }
    return NULL;
}

int main(int argc, char **argv)
{
    uint64_t start = 0, count = 1ULL << 32, stride = 1;
    int do_cexp = 0, do_expf = 0, nthreads = 8, pos = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--stride") && i + 1 < argc) stride = strtoull(argv[++i], 0, 0);
        else if (!strcmp(argv[i], "--threads") && i + 1 < argc) nthreads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--cexp")) do_cexp = 1;
        else if (!strcmp(argv[i], "--expf")) do_expf = 1;
        else if (pos == 0) { start = strtoull(argv[i], 0, 0); pos++; }
        else if (pos == 1) { count = strtoull(argv[i], 0, 0); pos++; }
    }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    job_t jobs[256];
    uint64_t per = (count + nthreads - 1) / nthreads;
    per = (per + stride - 1) / stride * stride;
    for (int t = 0; t < nthreads; ++t) {
        memset(&jobs[t], 0, sizeof(job_t));
        jobs[t].start = start + per * t;
        jobs[t].count = (per * t >= count) ? 0 : (per * (t + 1) > count ? count - per * t : per);
        jobs[t].stride = stride;
        jobs[t].do_cexp = do_cexp;
        jobs[t].do_expf = do_expf;
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    uint64_t mism[2] = {0, 0}, mc[2] = {0, 0}, me[2] = {0, 0};
    uint32_t fb[2] = {0, 0};
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        for (int v = 0; v < 2; ++v) {
            if (jobs[t].mism[v] && !mism[v]) fb[v] = jobs[t].first_bad[v];
            mism[v] += jobs[t].mism[v];
            mc[v] += jobs[t].mism_cexp[v];
            me[v] += jobs[t].mism_expf[v];
        }
    }
    int variant = orc_detect_libm_variant();
    printf("{\"start\": %llu, \"count\": %llu, \"stride\": %llu, \"libm_variant\": %d, "
           "\"mismatch_sse2\": %llu, \"mismatch_fma\": %llu, \"first_bad_sse2\": \"0x%08x\", "
           "\"first_bad_fma\": \"0x%08x\", \"cexp_checked\": %d, \"cexp_mismatch_sse2\": %llu, "
           "\"cexp_mismatch_fma\": %llu, \"expf_checked\": %d, \"expf_mismatch_sse2\": %llu, \"expf_mismatch_fma\": %llu}\n",
           (unsigned long long)start, (unsigned long long)count, (unsigned long long)stride, variant,
           (unsigned long long)mism[0], (unsigned long long)mism[1], fb[0], fb[1], do_cexp,
           (unsigned long long)mc[0], (unsigned long long)mc[1], do_expf, (unsigned long long)me[0], (unsigned long long)me[1]);
    if (variant < 0) return 2;
    return (mism[variant] == 0 && mc[variant] == 0 && me[variant] == 0) ? 0 : 1;
}
