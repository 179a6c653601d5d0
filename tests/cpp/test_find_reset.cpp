// find_reset (the Euclid-like descent per binade: csrc/dpx_planner.cpp, round 6) against find_reset_scan (every candidate
// tried with the reference's own arithmetic, dsp.rs:125-130): the same answer — found or not, and where — for
//   * the repository's named ratios from every start in [0, 2 P + 3] and windows that end before, at and after the reset;
//   * random ratios of every exponent a shift / samplerate pair can produce and far beyond (2^-60 .. 2^30), both signs,
//     random starts below and across 2^24, random window lengths;
//   * ratios built so that ratio * n sits exactly on a tie (half an ulp from an integer) or one bit either side of it;
//   * small integers over powers of two, subnormal ratios, ratios whose products overflow, zero, inf, nan.
// Host only, no sanitizer (the scan side is the cost: ~0.1 ms per query): tests/test_host_logic.py runs it.
//   g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math tests/cpp/test_find_reset.cpp doppler_amd/csrc/dpx_planner.cpp doppler_amd/csrc/dpx_simulate.cpp
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>

#include "../../doppler_amd/csrc/dpx_planner.h"

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static float from_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static uint32_t bits_of(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

static long n_checked = 0;
static double t_fast = 0, t_scan = 0;

static bool check(float ratio, uint32_t n_start, uint64_t span, const char *what)
{
    using clk = std::chrono::steady_clock;
    uint32_t a = 0xdeadbeef, b = 0xdeadbeef;
    const clk::time_point t0 = clk::now();
    const bool fa = dpx::find_reset(ratio, n_start, span, &a);
    const clk::time_point t1 = clk::now();
    const bool fb = dpx::find_reset_scan(ratio, n_start, span, &b);
    const clk::time_point t2 = clk::now();
    t_fast += std::chrono::duration<double>(t1 - t0).count();
    t_scan += std::chrono::duration<double>(t2 - t1).count();
    ++n_checked;
    if (fa != fb || (fa && a != b)) {
        fprintf(stderr, "%s: ratio %.9g (bits %08x) n_start %u span %llu: closed form %s %u, scan %s %u\n", what, ratio, bits_of(ratio), n_start,
                (unsigned long long)span, fa ? "found" : "none", a, fb ? "found" : "none", b);
        return false;
    }
    return true;
}

int main(int argc, char **argv)
{
    const int scale = argc > 1 ? atoi(argv[1]) : 1;
    // ---- named ratios, every start
    const struct { float hz; uint32_t rate; } named[] = {{5000.f, 1024000}, {-15000.f, 256000}, {815000.f, 2400000}, {9876.543f, 1024000},
                                                         {-5234.17f, 1024000}, {5001.f, 1024000}, {1.f, 3}, {7.f, 2}, {123456.f, 48000}, {0.f, 1024000}};
    for (auto &nr : named) {
        const float ratio = dpx::ratio_of(nr.hz, nr.rate);
        uint32_t P = 0;
        if (!dpx::find_reset_scan(ratio, 1, 1u << 22, &P)) { fprintf(stderr, "no period for %g / %u\n", nr.hz, nr.rate); return 1; }
        const uint32_t top = P > 20000 ? 20000 : 2 * P + 3;
        for (uint32_t s = 0; s <= top; ++s) {
            const uint32_t start = P > 20000 ? (uint32_t)(rnd() % (2ull * P + 3)) : s;
            if (!check(ratio, start, 3ull * P + 7, "named")) return 1;
            if (!check(ratio, start, 1 + rnd() % (P + 2), "named, short window")) return 1;
        }
    }
    // 3 Hz at 1.024 Msps (P = 1 024 000) and a shift of millihertz (nothing below 2^24): a few starts each
    for (int i = 0; i < 6 * scale; ++i) {
        if (!check(dpx::ratio_of(3.f, 1024000), (uint32_t)(rnd() % 2100000), 1100000, "3 Hz")) return 1;
        if (!check(dpx::ratio_of(0.004f, 1024000), (uint32_t)(rnd() % 1000), (1u << 24) + 300000, "4 mHz")) return 1;
        if (!check(dpx::ratio_of(0.004f, 1024000), (1u << 24) - 1000 + (uint32_t)(rnd() % 2000), 4000000, "4 mHz across 2^24")) return 1;
    }
    // ---- random ratios: exponent uniform in [-60, 30], random mantissa and sign
    for (int i = 0; i < 60000 * scale; ++i) {
        const int e = -60 + (int)(rnd() % 91);
        const uint32_t bits = ((uint32_t)(rnd() & 1) << 31) | ((uint32_t)(e + 127) << 23) | (uint32_t)(rnd() & 0x7fffff);
        const float ratio = from_bits(bits);
        uint32_t start;
        switch (rnd() % 6) {
        case 0: start = (uint32_t)(rnd() % 3); break;
        case 1: start = (uint32_t)(rnd() % 5000); break;
        case 2: start = (uint32_t)(rnd() % (1u << 22)); break;
        case 3: start = (1u << 24) - 2000 + (uint32_t)(rnd() % 4000); break;
        case 4: start = (uint32_t)(rnd() % (1u << 24)); break;
        default: start = (uint32_t)(rnd() % 100000); break;
        }
        const uint64_t span = 1 + rnd() % ((rnd() & 3) ? 300000 : 3000000);
        if (!check(ratio, start, span, "random")) return 1;
    }
    // ---- counters from 2^24 on (the counter itself is rounded before the multiply): small ratios whose first reset lies
    // there, starts anywhere up to 2^32, windows that end before / at / after the reset and across binades of the counter
    for (int i = 0; i < 1500 * scale; ++i) {
        const int e = -36 + (int)(rnd() % 14);                  // 2^-36 .. 2^-23: first resets between ~2^20 and 2^29
        const float ratio = from_bits(((uint32_t)(rnd() & 1) << 31) | ((uint32_t)(e + 127) << 23) | (uint32_t)(rnd() & 0x7fffff));
        uint32_t start;
        switch (rnd() % 5) {
        case 0: start = 1; break;
        case 1: start = (1u << 24) - 3 + (uint32_t)(rnd() % 6); break;
        case 2: start = (1u << (24 + rnd() % 8)) - 5 + (uint32_t)(rnd() % 10); break;
        case 3: start = (uint32_t)(rnd() % 0xffffffffu); break;
        default: start = (1u << 24) + (uint32_t)(rnd() % (1u << 26)); break;
        }
        const uint64_t span = 1 + rnd() % ((i % 7 == 0) ? 90000000ull : 3000000ull);
        if (!check(ratio, start, span, "counters beyond 2^24")) return 1;
    }
    for (uint32_t bits : {0x33000000u, 0x32ffffffu, 0x33000001u, 0x2f800000u, 0x30c90fdbu, 0x4b000000u, 0x3f800000u, 0x3e99999au, 0x7f7fffffu, 0x00000001u})
        for (uint32_t start : {(1u << 24), (1u << 24) + 1, (1u << 25) - 1, (1u << 25), (1u << 25) + 1, (1u << 25) + 2, (1u << 25) + 3, (1u << 31) - 2, (1u << 31) + 128,
                               0xffffff00u, 0xffffffffu, 0xfffffffeu})
            for (uint64_t span : {1ull, 2ull, 3ull, 5ull, 300ull, 100000ull})
                if (!check(from_bits(bits), start, span, "rounded counters, short windows")) return 1;
    // ---- the shifts a receiver really sees: |hz| < 50 kHz with a fractional part, the usual rates
    const uint32_t rates[] = {8000, 48000, 256000, 300000, 1024000, 2400000};
    for (int i = 0; i < 30000 * scale; ++i) {
        const uint32_t rate = rates[rnd() % 6];
        const float hz = (float)(((double)(rnd() >> 11) / 9007199254740992.0 - 0.5) * 1.0e5);
        const float ratio = dpx::ratio_of(hz, rate);
        if (!check(ratio, (uint32_t)(rnd() % 600000), 1 + rnd() % 1200000, "receiver shifts")) return 1;
    }
    // ---- ties: ratio = (I + d) / n with d = +-half an ulp of I (and one f32 either side), n and I random
    for (int i = 0; i < 40000 * scale; ++i) {
        const uint32_t n = 1 + (uint32_t)(rnd() % (1u << (4 + rnd() % 19)));
        const uint32_t I = 1 + (uint32_t)(rnd() % (1u << (1 + rnd() % 22)));
        int k = 31 - __builtin_clz(I);
        const double half_ulp = ldexp(1.0, k - 24);
        const double target = (double)I + ((rnd() & 1) ? half_ulp : -half_ulp) * ((rnd() & 3) ? 1.0 : 0.5);
        float ratio = (float)(target / (double)n);
        const int nudge = (int)(rnd() % 5) - 2;
        ratio = from_bits(bits_of(ratio) + (uint32_t)nudge);
        const uint32_t start = n > 50 ? n - (uint32_t)(rnd() % 50) : 1;
        if (!check(ratio, start, 100 + rnd() % 5000, "tie")) return 1;
        if (!check(ratio, 1, (uint64_t)n + 10, "tie from 1")) return 1;
    }
    // ---- small integers over powers of two, subnormals, overflow, specials
    for (int num = -40; num <= 40; ++num)
        for (int sh = 0; sh <= 30; sh += 1) {
            const float ratio = (float)ldexp((double)num, -sh);
            for (uint32_t start : {0u, 1u, 2u, 3u, 1000u, (1u << 24) - 5})
                if (!check(ratio, start, 70000, "dyadic")) return 1;
        }
    for (uint32_t bits : {0x00000001u, 0x00000003u, 0x007fffffu, 0x00800000u, 0x00ffffffu, 0x7f7fffffu, 0x7f000000u, 0x7e800001u, 0x4b000000u, 0x4affffffu,
                          0x4b000001u, 0x3f800000u, 0x3f7fffffu, 0x3f800001u, 0x3effffffu, 0x00000000u, 0x80000000u, 0x7f800000u, 0xff800000u, 0x7fc00000u,
                          0x33000000u, 0x32ffffffu, 0x33000001u})
        for (uint32_t start : {0u, 1u, 2u, 5u, 77777u, (1u << 23), (1u << 24) - 1, (1u << 24), 0xfffffff0u})
            if (!check(from_bits(bits), start, 200000, "special")) return 1;
    // ---- the counters at which a stretch's correctors change their sincos path: estimate-and-step against the bisection
    long n_bounds = 0;
    for (int i = 0; i < 200000 * scale; ++i) {
        const int e = -70 + (int)(rnd() % 110);
        const float ratio = from_bits(((uint32_t)(rnd() & 1) << 31) | ((uint32_t)(e + 127) << 23) | (uint32_t)(rnd() & 0x7fffff));
        const uint32_t bounds[] = {0x39800000u /* 2^-12 */, 0x42f00000u /* 120 */, 0x4e000000u /* 2^29 */, (uint32_t)(rnd() % 0x7f800000u)};
        for (uint32_t bound : bounds) {
            const uint32_t a = dpx::first_counter_reaching(ratio, bound), b = dpx::first_counter_reaching_bisect(ratio, bound);
            if (a != b) { fprintf(stderr, "first_counter_reaching(%.9g, %08x): %u, bisection %u\n", ratio, bound, a, b); return 1; }
            ++n_bounds;
        }
    }
    for (uint32_t bits : {0u, 0x80000000u, 1u, 0x00800000u, 0x7f7fffffu, 0x7f800000u, 0x7fc00000u, 0x3f800000u})
        for (uint32_t bound : {0x39800000u, 0x42f00000u, 0x4e000000u, 0u, 1u, 0x7f7fffffu})
            if (dpx::first_counter_reaching(from_bits(bits), bound) != dpx::first_counter_reaching_bisect(from_bits(bits), bound)) {
                fprintf(stderr, "first_counter_reaching(bits %08x, %08x) differs\n", bits, bound);
                return 1;
            }
    printf("first_counter_reaching: %ld cases equal to the bisection\n", n_bounds);
    printf("find_reset: %ld queries equal to the scan; closed form %.3f us per query, scan %.1f us per query\n", n_checked, t_fast / n_checked * 1e6,
           t_scan / n_checked * 1e6);
    return 0;
}
