// Host-only fuzz of the planner (csrc/dpx_planner.cpp), built with -fsanitize=address,undefined by `make planner-fuzz`:
// random const- and track-shaped segment lists -> plan_append -> finalize (every KernelChoice, two thirds of the cases with
// the PlanTuning knobs turned) -> simulate; every sample
// must be produced exactly once with the counter of the sequential rule (dsp.rs:125-130), and the sanitizers must stay
// silent (the hint / sentinel scans index vectors by hand).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../doppler_amd/csrc/dpx_planner.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static double uni() { return (double)(rnd() >> 11) / 9007199254740992.0; }

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 400;
    const uint32_t rates[] = {8000, 48000, 256000, 1024000, 300000};
    long checked = 0;
    for (int c = 0; c < cases; ++c) {
        const uint32_t rate = rates[rnd() % 5];
        const int nseg = 1 + (int)(rnd() % (c % 3 == 0 ? 14 : 3));
        std::vector<std::pair<uint64_t, float>> segs;
        for (int i = 0; i < nseg; ++i) {
            float hz;
            switch (rnd() % 4) {
            case 0: hz = (float)((int)(rnd() % 40000) - 20000); break;
            case 1: hz = (float)(uni() * 24000.0 - 12000.0); break;
            case 2: hz = (float)rate / (float)(2 + rnd() % 1000); break;
            default: hz = (float)(uni() * 6.0 - 3.0); break;
            }
            uint64_t cnt;
            switch (rnd() % 5) {
            case 0: cnt = 1 + rnd() % 5000; break;
            case 1: cnt = 2048 * (1 + rnd() % 64); break;          // whole blocks: aligned stretches, no heads or tails
            case 2: cnt = 16384 * (1 + rnd() % 8); break;
            case 3: cnt = 60000 + rnd() % 40000; break;
            default: cnt = 32 * (1 + rnd() % 4000); break;
            }
            segs.push_back({cnt, hz});
        }
        const uint32_t sn0s[] = {0, 1, 2, 1000, 65535, 1u << 20};
        const uint32_t sn0 = sn0s[rnd() % 6];
        // the sequential rule
        std::vector<uint32_t> want;
        {
            uint32_t n = sn0;
            for (auto &sg : segs) {
                const float ratio = dpx::ratio_of(sg.second, rate);
                for (uint64_t k = 0; k < sg.first; ++k) {
                    want.push_back(n);
                    n = dpx::is_reset(ratio, n) ? 1u : n + 1u;
                }
            }
        }
        for (int choice = 0; choice < 4; ++choice) {
            dpx::PlanResult plan;
            uint32_t sn = sn0;
            for (auto &sg : segs) dpx::plan_append(plan, dpx::ratio_of(sg.second, rate), sg.first, sn, 0);
            const uint32_t tile = (c & 1) ? 1024 : 512;
            // every third case with the measurement knobs turned: they change launch shapes, never counters
            dpx::PlanTuning tn;
            if (c % 3 == 1) {
                tn.rows_compute = (c & 4) ? 1u : 0xffffffffu;
                tn.rows_r = (c & 8) ? 8u : 4u;
                tn.walk_waves = (c & 16) ? 4u : 5u;                    // the planner's own spans: several windows per workgroup where the wavefronts divide
            } else if (c % 3 == 2) {
                tn.walk_span = (c & 16) ? 0u : 2u + (uint32_t)(c % 37);   // explicit span heights 2..38
                tn.walk_flags = ((c & 32) ? 1u : 0u) | ((c & 64) ? (uint32_t)(8 + c % 300) << 8 : 0u);   // descriptors from memory; row-length target
                tn.rows_compute = 3000u;
                tn.walk_waves = (c & 4) ? 2u : 8u;
            }
            dpx::finalize(plan, tile, choice, tn);
            if (plan.error) { fprintf(stderr, "case %d: %s\n", c, plan.error); return 1; }
            std::vector<uint32_t> got(plan.n_samples + 1, 0);
            std::vector<uint8_t> writes(plan.n_samples + 1, 0);
            // every format pair launches its own grid from the same plan (windows shared by two workgroups for f32 output,
            // one-matrix spans cut again, 8 wavefronts for f32 -> i16): dpx_planner.cpp, span_launch_shape
            for (int pair = 0; pair < 4; ++pair) {
                if (pair != 0 && plan.walk.empty()) break;              // only span launches depend on the pair
                std::fill(got.begin(), got.end(), 0u);
                std::fill(writes.begin(), writes.end(), (uint8_t)0);
                dpx::simulate(plan, got.data(), writes.data(), pair >> 1, pair & 1);
                for (uint64_t g = 0; g < plan.n_samples; ++g) {
                    if (writes[g] != 1 || got[g] != want[g]) {
                        fprintf(stderr, "case %d choice %d pair %d: sample %llu written %u times, counter %u, want %u\n", c, choice, pair,
                                (unsigned long long)g, writes[g], got[g], want[g]);
                        return 1;
                    }
                }
                checked += (long)plan.n_samples;
            }
        }
    }
    {
        // the period cache stays bounded however many distinct ratios a long-lived context sees (live track mode: one per block),
        // and an emptied cache gives the same periods again
        dpx::PeriodCache cache;
        uint32_t first = 0;
        for (uint32_t k = 0; k < 3 * (uint32_t)dpx::PeriodCache::kMaxEntries; ++k) {
            const float ratio = dpx::ratio_of(1000.0f + 0.25f * (float)k, 1024000);
            const uint32_t p = cache.period(ratio, 4096);
            if (k == 0) first = p;
            if (cache.first_reset.size() > dpx::PeriodCache::kMaxEntries) { fprintf(stderr, "period cache grew to %zu entries\n", cache.first_reset.size()); return 1; }
        }
        if (cache.period(dpx::ratio_of(1000.0f, 1024000), 4096) != first) { fprintf(stderr, "period changed after the cache was emptied\n"); return 1; }
        std::vector<float> ratios;
        std::vector<uint64_t> counts;
        for (uint32_t k = 0; k < 5000; ++k) { ratios.push_back(dpx::ratio_of(-7000.0f + 1.5f * (float)k, 1024000)); counts.push_back(1024000); }
        cache.prefetch(ratios.data(), counts.data(), ratios.size());
        dpx::PeriodCache serial;
        for (uint32_t k = 0; k < 5000; k += 97)
            if (cache.period(ratios[k], counts[k] + 1) != serial.period(ratios[k], counts[k] + 1)) { fprintf(stderr, "prefetched period differs\n"); return 1; }
    }
    printf("planner fuzz: %d cases x 4 kernel choices, %ld samples checked, ok\n", cases, checked);
    return 0;
}
