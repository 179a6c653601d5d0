// dpx_planner.cpp — see dpx_planner.h.  Host code, plain C++ (no HIP), compiled
// with -ffp-contract=off: the f32 products below must round exactly like the
// reference's (src/dsp.rs:121,125).
#include "dpx_planner.h"

#include <math.h>

#include <algorithm>

namespace dpx {

float ratio_of(float shift_hz, uint32_t samplerate)
{
    volatile float r = shift_hz / (float)samplerate;   // one IEEE f32 division
    return r;
}

// fract(p) == 0.0 for p = fl32(ratio * fl32(n)):  true iff p is a finite integer.
// (f32::fract = p - p.trunc(); inf gives NaN, NaN stays NaN: neither equals 0.)
static inline bool product_is_integer(float p)
{
    const float ap = fabsf(p);
    if (!(ap < 8388608.0f)) return ap <= 3.4028234663852886e38f;   // >= 2^23: integer unless inf/nan
    return (float)(int32_t)p == p;
}

bool is_reset(float ratio, uint32_t n)
{
    volatile float p = ratio * (float)n;
    return product_is_integer(p);
}

bool find_reset(float ratio, uint32_t n_start, uint64_t max_scan, uint32_t *n_reset)
{
    const uint64_t to_wrap = (1ULL << 32) - (uint64_t)n_start;
    const uint64_t span = std::min(max_scan, to_wrap);
    uint64_t n = n_start;
    const uint64_t end = (uint64_t)n_start + span;
    for (; n < end; ++n) {
        const float p = ratio * (float)(uint32_t)n;
        if (product_is_integer(p)) {
            *n_reset = (uint32_t)n;
            return true;
        }
    }
    return false;
}

static uint32_t lut_len_for(uint32_t period, uint64_t count, int variant)
{
    // the table length must be >= 4 so that 4 consecutive entries wrap at most once
    const uint32_t reps = period >= 4 ? 1u : (4u + period - 1u) / period;
    const uint64_t len = (uint64_t)period * reps;
    if (period < 4) return (uint32_t)len;           // only the table path handles tiny periods
    if (variant == 1) return 0;
    if (len > kLutMaxEntries) return 0;
    if (variant == 2) return (uint32_t)len;
    return count >= 2 * len ? (uint32_t)len : 0;
}

static void emit(PlanResult &plan, uint64_t first, uint64_t count, float ratio, uint32_t n_start,
                 uint32_t period, uint32_t lut_len)
{
    DevSeg s;
    s.first = first;
    s.count = count;
    s.ratio = ratio;
    s.n_start = n_start;
    s.period = period;
    s.lut_len = lut_len;
    plan.segs.push_back(s);
    plan.max_lut_len = std::max(plan.max_lut_len, lut_len);
}

void plan_append(PlanResult &plan, float ratio, uint64_t count, uint32_t &samplenum, int variant)
{
    uint64_t pos = plan.n_samples;
    uint64_t remaining = count;
    uint32_t n = samplenum;
    while (remaining > 0) {
        const uint64_t to_wrap = (1ULL << 32) - (uint64_t)n;
        const uint64_t span = std::min(remaining, to_wrap);
        uint32_t n1 = 0;
        if (!find_reset(ratio, n, span, &n1)) {
            // no reset among the next `span` counter values: n = n_start + j
            emit(plan, pos, span, ratio, n, 0, 0);
            pos += span;
            remaining -= span;
            n = (uint32_t)((uint64_t)n + span);     // u32 `+= 1` wraps to 0 after 2^32-1
            continue;
        }
        // the sample that uses n1 resets the counter to 1
        uint32_t p1 = 0;
        const bool steady = n >= 1 && find_reset(ratio, 1, n1, &p1) && p1 == n1;
        if (steady) {
            // no reset in [1, n1): the counter cycles 1..P with P = n1 from here on
            const uint32_t P = n1;
            emit(plan, pos, remaining, ratio, n, P, lut_len_for(P, remaining, variant));
            n = (uint32_t)(((uint64_t)(n - 1u) + remaining) % P) + 1u;
            pos += remaining;
            remaining = 0;
        } else {
            // lead-in (n == 0, or a counter carried over from another ratio): linear up to n1
            const uint64_t len = (uint64_t)n1 - n + 1;
            emit(plan, pos, len, ratio, n, 0, 0);
            pos += len;
            remaining -= len;
            n = 1;
        }
    }
    plan.n_samples = pos;
    samplenum = n;
    plan.final_samplenum = n;
}

}  // namespace dpx
