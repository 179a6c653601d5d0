// dpx_planner.h — host-side closed form of the reference's sample counter.
//
// src/dsp.rs:125-130 advances `samplenum` sequentially: after the sample that
// used n, n <- 1 if fract(fl32(ratio * fl32(n))) == 0.0 else n + 1 (u32, wrapping).
// That recurrence is the only thing that keeps the stream from being processed
// in parallel.  The planner breaks it: it scans for the reset points once (pure
// f32/integer host work, exact) and describes the stream as a short list of
// DevSeg stretches in which n is a closed form of the sample index.
#pragma once
#include <stdint.h>

#include <vector>

#include "dpx_types.h"

namespace dpx {

// fl32(shift_hz / fl32(samplerate)) — dsp.rs:121, the one rounding of the ratio
float ratio_of(float shift_hz, uint32_t samplerate);

// the reset predicate of dsp.rs:125 on counter value n
bool is_reset(float ratio, uint32_t n);

// first n in [n_start, n_start + max_scan) (not past 2^32-1) with is_reset; false if none
bool find_reset(float ratio, uint32_t n_start, uint64_t max_scan, uint32_t *n_reset);

struct PlanResult {
    std::vector<DevSeg> segs;   // consecutive, covering [0, n_samples)
    uint64_t n_samples = 0;
    uint32_t final_samplenum = 0;
    uint32_t max_lut_len = 0;
};

// variant: 0 auto, 1 never use the LDS table (except period < 4), 2 table whenever it fits
void plan_append(PlanResult &plan, float ratio, uint64_t count, uint32_t &samplenum, int variant);

}  // namespace dpx
