// schedule.cpp — see schedule.h.  Compiled with -ffp-contract=off.
#include "schedule.h"

namespace dpx {

float ReplaySchedule::next_block_shift()
{
    const double SPEED_OF_LIGHT_M_S = 299792458.;        // main.rs:48
    update_dt_ = dt_;
    last_rr_ = rr_(dt_);
    doppler_hz_ = (last_rr_ * 1000.0 / SPEED_OF_LIGHT_M_S) * (double)frequency_ * (-1.0);
    volatile float q = (float)sample_count_ / (float)samplerate_;
    // Rust `as i64`: truncate toward zero, saturate, NaN -> 0
    if (q != q) dt_ = 0;
    else if (q >= 9223372036854775807.0f) dt_ = INT64_MAX;
    else if (q <= -9223372036854775808.0f) dt_ = INT64_MIN;
    else dt_ = (int64_t)q;
    volatile float a = (float)doppler_hz_, b = (float)offset_;
    volatile float s = a + b;
    return s;
}

}  // namespace dpx
