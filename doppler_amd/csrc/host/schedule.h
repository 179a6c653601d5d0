// schedule.h — per-block shift schedule of `doppler track --time`.
//
// Restates the loop of reference src/main.rs:156-184 around a pluggable range-rate source:
//   predict.update(start + dt)  ->  doppler_hz = (rr*1000/c) * f * (-1)   (f64, main.rs:162-163)
//   dt = seconds((sample_count as f32 / samplerate as f32) as i64)          (main.rs:166)
//   shift(intype, doppler_hz as f32 + offset as f32, samplerate)            (main.rs:177)
//   sample_count += count                                                   (main.rs:182)
// i.e. block b is shifted with the range rate of the dt computed one iteration earlier (first
// block: dt = 0), dt is whole seconds truncated through f32, and the offset is added in f32.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>

namespace dpx {

class ReplaySchedule {
public:
    // range_rate_km_s(dt_seconds): range rate at start_time + dt
    ReplaySchedule(std::function<double(int64_t)> range_rate_km_s, uint32_t samplerate, uint32_t frequency_hz,
                   bool has_offset, int32_t offset_hz)
        : rr_(range_rate_km_s), samplerate_(samplerate), frequency_(frequency_hz),
          offset_(has_offset ? offset_hz : 0) {}

    // shift_hz for the block about to be read (advances dt exactly as main.rs:162-166 does)
    float next_block_shift();
    // the block held `count` samples (main.rs:182)
    void block_done(size_t count) { sample_count_ += count; }

    int64_t dt_seconds() const { return dt_; }
    int64_t update_dt_seconds() const { return update_dt_; }   // the dt the last predict.update() was given (one block behind)
    double last_doppler_hz() const { return doppler_hz_; }
    double last_range_rate() const { return last_rr_; }

private:
    std::function<double(int64_t)> rr_;
    uint32_t samplerate_, frequency_;
    int32_t offset_;
    uint64_t sample_count_ = 0;
    int64_t dt_ = 0, update_dt_ = 0;
    double doppler_hz_ = 0, last_rr_ = 0;
};

}  // namespace dpx
