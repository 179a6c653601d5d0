// dpx_types.h — structures shared by the host planner and the gfx950 kernels.
#pragma once
#include <stdint.h>

namespace dpx {

// One stretch of the stream over which the counter of dsp.rs:125-130 follows a
// single closed form.  For sample j = g - first of the stretch:
//   period == 0 ("linear"):   n = n_start + j            (no reset inside)
//   period  > 0 ("periodic"): n = ((n_start - 1 + j) mod period) + 1
// lut_len > 0 asks the kernel to keep the correctors of one period in LDS
// (lut_len is a multiple of period, >= 4); 0 means evaluate sincos per sample.
struct DevSeg {
    uint64_t first;     // global sample index of the first sample
    uint64_t count;     // samples in this stretch (> 0)
    float ratio;        // fl32(shift_hz / fl32(samplerate)), dsp.rs:121
    uint32_t n_start;   // counter value used by the first sample
    uint32_t period;
    uint32_t lut_len;
};
static_assert(sizeof(DevSeg) == 32, "DevSeg is read with scalar loads; keep it 32 bytes");

struct LaunchGeom {
    int grid;           // workgroups
    int unroll;         // 16-byte vectors in flight per lane (template instance)
    uint32_t lds_bytes; // dynamic LDS for the corrector table
};

constexpr int kBlock = 256;          // 4 wavefronts of 64
constexpr int kSamplesPerLane = 4;   // one 16-byte i16 vector = 4 IQ samples
constexpr uint32_t kLutMaxEntries = 4096;   // 32 KiB of LDS per workgroup

// launch wrappers implemented in dpx_kernels.hip (all asynchronous on `stream`)
int launch_shift(const void *d_in, int in_fmt, void *d_out, int out_fmt, const DevSeg *d_segs,
                 uint32_t n_segs, uint64_t n_samples, bool fma, const LaunchGeom &g, void *stream);
int launch_copy(const void *d_in, void *d_out, uint64_t n_bytes, int grid, void *stream);
int launch_unpack_i16(const void *d_in, void *d_out, uint64_t n_samples, void *stream);
int launch_pack_i16(const void *d_in, void *d_out, uint64_t n_samples, void *stream);
int launch_ccexpf_imag(void *d_z, uint64_t n, bool fma, void *stream);

}  // namespace dpx
