/*
 * doppler_hip_host.h — host-only entry points of libdoppler_hip.so (no device needed, no context).
 *
 * The callers either side of the hot path: the counter rule's closed form as plain
 * functions, the per-block shift schedule of `doppler track --time` (reference src/main.rs:156-184) and the orbit
 * provider that stands in for libgpredict (reference src/main.rs:141-149,162-173; NOT part of the reference tree:
 * ORBIT PARITY UNPINNED).  Nothing here is needed to bind the hot path itself: see doppler_hip.h.
 */
#ifndef DOPPLER_HIP_HOST_H
#define DOPPLER_HIP_HOST_H

#include "doppler_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------- host-side counter algebra
 * Closed form of the counter rule dsp.rs:125-130 (pure host integer/f32 code). */

/* first n in [n_start, n_start + max_scan) with fract(fl32(ratio*fl32(n))) == 0; *found = 0 if none in range.
 * ratio = shift_hz/(f32)samplerate.  Since round 6 computed, not searched: an Euclid-like descent per binade of the product
 * (and, from 2^24 on where the counter is rounded before the multiply, per binade of the counter): csrc/dpx_planner.cpp,
 * ~0.4 us whatever the period.  dpx_find_reset_scan tries every candidate — the definition, rounds 1-5's implementation,
 * kept as the checker; the library itself no longer calls it. */
int dpx_find_reset(float shift_hz, uint32_t samplerate, uint32_t n_start, uint64_t max_scan,
                   uint32_t *n_reset, int *found);
int dpx_find_reset_scan(float shift_hz, uint32_t samplerate, uint32_t n_start, uint64_t max_scan,
                        uint32_t *n_reset, int *found);
/* value of the counter after `k` samples starting from samplenum0 (constant shift) */
int dpx_samplenum_after(float shift_hz, uint32_t samplerate, uint32_t samplenum0, uint64_t k,
                        uint32_t *samplenum);
/* ... after a whole list of constant-shift segments (what a rank of a time-chunk sharded run needs for the chunks before
 * its own: the seed of its first sample, reference src/main.rs:60 carried by closed form instead of by running the stream) */
int dpx_samplenum_after_segments(const dpx_segment *segs, size_t n_segs, uint32_t samplerate, uint32_t samplenum0,
                                 uint32_t *samplenum);

/* ------------------------------------------------ track mode, host side (N2)
 * Host-only.  The per-block shift schedule of `doppler track --time` (reference
 * src/main.rs:156-184: one-block lag, whole seconds truncated through f32, f32 offset add) for a
 * stream of in_bytes, with the range rate supplied per whole second (entry t = range rate at
 * start_time + t s; the last entry is held).  Writes one shift per loop iteration of the reference
 * (floor(in_bytes / 8192) + 1, the last one belonging to the short or empty final block). */
int dpx_track_schedule(const double *range_rate_km_s, size_t n_table, uint32_t samplerate,
                       uint32_t frequency_hz, int32_t offset_hz, int has_offset, int in_fmt,
                       uint64_t in_bytes, float *shift_hz, size_t cap, size_t *n_blocks);

/* Host-only.  NORAD SGP4 (near-earth) + geodetic observer: what the reference reads from
 * predict.sat after predict.update(time) (src/main.rs:162-173).  out[4] = azimuth deg,
 * elevation deg, range km, range rate km/s.  libgpredict is not part of the reference tree:
 * ORBIT PARITY UNPINNED. */
int dpx_orbit_observe(const char *tle_line1, const char *tle_line2, double lat_deg, double lon_deg,
                      double alt_m, double unix_time_s, double out[4]);

/* Host-only.  SGP4 state vector `tsince_min` minutes after the element-set epoch:
 * out[6] = x, y, z (km), xdot, ydot, zdot (km/s), true-equator mean-equinox frame. */
int dpx_orbit_propagate(const char *tle_line1, const char *tle_line2, double tsince_min, double out[6]);

#ifdef __cplusplus
}
#endif
#endif
