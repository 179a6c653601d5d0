// dpx_planner.h — host-side closed form of the reference's sample counter.
//
// src/dsp.rs:125-130 advances `samplenum` sequentially: after the sample that
// used n, n <- 1 if fract(fl32(ratio * fl32(n))) == 0.0 else n + 1 (u32, wrapping).
// That recurrence is the only thing that keeps the stream from being processed
// in parallel.  The planner breaks it: it scans for the reset points once (pure
// f32/integer host work, exact) and describes the stream as a short list of
// DevSeg stretches in which n is a closed form of the sample index.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <unordered_map>
#include <vector>

#include "dpx_types.h"

namespace dpx {

// fl32(shift_hz / fl32(samplerate)) — dsp.rs:121, the one rounding of the ratio
float ratio_of(float shift_hz, uint32_t samplerate);

// the reset predicate of dsp.rs:125 on counter value n
bool is_reset(float ratio, uint32_t n);

// first n in [n_start, n_start + max_scan) (not past 2^32-1) with is_reset; false if none
bool find_reset(float ratio, uint32_t n_start, uint64_t max_scan, uint32_t *n_reset);
// the same by trying every candidate (AVX2 where the host has it): the definition find_reset is held against
// (tests/cpp/test_find_reset.cpp); the library itself no longer calls it
bool find_reset_scan(float ratio, uint32_t n_start, uint64_t max_scan, uint32_t *n_reset);

struct TableBuild {     // one corrector table to fill at plan time
    uint64_t off;       // pool entry index
    uint32_t period, n_first, n_entries;
    float ratio;
};

struct Launch {
    int kind;           // 0 = rows kernel, 1 = tile kernel, 2 = span kernel
    RowsArgs rows;
    TileArgs tiles;
    WalkArgs walk;
};

// finalize(): which kernels a plan may use
enum KernelChoice {
    kChooseAuto = 0,      // rows kernel for up to 8 long stretches, else the span kernel, else tiles
    kChooseTileOnly = 1,  // tile kernel only (measurement A/B)
    kChooseWalk = 2,      // span kernel wherever a stretch qualifies (measurement A/B)
    kChooseRows = 3,      // rows kernel or tiles, never the span kernel (measurement A/B)
};

// Measurement knobs of finalize() (dpx_set_option; 0 = the planner's own choice everywhere).  None of them changes
// a result, only which kernel shape produces it.
struct PlanTuning {
    uint32_t rows_mult = 0;      // rows kernel: row length = rows_mult * lcm(period, 4)
    uint32_t rows_maxl = 0;      // rows kernel: longest row considered
    uint32_t rows_r = 0;         // rows kernel: rows per wavefront (2 or 4)
    uint32_t walk_waves = 0;     // span kernel: wavefronts per workgroup (2, 4, 5 or 8)
    uint32_t rows_compute = 0;   // rows kernel: periods from this many samples on are evaluated in the kernel (0 = the
                                 // planner's default, 0xffffffff = never, 1 = always)
    uint64_t walk_tilemin = 0;   // span plans: an uncovered gap at least this long gets its own tile launch
    uint32_t walk_span = 0;      // span kernel: most rows per span (0 = the planner's default: kSpanRows, whole matrices up to
                                 // kSpanWhole rows, several windows per workgroup for spans of up to 4 rows); >= 2: spans of at
                                 // most that many rows, one window per workgroup, two rows per wavefront per turn
    uint32_t walk_flags = 0;     // bit 0: a one-matrix span launch reads descriptors like any other (measurement of what the
                                 // descriptor load costs); bits 8..: row-length target in KiSamples (measurement)
    uint32_t sub_lg = 0;         // span launches are dealt out in pieces of about 2^sub_lg samples (0 = kSubLaunchLg,
                                 // >= 48 = never cut)
    bool operator==(const PlanTuning &o) const
    {
        return sub_lg == o.sub_lg && rows_mult == o.rows_mult && rows_maxl == o.rows_maxl && rows_r == o.rows_r && walk_waves == o.walk_waves &&
               rows_compute == o.rows_compute && walk_tilemin == o.walk_tilemin && walk_span == o.walk_span &&
               walk_flags == o.walk_flags;
    }
};

struct PlanResult {
    std::vector<DevSeg> segs;   // consecutive, covering [0, n_samples)
    uint64_t n_samples = 0;
    uint32_t final_samplenum = 0;
    // filled by finalize():
    uint64_t lut_entries = 0;        // size of the corrector-table pool, in (cos, sin) entries
    uint32_t tile = 0;               // tile-kernel samples per workgroup the tables were laid out for
    bool tile_tables = false;        // some stretch is served from a tile-kernel table
    std::vector<uint32_t> hint;      // stretch index per 2^kHintShift samples
    std::vector<TableBuild> tables;
    std::vector<Launch> launches;
    // f32 -> i16 and i16 -> f32: ONE tile launch over the whole stream, filled when `launches` holds a span launch of many matrices
    // (track mode) and the kernel choice is the planner's own — see launches_for()
    std::vector<Launch> whole_tiles;
    // span-kernel launch (at most one per plan), each list closed by a sentinel
    std::vector<WalkSeg> walk;
    std::vector<uint32_t> walk_hint;  // WalkSeg index per 2^kWalkHintShift workgroups
    std::vector<LeftRange> left;
    std::vector<uint32_t> left_hint;  // LeftRange index per 2^kLeftHintShift leftover workgroups
    const char *error = nullptr;     // set by finalize() when the plan cannot be laid out
};

// period (first reset from counter 1) per ratio bit pattern; a context or stream keeps one so that a ratio is scanned once
struct PeriodCache {
    struct Entry { uint32_t period = 0; uint64_t scanned_to = 1; };
    std::unordered_map<uint32_t, Entry> first_reset;
    static constexpr size_t kMaxEntries = 16384;      // bounded: emptied when full (a live track stream brings a new ratio per block)
    void make_room(size_t incoming);
    uint32_t period(float ratio, uint64_t limit);     // first reset in [1, limit), or 0
    void prefetch(const float *ratios, const uint64_t *counts, size_t n);   // scan many ratios on several threads
};

// first counter whose |theta| (as the kernels compute it: two separately rounded f32 products) reaches the f32 whose bits are
// `bound`, 0xffffffff if none: where a stretch's correctors change their sincos path (DevSeg.n_plain / n_large / n_huge)
uint32_t first_counter_reaching(float ratio, uint32_t bound);
uint32_t first_counter_reaching_bisect(float ratio, uint32_t bound);

// variant: 0 auto, 1 sincos per sample wherever the period allows (>= 4), 2 tables whenever they fit
void plan_append(PlanResult &plan, float ratio, uint64_t count, uint32_t &samplenum, int variant, PeriodCache *cache = nullptr);

// after the last plan_append: choose the kernel for every stretch, lay out the
// corrector tables, build the hint table and the launch list.
void finalize(PlanResult &plan, uint32_t tile, int choice /* KernelChoice */, const PlanTuning &tuning = PlanTuning());

// wavefronts per workgroup the span kernel is built for (PlanTuning::walk_waves, SpanLaunch::waves)
bool walk_waves_ok(uint32_t waves, bool uni);

// The launches of a finalized plan for one format pair (DPX_FMT_*: 0 = i16, 1 = f32): `plan.launches`, except that a
// plan of many matrices (one span launch driven by descriptors: track mode) runs the two MIXED pairs through the tile
// kernel alone, every corrector evaluated per sample.  With 12 bytes per sample that arithmetic hides behind the memory
// side, and the tile kernel's one-shot kilobyte tiles in address order stream better than spans whose rows are 8 bytes
// per sample on one side and 4 on the other: 300 s replays, same process, tile against span kernel — f32 -> i16 78.8 / 76.6 %
// (round 4: 79.6-80.4 against 75.8-78.1 on four boxes of five), i16 -> f32 77.6 / 74.1 % since round 5's cheaper sincos
// (round 4: a tie, 72-75); f32 -> f32 stays on the span kernel (80.1 against 78.4), i16 -> i16 by far (80.0 against 70.7)
// (profiles/raw/r05_ab_route_poly9.log).  One-matrix launches of the mixed pairs tie or prefer the span kernel.
// dpx_run_device and simulate() both take their launches from here.
const std::vector<Launch> &launches_for(const PlanResult &plan, int in_fmt, int out_fmt);

// Host mirror of the kernels' index arithmetic (no arithmetic on samples): for every
// sample of a finalized plan, the counter value the launches would use, and how many
// times the sample is written.  n_out / writes hold plan.n_samples entries.  in_fmt / out_fmt: the format pair of the
// launch being mirrored (DPX_FMT_*: 0 = i16, 1 = f32) — a span launch cuts its grid per pair (span_launch_shape).
void simulate(const PlanResult &plan, uint32_t *n_out, uint8_t *writes, int in_fmt = 0, int out_fmt = 0);

// counter value at sample j of a stretch (host mirror of the kernels' counter_at)
uint32_t counter_at(const DevSeg &s, uint64_t j);

}  // namespace dpx
