// dpx_types.h — structures shared by the host planner and the gfx950 kernels.
#pragma once
#include <stdint.h>

namespace dpx {

// One stretch of the stream over which the counter of dsp.rs:125-130 follows a
// single closed form.  For sample j = g - first of the stretch:
//   period == 0 ("linear"):   n = n_start + j            (no reset inside)
//   period  > 0 ("periodic"): n = ((n_start - 1 + j) mod period) + 1
// lut_len > 0 ("tabulated"): the correctors of one period (lut_len == period)
// are precomputed, in up to two layouts (a stretch may have both):
//   rows kernel (RowsArgs::tab_off): period + 3 entries starting at the phase of
//              sample RowsArgs::A; index = column mod period, no phase arithmetic;
//   tile kernel (kSegTileTable; lut_off, c0, tmod): period + tile entries, entry e
//              holding corrector((e mod period) + 1), indexed from the tile's
//              phase (c0 + tile_index * tmod) mod period.
// lut_len == 0: sincos is evaluated per sample.
struct DevSeg {
    uint64_t first;     // global sample index of the first sample
    uint64_t count;     // samples in this stretch (> 0)
    float ratio;        // fl32(shift_hz / fl32(samplerate)), dsp.rs:121
    uint32_t n_start;   // counter value used by the first sample
    uint32_t period;
    uint32_t lut_len;
    uint32_t lut_off;   // table-pool entry index of this stretch's table
    uint32_t c0;        // (n_start - 1 - first) mod period: phase of global sample 0
    uint32_t tmod;      // tile mod period
    uint32_t flags;
    // |theta(n)| never decreases with the counter n (every rounding in fl32(-2 pi * fl32(ratio * fl32(n))) is monotone), so
    // the range of sincosf's argument paths is three counters, found on the host by bisection with the same f32 products:
    // the first n whose |theta| bits reach kThetaPlain / kThetaLarge / kThetaHuge = 2^-12 / 120 / 2^29 (0xffffffff: none).  A tile whose counters lie inside
    // [n_plain, n_large) or [n_large, n_huge) knows its path from two scalar comparisons (dpx_sincos.h, corrector4_f).
    uint32_t n_plain, n_large, n_huge;
    uint32_t pad;
};
static_assert(sizeof(DevSeg) == 64, "DevSeg is read with scalar loads");
// float bit patterns of the |theta| bounds of sincosf's fast paths (dpx_sincos.h): [2^-12, 120) and [120, 2^29) are the two
// ranges over which the device's cheaper operation sequences are proved, by enumeration, to give glibc's floats
constexpr uint32_t kThetaPlain = 0x39800000u, kThetaLarge = 0x42f00000u, kThetaHuge = 0x4e000000u;

constexpr uint32_t kSegRows = 2u;        // (most of) this stretch is served by a rows-kernel launch
constexpr uint32_t kSegTileTable = 4u;   // lut_off / c0 / tmod describe a tile-kernel table
constexpr uint32_t kSegWalk = 8u;        // (most of) this stretch is a matrix of the span-kernel launch

// the first 32 bytes, as exposed through the C ABI (dpx_stretch)
struct StretchView {
    uint64_t first, count;
    float ratio;
    uint32_t n_start, period, lut_len;
};

constexpr int kSamplesPerLane = 4;          // tile kernel: one 16-byte i16 vector = 4 IQ samples
constexpr uint32_t kLutMaxEntries = 4194304; // longest tile-kernel table (32 MiB: beyond L2, inside the 256 MiB Infinity Cache)
constexpr uint32_t kRowsMaxL = 1u << 24;     // longest rows-kernel row, in samples (the table is one period whatever L)
constexpr uint32_t kRowsMultMaxL = 131072;  // multiples of the basic row length are only considered up to this
constexpr int kHintShift = 16;              // one stretch hint per 65536 samples
constexpr int kRowsLanes = 64;              // rows kernel: one wavefront per workgroup

// tile-kernel geometry
struct LaunchGeom {
    int block;    // 128 or 256 lanes per workgroup
    int vecs;     // 4-sample groups per lane: 1 or 2
    int autosel = 0;   // the caller set neither: 128 x 2 or 256 x 1 (the same 1024-sample tile) is chosen per launch
    int legacy_cast = 0;   // dpx_set_i16_cast(DPX_CAST_LEGACY_X86): i16 output wraps instead of saturating (every kernel: a launch-uniform flag)
    uint32_t tile() const { return (uint32_t)block * kSamplesPerLane * (uint32_t)vecs; }
};

// One rows-kernel launch: a tabulated periodic stretch processed as a matrix whose
// rows are L samples long (L a multiple of the period and of 4), so that a column
// always sees the same corrector.  A workgroup (one wavefront) takes R (2 or 4)
// consecutive rows x 64 lanes x S samples; column slices of one row group are
// consecutive workgroups, so the grid sweeps HBM contiguously.
// Workgroups past n_rg * cols evaluate the ragged ranges [r0, A) and [B, r1)
// sample by sample (generic stretch lookup, sincos per sample).
struct RowsArgs {
    uint64_t A;          // first sample of the matrix (multiple of 256)
    uint64_t B;          // one past its last sample: A + n_rg * R * L
    uint64_t r0, r1;     // ragged ranges [r0, A) and [B, r1) handled by the extra workgroups
    uint64_t n_rg;       // row groups
    uint32_t L;          // row length in samples
    uint32_t tab_off;    // table-pool entry index of the (P + 3)-entry table (origin: sample A)
    uint32_t seg_lo;     // index of the stretch holding r0
    uint32_t n_segs;
    uint32_t R;          // rows per workgroup: 2 or 4
    uint32_t P;          // period of the stretch (L is a multiple of it)
    // compute != 0: the launch leaves the table alone — every wavefront evaluates the correctors of its 64 x S
    // columns itself, once for its R rows (R = 2: half a sincos per sample), while the rows' loads are in flight.
    // For periods whose table does not stay in the CUs' vector caches (the planner's kRowsComputeMinP).
    uint32_t compute;
    uint32_t idx0;       // counter of sample A, minus one: column c uses counter ((idx0 + c) mod P) + 1
    float ratio;
    uint32_t pad;
};

// ---- span kernel: many periodic stretches in ONE launch (track mode: one stretch per second of stream), and the
// const-mode stretches whose period does not allow rows of whole 4 KiB pages.
// A stretch [A, E) is viewed as rows of L samples (L a multiple of the period), row r starting at the 32-sample
// boundary at or below A + r * L, so every wavefront access is a whole number of 128-byte lines whatever the
// period.  Row r is shifted left by delta_r = (A + r * L) mod 32 samples, so its lanes use the correctors of the
// columns delta_r to the right.  A workgroup takes a 256-sample column window of a SPAN of rows: it evaluates the
// window's 288 correctors once with the bit-exact sincos (while its first rows' loads are in flight), shares them
// through LDS, and its wavefronts mix and store the rows, `upw` per wavefront per turn.
// The launch is a list of spans (WalkSeg), stretch by stretch, each padded to a multiple of 8 workgroups.
//
// Short matrices (round 4).  A span of up to 5 rows leaves half or three quarters of a 4-wavefront workgroup without rows
// (the replay's seconds near the closest approach: periods above a quarter of a second), and measures 6-9 points better
// under 2 wavefronts per window (profiles/r04_walk.md).  A launch has one workgroup size, so such a span's workgroups take
// 2^wshift ADJACENT windows instead, WAVES >> wshift wavefronts each, one contiguous slice: every wavefront has rows, and
// the span needs a half or a quarter of the workgroups.  (Three rows per wavefront in one turn for spans of 9-12 rows —
// instead of a second turn for a part of the wavefronts — was built too and dropped: no gain for those spans, and the
// second instantiation of the body cost every other span 1.5 points, profiles/r04_walk.md.)
struct WalkSeg {           // one span of one stretch's matrix, or (upw == 0) a group of up to 8 leftover blocks
    uint64_t A;            // first sample of the matrix (multiple of 32)
    uint64_t E;            // one past its last sample (multiple of 32)
    uint32_t L;            // row length in samples
    uint32_t wshift;       // log2 of the column windows a workgroup of this span takes (0, 1 or 2)
    uint32_t wg_base;      // first workgroup of this span (multiple of 8); workgroup wg_base + w takes windows w << wshift ...
    uint32_t nw;           // column windows per row
    uint32_t rows;         // rows of the matrix
    uint32_t row0;         // first row of this span (leftover group: first leftover block)
    uint32_t period;       // of the stretch (L is a multiple of it)
    uint32_t phase;        // counter of sample A, minus 1: column c uses counter ((phase + c) mod period) + 1
    float ratio;           // of the stretch (dsp.rs:121)
    uint32_t upw;          // rows per wavefront per turn: 2 (0: a group of leftover blocks)
    uint32_t row_end;      // one past the span's last row (<= rows)
    uint32_t nwg;          // workgroups of this span that have work: ceil(nw >> wshift) (leftover group: blocks in the group)
};
static_assert(sizeof(WalkSeg) == 64, "WalkSeg is read with scalar loads");

// what the span launch's matrices do not cover (heads, tails, lead-ins, stretches too short for a matrix or a tile
// launch): ranges inside ONE stretch each, evaluated sample by sample in blocks of kLeftBlock samples
struct LeftRange {
    uint64_t start;
    uint32_t len;
    uint32_t seg;          // index of the stretch holding the range
    uint32_t wg_off;       // first block of this range among the leftover workgroups
    uint32_t pad;
};
static_assert(sizeof(LeftRange) == 24, "LeftRange is read with scalar loads");

constexpr uint32_t kWalkPad = 32;          // slice entries before column 0 (the largest row shift is 31)
constexpr uint32_t kWalkWindow = 256;      // samples per column window
constexpr uint32_t kWalkMinL = 8192;       // no row is shorter (a multiple of the period otherwise)
constexpr uint32_t kWalkRowTarget = 262144; // long matrices: the multiple of the period that reaches this many samples (1 MB of i16) per row
constexpr int kWalkHintShift = 3;          // one WalkSeg index per 8 workgroups: exact, every span is padded to a multiple of 8
constexpr int kLeftHintShift = 4;          // one LeftRange hint per 16 leftover workgroups
constexpr uint32_t kLeftBlock = 1024;      // samples per leftover workgroup
constexpr uint32_t kWalkSlice = kWalkWindow + kWalkPad;   // slice entries a window needs: 288
constexpr uint32_t kSpanRows = 8;          // most rows of a matrix one workgroup keeps its window for (dpx_options.walk_span): one turn of 4 x 2
constexpr uint32_t kSpanWhole = 12;        // ... but a matrix of up to this many rows is one span (rows 9-12: a second turn of the first wavefronts)
constexpr uint32_t kSpanWaves = 4;         // wavefronts per workgroup
#ifndef DPX_SPAN_MAX_SHIFT
#define DPX_SPAN_MAX_SHIFT 2
#endif
constexpr uint32_t kSpanMaxShift = DPX_SPAN_MAX_SHIFT;   // a workgroup takes at most 4 windows (tools/build_variant.sh: 0 or 1 for A/B builds)

// A span launch that consists of ONE matrix (const mode: an odd period, or a long one) needs no descriptor table at all:
// every span is the same WalkSeg but for its rows, which follow from the span's index.  The span kernel then takes the
// matrix from its kernel arguments (the first dwords preloaded into scalar registers at wavefront launch) and the span
// from blockIdx.y, and issues its sample loads without the scalar-load round trip that every workgroup of a
// many-matrix launch starts with (to HBM: each group of 8 workgroups has its own descriptor line).  Such spans take one
// window per workgroup.
struct WalkUni {
    WalkSeg seg;           // the matrix; row0 / row_end / wg_base unused
    uint32_t n_spans;      // 0: not a one-matrix launch
    uint32_t base, rem;    // span c takes rows [c * base + min(c, rem), + base + (c < rem))
    uint32_t nw8;          // workgroups per span in the descriptor list: nw rounded up to a multiple of 8
};

struct WalkArgs {
    uint32_t n_walk_wg;    // workgroups of the whole grid (spans and leftover groups)
    uint32_t n_left_wg;    // leftover blocks among them
    uint32_t n_segs;
    uint32_t waves;        // wavefronts per workgroup the descriptors were laid out for (2, 4, 5 or 8)
    uint32_t span;         // most rows per span the descriptors were cut for
    WalkUni uni;           // launches of one matrix
    uint32_t auto_shape;   // the caller named neither walk_waves nor walk_span: the launch may choose per format pair
    uint32_t sub_lg;       // the grid is dealt out in pieces of about 2^sub_lg samples (0: kSubLaunchLg; sub_launch_pieces)
    uint64_t cover;        // samples the launch produces (matrices and leftover ranges)
};

// Long launches are cut into sub-launches back to back on the stream (dpx_kernels.hip, span_t: 2-3 points at 4-8 GiB):
// how many pieces a launch over n samples becomes.  sub_lg >= 48 never cuts (measurement).
constexpr uint32_t kSubLaunchLg = 28;                    // 2^28 samples: 1 GiB of i16 in
inline uint32_t sub_launch_pieces(uint64_t n, uint32_t sub_lg)
{
    const uint32_t lg = sub_lg ? sub_lg : kSubLaunchLg;
    if (lg >= 48) return 1;
    const uint64_t per = 1ull << lg, k = (n + per / 2) / per;      // 1.4 pieces' worth stays one launch
    return k < 1 ? 1u : k > 4096 ? 4096u : (uint32_t)k;
}

// What ONE launch makes of a plan's span shape for its format pair (the plan does not know the formats): dpx_planner.cpp,
// span_launch_shape — used by the launch wrapper and by the planner's host simulation alike.
struct SpanLaunch {
    WalkUni uni;           // the one-matrix spans as launched (cut again for pairs with an f32 side)
    uint32_t waves;        // wavefronts per workgroup as launched
    uint32_t left_rows;    // one-matrix launches: grid rows that hold the leftover blocks
};
bool span_launch_shape(const WalkArgs &w, int in_fmt, int out_fmt, SpanLaunch *out);
// windows of 256 columns are shared by this many workgroups of a format pair (f32 output without the LDS transposition: 2)
constexpr uint32_t span_split(int in_fmt, int out_fmt) { return (out_fmt == 1 /* DPX_FMT_F32 */) ? 2u : 1u; }

struct TileArgs {
    uint64_t tile_lo;    // first tile of this launch (global tiling from sample 0)
    uint64_t n_tiles;
    uint64_t m0, m1;     // only samples in [m0, m1) are produced
    uint32_t legacy;     // dpx_set_i16_cast(DPX_CAST_LEGACY_X86): set by the launch wrapper
    uint32_t pad;
};

// ---- resident block kernel (round 4): one 8 KiB block per call without a launch per call.
// The reference's loop hands ONE block at a time to its operator (src/main.rs:113-118).  A launch and an event per block
// cost 8.6-15.8 us; the kernel below is launched ONCE for a context's staging slots and stays: workgroup s polls the
// doorbell of slot s in host-mapped memory, and when the host rings it reads the block (samples + stretch list) over PCIe,
// evaluates it with the per-sample path and writes result and completion word back to host memory.  A block then costs
// PCIe round trips, not a launch, and the slots' round trips overlap.  The kernel is ONE unit on ONE HIP stream (a
// resident kernel holds its hardware queue; launches of other streams that share the queue wait for it): all its workgroups
// leave together — on request (kDoorExit in every doorbell: the context does that before any launch of its own), or once
// none of them has had work for `idle_ticks` of the 100 MHz wall clock (a word in device memory the first idle workgroup
// sets), so an abandoned or crashed caller leaves nothing running.  The next block after that pays one launch.
constexpr int kResidentSlots = 4;
constexpr int kResidentThreads = 512;      // one quad per thread for the reference's 2048-sample block
struct BlockCtl {
    // ---- written by the host, read by the kernel (one 64-byte line).  The kernel takes the first 16 bytes in ONE load, but
    // nothing guarantees that a 16-byte read of host memory over PCIe observes one state of the line, so the word proves
    // itself (ctl_word_valid): the ticket stands at BOTH ends, and the payload dword carries the ticket's low bits — a read
    // that mixes dwords of two writes of the slot (consecutive tickets of a slot differ by kResidentSlots) fails the test
    // and counts as "not rung yet"; the next poll sees the settled word.
    uint32_t doorbell;     // ticket of the block to process; the kernel acts when it differs from `done`; kDoorExit: leave
    uint32_t payload;      // ctl_payload(): n_samples, n_segs, cast mode, the kernel instance the block is meant for, ticket tag
    uint32_t reserved;
    uint32_t doorbell2;    // the ticket again
    uint32_t pad0[12];
    // ---- written by the kernel, read by the host (another line)
    uint32_t done;         // ticket of the last block finished
    uint32_t state;        // kResidentRunning while the workgroup polls; kResidentParked once it has left (its last store)
    uint32_t blocks;       // blocks processed since launch (statistics)
    uint32_t pad1[13];
};
// payload dword: bits 0-13 n_samples (<= 8192), 14-18 n_segs (<= 16), 19 legacy cast, 20 in_fmt, 21 out_fmt, 22 fma
// (20-22: the kernel instance — a resident kernel of another instance does not serve the ticket, it leaves), 23-31 ticket tag
constexpr uint32_t ctl_instance(int in_fmt, int out_fmt, bool fma) { return (uint32_t)in_fmt | ((uint32_t)out_fmt << 1) | (fma ? 4u : 0u); }
constexpr uint32_t ctl_payload(uint32_t ticket, uint32_t n_samples, uint32_t n_segs, uint32_t legacy, uint32_t instance)
{
    return (n_samples & 0x3fffu) | ((n_segs & 0x1fu) << 14) | ((legacy & 1u) << 19) | ((instance & 7u) << 20) | ((ticket & 0x1ffu) << 23);
}
constexpr bool ctl_word_valid(uint32_t doorbell, uint32_t payload, uint32_t doorbell2)
{
    return doorbell == doorbell2 && (payload >> 23) == (doorbell & 0x1ffu);
}
static_assert(sizeof(BlockCtl) == 128, "two lines: host-written and kernel-written");
struct ResidentShared {    // device memory, one per context
    unsigned long long activity;   // wall clock of the last block any workgroup finished
    uint32_t leaving;              // set by the first workgroup that finds the kernel idle: everyone leaves
    uint32_t pad;
};
struct ResidentArgs {
    BlockCtl *ctl[kResidentSlots];
    const uint8_t *in[kResidentSlots];
    uint8_t *out[kResidentSlots];
    const DevSeg *segs[kResidentSlots];
    ResidentShared *shared;
    uint64_t idle_ticks;
};
constexpr uint32_t kDoorExit = 0xffffffffu;
constexpr uint32_t kResidentRunning = 1, kResidentParked = 2;
constexpr uint32_t kResidentMaxSegs = 16;
int launch_resident_block(const ResidentArgs &args, int in_fmt, int out_fmt, bool fma, void *stream);

// launch wrappers implemented in dpx_kernels.hip (all asynchronous on `stream`)
int launch_tiles(const void *d_in, int in_fmt, void *d_out, int out_fmt, const DevSeg *d_segs,
                 uint32_t n_segs, const uint32_t *d_hint, const void *d_lut, const TileArgs &t,
                 bool fma, const LaunchGeom &g, void *stream);
int launch_rows(const void *d_in, int in_fmt, void *d_out, int out_fmt, const DevSeg *d_segs,
                const void *d_lut, const RowsArgs &r, bool fma, int legacy_cast, void *stream);
int launch_span(const void *d_in, int in_fmt, void *d_out, int out_fmt, const DevSeg *d_segs,
                const WalkSeg *d_walk_desc /* one per 2^kWalkHintShift workgroups */, const LeftRange *d_left,
                const uint32_t *d_left_hint, const WalkArgs &w, bool fma, int legacy_cast, void *stream);
int launch_build_lut(void *d_lut_entries, uint32_t period, uint32_t n_first, uint32_t n_entries,
                     float ratio, bool fma, void *stream);
int launch_copy(const void *d_in, void *d_out, uint64_t n_bytes, void *stream);
// desc[h] = spans[index[h]] for h < n_desc (the span launch's descriptor per group of 8 workgroups, written on the device)
int launch_expand_walk(const void *d_spans, const void *d_index, void *d_desc, uint32_t n_desc, void *stream);
int launch_unpack_i16(const void *d_in, void *d_out, uint64_t n_samples, void *stream);
int launch_pack_i16(const void *d_in, void *d_out, uint64_t n_samples, void *stream, int legacy_cast = 0);
int launch_ccexpf_imag(void *d_z, uint64_t n, bool fma, void *stream);
int launch_ccexpf(void *d_z, uint64_t n, bool fma, void *stream);

}  // namespace dpx
