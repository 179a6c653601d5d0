// dpx_simulate.cpp — host mirror of the kernels' index arithmetic (dpx_planner.h, simulate): for every sample of a
// finalized plan, the counter value the launches would use and how many launches write it.  No arithmetic on samples;
// walks exactly the grids dpx_kernels.hip launches (same helper functions: launches_for, span_launch_shape, span_split).
// Used by dpx_plan_simulate (tests/test_host_logic.py, tests/cpp/test_planner_fuzz.cpp), never by a launch.
#include <string.h>

#include <algorithm>

#include "dpx_planner.h"

namespace dpx {

void simulate(const PlanResult &plan, uint32_t *n_out, uint8_t *writes, int in_fmt, int out_fmt)
{
    const uint32_t ns = (uint32_t)plan.segs.size();
    auto put = [&](uint64_t g, uint32_t n) {
        n_out[g] = n;
        if (writes[g] < 255) ++writes[g];
    };
    auto generic = [&](uint32_t si, uint64_t g) {     // one_sample()
        while (si + 1 < ns && plan.segs[si].first + plan.segs[si].count <= g) ++si;
        put(g, counter_at(plan.segs[si], g - plan.segs[si].first));
    };
    for (const Launch &ln : launches_for(plan, in_fmt, out_fmt)) {
        if (ln.kind == 0) {
            const RowsArgs &r = ln.rows;
            const DevSeg &s = plan.segs[0];
            (void)s;
            // the table this launch reads: entry e -> ((n_first - 1 + e) mod P) + 1; a launch that evaluates instead
            // uses the same counters, from idx0 — the two must agree
            const TableBuild *tb = nullptr;
            for (const TableBuild &t : plan.tables) if (t.off == r.tab_off) tb = &t;
            if (r.compute && (r.idx0 != tb->n_first - 1u || r.P != tb->period)) put(0, 0xfffffff9u);
            const uint32_t first_idx = tb->n_first - 1u, period = tb->period;
            for (uint64_t rg = 0; rg < r.n_rg; ++rg)
                for (uint32_t row = 0; row < r.R; ++row)
                    for (uint32_t cs = 0; cs < r.L; ++cs) {
                        const uint64_t g = r.A + (rg * r.R + row) * (uint64_t)r.L + cs;
                        const uint32_t e = (r.L == r.P) ? cs : cs % r.P;          // the kernel's table index
                        put(g, (uint32_t)(((uint64_t)first_idx + e) % period) + 1u);
                    }
            for (uint64_t g = r.r0; g < r.A; ++g) generic(r.seg_lo, g);
            for (uint64_t g = r.B; g < r.r1; ++g) generic(r.seg_lo, g);
        } else if (ln.kind == 2) {
            const WalkArgs &wa = ln.walk;
            SpanLaunch sl;
            if (!span_launch_shape(wa, in_fmt, out_fmt, &sl)) { put(0, 0xfffffff5u); continue; }
            const uint32_t split = span_split(in_fmt, out_fmt), cols = kWalkWindow / split;
            if (wa.uni.n_spans) {
                // a one-matrix launch takes the spans from its arguments: the plan's must be the descriptor list's spans
                uint32_t c = 0;
                for (size_t i = 0; i + 1 < plan.walk.size(); ++i) {
                    const WalkSeg &d = plan.walk[i];
                    if (d.upw == 0) continue;
                    const WalkUni &u = wa.uni;
                    const uint32_t row0 = c * u.base + std::min(c, u.rem), row_end = row0 + u.base + (c < u.rem ? 1u : 0u);
                    if (d.A != u.seg.A || d.E != u.seg.E || d.L != u.seg.L || d.nw != u.seg.nw || d.period != u.seg.period || d.phase != u.seg.phase ||
                        memcmp(&d.ratio, &u.seg.ratio, 4) != 0 || d.row0 != row0 || d.row_end != row_end || d.wshift != 0 || d.upw != 2 || d.nwg != d.nw ||
                        u.nw8 != ((d.nw + 7u) & ~7u)) put(0, 0xfffffff8u);
                    ++c;
                }
                if (c != wa.uni.n_spans) put(0, 0xfffffff7u);
            }
            // one workgroup of a span (span_body): windows (w << wshift) ... of rows [row0, row_end), `half` of each where
            // a window is shared by two workgroups
            auto span_wg = [&](const WalkSeg &ws, uint32_t w, uint32_t half, bool multi) {
                const uint32_t wshift = multi ? ws.wshift : 0u;
                // the launch's wavefronts must split evenly over the windows, and hold every row's staging (xpose) slot
                if (ws.upw != 2 || wshift > kSpanMaxShift || sl.waves % (1u << wshift) != 0 ||
                    ws.row_end > ws.rows || ws.row_end <= ws.row0) { put(0, 0xfffffffbu); return; }
                const uint32_t P = ws.period;
                const uint32_t colbase = ((w * split + half) << wshift) * cols;
                const uint32_t n_entries = (cols << wshift) + kWalkPad;
                if ((uint64_t)ws.phase + (uint64_t)colbase + (uint64_t)P * kWalkPad > 0xffffffffull) { put(0, 0xfffffff6u); return; }
                const uint32_t ub = (ws.phase + colbase + P * kWalkPad - kWalkPad) % P;                 // the kernel's 32-bit arithmetic
                for (uint32_t sub = 0; sub < (1u << wshift); ++sub) {
                    const uint32_t col0 = colbase + sub * cols;
                    for (uint32_t r = ws.row0; r < ws.row_end; ++r) {
                        const uint64_t ideal = ws.A + (uint64_t)r * ws.L;
                        const uint64_t row0 = ideal & ~31ull;
                        const uint32_t delta = (uint32_t)ideal & 31u;
                        const uint64_t nxt = (ideal + ws.L) & ~31ull;
                        const uint32_t rowlen = (uint32_t)((nxt < ws.E ? nxt : ws.E) - row0);
                        for (uint32_t cl = 0; cl < cols; ++cl) {
                            const uint32_t c = col0 + cl;
                            if (c >= rowlen) break;                                   // lanes past the row are masked
                            const uint32_t j = kWalkPad - delta + sub * cols + cl;    // index in the workgroup's slice
                            if (j >= n_entries) { put(row0 + c, 0xffffffffu); continue; }
                            uint32_t t = ub + j;
                            if (P > n_entries) t = t >= P ? t - P : t;
                            else               t %= P;
                            if (t >= P) { put(row0 + c, 0xfffffffdu); continue; }
                            put(row0 + c, t + 1u);
                        }
                    }
                }
            };
            auto leftover = [&](uint32_t e) {
                if (e >= wa.n_left_wg) { put(0, 0xfffffffau); return; }
                uint32_t li = plan.left_hint[e >> kLeftHintShift];
                while (plan.left[li + 1].wg_off <= e) ++li;
                const LeftRange &lr = plan.left[li];
                const DevSeg &sg = plan.segs[lr.seg];
                const uint32_t o0 = (e - lr.wg_off) * kLeftBlock;
                for (uint32_t o = 0; o < kLeftBlock && o0 + o < lr.len; ++o) {
                    const uint64_t g = lr.start + o0 + o;
                    put(g, counter_at(sg, g - sg.first));
                }
            };
            if (sl.uni.n_spans) {
                // the 2-D grid of a one-matrix launch, with the spans as THIS format pair cuts them
                const WalkUni &u = sl.uni;
                if ((uint64_t)u.n_spans + sl.left_rows > 65535u) { put(0, 0xfffffff4u); continue; }
                for (uint32_t c = 0; c < u.n_spans + sl.left_rows; ++c)
                    for (uint32_t w = 0; w < u.nw8; ++w)
                        for (uint32_t half = 0; half < split; ++half) {
                            if (c < u.n_spans) {
                                if (w >= u.seg.nw) continue;
                                WalkSeg ws = u.seg;
                                ws.row0 = c * u.base + std::min(c, u.rem);
                                ws.row_end = ws.row0 + u.base + (c < u.rem ? 1u : 0u);
                                span_wg(ws, w, half, false);
                            } else {
                                const uint32_t e = (c - u.n_spans) * u.nw8 + w;
                                if (half != 0 || e >= wa.n_left_wg) continue;
                                leftover(e);
                            }
                        }
                continue;
            }
            for (uint32_t b = 0; b < wa.n_walk_wg; ++b) {
                const WalkSeg &ws = plan.walk[plan.walk_hint[b >> kWalkHintShift]];
                if (b < ws.wg_base || b - ws.wg_base >= ((ws.nwg + 7u) & ~7u)) { put(0, 0xfffffffcu); continue; }   // hint not exact
                const uint32_t w = b - ws.wg_base;
                if (w >= ws.nwg) continue;
                if (ws.upw == 0) {                                                 // a group of leftover blocks
                    leftover(ws.row0 + w);
                    continue;
                }
                if (ws.nwg != ((ws.nw + (1u << ws.wshift) - 1) >> ws.wshift)) { put(0, 0xfffffff3u); continue; }
                for (uint32_t half = 0; half < split; ++half) span_wg(ws, w, half, true);
            }
        } else {
            const TileArgs &t = ln.tiles;
            for (uint64_t tile = t.tile_lo; tile < t.tile_lo + t.n_tiles; ++tile) {
                const uint64_t t0 = tile * plan.tile;
                const bool in_mask = t0 >= t.m0 && t0 + plan.tile <= t.m1;
                const uint64_t gs = t0 > t.m0 ? t0 : t.m0;
                uint32_t si = plan.hint[gs >> kHintShift];
                while (si + 1 < ns && plan.segs[si].first + plan.segs[si].count <= gs) ++si;
                const DevSeg &sg = plan.segs[si];
                const bool whole = in_mask && t0 >= sg.first && t0 + plan.tile <= sg.first + sg.count;
                if (whole && sg.lut_len != 0 && (sg.flags & kSegTileTable)) {
                    const uint32_t P = sg.period;
                    uint32_t ph = P <= (1u << 18) ? sg.c0 + (((uint32_t)tile % P) * sg.tmod) % P
                                                  : sg.c0 + (uint32_t)(t0 % P);
                    ph = ph >= P ? ph - P : ph;
                    for (uint32_t e = 0; e < plan.tile; ++e) put(t0 + e, ((ph + e) % P) + 1u);
                } else if (whole && (sg.period == 0 || sg.period >= 4)) {
                    const uint32_t P = sg.period;
                    const uint64_t j0 = t0 - sg.first;
                    for (uint32_t e = 0; e < plan.tile; ++e) {
                        if (P) put(t0 + e, (uint32_t)((((uint64_t)(sg.n_start - 1u) + j0) % P + e) % P) + 1u);
                        else put(t0 + e, sg.n_start + (uint32_t)j0 + e);
                    }
                } else {
                    for (uint32_t e = 0; e < plan.tile; ++e) {
                        const uint64_t g = t0 + e;
                        if (g < t.m0) continue;
                        if (g >= t.m1) break;
                        generic(si, g);
                    }
                }
            }
        }
    }
}

}  // namespace dpx
