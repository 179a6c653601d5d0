// dpx_planner.cpp — see dpx_planner.h.  Host code, plain C++ (no HIP), compiled
// with -ffp-contract=off: the f32 products below must round exactly like the
// reference's (src/dsp.rs:121,125).
#include "dpx_planner.h"

#include <immintrin.h>
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <utility>

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

// The scan, candidate by candidate — the definition of the counter rule's reset points, the planner's whole cost in rounds
// 1-5, and since round 6 what find_reset() is held against.  Where the host has
// AVX2 it runs eight candidates per instruction in blocks, and only a block that holds a reset is looked at candidate by candidate.
// (The library itself no longer calls it: dpx_find_reset_scan exports it for the comparison.)  Every
// candidate still gets exactly the reference's arithmetic: fl32(ratio * fl32(n)) — one IEEE multiply per lane, the
// int -> f32 conversion in the current (nearest-even) rounding mode — and the same test as product_is_integer().
__attribute__((target("avx2")))
static int block_has_reset_avx2(float ratio, uint32_t n0, int len)      // candidates n0 .. n0 + len - 1 < 2^31, len % 8 == 0
{
    const __m256 r = _mm256_set1_ps(ratio);
    const __m256 two23 = _mm256_set1_ps(8388608.0f), fmax = _mm256_set1_ps(3.4028234663852886e38f);
    const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7fffffff));
    __m256i n = _mm256_add_epi32(_mm256_set1_epi32((int)n0), _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7));
    const __m256i step = _mm256_set1_epi32(8);
    __m256 acc = _mm256_setzero_ps();
    for (int i = 0; i < len; i += 8) {
        const __m256 p = _mm256_mul_ps(r, _mm256_cvtepi32_ps(n));
        const __m256 ap = _mm256_and_ps(p, absmask);
        const __m256 t = _mm256_cvtepi32_ps(_mm256_cvttps_epi32(p));       // only looked at where |p| < 2^23
        const __m256 small = _mm256_cmp_ps(ap, two23, _CMP_LT_OQ);
        const __m256 same = _mm256_cmp_ps(t, p, _CMP_EQ_OQ);
        const __m256 finite = _mm256_cmp_ps(ap, fmax, _CMP_LE_OQ);         // false for inf and nan
        acc = _mm256_or_ps(acc, _mm256_or_ps(_mm256_and_ps(small, same), _mm256_andnot_ps(small, finite)));
        n = _mm256_add_epi32(n, step);
    }
    return _mm256_movemask_ps(acc) != 0;
}

static const bool kHaveAvx2 = __builtin_cpu_supports("avx2");

bool find_reset_scan(float ratio, uint32_t n_start, uint64_t max_scan, uint32_t *n_reset)
{
    const uint64_t to_wrap = (1ULL << 32) - (uint64_t)n_start;
    const uint64_t span = std::min(max_scan, to_wrap);
    uint64_t n = n_start;
    const uint64_t end = (uint64_t)n_start + span;
    constexpr int kBlock = 256;
    while (kHaveAvx2 && n + kBlock <= end && n + kBlock <= (1ULL << 31)) {
        if (block_has_reset_avx2(ratio, (uint32_t)n, kBlock)) break;   // the first reset is in this block: find it below
        n += kBlock;
    }
    for (; n < end; ++n) {
        const float p = ratio * (float)(uint32_t)n;
        if (product_is_integer(p)) {
            *n_reset = (uint32_t)n;
            return true;
        }
    }
    return false;
}

// ---- the first reset without a scan (round 6).
//
// For n < 2^24 the counter converts exactly, so p = fl32(ratio * n) is the rounding of the EXACT product X = M n 2^E
// (|ratio| = M 2^E, M < 2^24 an integer; the sign plays no part: rounding to nearest even is symmetric).  With
// k = floor(log2 X) and E < 0, Q = 2^-E:
//     k >= 23            every float is an integer: reset (unless the product overflows)
//     k <  23            p is an integer I  <=>  |X - I| <= ulp(X) / 2 = 2^(k-24)   [ties go to I: its significand is even
//                        in every binade below 2^23; I = 2^(k+1), the binade's upper end, included; I = 0 never: X >= 2^k]
//                        <=>  (M n mod Q) in [0, T] or [Q - T, Q),  T = floor(Q 2^(k-24)) = floor(2^(s-24)),  s = floor(log2(M n))
// Inside one binade the tolerance T is a constant and n runs over an interval, so the first reset in it is the smallest
// x >= 0 with  l <= (a x mod Q) <= r  for a = M mod Q and an interval [l, r] that follows from where the binade starts:
// the Euclid-like descent of first_in_window() below, O(log Q) steps whatever the period.  At most ~25 binades lie
// between n_start and X >= 2^23; binades in which even the best approximation so far (the remainder sequence of
// Euclid's algorithm on (Q, a): |q_i a - p_i Q|, the smallest distance any n < q_(i+1) reaches) stays outside the
// tolerance are skipped without a descent.  From 2^24 on (ratios below ~3e-8 with nothing found yet: a shift of
// centihertz at megasamples per second) the counter itself is rounded first; the same search then runs over its 24-bit
// significand, binade of the counter by binade (first_reset_exact).  No candidate is ever tried one by one.
// find_reset_scan is the definition; tests/cpp/test_planner_fuzz.cpp and tests/test_host_logic.py hold the two against
// each other (every start below 2^24 of selected ratios, random ratios of every exponent, ties, subnormals, overflow).
typedef unsigned __int128 u128;
static constexpr uint64_t kNone = ~0ull;

// smallest x >= 0 with l <= (a x mod m) <= r, or kNone.  0 <= a < m, 0 <= l <= r < m; results beyond `cap` are reported
// as kNone (the caller's binade ends there: no need to finish the arithmetic exactly).
static uint64_t first_in_window(uint64_t a, uint64_t m, uint64_t l, uint64_t r, uint64_t cap)
{
    if (l == 0) return 0;
    if (a == 0) return kNone;
    const uint64_t c = (l + a - 1) / a;
    if ((u128)a * c <= r) return c <= cap ? c : kNone;
    // no multiple of a inside [l, r]: a x - m y in [l, r] needs y >= 1, and y must put (m y mod a) into [-r, -l] mod a
    const uint64_t l2 = (a - r % a) % a, r2 = (a - l % a) % a;
    // x = ceil((l + m y) / a) <= cap  =>  y <= (a cap + a - l) / m: the descent only ever needs y up to there
    const u128 ycap = ((u128)a * cap + a) / m + 1;
    const uint64_t y = first_in_window(m % a, a, l2, r2, ycap > (u128)kNone - 1 ? kNone - 1 : (uint64_t)ycap);
    if (y == kNone) return kNone;
    const u128 x = ((u128)l + (u128)m * y + a - 1) / a;
    return x <= cap ? (uint64_t)x : kNone;
}

// smallest m in [lo, hi), hi <= 2^24 + 1, for which the rounding of the exact product X = M m 2^E (M < 2^24) to f32 is an
// integer by the rule above; kNone if there is none.  *big: the answer is the first m with X >= 2^23 (an integer unless the
// product overflowed: the caller looks).  m is the counter itself below 2^24 and the counter's 24-bit significand beyond.
static uint64_t first_integer_product(uint64_t M, int E, uint64_t lo, uint64_t hi, bool *big)
{
    *big = false;
    if (lo >= hi) return kNone;
    if (E >= 0) { *big = true; return lo; }              // every product is an integer of at least 2^23
    if (-E > 62) return kNone;                           // X < 2^25 * 2^24 * 2^-63: every product below 1/2
    const uint64_t Q = 1ull << -E;
    const uint64_t a = M % Q;
    // remainder sequence of Euclid on (Q, a): rem[i] = |q[i] a - p Q| is the smallest distance to a multiple of Q that any
    // 1 <= m < q[i+1] reaches
    uint64_t qs[96], rems[96];
    int n_conv = 0;
    {
        uint64_t r0 = Q, r1 = a, q0 = 0, q1 = 1;
        while (r1 != 0 && n_conv < 94) {
            qs[n_conv] = q1;
            rems[n_conv] = r1;
            ++n_conv;
            const uint64_t t = r0 / r1, r2 = r0 - t * r1;
            const u128 q2 = (u128)q0 + (u128)t * q1;
            r0 = r1; r1 = r2;
            q0 = q1; q1 = q2 > (u128)1 << 40 ? 1ull << 40 : (uint64_t)q2;
        }
        qs[n_conv] = q1;                                 // the denominator at which the remainder reaches 0 (or the cut-off)
        rems[n_conv] = r1;
        ++n_conv;
    }
    uint64_t m = lo;
    int conv = 0;
    while (m < hi) {
        const uint64_t mm = M * m;                       // < 2^49
        const int s = 63 - __builtin_clzll(mm);          // floor(log2(M m)); k = s + E
        if (s + E >= 23) { *big = true; return m; }      // X >= 2^23
        // the binade's last m: M m < 2^(s+1)
        const uint64_t m_last = std::min<uint64_t>(hi - 1, ((1ull << (s + 1)) - 1) / M);
        if (s + E < -1) { m = m_last + 1; continue; }    // X < 1/2: no integer within half an ulp
        const uint64_t T = s >= 24 ? 1ull << (s - 24) : 0;
        // skip the binade if no m up to its end comes within T of a multiple of Q at all
        while (conv + 1 < n_conv && qs[conv + 1] <= m_last) ++conv;
        if (qs[conv] <= m_last && rems[conv] > T && conv + 1 < n_conv) { m = m_last + 1; continue; }
        const uint64_t b = (uint64_t)(((u128)a * m + T) % Q);
        uint64_t x;
        if (b <= 2 * T) x = 0;
        else x = first_in_window(a, Q, Q - b, Q - b + 2 * T, m_last - m);
        if (x != kNone && x <= m_last - m) return m + x;
        m = m_last + 1;
    }
    return kNone;
}

// first n in [n_start, end), end <= 2^32, with fl32(ratio * fl32(n)) an integer.  Below 2^24 the counter is the multiplicand;
// from 2^24 on it is rounded first (to nearest, ties to even): in the binade [2^(23+j), 2^(24+j)) the multiplicand is
// c 2^j with c = RNE(n / 2^j) in [2^23, 2^24], so the same search runs over c with the exponent E + j, and the first counter
// that rounds to the c it finds is c 2^j - 2^(j-1) (+ 1 when c is odd: the tie goes to the even neighbour).
static bool first_reset_exact(float ratio, uint64_t n_start, uint64_t end, uint32_t *n_reset)
{
    uint32_t bits;
    memcpy(&bits, &ratio, sizeof bits);
    const uint32_t ef = (bits >> 23) & 0xffu, frac = bits & 0x7fffffu;
    if (n_start >= end) return false;
    if (ef == 0xffu) return false;                       // inf / nan: no product is an integer (0 * inf = nan as well)
    if (n_start == 0 || (ef == 0 && frac == 0)) {        // n = 0, or ratio = +-0: the product is 0
        *n_reset = (uint32_t)n_start;
        return true;
    }
    const uint64_t M = ef ? (frac | 0x800000u) : frac;   // |ratio| = M * 2^E
    const int E = ef ? (int)ef - 150 : -149;
    constexpr uint64_t kExact = 1ull << 24;              // counters below convert to f32 exactly
    uint64_t n = n_start;
    bool big = false;
    if (n < kExact) {
        const uint64_t m = first_integer_product(M, E, n, std::min(end, kExact), &big);
        if (m != kNone) {
            if (big && !is_reset(ratio, (uint32_t)m)) return false;      // overflowed: inf from here on (the product grows with n)
            *n_reset = (uint32_t)m;
            return true;
        }
        n = kExact;
    }
    for (int j = 1; j <= 8 && n < end; ++j) {
        const uint64_t top = 1ull << (24 + j), h = 1ull << (j - 1);     // the binade's end; half a step of the rounded counter
        if (n >= top) continue;
        auto significand = [&](uint64_t v) {             // RNE(v / 2^j)
            const uint64_t c = v >> j, rem = v & ((1ull << j) - 1);
            return c + (rem > h || (rem == h && (c & 1u)) ? 1u : 0u);
        };
        const uint64_t last = std::min(end, top) - 1;    // the binade's last counter in the window
        const uint64_t c_lo = significand(n), c_hi = significand(last) + 1;
        const uint64_t c = first_integer_product(M, E + j, c_lo, c_hi, &big);
        if (c != kNone) {
            const uint64_t first_n = (c << j) - h + ((c & 1u) ? 1u : 0u);  // the smallest counter that rounds to c 2^j
            const uint64_t m = std::max(n, first_n);
            if (big && !is_reset(ratio, (uint32_t)m)) return false;
            *n_reset = (uint32_t)m;
            return true;
        }
        n = top;
    }
    return false;
}

bool find_reset(float ratio, uint32_t n_start, uint64_t max_scan, uint32_t *n_reset)
{
    const uint64_t to_wrap = (1ULL << 32) - (uint64_t)n_start;
    const uint64_t span = std::min(max_scan, to_wrap);
    return first_reset_exact(ratio, n_start, (uint64_t)n_start + span, n_reset);
}

// first reset from counter 1 on, remembered per ratio (bit pattern): the period of every steady stretch with that ratio.
// Only candidates below `limit` are ever scanned (a caller never needs to know about resets beyond its own samples);
// what has been scanned without finding one is remembered too.  Returns 0 if there is no reset in [1, limit).
// scan one entry further (no map access: what the worker threads of prefetch() run on entries they own)
static uint32_t scan_entry(PeriodCache::Entry &e, float ratio, uint64_t limit)
{
    if (e.period != 0) return e.period < limit ? e.period : 0;
    limit = std::min<uint64_t>(limit, 1ULL << 32);
    if (e.scanned_to >= limit) return 0;
    uint32_t p1 = 0;
    if (find_reset(ratio, (uint32_t)e.scanned_to, limit - e.scanned_to, &p1)) {
        e.period = p1;
        return p1;
    }
    e.scanned_to = limit;
    return 0;
}

// The cache is bounded: a live `doppler track` re-evaluates the shift for every 8 KiB block, so almost every block
// brings a new ratio (hundreds per second) and a context lives for days.  An entry only has to survive while its plan
// is being built, so a full cache is simply emptied (the next lookups scan again: microseconds to milliseconds each).
void PeriodCache::make_room(size_t incoming)
{
    if (first_reset.size() + incoming > kMaxEntries) first_reset.clear();
}

uint32_t PeriodCache::period(float ratio, uint64_t limit)
{
    uint32_t bits;
    memcpy(&bits, &ratio, sizeof bits);
    auto it = first_reset.find(bits);
    if (it == first_reset.end()) {
        make_room(1);
        it = first_reset.emplace(bits, Entry()).first;      // {0, 1}: nothing found, scanned up to (excluding) 1
    }
    return scan_entry(it->second, ratio, limit);
}

// Periods of many ratios at once.  Rounds 3-5 scanned them on up to 16 host threads before the sequential pass (a period was
// 10^4..10^6 candidates tried one by one); with the closed form a period costs ~0.4 us, less than handing it to a thread,
// so this only makes room and the sequential pass finds each period on demand.
void PeriodCache::prefetch(const float *ratios, const uint64_t *counts, size_t n)
{
    (void)ratios;
    (void)counts;
    if (n <= kMaxEntries) make_room(n);
}

static uint32_t lut_len_for(uint32_t period, uint64_t count, int variant)
{
    if (period < 4) return period;                 // only the table path handles tiny periods
    if (variant == 1) return 0;
    if (period > kLutMaxEntries) return 0;
    if (variant == 2) return period;
    return count >= 2ull * period ? period : 0;    // a table pays once it is reused
}

// bits of |theta(n)| as the kernels compute it (dpx_sincos.h, corrector): two separately rounded f32 products
static uint32_t theta_abs_bits(float ratio, uint32_t n)
{
    const float p = ratio * (float)n;
    const float theta = -6.28318530717958647692f * p;
    uint32_t b;
    memcpy(&b, &theta, sizeof b);
    return b & 0x7fffffffu;
}

// first counter whose |theta| bits are >= bound (0xffffffff: none below 2^32 - 1); |theta| is monotone in n.
// An estimate from the bound's value, then stepped to the exact boundary (1-3 evaluations; rounds 2-5 bisected: 32).
uint32_t first_counter_reaching_bisect(float ratio, uint32_t bound)       // rounds 2-5's form: what the estimate-and-step form is held against
{
    if (theta_abs_bits(ratio, 0xfffffffeu) < bound) return 0xffffffffu;
    uint32_t lo = 0, hi = 0xfffffffeu;                     // invariant: bits(hi) >= bound
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (theta_abs_bits(ratio, mid) >= bound) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

uint32_t first_counter_reaching(float ratio, uint32_t bound)
{
    if (theta_abs_bits(ratio, 0xfffffffeu) < bound) return 0xffffffffu;
    if (theta_abs_bits(ratio, 0) >= bound) return 0;
    float bv;
    memcpy(&bv, &bound, sizeof bv);
    const double est = (double)bv / (6.283185307179586 * fabs((double)ratio));
    uint32_t n = est >= 4294967294.0 ? 0xfffffffeu : est < 1.0 ? 1u : (uint32_t)est;
    // |theta(n)| >= bound from the boundary on: walk down while the predecessor still reaches it, up while n does not
    uint32_t steps = 0;
    while (n > 0 && theta_abs_bits(ratio, n - 1) >= bound && ++steps < 64) --n;
    while (theta_abs_bits(ratio, n) < bound && ++steps < 64) ++n;
    if (steps >= 64) return first_counter_reaching_bisect(ratio, bound);      // an estimate far off (counters beyond 2^24 round)
    return n;
}

static void emit(PlanResult &plan, uint64_t first, uint64_t count, float ratio, uint32_t n_start,
                 uint32_t period, uint32_t lut_len)
{
    DevSeg s;
    if (!plan.segs.empty() && memcmp(&plan.segs.back().ratio, &ratio, sizeof ratio) == 0) {
        s.n_plain = plan.segs.back().n_plain;
        s.n_large = plan.segs.back().n_large;
        s.n_huge = plan.segs.back().n_huge;
    } else {
        s.n_plain = first_counter_reaching(ratio, kThetaPlain);     // 2^-12
        s.n_large = first_counter_reaching(ratio, kThetaLarge);     // 120
        s.n_huge = first_counter_reaching(ratio, kThetaHuge);       // 2^29 (dpx_sincos.h, kLargeQuickEnd)
    }
    s.pad = 0;
    s.first = first;
    s.count = count;
    s.ratio = ratio;
    s.n_start = n_start;
    s.period = period;
    s.lut_len = lut_len;
    s.lut_off = 0;   // assigned by finalize()
    s.c0 = 0;
    s.tmod = 0;
    s.flags = 0;
    plan.segs.push_back(s);
}

void plan_append(PlanResult &plan, float ratio, uint64_t count, uint32_t &samplenum, int variant, PeriodCache *cache)
{
    PeriodCache local;
    if (!cache) cache = &local;
    uint64_t pos = plan.n_samples;
    uint64_t remaining = count;
    uint32_t n = samplenum;
    while (remaining > 0) {
        // P = first reset from 1 (one scan per ratio).  From a counter in [1, P] the next reset is P itself and the
        // counter then cycles 1..P: no scan at all.  Otherwise (counter 0, or carried over from another ratio beyond
        // P) the counter runs linearly to its next reset, which is found by scanning from it.
        const uint32_t P = n >= 1 ? cache->period(ratio, (uint64_t)n + remaining) : 0;
        if (P != 0 && n <= P) {
            emit(plan, pos, remaining, ratio, n, P, lut_len_for(P, remaining, variant));
            n = (uint32_t)(((uint64_t)(n - 1u) + remaining) % P) + 1u;
            pos += remaining;
            remaining = 0;
            break;
        }
        const uint64_t to_wrap = (1ULL << 32) - (uint64_t)n;
        const uint64_t span = std::min(remaining, to_wrap);
        uint32_t n1 = 0;
        const bool none_ahead = P == 0 && n >= 1;        // no reset in [1, n + remaining): none among this call's counters
        if (none_ahead || !find_reset(ratio, n, span, &n1)) {
            // no reset among the next `span` counter values: n = n_start + j
            emit(plan, pos, span, ratio, n, 0, 0);
            pos += span;
            remaining -= span;
            n = (uint32_t)((uint64_t)n + span);     // u32 `+= 1` wraps to 0 after 2^32-1
            continue;
        }
        // lead-in: linear up to n1; the sample that uses n1 resets the counter to 1
        const uint64_t len = (uint64_t)n1 - n + 1;
        emit(plan, pos, len, ratio, n, 0, 0);
        pos += len;
        remaining -= len;
        n = 1;
    }
    plan.n_samples = pos;
    samplenum = n;
    plan.final_samplenum = n;
}

uint32_t counter_at(const DevSeg &s, uint64_t j)
{
    if (s.period == 0) return s.n_start + (uint32_t)j;
    return (uint32_t)(((uint64_t)(s.n_start - 1u) + j) % s.period) + 1u;
}

namespace {

constexpr uint64_t kRowsMinSamples = 1u << 16;   // below this a stretch stays on the tile kernel
constexpr uint64_t kAbsorbMax = 65536;           // neighbouring crumbs and tail a rows launch evaluates itself, sample by sample
constexpr size_t kRowsMaxLaunches = 8;           // more tabulated stretches than this: one tile launch instead

uint64_t lcm_u64(uint64_t a, uint64_t b)
{
    uint64_t g = a, h = b;
    while (h) { const uint64_t t = g % h; g = h; h = t; }
    return a / g * b;
}

// Row length for a stretch of period P with `body` samples available from the matrix origin and R rows per
// wavefront.  L must be a multiple of P (a column then always sees the same corrector) and of 4 (16-byte rows);
// the table is one period whatever L is, so L is chosen for the memory system alone, scored from measurements
// on MI355X (profiles/r01_membench.md section 4, i16 stream, relative to the best case):
//   rows that do not start on a 128-byte line (L % 32 != 0): -20 % (a wavefront's 1 KiB piece then shares
//     lines with wavefronts running on other XCDs);
//   rows that are not a whole number of 4 KiB pages (L % 1024 != 0): about -3 %;
//   rows shorter than 8192 samples or longer than 16384: -2 % / -3 % (the two rows of a wavefront are then 4-16 KiB
//     or more than 128 KiB apart);
//   idle lanes in the last 256-sample column slice: proportional.
// At least 16 row groups must fit.  Returns 0 if no such L exists.
uint32_t pick_row_length(uint32_t P, uint64_t body, uint32_t R, const PlanTuning &tn)
{
    uint64_t cap = std::min<uint64_t>(kRowsMaxL, body / (16ull * R));
    if (tn.rows_maxl) cap = std::min<uint64_t>(cap, tn.rows_maxl);            // measurement override
    const uint64_t steps[3] = {lcm_u64(P, 1024), lcm_u64(P, 32), lcm_u64(P, 4)};
    if (tn.rows_mult) {                                                        // measurement override: L = mult * lcm(P, 4)
        const uint64_t L = steps[2] * (uint64_t)tn.rows_mult;
        return (L >= steps[2] && L <= cap) ? (uint32_t)L : 0;
    }
    uint32_t best = 0;
    double best_score = -1e9;
    for (uint64_t step : steps) {
        int tried = 0;
        for (uint64_t L = step; L <= cap && (L == step || L <= kRowsMultMaxL) && tried < 64; L += step, ++tried) {
            double score = (double)L / (256.0 * (double)((L + 255) / 256));
            if (L % 32 != 0) score -= 0.20;
            if (L % 1024 != 0) score -= 0.03;
            // the row length itself (headline stream, all four format pairs, several boxes): 8192 samples is the best,
            // 16384 within 1 %, 1024-4096 cost 1.5-3 %, 32768 and more 3-5 %
            if (L < 8192) score -= 0.02;
            else if (L > 16384) score -= 0.03;
            else if (L > 8192) score -= 0.005;
            if (score > best_score + 1e-9) { best_score = score; best = (uint32_t)L; }
        }
    }
    return best;
}

// rows per wavefront, and table or evaluation.
// Rows of 8192 or 16384 samples (the two rows of a wavefront 32 or 64 KiB apart) run 3 points faster than any other
// length (83.7 / 84.3 % against 80.1-81.7 % for L = 4096 ... 131072 with the headline's own table), which needs a period
// that divides 16384.  Every other period pays, on top, for its table: one period is read by every wavefront of the launch,
// 8 bytes per R samples beside the 8-16 bytes of stream a sample moves.  From kRowsComputeMinP on a launch lets every
// wavefront evaluate its columns' correctors itself, once for its rows, while the rows' loads are in flight.
// Round 5: with the sincos at 22-24 instructions (dpx_sincos.h) that is TWO rows per wavefront and every format pair —
// one box, same process (profiles/raw/r05_ab_rows_eval.log), table under 4 rows (round 4's default) / evaluation under 4 /
// under 2:   3 Hz (P = 1 024 000): i16->i16 71.3 / 76.5 / 80.2 %, f32->f32 71.5 / 72.7 / 80.4, i16->f32 70.5 / 72.0 / 80.3,
// f32->i16 75.2 / 75.2 / 82.3;   100 Hz (P = 10 240): i16->i16 - / 80.2 / 82.7, f32->f32 79.9 / 78.7 / 81.6, i16->f32 77.7 /
// 78.5 / 80.5, f32->i16 82.6 / 83.5 / 85.6;   9876.543 Hz (P = 2592): 80.5 -> 83.8.  (Rounds 2-4, sincos at 31-52
// instructions: four rows, i16 -> i16 only.)  Short periods keep their table: P = 480 at large angles 81.5 % against 76.7
// evaluated; the headline (P = 1024, an 8 KiB table, rows of 8 periods) stays on its table as well — evaluation measured
// +1 point there on a warm chip, but a kernel whose rate does not depend on the shader clock is the safer headline.
// The launch still carries both — the table and (ratio, idx0) — so that `rows_compute` can A/B them.
constexpr uint32_t kRowsComputeMinP = 2049;

// 0: the launch reads its table; 2: every wavefront evaluates its columns' correctors
uint32_t rows_compute(uint32_t P, const PlanTuning &tn)
{
    if (P < 4) return 0;
    if (tn.rows_compute == 1) return 2;
    return P >= (tn.rows_compute ? tn.rows_compute : kRowsComputeMinP) ? 2 : 0;
}

uint32_t pick_rows_per_wave(uint32_t P, const PlanTuning &tn, uint32_t compute)
{
    // tables: 2 rows while one period of correctors stays L1/L2-hot, 4 for large tables (measured, GB/s at R = 2 / 4 / 8:
    // 8 KB table 6615 / 6264 / 6194; 876 KB table 5406 / 5660 / 5777); evaluation: 2 (above)
    uint32_t R = ((uint64_t)P + 3) * 8 <= (128u << 10) ? 2 : 4;
    if (compute) R = 2;
    if (tn.rows_r) R = tn.rows_r;                          // measurement override
    return (R == 2 || R == 4) ? R : 2;
}

struct Interval { uint64_t lo, hi; };

constexpr uint64_t kWalkTileMinDefault = 1ull << 22;   // an uncovered gap at least this long gets its own tile launch ...
constexpr size_t kWalkMaxTileLaunches = 8;       // ... up to this many; the rest is evaluated by leftover workgroups

struct WalkShape { uint32_t waves, span; bool fixed; };
// waves: wavefronts per workgroup; span: most rows per span; fixed: the caller named a span height (measurement): spans of
// exactly that cut, one window per workgroup, two rows per wavefront per turn
WalkShape walk_shape(const PlanTuning &tn)
{
    WalkShape g = {kSpanWaves, kSpanRows, false};
    if (tn.walk_waves != 0 && walk_waves_ok(tn.walk_waves, false)) g.waves = tn.walk_waves;
    if (tn.walk_span >= 2) { g.span = tn.walk_span; g.fixed = true; }
    return g;
}

// One span of a matrix: `h` rows under `waves` wavefronts.
//   h <= 5 (waves = 4): the workgroup takes 2 (h <= 3: 4) adjacent windows, 2 (1) wavefronts each — otherwise half
//     (three quarters) of its wavefronts would hold no row.  Measured on the replay's seconds of 3-4 rows: 72 -> 78 % under
//     two wavefronts per window, 0-2 rows: 54 -> 62 % (profiles/r04_walk.md).  In general: the largest 2^s <= 4 that divides
//     the wavefronts and leaves (waves >> s) x 2 + 1 >= h — one row more than a turn holds (one wavefront of the window
//     takes a second turn) is still better than the next larger workgroup share: 3 rows under 1 wavefront per window 70 %
//     against 58 under 2, 5 rows under 2 per window 78 against 71 under 4; from 6 rows on 4 per window wins (79 against 73-75).
struct SpanShape { uint32_t wshift, upw; };
SpanShape span_shape(uint32_t h, const WalkShape &g)
{
    SpanShape sh = {0, 2};
    if (g.fixed) return sh;
    for (uint32_t s2 = kSpanMaxShift; s2 > 0; --s2) {
        if (g.waves % (1u << s2) == 0 && (g.waves >> s2) * 2 + 1 >= h) { sh.wshift = s2; break; }
    }
    return sh;
}

// Row length of a walk / span matrix: a multiple of the period.  Measured (profiles/r03_walk.md, `tools/ab.py --set minl`,
// replay classes): rows of 0.5-1.3 MB run 1-4 points faster than rows of one period of 10-100 thousand samples (fewer,
// longer fronts in flight), so a long stretch takes the multiple that reaches kWalkRowTarget; a short one (a second of
// stream) stops where its rows still fill one span of 8, and keeps the period itself while that gives no more than 12
// rows (one span).  A rule that scored every multiple against the pitch bands of `tools/rowbench pitch3` (70-76 % around
// 1 and 2 MiB between the rows of a workgroup, 80-84 % elsewhere) was tried and measured no better on the real kernels
// (`tools/ab.py --set rowrule`): the bands of a bare copy are not the bands of rows shifted against their lines.
uint64_t walk_row_length_target(uint64_t P, uint64_t target, uint64_t len)
{
    const uint64_t r = len / P;
    uint64_t m = 1;
    if (r > 12) m = std::min((target + P - 1) / P, (r + 7) / 8);     // (rows of two periods from 9 rows on: measured equal, profiles/r04_walk.md)
    if (P * m < kWalkMinL) m = (kWalkMinL + P - 1) / P;
    return P * m;
}

// matrix of the span kernel for one stretch (dpx_types.h, WalkSeg); false if the stretch does not qualify
bool walk_geometry(const DevSeg &s, WalkSeg *w, const PlanTuning &tn)
{
    if (s.lut_len == 0 || s.period == 0) return false;
    const uint64_t end = s.first + s.count;
    const uint64_t A = (s.first + 31) & ~31ull, E = end & ~31ull;
    if (E <= A) return false;
    const uint64_t P = s.period;
    const uint64_t L = walk_row_length_target(P, (tn.walk_flags >> 8) ? (uint64_t)(tn.walk_flags >> 8) * 1024u : kWalkRowTarget, E - A);   // (bits 8.. of walk_flags: measurement)
    if (L > kLutMaxEntries || E - A < 2 * L) return false;     // the table must be reused at least once
    const uint64_t rows = (E - A + L - 1) / L;
    const uint64_t longest = (L % 32 == 0) ? L : (L & ~31ull) + 32;   // rows start on 32-sample boundaries
    const uint64_t nw = (longest + kWalkWindow - 1) / kWalkWindow;
    if (rows > 0xffffffffull) return false;
    w->A = A;
    w->E = E;
    w->L = (uint32_t)L;
    w->wshift = 0;
    w->wg_base = 0;
    w->nw = (uint32_t)nw;
    w->rows = (uint32_t)rows;
    w->row0 = 0;
    w->period = s.period;
    w->phase = counter_at(s, A - s.first) - 1u;
    w->ratio = s.ratio;
    w->upw = 0;
    w->row_end = 0;
    w->nwg = 0;
    return true;
}

// spans of a matrix: one for up to kSpanWhole rows (every one-second matrix under the row-length rule above: its slice is
// evaluated once; replay 76.3 -> 78.0 % against spans of 8, measured), spans of 8 = one turn of 4 x 2 each for a taller
// one (const mode: 80.1 % with 8, 79.2 with 12, 77.5 with 16)
uint32_t walk_chunks(const WalkSeg &w, const PlanTuning &tn)
{
    const WalkShape g = walk_shape(tn);
    if (!g.fixed && w.rows <= kSpanWhole) return 1;
    return (uint32_t)(((uint64_t)w.rows + g.span - 1) / g.span);
}

// span c of k: rows [row0, row0 + h)
inline void span_rows(const WalkSeg &w, uint32_t k, uint32_t c, uint32_t *row0, uint32_t *h)
{
    const uint32_t base = w.rows / k, rem = w.rows % k;
    *row0 = c * base + std::min(c, rem);
    *h = base + (c < rem ? 1u : 0u);
}

// workgroups of a matrix in the descriptor list: every span padded to a multiple of 8
uint64_t walk_workgroups(const WalkSeg &w, const PlanTuning &tn)
{
    const WalkShape g = walk_shape(tn);
    const uint32_t k = walk_chunks(w, tn);
    uint64_t n = 0;
    for (uint32_t c = 0; c < k; ++c) {
        uint32_t row0, h;
        span_rows(w, k, c, &row0, &h);
        const uint32_t nwg = (w.nw + (1u << span_shape(h, g).wshift) - 1) >> span_shape(h, g).wshift;
        n += (nwg + 7) & ~7u;
    }
    return n;
}

}  // namespace

// What one launch makes of a plan's span shape for its format pair (dpx_types.h, SpanLaunch).
//
// One matrix (const mode).  Its spans follow from the kernel arguments, so the LAUNCH may cut them per pair — and the rule
// is about the memory side, not about arithmetic (profiles/r04_pairs_pmc.md: request mix, vector-ALU load and LDS are the same
// for every shape; what differs is how long requests queue at the L2's memory port): a workgroup should keep about 8 KiB
// of its column window in flight per direction.  A row vector (256 samples) is 1 KiB of i16 or 2 KiB of f32, so a span is
//     8 KiB / (the wider side's row vector)  =  8 rows for i16 -> i16,  4 rows for every pair with an f32 side
// (f32 -> f32: DRAM credit stalls per request 0.32 -> 0.06, requests in flight 3788 -> 3306 at 83 % instead of 78 %;
// i16 -> i16 with spans of 4 only doubles the slice arithmetic: 75.8 % against 79.6).  With so few rows the wavefronts are
// there for the slice: 4 for f32 -> f32 (half windows: 160 entries), 5 for f32 -> i16 (whole windows of 288 entries and the
// LDS transposition), 2 for i16 -> f32 (half windows, 8-byte loads: 79.4 % against 69.9 under 4).  Measured on four shifts,
// two processes each, in round 3 (`tools/ab.py --set pairs3 / pairs4`).  Not when the caller fixed a shape (auto_shape == 0).
// Many matrices (track mode).  The descriptors fix the spans, not the workgroup size: f32 -> i16 (two 16-byte loads per
// lane per row) runs its replays 1.5-2 points faster under 8 wavefronts (78.4 against 76.4 %, this round); the other pairs
// lose under more than 4 (i16 -> f32 74.6 -> 72.6).  The planner's window shifts divide 4, hence 8.
bool walk_waves_ok(uint32_t waves, bool uni)
{
    return waves == 2 || waves == 4 || waves == 5 || (waves == 8 && !uni);
}

const std::vector<Launch> &launches_for(const PlanResult &plan, int in_fmt, int out_fmt)
{
    return (in_fmt != out_fmt && !plan.whole_tiles.empty()) ? plan.whole_tiles : plan.launches;
}

bool span_launch_shape(const WalkArgs &w, int in_fmt, int out_fmt, SpanLaunch *o)
{
    const bool in_f32 = in_fmt == 1, out_f32 = out_fmt == 1;
    o->uni = w.uni;
    o->waves = w.waves;
    const bool uni = w.uni.n_spans != 0;
    o->left_rows = uni ? (w.n_left_wg + w.uni.nw8 - 1) / w.uni.nw8 : 0;
    constexpr uint32_t kWindowBytesInFlight = 8192;
    const uint32_t row_vector = kWalkWindow * std::max(in_f32 ? 8u : 4u, out_f32 ? 8u : 4u);
    const uint32_t span_rows = kWindowBytesInFlight / row_vector;                 // 8 or 4
    if (uni && w.auto_shape && span_rows < kSpanRows && w.uni.seg.rows > kSpanWhole) {
        const uint32_t k = (w.uni.seg.rows + span_rows - 1) / span_rows;
        if ((uint64_t)k + o->left_rows <= 65535u) {
            o->uni.n_spans = k;
            o->uni.base = w.uni.seg.rows / k;
            o->uni.rem = w.uni.seg.rows % k;
            o->waves = !out_f32 ? 5 : in_f32 ? 4 : 2;
        }
    }
    if (!uni && w.auto_shape && in_f32 && !out_f32) o->waves = 8;
    if (uni && (uint64_t)o->uni.n_spans + o->left_rows > 65535u) return false;
    return walk_waves_ok(o->waves, uni);
}

void finalize(PlanResult &plan, uint32_t tile, int choice, const PlanTuning &tn)
{
    const uint64_t kWalkTileMin = tn.walk_tilemin ? tn.walk_tilemin : kWalkTileMinDefault;
    plan.tile = tile;
    plan.error = nullptr;
    plan.tables.clear();
    plan.launches.clear();
    plan.whole_tiles.clear();
    plan.walk.clear();
    plan.walk_hint.clear();
    plan.left.clear();
    plan.left_hint.clear();
    const size_t ns = plan.segs.size();
    uint64_t pool = 0;
    for (DevSeg &s : plan.segs) s.flags = 0;
    plan.tile_tables = false;

    // ---- which kernel serves the tabulated stretches.
    // rows kernel: one launch per stretch, the fastest shape for a long stretch (const mode);
    // walk kernel: any number of stretches in one launch (track mode: one stretch per second of stream);
    // tile kernel: whatever is left, and everything when neither of the above applies.
    auto rows_geometry = [&](const DevSeg &s, uint64_t *A, uint32_t *R, uint32_t *L, uint32_t *compute = nullptr) {
        if (s.lut_len == 0 || s.count < kRowsMinSamples) return false;
        const uint64_t end = s.first + s.count;
        // matrix origin on a 256-sample boundary: 1 KiB of i16 / 2 KiB of f32 per wavefront, aligned
        *A = (s.first + 255) & ~255ull;
        if (*A >= end) return false;
        uint32_t comp = rows_compute(s.period, tn);
        *R = pick_rows_per_wave(s.period, tn, comp);
        *L = pick_row_length(s.period, end - *A, *R, tn);
        if (comp != 0 && tn.rows_compute != 1 && *L != 0 && *L % 8192 == 0 && *L <= 16384) {       // the fast row lengths: table, two rows
            comp = 0;
            *R = pick_rows_per_wave(s.period, tn, 0);
            *L = pick_row_length(s.period, end - *A, *R, tn);
        }
        if (compute) *compute = comp;
        return *L != 0;
    };
    bool use_rows = false, use_walk = false;
    if (choice == kChooseAuto || choice == kChooseRows) {
        // a plan of a few long tabulated stretches (const mode) is a rows plan; one of many (track mode: a stretch per
        // second of stream) belongs to the walk kernel whatever rows geometry its stretches would allow
        size_t eligible = 0, long_tabulated = 0;
        for (const DevSeg &s : plan.segs) {
            uint64_t A;
            uint32_t R, L;
            if (rows_geometry(s, &A, &R, &L)) ++eligible;
            if (s.lut_len != 0 && s.count >= kRowsMinSamples) ++long_tabulated;
        }
        use_rows = eligible > 0 && long_tabulated <= kRowsMaxLaunches;
    }
    if ((choice == kChooseAuto && !use_rows) || choice == kChooseWalk) {
        uint64_t in_matrices = 0, n_wg = 0;
        for (const DevSeg &s : plan.segs) {
            WalkSeg w;
            if (!walk_geometry(s, &w, tn)) continue;
            in_matrices += w.E - w.A;
            n_wg += walk_workgroups(w, tn);
        }
        // worth it when most of the stream is in matrices (the rest is evaluated sample by sample)
        use_walk = in_matrices > 0 && in_matrices >= plan.n_samples / 2 && n_wg < 0x40000000ull;
    }

    const PlanTuning &tnw = tn;

    // Const-mode plans: a stretch whose period does not allow rows of whole 4 KiB pages (odd periods: the common case
    // for an arbitrary integer --shift) runs 4-23 % faster as a walk-kernel matrix than as a rows launch with
    // L = 32 P (measured, profiles/r01_secondary_workloads.md), and a stretch too short for a rows launch is better off
    // there than on the tile kernel; page-aligned rows stay on the rows kernel.
    std::vector<uint8_t> to_walk(ns, 0);
    if (use_rows && choice == kChooseAuto) {
        for (size_t i = 0; i < ns; ++i) {
            uint64_t A;
            uint32_t R, L;
            WalkSeg w;
            const bool page_rows = rows_geometry(plan.segs[i], &A, &R, &L) && L % 1024 == 0;
            if (!page_rows && walk_geometry(plan.segs[i], &w, tn)) {
                to_walk[i] = 1;
                plan.segs[i].flags |= kSegWalk;
            }
        }
    }

    std::vector<Interval> covered;                   // what rows launches / walk matrices produce, in stream order
    std::vector<uint64_t> seg_covered_hi(ns, 0);     // per stretch: end of the part a rows launch covers (0 = none)
    for (size_t i = 0; use_rows && i < ns; ++i) {
        DevSeg &s = plan.segs[i];
        uint64_t A;
        uint32_t R, L;
        uint32_t comp = 0;
        if (to_walk[i] || !rows_geometry(s, &A, &R, &L, &comp)) continue;
        const uint64_t end = s.first + s.count;
        const uint64_t n_rg = (end - A) / ((uint64_t)R * L);
        s.flags |= kSegRows;
        Launch ln;
        ln.kind = 0;
        ln.rows.A = A;
        ln.rows.B = A + n_rg * R * L;
        ln.rows.R = R;
        ln.rows.P = s.period;
        ln.rows.n_rg = n_rg;
        ln.rows.L = L;
        ln.rows.r0 = s.first;
        // a short tail is evaluated sample by sample by the launch's extra workgroups; a long one (up to R rows)
        // is left to a tile-kernel launch
        ln.rows.r1 = (end - ln.rows.B <= kAbsorbMax) ? end : ln.rows.B;
        ln.rows.seg_lo = (uint32_t)i;
        ln.rows.n_segs = (uint32_t)ns;
        ln.rows.compute = comp;
        ln.rows.idx0 = counter_at(s, A - s.first) - 1u;
        ln.rows.ratio = s.ratio;
        ln.rows.pad = 0;
        // table: one period (+3 so that four consecutive entries never wrap), origin = sample A
        ln.rows.tab_off = (uint32_t)pool;
        plan.tables.push_back({pool, s.period, counter_at(s, A - s.first), s.period + 3, s.ratio});
        pool += ((uint64_t)s.period + 3 + 3) & ~3ull;
        plan.launches.push_back(ln);
    }
    // a rows launch also evaluates small neighbouring crumbs (e.g. the one-sample lead-in of a
    // stream that starts at counter 0), so that such a plan is a single launch.  Launches are in
    // stretch order, so "not below the previous launch's r1" keeps the ranges disjoint.
    uint64_t covered_hi = 0;
    for (Launch &ln : plan.launches) {
        const uint32_t own = ln.rows.seg_lo;
        uint32_t lo = own;
        while (lo > 0) {
            const DevSeg &p = plan.segs[lo - 1];
            if ((p.flags & (kSegRows | kSegWalk)) || p.first < covered_hi || ln.rows.A - p.first > kAbsorbMax) break;
            --lo;
        }
        ln.rows.seg_lo = lo;
        ln.rows.r0 = plan.segs[lo].first;
        const uint64_t own_end = plan.segs[own].first + plan.segs[own].count;
        if (ln.rows.r1 == own_end) {                 // the whole tail is ours: following crumbs may join
            uint32_t hi = own + 1;
            while (hi < ns) {
                const DevSeg &q = plan.segs[hi];
                if ((q.flags & (kSegRows | kSegWalk)) || q.first + q.count - ln.rows.B > kAbsorbMax) break;
                ln.rows.r1 = q.first + q.count;
                ++hi;
            }
        }
        seg_covered_hi[own] = std::min(ln.rows.r1, own_end);
        covered_hi = ln.rows.r1;
        covered.push_back({ln.rows.r0, ln.rows.r1});
    }

    // ---- walk matrices, and the pieces of the stream they leave out
    struct Piece { uint64_t lo, hi; uint32_t seg; };
    std::vector<Piece> pieces;
    bool any_to_walk = false;
    for (size_t i = 0; i < ns; ++i) any_to_walk = any_to_walk || to_walk[i];
    if (use_walk || any_to_walk) {
        std::vector<WalkSeg> mats;                   // one per stretch that becomes a matrix, in stream order
        const std::vector<Interval> rows_cov = covered;        // rows launches, disjoint, in stream order
        size_t rc = 0;
        // the parts of [lo, hi) that no rows launch produces
        auto uncovered = [&](uint64_t lo, uint64_t hi, uint32_t seg) {
            while (rc < rows_cov.size() && rows_cov[rc].hi <= lo) ++rc;
            uint64_t pos2 = lo;
            for (size_t k = rc; k < rows_cov.size() && rows_cov[k].lo < hi; ++k) {
                if (rows_cov[k].lo > pos2) pieces.push_back({pos2, rows_cov[k].lo, seg});
                pos2 = std::max(pos2, rows_cov[k].hi);
            }
            if (pos2 < hi) pieces.push_back({pos2, hi, seg});
        };
        for (size_t i = 0; i < ns; ++i) {
            DevSeg &s = plan.segs[i];
            const uint64_t end = s.first + s.count;
            WalkSeg w;
            if (!(use_walk || to_walk[i]) || !walk_geometry(s, &w, tn)) {
                s.flags &= ~kSegWalk;
                uncovered(s.first, end, (uint32_t)i);
                continue;
            }
            s.flags |= kSegWalk;
            mats.push_back(w);
            if (s.first < w.A) uncovered(s.first, w.A, (uint32_t)i);
            if (w.E < end) uncovered(w.E, end, (uint32_t)i);
            covered.push_back({w.A, w.E});
        }
        // uncovered pieces that touch form a gap; long gaps become tile launches, the rest leftover ranges
        std::vector<Interval> gaps_for_tiles;
        uint64_t left_wg = 0;
        for (size_t i = 0; i < pieces.size();) {
            size_t j = i + 1;
            while (j < pieces.size() && pieces[j].lo == pieces[j - 1].hi) ++j;
            const uint64_t lo = pieces[i].lo, hi = pieces[j - 1].hi;
            if (hi - lo >= kWalkTileMin && gaps_for_tiles.size() < kWalkMaxTileLaunches) {
                gaps_for_tiles.push_back({lo, hi});
            } else {
                for (size_t k = i; k < j; ++k) {
                    for (uint64_t p0 = pieces[k].lo; p0 < pieces[k].hi;) {
                        const uint64_t len = std::min<uint64_t>(pieces[k].hi - p0, 1u << 30);
                        plan.left.push_back({p0, (uint32_t)len, pieces[k].seg, (uint32_t)left_wg, 0});
                        left_wg += (len + kLeftBlock - 1) / kLeftBlock;
                        p0 += len;
                    }
                    covered.push_back({pieces[k].lo, pieces[k].hi});
                }
            }
            i = j;
        }
        // dispatch order: stretch by stretch, span by span, window fastest (a span sweeps its rows contiguously);
        // every span is padded to a multiple of 8 workgroups so that window w of every span runs on XCD w % 8.
        // The leftover workgroups (sincos per sample: VALU-bound) are dealt out in groups of 8 between the spans, evenly
        // over the grid, so that their arithmetic runs beside memory-bound matrix workgroups on every CU instead of in a
        // block of its own (all at the front: 5-8 % of the 600-second replay for 1 % of its samples; at the end: worse).
        const WalkShape shape = walk_shape(tnw);
        uint64_t m_groups = 0;
        for (const WalkSeg &m : mats) m_groups += walk_workgroups(m, tnw) / 8;
        const uint64_t l_groups = (left_wg + 7) / 8;
        uint64_t wg = 0, m_done = 0, l_done = 0;
        bool plain_spans = true;                            // every span: one window per workgroup, two rows per wavefront
        auto deal_leftovers = [&](uint64_t upto) {          // leftover groups [l_done, upto) go here
            for (; l_done < upto; ++l_done) {
                WalkSeg g;
                memset(&g, 0, sizeof g);
                g.wg_base = (uint32_t)wg;
                g.row0 = (uint32_t)(l_done * 8);                                        // first leftover block of the group
                g.nwg = (uint32_t)std::min<uint64_t>(8, left_wg - l_done * 8);
                g.upw = 0;                                                                // marks a leftover group
                wg += 8;
                plan.walk.push_back(g);
            }
        };
        for (size_t mi = 0; mi < mats.size(); ++mi) {
            const uint32_t k = walk_chunks(mats[mi], tnw);
            for (uint32_t c = 0; c < k; ++c) {
                WalkSeg w = mats[mi];
                uint32_t h;
                span_rows(w, k, c, &w.row0, &h);
                const SpanShape sh = span_shape(h, shape);
                w.row_end = w.row0 + h;
                w.upw = sh.upw;
                w.wshift = sh.wshift;
                w.nwg = (w.nw + (1u << sh.wshift) - 1) >> sh.wshift;
                w.wg_base = (uint32_t)wg;
                if (sh.wshift != 0 || sh.upw != 2) plain_spans = false;
                wg += (w.nwg + 7) & ~7u;
                plan.walk.push_back(w);
                m_done += (w.nwg + 7) / 8;
                deal_leftovers(m_groups ? l_groups * m_done / m_groups : 0);
            }
        }
        deal_leftovers(l_groups);
        if (wg > 0x7fffffffull) plan.error = "stream needs more than 2^31 workgroups";
        Launch ln;
        ln.kind = 2;
        ln.walk.n_walk_wg = (uint32_t)wg;            // the whole grid, leftover groups included
        ln.walk.n_left_wg = (uint32_t)left_wg;       // leftover blocks among them
        ln.walk.n_segs = (uint32_t)ns;
        ln.walk.waves = shape.waves;
        ln.walk.span = shape.span;
        memset(&ln.walk.uni, 0, sizeof ln.walk.uni);
        ln.walk.auto_shape = (!tn.walk_waves && !tn.walk_span) ? 1u : 0u;
        ln.walk.sub_lg = tn.sub_lg;
        ln.walk.cover = 0;
        for (const WalkSeg &m : mats) ln.walk.cover += m.E - m.A;
        for (const LeftRange &lr : plan.left) ln.walk.cover += lr.len;
        // one matrix: the kernel takes it from its arguments (no such kernel is built for 8 wavefronts: no pair's cut wants them)
        if (mats.size() == 1 && plain_spans && !(tn.walk_flags & 1u) && shape.waves != 8) {
            const uint32_t k = walk_chunks(mats[0], tnw);
            ln.walk.uni.seg = mats[0];
            ln.walk.uni.seg.upw = 2;
            ln.walk.uni.seg.nwg = mats[0].nw;
            ln.walk.uni.n_spans = k;
            ln.walk.uni.base = mats[0].rows / k;
            ln.walk.uni.rem = mats[0].rows % k;
            ln.walk.uni.nw8 = (mats[0].nw + 7) & ~7u;
            if ((uint64_t)k + (left_wg + ln.walk.uni.nw8 - 1) / ln.walk.uni.nw8 > 65535u) ln.walk.uni.n_spans = 0;   // grid rows
        }
        plan.launches.push_back(ln);
        // sentinels end the kernels' forward scans; hints give the scan its starting point
        WalkSeg wend;
        memset(&wend, 0, sizeof wend);
        wend.wg_base = 0xffffffffu;
        wend.upw = 2;
        plan.walk.push_back(wend);
        plan.left.push_back({0, 0, 0, 0xffffffffu, 0});
        const uint64_t n_wh = (wg >> kWalkHintShift) + 1;
        plan.walk_hint.assign(n_wh, 0);
        uint32_t wi = 0;
        for (uint64_t h = 0; h < n_wh; ++h) {
            while (wi + 2 < plan.walk.size() && plan.walk[wi + 1].wg_base <= (h << kWalkHintShift)) ++wi;
            plan.walk_hint[h] = wi;
        }
        const uint64_t n_lh = (left_wg >> kLeftHintShift) + 1;
        plan.left_hint.assign(n_lh, 0);
        uint32_t li = 0;
        for (uint64_t h = 0; h < n_lh; ++h) {
            while (li + 2 < plan.left.size() && plan.left[li + 1].wg_off <= (h << kLeftHintShift)) ++li;
            plan.left_hint[h] = li;
        }
    }

    // ---- everything nothing above covers goes to tile-kernel launches
    std::sort(covered.begin(), covered.end(), [](const Interval &a, const Interval &b) { return a.lo < b.lo; });
    std::vector<Interval> tile_ranges;
    uint64_t pos = 0;
    auto add_tiles = [&](uint64_t lo, uint64_t hi) {
        if (lo >= hi) return;
        Launch ln;
        ln.kind = 1;
        ln.tiles.m0 = lo;
        ln.tiles.m1 = hi;
        ln.tiles.tile_lo = lo / tile;
        ln.tiles.n_tiles = (hi + tile - 1) / tile - ln.tiles.tile_lo;
        ln.tiles.legacy = 0;
        ln.tiles.pad = 0;
        plan.launches.push_back(ln);
        tile_ranges.push_back({lo, hi});
    };
    for (const Interval &c : covered) {
        add_tiles(pos, c.lo);
        pos = std::max(pos, c.hi);
    }
    add_tiles(pos, plan.n_samples);
    // launches_for(): the alternative of a many-matrix plan for f32 -> i16 and i16 -> f32
    if (choice == kChooseAuto && plan.n_samples != 0) {
        bool many = false;
        for (const Launch &ln : plan.launches) many = many || (ln.kind == 2 && ln.walk.uni.n_spans == 0);
        if (many) {
            Launch ln;
            memset(&ln, 0, sizeof ln);
            ln.kind = 1;
            ln.tiles.m0 = 0;
            ln.tiles.m1 = plan.n_samples;
            ln.tiles.tile_lo = 0;
            ln.tiles.n_tiles = (plan.n_samples + tile - 1) / tile;
            plan.whole_tiles.push_back(ln);
        }
    }

    // ---- tile-kernel tables: every tabulated stretch with at least a tile's worth of samples in a tile launch
    const DevSeg *prev = nullptr;   // same (ratio, period) shares a table
    size_t tr = 0;
    for (size_t i = 0; i < ns; ++i) {
        DevSeg &s = plan.segs[i];
        if (s.lut_len == 0) continue;
        const uint64_t end = s.first + s.count;
        while (tr < tile_ranges.size() && tile_ranges[tr].hi <= s.first) ++tr;
        bool needs_tile_table = false;
        for (size_t k = tr; k < tile_ranges.size() && tile_ranges[k].lo < end; ++k) {
            const uint64_t lo = std::max(tile_ranges[k].lo, s.first), hi = std::min(tile_ranges[k].hi, end);
            if (hi > lo && hi - lo >= tile) needs_tile_table = true;
        }
        if (!needs_tile_table) continue;
        const uint32_t P = s.period;
        s.flags |= kSegTileTable;
        plan.tile_tables = true;
        s.c0 = (uint32_t)(((uint64_t)((s.n_start - 1u) % P) + P - (s.first % P)) % P);
        s.tmod = tile % P;
        if (prev && prev->period == P && memcmp(&prev->ratio, &s.ratio, sizeof(float)) == 0) {
            s.lut_off = prev->lut_off;
            prev = &s;
            continue;
        }
        prev = &s;
        s.lut_off = (uint32_t)pool;
        plan.tables.push_back({pool, P, 1u, P + tile, s.ratio});
        pool += ((uint64_t)P + tile + 3) & ~3ull;
    }
    plan.lut_entries = pool;
    if (pool > 0xffffffffull) plan.error = "corrector tables exceed 2^32 entries";
    if ((plan.n_samples + tile - 1) / tile > 0xffffffffull) plan.error = "stream longer than 2^32 tiles";

    // ---- one hint per 2^kHintShift samples: the stretch holding the first sample of that span
    const uint64_t n_hint = (plan.n_samples >> kHintShift) + 1;
    plan.hint.assign(n_hint, 0);
    uint32_t si = 0;
    for (uint64_t h = 0; h < n_hint; ++h) {
        const uint64_t g = h << kHintShift;
        while (si + 1 < ns && plan.segs[si].first + plan.segs[si].count <= g) ++si;
        plan.hint[h] = si;
    }
}

}  // namespace dpx
