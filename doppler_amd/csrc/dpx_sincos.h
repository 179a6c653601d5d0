// dpx_sincos.h — device sincosf that is bit-identical to glibc 2.35's.
//
// The reference builds its corrector with libm: src/dsp.rs:121-122 ->
// src/complex.c:35 cexpf(0 + i*theta), which for a zero real part is
// (cosf(theta), sinf(theta)) from glibc's sincosf (Szabolcs Nagy's
// optimized-routines algorithm: double-precision polynomial, three argument
// ranges).  "Within 1 ulp" of the output cannot be met with a different
// sincos (theta reaches tens to thousands of radians, and the complex multiply
// cancels), so this file evaluates the same double-precision operation
// sequence, with the same products fused as the x86-64 FMA build of libm fuses
// them (FMA=true) or none fused (FMA=false, the SSE2 build).
//
// The polynomial coefficients, the 2/pi and pi/2 constants, the 4/pi bit string
// and the 2^(i/32) table below are the published constants of that algorithm
// (glibc 2.35 sysdeps/ieee754/flt-32/{s_sincosf.h,s_sincosf_data.c,e_exp2f_data.c}):
//   Copyright (C) 2018-2022 Free Software Foundation, Inc. — GNU Lesser General
//   Public License 2.1 or later; originally ARM optimized-routines,
//   Copyright (c) 2018 Arm Ltd., MIT licence ("Permission is hereby granted, free of
//   charge, to any person obtaining a copy of this software ... to deal in the
//   Software without restriction ... THE SOFTWARE IS PROVIDED "AS IS", WITHOUT
//   WARRANTY OF ANY KIND").  Only the constants and the operation order are taken
//   over (they ARE the function being reproduced); the code around them is written
//   for 64-wide wavefronts.
//
// Structure for a 64-wide wavefront.  Every VALU instruction here costs the same
// four cycles per wavefront, f64 included (measured, profiles/r02_valubench.md), so
// the function is priced in instructions:
//   * fast path, taken when EVERY active lane has 2^-12 <= |y| < 120 (one ballot,
//     a wave-uniform branch): quadrant reduction (4 instructions: the rounding is a
//     fused multiply-add against 1.5 * 2^52) + the two polynomials and nothing
//     else — no range selects, no clamps;
//   * every active lane in [120, 2^29) — all a stream produces beyond 120, since
//     |theta| < 2^26: the remainder of x * 2/pi from a three-term 2/pi in double
//     precision (7 instructions; rounds 1-3: glibc's 32x96-bit integer product,
//     31 instructions).  Not glibc's operation sequence — its RESULT, proved over
//     the whole range by enumeration (reduce_large_quick below);
//   * general path otherwise.  |y| < 2^-12 ("tiny") and inf/nan are per-lane
//     selects; the "< pi/4" range goes through the quadrant-reduction formula,
//     which yields quadrant 0 and an unchanged argument there (n*hpi = 0), so it
//     is exactly the direct polynomial; |y| >= 120 uses the 32x96-bit fixed-point
//     product with 4/pi, whose three 32-bit windows are cut out of the bit string
//     with 64-bit register shifts, or read from the table for |y| >= 2^33.
// The two polynomial tables of glibc (cos / -cos) and the sign[] table are
// replaced by exact sign flips of the results (round-to-nearest is symmetric).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace dpx {

// 4/pi in overlapping 32-bit windows (glibc __inv_pio4), 192 bits
__device__ __constant__ const uint32_t kInvPio4[24] = {
    0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
    0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
    0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
    0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041,
};

template <bool FMA>
__device__ __forceinline__ double mad(double a, double b, double c)
{
    if constexpr (FMA) {
        return __builtin_fma(a, b, c);
    } else {
        double p = a * b;   // contract(off): two roundings
        return p + c;
    }
}

// quadrant reduction for |x| < 120 (glibc: ranges "< pi/4" and "< 120"): n = round(x * 2/pi), xr = x - n*pi/2.
//
// glibc rounds with a scaled truncating conversion, n = ((int32)(x * 2/pi * 2^24) + 2^23) >> 24.  Here (round 4) the
// rounding is ONE fused multiply-add against 1.5 * 2^52: the sum's low mantissa word IS n (two's complement), the double
// n comes back with one subtraction — four instructions instead of seven, no conversion.  The two roundings agree for
// every float below 120 except five negative arguments within 2^-24 of a quadrant boundary (glibc truncates toward zero
// before adding the half, so it rounds those ties-by-truncation toward +inf: -0x1.921fb6p-1, -0x1.921fb8p-1,
// -0x1.2d97c8p+1, -0x1.c463acp+2, -0x1.78fdbap+3), where the neighbouring quadrant with the mirrored remainder gives the
// same two floats.  That is a finite statement and it is checked, not argued: all 2^32 arguments, both libm builds, on
// the CPU model (tests/extended/sincos_model.c) and on the device (tests/extended/exhaustive_device_sincos.py): 0 mismatches.
constexpr double kTwoOverPi = 0x1.45F306DC9C883p-1;        // 2/pi rounded to double: glibc's hpi_inv / 2^24
constexpr double kRoundMagic = 0x1.8p52;                   // 1.5 * 2^52: ulp 1, room for |n| < 2^31
constexpr uint32_t kLargeQuickEnd = 0x4e000000u;           // 2^29: where the quick reductions' proofs end (DevSeg::n_huge)
template <bool FMA>
__device__ __forceinline__ double reduce_small(double x, uint32_t &n_out)
{
    constexpr double HPI = 0x1.921FB54442D18p0;         // pi/2
    const double pm = __builtin_fma(x, kTwoOverPi, kRoundMagic);
    n_out = (uint32_t)__double2loint(pm);
    const double nd = pm - kRoundMagic;
    if constexpr (FMA) return __builtin_fma(-nd, HPI, x);
    else               return x - nd * HPI;
}

// 120 <= |x| < 2^30 (round 4; since round 5 the SSE2 build's fast path only, and used below 2^29): the remainder of x * 2/pi from a three-term 2/pi in double precision instead of glibc's
// 32x96-bit integer product (reduce_large below: 31 instructions; this: 7, the conversion included).
//   n  = round(x * 2/pi)                 as above (|n| < 2^30: the magic sum is exact to the integer)
//   r  = x*c1 - n                        c1 = the leading 29 bits of 2/pi: 24 x 29 bits, the product and the difference exact
//   r += x*c2 ; r += x*c3                c2, c3 = the next 53 + 53 bits (fused: one rounding each, relative to r)
//   xr = r * pi/2                        the same double constant glibc multiplies its remainder by (pi63 = pi/2 * 2^-62)
// r differs from glibc's remainder (a 64-bit fixed-point value truncated below 2^-62 of a quadrant, then rounded to double)
// in the last place in about one argument of a hundred — and never enough to move either result across a float rounding
// boundary, nor to pick another quadrant: all 387 973 120 arguments of the range, both signs, both libm builds, give the
// floats of the integer path (tests/extended/sincos_model.c; without c3 two arguments differ).  Odd symmetry (round-to-nearest
// is symmetric) lets the signed x go through: quadrant -n and remainder -r reproduce glibc's "n + sign" bookkeeping,
// so quad = sidx = n exactly as in the small range.  Checked like the small range: exhaustively, CPU model and device.
__device__ __forceinline__ double reduce_large_quick(double x, uint32_t &n_out)
{
    constexpr double C1 = 0x1.45F306Dp-1, C2 = 0x1.9391054A7F09Dp-30, C3 = 0x1.7D1F534DDC0DBp-84;
    constexpr double HPI = 0x1.921FB54442D18p0;
    const double pm = __builtin_fma(x, kTwoOverPi, kRoundMagic);
    n_out = (uint32_t)__double2loint(pm);
    const double nd = pm - kRoundMagic;
    double r = __builtin_fma(x, C1, -nd);
    r = __builtin_fma(x, C2, r);
    r = __builtin_fma(x, C3, r);
    return r * HPI;
}

// quadrant signs and the sin/cos exchange on the two float results (glibc: argument * sign[sidx&3] with sign = {+,-,-,+};
// table[1] (sidx&2) is -cos; odd quadrants trade places): seven instructions.
__device__ __forceinline__ void sincos_signs(uint32_t fs, uint32_t fc, uint32_t quad, uint32_t sidx, float &rs, float &rc)
{
    // sign flips: fs ^= ((sidx + 1) & 2) << 30, fc ^= (sidx & 2) << 30 — one three-input bit operation each
    // (bitop3 0x6c = b ^ (a & c)); odd quadrant: the two trade places — a bit-field insert under an all-ones /
    // all-zeros mask, without a compare, its wait states and the condition register.  Spelled as instructions: the
    // compiler otherwise splits them into and / xor / or chains (12 instructions instead of 7).
    const uint32_t t = sidx << 30;                       // bit 31 = sidx & 2
    const uint32_t t1 = t + 0x40000000u;                 // bit 31 = (sidx + 1) & 2
    const uint32_t sign = 0x80000000u;
    uint32_t m;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6c" : "=v"(fs) : "v"(t1), "v"(fs), "s"(sign));
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6c" : "=v"(fc) : "v"(t), "v"(fc), "s"(sign));
    asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(m) : "v"(quad));                 // -1 if the quadrant is odd
    uint32_t rsb, rcb;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(rsb) : "v"(m), "v"(fc), "v"(fs));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(rcb) : "v"(m), "v"(fs), "v"(fc));
    rs = __uint_as_float(rsb);
    rc = __uint_as_float(rcb);
}

// the two polynomials on the reduced argument in glibc's own operation order (the general path, and the SSE2 build)
template <bool FMA>
__device__ __forceinline__ void sincos_poly(double xr, uint32_t quad, uint32_t sidx, float &rs, float &rc)
{
    constexpr double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
                     C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    constexpr double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7,
                     S3 = -0x1.994eb3774cf24p-13;
    const double x2 = xr * xr;
    const double x3 = x2 * xr;
    const double x4 = x2 * x2;
    const double c2 = mad<FMA>(x2, C4, C3);
    const double s1 = mad<FMA>(x2, S3, S2);
    const double c1 = mad<FMA>(x2, C1, C0);
    const double x5 = x3 * x2;
    const double x6 = x4 * x2;
    const double s = mad<FMA>(x3, S1, xr);
    const double c = mad<FMA>(x4, C2, c1);
    sincos_signs(__float_as_uint((float)mad<FMA>(x5, s1, s)), __float_as_uint((float)mad<FMA>(x6, c2, c)), quad, sidx, rs, rc);
}

// ---- round 5: the two polynomials in 9 operations instead of glibc's 12, for the FMA build's two fast paths.
// glibc evaluates  sin = (x + x3*S1) + x5*(S2 + x2*S3)  and  cos = ((C0 + x2*C1) + x4*C2) + x6*(C3 + x2*C4)  (twelve double
// operations with x2 ... x6).  Horner's rule needs nine:
//     sin = x + x * (x2 * (S1 + x2*(S2 + x2*S3)))          cos = C0 + x2*(C1 + x2*(C2 + x2*(C3 + x2*C4)))
// The two differ in the last places of the DOUBLE result, and the float result only changes when the double lies within
// those last places of a float rounding boundary.  Whether that ever happens is a finite question, and it is answered
// by enumeration, not by an error bound: for every argument of [2^-12, 120) (remainder as glibc forms it) and of
// [120, 2^29) (remainder from reduce_large_two), both signs, the two floats are those of the restated glibc sincosf —
// 316 669 952 + 371 195 904 arguments, 0 mismatches (tests/extended/sincos_model.c, part of the CPU suite; the device
// itself: tests/extended/exhaustive_device_sincos.py -> profiles/r05_exhaustive_device_sincos.json).  Ten other
// associations of the same polynomials were enumerated as well and none of them ever differs (profiles/r05_sincos.md):
// the function's float results are far more robust than its operation order suggests.
// One corrector: 25 -> 22 instructions in [2^-12, 120), 28 -> 24 in [120, 2^29).
__device__ __forceinline__ void sincos_horner(double xr, uint32_t n, float &rs, float &rc)
{
    constexpr double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
                     C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    constexpr double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7,
                     S3 = -0x1.994eb3774cf24p-13;
    const double x2 = xr * xr;
    const double u = __builtin_fma(x2, __builtin_fma(x2, S3, S2), S1);
    const double sn = __builtin_fma(xr, x2 * u, xr);
    const double h = __builtin_fma(x2, __builtin_fma(x2, C4, C3), C2);
    const double cs = __builtin_fma(x2, __builtin_fma(x2, h, C1), C0);
    sincos_signs(__float_as_uint((float)sn), __float_as_uint((float)cs), n, n, rs, rc);
}

// The same on the remainder r of x * 2/pi itself (|r| <= 1/2): glibc multiplies its remainder by pi/2 and evaluates the
// polynomials above; here pi/2 is inside the coefficients — SPk = Sk * hpi^(2k+1), CPk = Ck * hpi^(2k), hpi = glibc's double
// pi/2, each product formed exactly (rational arithmetic) and rounded once — and the multiplication is gone.
//     sin = r * (SP0 + r2*(SP1 + r2*(SP2 + r2*SP3)))       cos = 1 + r2*(CP1 + r2*(CP2 + r2*(CP3 + r2*CP4)))
__device__ __forceinline__ void sincos_horner_quadrants(double r, uint32_t n, float &rs, float &rc)
{
    constexpr double SP0 = 0x1.921fb54442d18p+0, SP1 = -0x1.4abbbf2376856p-1, SP2 = 0x1.466031025d4cdp-4, SP3 = -0x1.2dd0472562ec7p-8;
    constexpr double CP1 = -0x1.3bd3cc7ec2ba7p+0, CP2 = 0x1.03c1decc70af8p-2, CP3 = -0x1.55c6643b8d8a8p-6, CP4 = 0x1.d9f7bc1fcaa24p-11;
    const double r2 = r * r;
    const double w = __builtin_fma(r2, __builtin_fma(r2, __builtin_fma(r2, SP3, SP2), SP1), SP0);
    const double sn = r * w;
    const double h = __builtin_fma(r2, __builtin_fma(r2, CP4, CP3), CP2);
    const double cs = __builtin_fma(r2, __builtin_fma(r2, h, CP1), 1.0);
    sincos_signs(__float_as_uint((float)sn), __float_as_uint((float)cs), n, n, rs, rc);
}

// 120 <= |x| < 2^29: the remainder of x * 2/pi in quadrants from TWO terms of 2/pi (c1 = its leading 29 bits: x*c1 - n
// exact; c2 = the next 53).  The third term of reduce_large_quick matters for one magnitude of the whole range up to 2^30,
// 0x4e4dc501 (8.6e8) — beyond 2^29, and a stream's |theta| stays below 2^26 — so the quick range ends at 2^29 since round 5.
__device__ __forceinline__ double reduce_large_two(double x, uint32_t &n_out)
{
    constexpr double C1 = 0x1.45F306Dp-1, C2 = 0x1.9391054A7F09Dp-30;
    const double pm = __builtin_fma(x, kTwoOverPi, kRoundMagic);
    n_out = (uint32_t)__double2loint(pm);
    const double nd = pm - kRoundMagic;
    return __builtin_fma(x, C2, __builtin_fma(x, C1, -nd));
}

// |y| >= 120: exact 32x96-bit fixed-point product with 4/pi (glibc reduce_large).
// IN_REGS: the caller has established that every active lane has |y| < 2^33 (window offset <= 3).
template <bool IN_REGS>
__device__ __forceinline__ double reduce_large(uint32_t xi, uint32_t &quad, uint32_t &sidx)
{
    constexpr double PI63 = 0x1.921FB54442D18p-62;
    const uint32_t idx = (xi >> 26) & 15u;               // byte offset of the 96-bit window in the 4/pi string
    uint32_t a0, a4, a8;
    if (IN_REGS || __builtin_amdgcn_ballot_w64(idx > 3u) == 0) {
        // the string's first 16 bytes, big-endian: 000000a2 f9836e4e 441529fc 2757d1f5; window k = bytes idx+4k .. idx+4k+3
        const uint32_t sh = idx * 8u;
        a0 = (uint32_t)((0x000000a2f9836e4eull << sh) >> 32);
        a4 = (uint32_t)((0xf9836e4e441529fcull << sh) >> 32);
        a8 = (uint32_t)((0x441529fc2757d1f5ull << sh) >> 32);
    } else {
        const uint32_t *arr = &kInvPio4[idx];
        a0 = arr[0];
        a4 = arr[4];
        a8 = arr[8];
    }
    const uint32_t shift = (xi >> 23) & 7u;
    uint32_t m = (xi & 0xffffffu) | 0x800000u;
    m <<= shift;
    uint64_t res0 = (uint32_t)(m * a0);
    const uint64_t res1 = (uint64_t)m * a4;
    const uint64_t res2 = (uint64_t)m * a8;
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    // n = round(res0 / 2^62); res0 - n*2^62 only changes the high word, and the signed 64-bit remainder converts
    // to double with one rounding either way: hi*2^32 is exact, lo is exact, the fused add rounds their sum once.
    const uint32_t hi = (uint32_t)(res0 >> 32), lo = (uint32_t)res0;
    const uint32_t t = hi + 0x20000000u;
    const uint32_t n = t >> 30;
    const int32_t rem_hi = (int32_t)((t & 0x3fffffffu) - 0x20000000u);
    quad = n;
    sidx = n + (xi >> 31);
    return __builtin_fma((double)rem_hi, 4294967296.0, (double)lo) * PI63;
}

// A wavefront whose lanes lie in BOTH fast ranges, or below 2^-12 (round 5): a tile in which the counter wraps from the end
// of a period (|theta| in the thousands) to 1, 2, 3 ... — one tile in 25-75 of a replay's per-sample launches, and it went
// through sincosf_general (the integer product for every lane beyond 120: ~75 instructions per corrector).  Here the quadrant
// rounding is shared, both remainders are formed — glibc's own x - n*pi/2 below 120, (x*c1 - n + x*c2) * pi/2 from 120 on —
// one is selected per lane, and Horner's polynomials in radians follow: 32 instructions, valid for |y| < 2^29 (tiny lanes:
// sin = y, cos = 1 as in glibc).  Enumerated like the other paths (tests/extended/sincos_model.c --mixed): every argument of
// [0, 2^29), both signs, 0 mismatches against the restated glibc; FMA build only.
__device__ __forceinline__ void sincosf_mixed(float y, float &sn, float &cs)
{
    constexpr double C1 = 0x1.45F306Dp-1, C2 = 0x1.9391054A7F09Dp-30, HPI = 0x1.921FB54442D18p0;
    const uint32_t ax = __float_as_uint(y) & 0x7fffffffu;
    const double x = (double)y;
    const double pm = __builtin_fma(x, kTwoOverPi, kRoundMagic);
    const uint32_t n = (uint32_t)__double2loint(pm);
    const double nd = pm - kRoundMagic;
    const double xs = __builtin_fma(-nd, HPI, x);
    const double xl = __builtin_fma(x, C2, __builtin_fma(x, C1, -nd)) * HPI;
    const double xr = ax < 0x42f00000u ? xs : xl;
    float rs, rc;
    sincos_horner(xr, n, rs, rc);
    const bool tiny = ax < 0x39800000u;
    sn = tiny ? y : rs;
    cs = tiny ? 1.0f : rc;
}

template <bool FMA>
__device__ __forceinline__ void sincosf_general(float y, float &sn, float &cs);

// sin and cos of y, bit-identical to glibc 2.35 sincosf (see header comment).
// Must be called from converged or divergent code alike: the ballots below see the active lanes only.
template <bool FMA>
__device__ __forceinline__ void sincosf_glibc(float y, float &sn, float &cs)
{
    const uint32_t xi = __float_as_uint(y);
    const uint32_t ax = xi & 0x7fffffffu;
    // 2^-12 <= |y| < 120 (0x39800000 = 2^-12, 0x42f00000 = 120)
    const bool plain = (ax - 0x39800000u) < (0x42f00000u - 0x39800000u);
    if (__builtin_amdgcn_ballot_w64(!plain) == 0) {
        uint32_t n;
        const double xr = reduce_small<FMA>((double)y, n);
        if constexpr (FMA) sincos_horner(xr, n, sn, cs);
        else               sincos_poly<FMA>(xr, n, n, sn, cs);
        return;
    }
    // 120 <= |y| < 2^29: every lane takes the quick double-precision reduction
    // (the common case for a stream: theta = 2 pi ratio n passes 120 after a few thousand counters, and stays below 2^26)
    const bool large = (ax - 0x42f00000u) < (kLargeQuickEnd - 0x42f00000u);
    if (__builtin_amdgcn_ballot_w64(!large) == 0) {
        uint32_t n;
        if constexpr (FMA) {
            const double r = reduce_large_two((double)y, n);
            sincos_horner_quadrants(r, n, sn, cs);
        } else {
            const double xr = reduce_large_quick((double)y, n);
            sincos_poly<FMA>(xr, n, n, sn, cs);
        }
        return;
    }
    if constexpr (FMA) {
        if (__builtin_amdgcn_ballot_w64(!(ax < kLargeQuickEnd)) == 0) {      // both fast ranges and tiny lanes in one wavefront
            sincosf_mixed(y, sn, cs);
            return;
        }
    }
    sincosf_general<FMA>(y, sn, cs);
}

// any mixture of ranges in one wavefront: per-lane selects
template <bool FMA>
__device__ __forceinline__ void sincosf_general(float y, float &sn, float &cs)
{
    const uint32_t xi = __float_as_uint(y);
    const uint32_t top = xi >> 20 & 0x7ffu;             // abstop12
    // clamp keeps the conversion defined for lanes that take another range
    uint32_t quad, sidx;
    double xr = reduce_small<FMA>((top < 0x42fu) ? (double)y : 0.0, quad);
    sidx = quad;
    if (top >= 0x42fu && top < 0x7f8u) xr = reduce_large<false>(xi, quad, sidx);
    float rs, rc;
    sincos_poly<FMA>(xr, quad, sidx, rs, rc);
    if (top < 0x398u) {          // |y| < 2^-12: sin = y, cos = 1
        rs = y;
        rc = 1.0f;
    }
    if (top >= 0x7f8u) {         // inf / nan
        rs = rc = y - y;
    }
    sn = rs;
    cs = rc;
}

// 2^(i/32) as IEEE-754 doubles (glibc __exp2f_data.tab)
__device__ __constant__ const uint64_t kExp2fTab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b,
    0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb,
    0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429,
    0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad,
    0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};

// expf, bit-identical to glibc 2.35 (optimized-routines expf: x*32/ln2 = k + r, 2^(k/32) from the
// table, cubic in r, all in double).  Not on the streaming path: ccexpf arguments with a real
// part only (reference src/dsp.rs:57-83).
template <bool FMA>
__device__ __forceinline__ float expf_glibc(float x)
{
    constexpr double SHIFT = 0x1.8p+52, INVLN2N = 0x1.71547652b82fep+5;
    constexpr double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
    const uint32_t xi = __float_as_uint(x);
    const uint32_t abstop = (xi >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {                       // |x| >= 88 or nan
        if (xi == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);    // 0x1p97f * 0x1p97f
        if (x < -0x1.9fe368p6f) return 0.0f;                           // 0x1p-95f * 0x1p-95f
        if (x < -0x1.9d1d9ep6f) return __uint_as_float(1u);            // 0x1.4p-75f squared rounds to 2^-149
    }
    const double xd = (double)x;
    double kd, r;
    if constexpr (FMA) {
        kd = __builtin_fma(INVLN2N, xd, SHIFT);
    } else {
        const double z = INVLN2N * xd;
        kd = z + SHIFT;
    }
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= SHIFT;
    if constexpr (FMA) {
        r = __builtin_fma(INVLN2N, xd, -kd);
    } else {
        const double z = INVLN2N * xd;
        r = z - kd;
    }
    const uint64_t t = kExp2fTab[ki & 31u] + (ki << 47);
    const double sc = __longlong_as_double((long long)t);
    const double z2 = mad<FMA>(C0, r, C1);
    const double r2 = r * r;
    double y = mad<FMA>(C2, r, 1.0);
    y = mad<FMA>(z2, r2, y);
    y = y * sc;
    return (float)y;
}

// ccexpf of reference src/complex.c:33-39 for ANY argument: the argument construction
// `real + imag * I` followed by glibc 2.35 cexpf (math/s_cexp_template.c), on the two functions above.
template <bool FMA>
__device__ __forceinline__ void ccexpf_glibc(float re, float im, float &out_re, float &out_im)
{
    const float FLT_MIN_ = 1.17549435e-38f, FLT_MAX_ = 3.40282347e+38f;
    const float inf = __uint_as_float(0x7f800000u), nan = __uint_as_float(0x7fc00000u);
    float xr = __fadd_rn(re, __fmul_rn(im, 0.0f));
    const float xi = im;
    const bool r_fin = fabsf(xr) <= FLT_MAX_, i_fin = fabsf(xi) <= FLT_MAX_;     // false for inf and nan
    float sinix = xi, cosix = 1.0f;
    if (i_fin && fabsf(xi) > FLT_MIN_) sincosf_glibc<FMA>(xi, sinix, cosix);
    if (r_fin) {
        if (!i_fin) { out_re = out_im = nan; return; }
        const float t = 88.0f;
        if (xr > t) {
            const float exp_t = expf_glibc<FMA>(t);
            xr = __fsub_rn(xr, t);
            sinix = __fmul_rn(sinix, exp_t);
            cosix = __fmul_rn(cosix, exp_t);
            if (xr > t) {
                xr = __fsub_rn(xr, t);
                sinix = __fmul_rn(sinix, exp_t);
                cosix = __fmul_rn(cosix, exp_t);
            }
        }
        if (xr > t) {
            out_re = __fmul_rn(FLT_MAX_, cosix);
            out_im = __fmul_rn(FLT_MAX_, sinix);
        } else {
            const float e = expf_glibc<FMA>(xr);
            out_re = __fmul_rn(e, cosix);
            out_im = __fmul_rn(e, sinix);
        }
    } else if (xr != xr) {                       // real part NaN
        out_re = nan;
        out_im = (xi == 0.0f) ? xi : nan;
    } else if (i_fin) {                          // real part +-inf, imaginary finite
        const float value = (__float_as_uint(xr) >> 31) ? 0.0f : inf;
        if (xi == 0.0f) {
            out_re = value;
            out_im = xi;
        } else {
            out_re = copysignf(value, cosix);
            out_im = copysignf(value, sinix);
        }
    } else if (!(__float_as_uint(xr) >> 31)) {   // +inf, imaginary inf/nan
        out_re = inf;
        out_im = xi - xi;
    } else {                                     // -inf, imaginary inf/nan
        out_re = 0.0f;
        out_im = copysignf(0.0f, xi);
    }
}

// The corrector of dsp.rs:121-122 for counter value n:
//   theta = (-2*PI) * (ratio * (n as f32))   — each product rounded to f32,
//   (c, s) = cexpf(0 + i*theta) = (cosf(theta), sinf(theta)).
// ratio = shift_hz / (samplerate as f32) is rounded once on the host.
template <bool FMA>
__device__ __forceinline__ void corrector(float ratio, uint32_t n, float &c, float &s)
{
    const float p = __fmul_rn(ratio, (float)n);          // u32 -> f32 rounds to nearest even
    const float theta = __fmul_rn(-6.28318530717958647692f, p);
    sincosf_glibc<FMA>(theta, s, c);
}

// Four correctors at once (the four samples of a lane's 16-byte vector, or four blocks of a strided loop): the two f32
// products are packed multiplies (each lane of a packed multiply rounds on its own, like the scalar one), ONE ballot
// decides for all four whether the wavefront may take the fast path of sincosf_glibc, and the four polynomial chains
// are independent instruction streams for the scheduler.  cs[k] = (cos, sin) for counter n[k].
typedef float sc_f32x2 __attribute__((ext_vector_type(2)));
// path: what the caller knows about all the counters of the WAVEFRONT (a wavefront-uniform value; DevSeg's n_plain /
// n_large / n_huge give it for free): kPathPlain — every |theta| in [2^-12, 120); kPathLarge — every |theta| in
// [120, 2^29); kPathAny — nothing known, the function looks and votes.
constexpr int kPathAny = 0, kPathPlain = 1, kPathLarge = 2, kPathMixed = 3;   // (kPathMixed: found by the vote only)

template <bool FMA>
__device__ __forceinline__ void corrector4_f(float ratio, sc_f32x2 f01, sc_f32x2 f23, sc_f32x2 cs[4], int path = kPathAny);

template <bool FMA>
__device__ __forceinline__ void corrector4(float ratio, const uint32_t n[4], sc_f32x2 cs[4], int path = kPathAny)
{
    corrector4_f<FMA>(ratio, sc_f32x2{(float)n[0], (float)n[1]}, sc_f32x2{(float)n[2], (float)n[3]}, cs, path);
}

// Four CONSECUTIVE counters n0 .. n0 + 3, all below 2^24 (the caller's uniform test): fl32(n0 + k) = fl32(n0) + k exactly,
// so one conversion and two packed additions replace four conversions and three integer additions.
template <bool FMA>
__device__ __forceinline__ void corrector4_consecutive(float ratio, uint32_t n0, sc_f32x2 cs[4], int path = kPathAny)
{
    const float f0 = (float)n0;
    corrector4_f<FMA>(ratio, sc_f32x2{f0, f0} + sc_f32x2{0.0f, 1.0f}, sc_f32x2{f0, f0} + sc_f32x2{2.0f, 3.0f}, cs, path);
}

// Two PAIRS of consecutive counters (na, na + 1, nb, nb + 1), all below 2^24: the tile kernel's f32 -> f32 lanes
template <bool FMA>
__device__ __forceinline__ void corrector4_pairs(float ratio, uint32_t na, uint32_t nb, sc_f32x2 cs[4], int path = kPathAny)
{
    const float fa = (float)na, fb = (float)nb;
    corrector4_f<FMA>(ratio, sc_f32x2{fa, fa} + sc_f32x2{0.0f, 1.0f}, sc_f32x2{fb, fb} + sc_f32x2{0.0f, 1.0f}, cs, path);
}

// fl32 of the four counters given: the two f32 products of theta, then the path decision
template <bool FMA>
__device__ __forceinline__ void corrector4_f(float ratio, sc_f32x2 f01, sc_f32x2 f23, sc_f32x2 cs[4], int path)
{
    const sc_f32x2 r2 = {ratio, ratio}, m2 = {-6.28318530717958647692f, -6.28318530717958647692f};
    const sc_f32x2 p01 = f01 * r2, p23 = f23 * r2;
    const sc_f32x2 t01 = p01 * m2, t23 = p23 * m2;
    const float th[4] = {t01.x, t01.y, t23.x, t23.y};
    bool all_plain = path == kPathPlain, all_large = path == kPathLarge;
    if (path == kPathAny) {
        uint32_t lo = 0xffffffffu, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t a = __float_as_uint(th[k]) & 0x7fffffffu;
            lo = a < lo ? a : lo;
            hi = a > hi ? a : hi;
        }
        // every |theta| in [2^-12, 120)  (nan / inf have the largest magnitudes: they fail the upper bound)
        all_plain = __builtin_amdgcn_ballot_w64(!(lo >= 0x39800000u && hi < 0x42f00000u)) == 0;
        all_large = !all_plain && __builtin_amdgcn_ballot_w64(!(lo >= 0x42f00000u && hi < kLargeQuickEnd)) == 0;
        if (FMA && !all_plain && !all_large && __builtin_amdgcn_ballot_w64(!(hi < kLargeQuickEnd)) == 0) path = kPathMixed;
    }
    if (all_plain) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t q;
            float sn, c;
            const double xr = reduce_small<FMA>((double)th[k], q);
            if constexpr (FMA) sincos_horner(xr, q, sn, c);
            else               sincos_poly<FMA>(xr, q, q, sn, c);
            cs[k] = sc_f32x2{c, sn};
        }
    } else if (all_large) {
        // every |theta| in [120, 2^29)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t q;
            float sn, c;
            if constexpr (FMA) {
                sincos_horner_quadrants(reduce_large_two((double)th[k], q), q, sn, c);
            } else {
                const double xr = reduce_large_quick((double)th[k], q);
                sincos_poly<FMA>(xr, q, q, sn, c);
            }
            cs[k] = sc_f32x2{c, sn};
        }
    } else if (FMA && path == kPathMixed) {
        // lanes on both sides of 120 (a counter wrapping inside the tile) or below 2^-12, none beyond 2^29
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float sn, c;
            sincosf_mixed(th[k], sn, c);
            cs[k] = sc_f32x2{c, sn};
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float sn, c;
            sincosf_general<FMA>(th[k], sn, c);
            cs[k] = sc_f32x2{c, sn};
        }
    }
}

}  // namespace dpx
