/*
 * TEST INFRASTRUCTURE — CPU oracle. Not shipped, not on the product path.
 * See sincosf_glibc.h for provenance (glibc 2.35 sincosf, restated).
 *
 * Compile with -ffp-contract=off: every fused operation below is an explicit
 * fma(); every unfused a*b+c must stay two roundings.
 */
#define _GNU_SOURCE /* sincosf */
#include "sincosf_glibc.h"

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    double sign[4];   /* sign of sine in quadrants 0..3 */
    double hpi_inv;   /* 2/pi * 2^24 */
    double hpi;       /* pi/2 */
    double c0, c1, c2, c3, c4; /* cosine polynomial */
    double s1, s2, s3;         /* sine polynomial */
} sincos_tab;

/* entry 1 evaluates -cos so that quadrants 2,3 get their sign for free */
static const sincos_tab TAB[2] = {
    {{1.0, -1.0, -1.0, 1.0},
     0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
     0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16,
     -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0},
     0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
     -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16,
     -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
};

/* 4/pi as overlapping 32-bit windows, 192 bits of precision */
static const uint32_t INV_PIO4[24] = {
    0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
    0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
    0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
    0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041,
};

static const double PI63 = 0x1.921FB54442D18p-62;

static inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t abstop12(float f) { return (asuint(f) >> 20) & 0x7ff; }

/* a*b + c, fused or not */
static inline double mad(double a, double b, double c, int fused)
{
    if (fused) return __builtin_fma(a, b, c);
    double p = a * b;
    return p + c;
}

static inline void poly(double x, double x2, const sincos_tab *p, int n,
                        float *sinp, float *cosp, int fused)
{
    double x3 = x2 * x;
    double x4 = x2 * x2;
    double c2 = mad(x2, p->c4, p->c3, fused);
    double s1 = mad(x2, p->s3, p->s2, fused);
    double c1 = mad(x2, p->c1, p->c0, fused);
    double x5 = x3 * x2;
    double x6 = x4 * x2;
    double s = mad(x3, p->s1, x, fused);
    double c = mad(x4, p->c2, c1, fused);
    float fs = (float)mad(x5, s1, s, fused);
    float fc = (float)mad(x6, c2, c, fused);
    if (n & 1) { *sinp = fc; *cosp = fs; }
    else       { *sinp = fs; *cosp = fc; }
}

void orc_sincosf_glibc235(float y, float *sinp, float *cosp, int fused)
{
    double x = (double)y;
    const sincos_tab *p = &TAB[0];
    uint32_t top = abstop12(y);

    if (top < abstop12(0x1.921FB6p-1f)) {           /* |y| < ~pi/4 (top-12-bit compare) */
        double x2 = x * x;
        if (top < abstop12(0x1p-12f)) {             /* tiny: sin=y, cos=1 */
            *sinp = y;
            *cosp = 1.0f;
            return;
        }
        poly(x, x2, p, 0, sinp, cosp, fused);
    } else if (top < abstop12(120.0f)) {            /* scaled-int quadrant reduction */
        double r = x * p->hpi_inv;
        int32_t n = ((int32_t)r + 0x800000) >> 24;
        double xr = fused ? __builtin_fma(-(double)n, p->hpi, x) : x - (double)n * p->hpi;
        double s = p->sign[n & 3];
        if (n & 2) p = &TAB[1];
        poly(xr * s, xr * xr, p, n, sinp, cosp, fused);
    } else if (top < abstop12(INFINITY)) {          /* 32x96-bit fixed-point reduction */
        uint32_t xi = asuint(y);
        int sign = (int)(xi >> 31);
        const uint32_t *arr = &INV_PIO4[(xi >> 26) & 15];
        int shift = (xi >> 23) & 7;
        uint64_t n, res0, res1, res2;
        uint32_t m = (xi & 0xffffff) | 0x800000;
        m <<= shift;
        res0 = (uint32_t)(m * arr[0]);
        res1 = (uint64_t)m * arr[4];
        res2 = (uint64_t)m * arr[8];
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        n = (res0 + (1ULL << 61)) >> 62;
        res0 -= n << 62;
        double xr = (double)(int64_t)res0 * PI63;
        int q = (int)n;
        double s = p->sign[(q + sign) & 3];
        if ((q + sign) & 2) p = &TAB[1];
        poly(xr * s, xr * xr, p, q, sinp, cosp, fused);
    } else {                                        /* inf / nan */
        *sinp = *cosp = y - y;
    }
}

void orc_cexpf_imag_glibc235(float theta, float *re, float *im, int fused)
{
    /* math/s_cexp_template.c, real part +0: exp(0)=1, so the result is
       (1*cos, 1*sin); sincos is skipped for |theta| <= FLT_MIN. */
    float s, c;
    if (fabsf(theta) > FLT_MIN) {
        orc_sincosf_glibc235(theta, &s, &c, fused);
    } else if (theta != theta) {
        s = c = theta - theta;
    } else {
        s = theta;
        c = 1.0f;
    }
    *re = c;
    *im = s;
}

/* ------------------------------------------------------------------ expf ---- */
static const uint64_t EXP2F_T[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b,
    0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb,
    0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429,
    0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad,
    0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};

float orc_expf_glibc235(float x, int fused)
{
    const double SHIFT = 0x1.8p+52, InvLn2N = 0x1.71547652b82fep+5;
    const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
    const double xd = (double)x;
    const uint32_t abstop = (asuint(x) >> 20) & 0x7ff;
    if (abstop >= 0x42b) {                      /* |x| >= 88 or nan */
        if (asuint(x) == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) { volatile float h = 0x1p97f; return h * h; }          /* overflow: +inf */
        if (x < -0x1.9fe368p6f) { volatile float t = 0x1p-95f; return t * t; }        /* underflow: +0 */
        if (x < -0x1.9d1d9ep6f) { volatile float t = 0x1.4p-75f; return t * t; }      /* may underflow: 2^-149 */
    }
    double kd, r;
    if (fused) {
        kd = __builtin_fma(InvLn2N, xd, SHIFT);
    } else {
        const double z = InvLn2N * xd;
        kd = z + SHIFT;
    }
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd -= SHIFT;
    if (fused) {
        r = __builtin_fma(InvLn2N, xd, -kd);
    } else {
        const double z = InvLn2N * xd;
        r = z - kd;
    }
    uint64_t t = EXP2F_T[ki % 32];
    t += ki << (52 - 5);
    double sc;
    memcpy(&sc, &t, 8);
    const double z2 = mad(C0, r, C1, fused);
    const double r2 = r * r;
    double y = mad(C2, r, 1.0, fused);
    y = mad(z2, r2, y, fused);
    y = y * sc;
    return (float)y;
}

/* ----------------------------------------------------------------- cexpf ---- */
void orc_ccexpf_glibc235(float re, float im, float *out_re, float *out_im, int fused)
{
    /* complex.c:34: float complex input = a->real + a->imag * I;   (imag * (0 + 1i), then real + that) */
    volatile float zero = 0.0f;
    float xr = re + im * zero;
    const float xi = im;
    /* math/s_cexp_template.c for float */
    const int r_fin = isfinite(xr), i_fin = isfinite(xi);
    float rr, ri;
    if (r_fin) {
        if (i_fin) {
            const int t = 88;                   /* (int)((FLT_MAX_EXP - 1) * M_LN2f) */
            float sinix, cosix;
            if (fabsf(xi) > FLT_MIN) orc_sincosf_glibc235(xi, &sinix, &cosix, fused);
            else { sinix = xi; cosix = 1.0f; }
            if (xr > t) {
                const float exp_t = orc_expf_glibc235((float)t, fused);
                xr -= t;
                sinix *= exp_t;
                cosix *= exp_t;
                if (xr > t) {
                    xr -= t;
                    sinix *= exp_t;
                    cosix *= exp_t;
                }
            }
            if (xr > t) {
                rr = FLT_MAX * cosix;
                ri = FLT_MAX * sinix;
            } else {
                const float exp_val = orc_expf_glibc235(xr, fused);
                rr = exp_val * cosix;
                ri = exp_val * sinix;
            }
        } else {
            rr = ri = NAN;
        }
    } else if (isinf(xr)) {
        if (i_fin) {
            const float value = signbit(xr) ? 0.0f : HUGE_VALF;
            if (xi == 0.0f) {
                rr = value;
                ri = xi;
            } else {
                float sinix, cosix;
                if (fabsf(xi) > FLT_MIN) orc_sincosf_glibc235(xi, &sinix, &cosix, fused);
                else { sinix = xi; cosix = 1.0f; }
                rr = copysignf(value, cosix);
                ri = copysignf(value, sinix);
            }
        } else if (!signbit(xr)) {
            rr = HUGE_VALF;
            ri = xi - xi;
        } else {
            rr = 0.0f;
            ri = copysignf(0.0f, xi);
        }
    } else {
        rr = NAN;
        ri = (xi == 0.0f) ? xi : NAN;
    }
    *out_re = rr;
    *out_im = ri;
}

/* The only 17 magnitudes (x2 signs) out of all 2^32 floats on which the two
   contraction variants give different results (found by exhaustive search,
   all in the scaled-int reduction range); they discriminate the ifunc choice. */
static const uint32_t DISCRIMINATORS[17] = {
    0x418a3adb, 0x418a3adc, 0x418a3add, 0x418a3ade, 0x41bc76d9, 0x4202eb4b,
    0x4255b0a9, 0x42687a55, 0x4280ce28, 0x42870e40, 0x42a35c07, 0x42a35d44,
    0x42a97360, 0x42c55faa, 0x42cf5854, 0x42d8d23e, 0x42e87a55,
};

int orc_detect_libm_variant(void)
{
    int ok[2] = {1, 1};
    uint32_t u = 0x12345u;
    for (int i = 0; i < 1000000 + 34; ++i) {
        uint32_t bits;
        if (i < 34) {
            bits = DISCRIMINATORS[i >> 1] | ((uint32_t)(i & 1) << 31);
        } else {
            u = u * 1664525u + 1013904223u;
            /* exponent range 2^-14 .. 2^30, both signs */
            bits = (u & 0x807fffffu) | ((113u + (u >> 8) % 45u) << 23);
        }
        float y, ls, lc;
        memcpy(&y, &bits, 4);
        sincosf(y, &ls, &lc);
        for (int v = 0; v < 2; ++v) {
            float s, c;
            orc_sincosf_glibc235(y, &s, &c, v);
            if (asuint(s) != asuint(ls) || asuint(c) != asuint(lc)) ok[v] = 0;
        }
    }
    if (ok[1]) return 1;
    if (ok[0]) return 0;
    return -1;
}
