/*
 * TEST INFRASTRUCTURE — CPU oracle for the cubehub/doppler hot path.
 * See doppler_oracle.h for scope, pinning status and build flags.
 * Every function cites the reference lines it restates.
 */
#define _GNU_SOURCE
#include "doppler_oracle.h"
#include "sincosf_glibc.h"

#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* std::f32::consts::PI (dsp.rs:44) */
static const float PI_F32 = 3.14159265358979323846264338327950288f;

/* ------------------------------------------------------------------ A7 ---- */
/* complex.c:33-39: z <- cexpf(z.real + z.imag*I), in place, returns void. */
void orc_ccexpf(orc_complex *z)
{
    float complex in = z->re + z->im * I;
    float complex out = cexpf(in);
    z->re = crealf(out);
    z->im = cimagf(out);
}

static orc_ccexpf_fn g_ccexpf = orc_ccexpf;
static int g_mode = 0;

void orc_set_ccexpf(orc_ccexpf_fn fn) { g_ccexpf = fn ? fn : orc_ccexpf; }
void orc_set_corrector_mode(int mode) { g_mode = mode; }
int orc_get_corrector_mode(void) { return g_mode; }

void orc_ccexpf_imag_array(const float *theta, size_t n, orc_complex *out, int mode)
{
    for (size_t k = 0; k < n; ++k) {
        if (mode == 0) {
            out[k].re = 0.0f;
            out[k].im = theta[k];
            g_ccexpf(&out[k]);
        } else {
            orc_cexpf_imag_glibc235(theta[k], &out[k].re, &out[k].im, mode == 1);
        }
    }
}

void orc_ccexpf_array(const orc_complex *z, size_t n, orc_complex *out, int mode)
{
    for (size_t k = 0; k < n; ++k) {
        if (mode == 0) {
            out[k] = z[k];
            g_ccexpf(&out[k]);
        } else {
            orc_ccexpf_glibc235(z[k].re, z[k].im, &out[k].re, &out[k].im, mode == 1);
        }
    }
}

/* ------------------------------------------------------------------ A1 ---- */
/* dsp.rs:85-99: assert len%4==0; per 4 bytes b:
 *   i = ((b[1] as i16) << 8 | b[0] as i16) as f32 / 32768.   (same for q) */
long orc_convert_iqi16_to_complex(const uint8_t *inbuf, size_t len, orc_complex *out)
{
    if (len % 4 != 0) return ORC_ERR_BLOCK_LEN;
    size_t n = len / 4;
    for (size_t k = 0; k < n; ++k) {
        const uint8_t *b = inbuf + 4 * k;
        int16_t i = (int16_t)(uint16_t)(((uint16_t)b[1] << 8) | (uint16_t)b[0]);
        int16_t q = (int16_t)(uint16_t)(((uint16_t)b[3] << 8) | (uint16_t)b[2]);
        out[k].re = (float)i / 32768.f;
        out[k].im = (float)q / 32768.f;
    }
    return (long)n;
}

/* ------------------------------------------------------------------ A2 ---- */
/* dsp.rs:101-115: assert len%8==0; per 8 bytes: two LE u32 transmuted to f32. */
long orc_convert_iqf32_to_complex(const uint8_t *inbuf, size_t len, orc_complex *out)
{
    if (len % 8 != 0) return ORC_ERR_BLOCK_LEN;
    size_t n = len / 8;
    for (size_t k = 0; k < n; ++k) {
        const uint8_t *b = inbuf + 8 * k;
        uint32_t ui = ((uint32_t)b[3] << 24) | ((uint32_t)b[2] << 16) | ((uint32_t)b[1] << 8) | b[0];
        uint32_t uq = ((uint32_t)b[7] << 24) | ((uint32_t)b[6] << 16) | ((uint32_t)b[5] << 8) | b[4];
        memcpy(&out[k].re, &ui, 4);
        memcpy(&out[k].im, &uq, 4);
    }
    return (long)n;
}

/* --------------------------------------------------------------- A3 / A4 -- */
/* f32::fract: self - self.trunc() */
static inline float fract_f32(float x) { return x - truncf(x); }

/* dsp.rs:125-130 on the pre-increment counter; u32 += 1 wraps (release build) */
static inline void counter_rule(uint32_t *samplenum, float shift_hz, uint32_t samplerate)
{
    if (fract_f32(shift_hz / (float)samplerate * (float)(*samplenum)) == 0.0f)
        *samplenum = 1;
    else
        *samplenum += 1;
}

/* dsp.rs:117-134.  theta = (-2*PI) * (shift_hz / samplerate as f32 * samplenum as f32),
 * three f32 roundings in this association (dsp.rs:121); corrector = ccexpf(0 + i*theta)
 * (dsp.rs:122); output = sample * corrector with num-complex 0.1.35's Mul:
 * (a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re), no contraction (dsp.rs:123). */
void orc_shift_frequency(const orc_complex *inbuf, size_t n, uint32_t *samplenum,
                         float shift_hz, uint32_t samplerate, orc_complex *out)
{
    for (size_t k = 0; k < n; ++k) {
        orc_complex corrector;
        corrector.re = 0.0f;
        corrector.im = -2.f * PI_F32 * (shift_hz / (float)samplerate * (float)(*samplenum));
        if (g_mode == 0) {
            g_ccexpf(&corrector);
        } else {
            float re, im;
            orc_cexpf_imag_glibc235(corrector.im, &re, &im, g_mode == 1);
            corrector.re = re;
            corrector.im = im;
        }
        orc_complex s = inbuf[k];
        float rr = s.re * corrector.re;
        float ii = s.im * corrector.im;
        float ri = s.re * corrector.im;
        float ir = s.im * corrector.re;
        out[k].re = rr - ii;
        out[k].im = ri + ir;
        counter_rule(samplenum, shift_hz, samplerate);
    }
}

void orc_advance_samplenum(uint32_t *samplenum, float shift_hz, uint32_t samplerate, uint64_t n)
{
    for (uint64_t k = 0; k < n; ++k) counter_rule(samplenum, shift_hz, samplerate);
}

/* ------------------------------------------------------------------ A5 ---- */
/* Rust `f32 as i16`: truncate toward zero, saturate, NaN -> 0 (defined since
 * Rust 1.45; stated as the reference's behaviour). */
static int g_i16_cast = 0;
/* 0: Rust >= 1.45 (`as` saturates, NaN -> 0) — the default.
 * 1: what a 2016 rustc (the reference's Cargo.lock pins crates of that year) compiled on x86-64: `fptosi float to i16`
 *    became CVTTSS2SI to a 32-bit register whose low half was kept — truncation toward zero, then wrap-around modulo
 *    2^16; NaN and |x| >= 2^31 give the "integer indefinite" 0x80000000, whose low half is 0.  (LLVM called the
 *    out-of-range case undefined; this is the instruction sequence it emitted.) */
void orc_set_i16_cast(int mode) { g_i16_cast = mode; }
int orc_get_i16_cast(void) { return g_i16_cast; }

static inline int16_t f32_as_i16(float x)
{
    if (g_i16_cast == 1) {
        int32_t w;
        if (x != x || x >= 2147483648.0f || x < -2147483648.0f) w = INT32_MIN;    /* CVTTSS2SI: integer indefinite */
        else w = (int32_t)x;                                                          /* truncates toward zero */
        return (int16_t)(uint16_t)((uint32_t)w & 0xffffu);
    }
    if (x != x) return 0;
    if (x >= 32767.0f) return 32767;
    if (x <= -32768.0f) return -32768;
    return (int16_t)x;
}

/* main.rs:74-84: i = (re * 32767.0) as i16; emit lo, hi bytes of i then q. */
void orc_pack_i16(const orc_complex *in, size_t n, uint8_t *out)
{
    for (size_t k = 0; k < n; ++k) {
        int16_t i = f32_as_i16(in[k].re * 32767.0f);
        int16_t q = f32_as_i16(in[k].im * 32767.0f);
        out[4 * k + 0] = (uint8_t)(i & 0xFF);
        out[4 * k + 1] = (uint8_t)((i >> 8) & 0xFF);
        out[4 * k + 2] = (uint8_t)(q & 0xFF);
        out[4 * k + 3] = (uint8_t)((q >> 8) & 0xFF);
    }
}

/* ------------------------------------------------------------------ A6 ---- */
/* main.rs:89-93: the Complex<f32> array reinterpreted as bytes. */
void orc_pack_f32(const orc_complex *in, size_t n, uint8_t *out)
{
    memcpy(out, in, n * sizeof(orc_complex));
}

/* ------------------------------------------------------------------ A8 ---- */
/* main.rs:62-99.  Keeps the reference's cost structure: three passes and a
 * fresh heap buffer for each (dsp.rs:89/105, dsp.rs:118, main.rs:74). */
int orc_shift_block(const uint8_t *in, size_t in_len, int intype, int outtype,
                    uint32_t *samplenum, float shift_hz, uint32_t samplerate,
                    uint8_t *out, size_t *n_out)
{
    if (in_len > ORC_BUFFER_SIZE) return ORC_ERR_ARG;
    if ((intype != ORC_FMT_I16 && intype != ORC_FMT_F32) ||
        (outtype != ORC_FMT_I16 && outtype != ORC_FMT_F32))
        return ORC_ERR_ARG;
    size_t cap = in_len / 4 + 1;
    orc_complex *input = (orc_complex *)malloc(cap * sizeof(orc_complex));
    long n = (intype == ORC_FMT_I16) ? orc_convert_iqi16_to_complex(in, in_len, input)
                                      : orc_convert_iqf32_to_complex(in, in_len, input);
    if (n < 0) { free(input); return (int)n; }
    orc_complex *output = (orc_complex *)malloc(((size_t)n + 1) * sizeof(orc_complex));
    orc_shift_frequency(input, (size_t)n, samplenum, shift_hz, samplerate, output);
    if (outtype == ORC_FMT_I16) {
        uint8_t *packed = (uint8_t *)malloc((size_t)n * 4 + 1);
        orc_pack_i16(output, (size_t)n, packed);
        memcpy(out, packed, (size_t)n * 4);     /* stdout.write */
        free(packed);
    } else {
        orc_pack_f32(output, (size_t)n, out);   /* stdout.write of the raw slice */
    }
    free(output);
    free(input);
    if (n_out) *n_out = (size_t)n;
    return in_len != ORC_BUFFER_SIZE;           /* main.rs:98 */
}

static inline size_t bps(int fmt) { return fmt == ORC_FMT_I16 ? 4 : 8; }

/* main.rs:102-119: shift_hz = shift as f32; loop until a short (or empty) block. */
long orc_const_stream(const uint8_t *in, size_t in_len, int intype, int outtype,
                      int32_t shift, uint32_t samplerate, uint32_t *samplenum, uint8_t *out)
{
    float shift_hz = (float)shift;              /* main.rs:110 */
    size_t pos = 0, opos = 0;
    for (;;) {
        size_t take = in_len - pos < ORC_BUFFER_SIZE ? in_len - pos : ORC_BUFFER_SIZE;
        size_t cnt = 0;
        int stop = orc_shift_block(in + pos, take, intype, outtype, samplenum, shift_hz,
                                   samplerate, out + opos, &cnt);
        if (stop < 0) return stop;
        pos += take;
        opos += cnt * bps(outtype);
        if (stop) break;
    }
    return (long)opos;
}

/* main.rs:156-184.  Order of operations inside one loop turn:
 *   predict.update(start+dt)            -> range rate for the CURRENT dt   (162)
 *   doppler_hz = (rr*1000/c)*f*(-1)      in f64                             (163)
 *   dt = seconds((sample_count as f32 / samplerate as f32) as i64)          (166)
 *   shift(intype, doppler_hz as f32 + offset as f32, samplerate)            (177)
 *   sample_count += count                                                   (182)
 * so block b is shifted with the range rate of the dt computed one turn earlier. */
long orc_track_stream(const uint8_t *in, size_t in_len, int intype, int outtype,
                      uint32_t samplerate, uint32_t frequency_hz, int32_t offset_hz, int has_offset,
                      const double *range_rate_km_s, size_t n_table,
                      uint32_t *samplenum, uint8_t *out, float *shift_log, size_t *n_blocks)
{
    const double SPEED_OF_LIGHT_M_S = 299792458.;   /* main.rs:48 */
    if (n_table == 0) return ORC_ERR_ARG;
    uint64_t sample_count = 0;
    int64_t dt = 0;
    size_t pos = 0, opos = 0, nb = 0;
    for (;;) {
        size_t idx = (dt < 0) ? 0 : ((uint64_t)dt >= n_table ? n_table - 1 : (size_t)dt);
        double rr = range_rate_km_s[idx];
        double doppler_hz = (rr * 1000.0 / SPEED_OF_LIGHT_M_S) * (double)frequency_hz * (-1.0);
        float q = (float)sample_count / (float)samplerate;
        /* Rust `as i64`: truncate, saturate, NaN -> 0 */
        if (q != q) dt = 0;
        else if (q >= 9223372036854775807.0f) dt = INT64_MAX;
        else if (q <= -9223372036854775808.0f) dt = INT64_MIN;
        else dt = (int64_t)q;
        float shift_hz = (float)doppler_hz + (float)(has_offset ? offset_hz : 0);
        if (shift_log) shift_log[nb] = shift_hz;
        nb++;
        size_t take = in_len - pos < ORC_BUFFER_SIZE ? in_len - pos : ORC_BUFFER_SIZE;
        size_t cnt = 0;
        int stop = orc_shift_block(in + pos, take, intype, outtype, samplenum, shift_hz,
                                   samplerate, out + opos, &cnt);
        if (stop < 0) return stop;
        pos += take;
        opos += cnt * bps(outtype);
        if (stop) break;
        sample_count += cnt;
    }
    if (n_blocks) *n_blocks = nb;
    return (long)opos;
}

/* ---------------------------------------------------------- CPU baseline -- */
typedef struct {
    const uint8_t *in; size_t in_len; int intype, outtype; int32_t shift; uint32_t samplerate;
    uint8_t *out; uint64_t samples_before; long result;
} mt_job;

static void *mt_worker(void *arg)
{
    mt_job *j = (mt_job *)arg;
    uint32_t samplenum = 0;                     /* main.rs:60 */
    orc_advance_samplenum(&samplenum, (float)j->shift, j->samplerate, j->samples_before);
    /* process whole blocks only, except in the last chunk (handled by caller) */
    j->result = orc_const_stream(j->in, j->in_len, j->intype, j->outtype, j->shift,
                                 j->samplerate, &samplenum, j->out);
    return NULL;
}

long orc_const_stream_mt(const uint8_t *in, size_t in_len, int intype, int outtype,
                         int32_t shift, uint32_t samplerate, uint8_t *out, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    size_t blocks = in_len / ORC_BUFFER_SIZE;
    size_t per = (blocks + (size_t)n_threads - 1) / (size_t)n_threads;
    if (per == 0) { n_threads = 1; }
    pthread_t th[256];
    mt_job jobs[256];
    int used = 0;
    size_t pos = 0;
    for (int t = 0; t < n_threads; ++t) {
        size_t len = (t == n_threads - 1 || pos + per * ORC_BUFFER_SIZE >= in_len)
                         ? in_len - pos : per * ORC_BUFFER_SIZE;
        mt_job *j = &jobs[used];
        j->in = in + pos; j->in_len = len; j->intype = intype; j->outtype = outtype;
        j->shift = shift; j->samplerate = samplerate;
        j->samples_before = pos / bps(intype);
        j->out = out + (pos / bps(intype)) * bps(outtype);
        j->result = 0;
        pthread_create(&th[used], NULL, mt_worker, j);
        used++;
        pos += len;
        if (pos >= in_len) break;
    }
    long total = 0;
    for (int t = 0; t < used; ++t) {
        pthread_join(th[t], NULL);
        if (jobs[t].result < 0) total = jobs[t].result;
        else if (total >= 0) total += jobs[t].result;
    }
    return total;
}

/* ------------------------------------------- checker for sharded streams -- */
/* A stream of piecewise-constant shifts (what the per-block loop of main.rs:156-184 produces once equal
 * neighbouring blocks are merged; one segment for main.rs:102-119), starting from a counter the caller carries.
 * The counter at every thread's first sample comes from the sequential rule (dsp.rs:125-130) run over all the
 * preceding samples of this call — no closed form is assumed anywhere in the oracle.  Threads then run the
 * reference's three passes (A1/A2 -> A3/A4 -> A5/A6) over pieces of at most one reference block. */
typedef struct {
    const uint8_t *in; uint8_t *out; int intype, outtype; uint32_t samplerate;
    const orc_segment *segs; size_t n_segs;
    uint64_t lo, hi;            /* sample range of this thread */
    size_t seg0; uint64_t seg0_first;   /* segment holding `lo`, and its first sample */
    uint32_t samplenum;         /* counter at `lo` */
} seg_job;

static void *seg_worker(void *arg)
{
    seg_job *j = (seg_job *)arg;
    const size_t ib = bps(j->intype), ob = bps(j->outtype);
    orc_complex a[2048], b[2048];
    size_t si = j->seg0;
    uint64_t first = j->seg0_first, pos = j->lo;
    uint32_t sn = j->samplenum;
    while (pos < j->hi) {
        while (pos >= first + j->segs[si].n_samples) { first += j->segs[si].n_samples; si++; }
        uint64_t end = first + j->segs[si].n_samples;
        if (end > j->hi) end = j->hi;
        size_t n = (size_t)(end - pos < 2048 / (ib / 4) ? end - pos : 2048 / (ib / 4));
        if (j->intype == ORC_FMT_I16) orc_convert_iqi16_to_complex(j->in + pos * ib, n * ib, a);
        else orc_convert_iqf32_to_complex(j->in + pos * ib, n * ib, a);
        orc_shift_frequency(a, n, &sn, j->segs[si].shift_hz, j->samplerate, b);
        if (j->outtype == ORC_FMT_I16) orc_pack_i16(b, n, j->out + pos * ob);
        else orc_pack_f32(b, n, j->out + pos * ob);
        pos += n;
    }
    return NULL;
}

long orc_segments_stream_mt(const uint8_t *in, int intype, int outtype, uint32_t samplerate,
                            const orc_segment *segs, size_t n_segs, uint32_t *samplenum,
                            uint8_t *out, int n_threads)
{
    if ((intype != ORC_FMT_I16 && intype != ORC_FMT_F32) || (outtype != ORC_FMT_I16 && outtype != ORC_FMT_F32))
        return ORC_ERR_ARG;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    uint64_t total = 0;
    for (size_t i = 0; i < n_segs; ++i) total += segs[i].n_samples;
    if (total == 0) return 0;
    const uint64_t per = (total + (uint64_t)n_threads - 1) / (uint64_t)n_threads;
    pthread_t th[256];
    seg_job jobs[256];
    int used = 0;
    /* one sequential pass of the counter rule over the whole call; thread starts are noted on the way */
    uint32_t sn = *samplenum;
    size_t si = 0;
    uint64_t first = 0, pos = 0;
    while (pos < total) {
        while (pos >= first + segs[si].n_samples) { first += segs[si].n_samples; si++; }
        seg_job *j = &jobs[used];
        j->in = in; j->out = out; j->intype = intype; j->outtype = outtype; j->samplerate = samplerate;
        j->segs = segs; j->n_segs = n_segs;
        j->lo = pos; j->hi = pos + per < total ? pos + per : total;
        j->seg0 = si; j->seg0_first = first; j->samplenum = sn;
        used++;
        uint64_t p = pos;
        size_t s2 = si;
        uint64_t f2 = first;
        while (p < j->hi) {
            while (p >= f2 + segs[s2].n_samples) { f2 += segs[s2].n_samples; s2++; }
            uint64_t end = f2 + segs[s2].n_samples;
            if (end > j->hi) end = j->hi;
            orc_advance_samplenum(&sn, segs[s2].shift_hz, samplerate, end - p);
            p = end;
        }
        pos = j->hi;
    }
    for (int t = 0; t < used; ++t) pthread_create(&th[t], NULL, seg_worker, &jobs[t]);
    for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
    *samplenum = sn;
    return (long)(total * bps(outtype));
}
