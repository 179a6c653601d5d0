/*
 * TEST INFRASTRUCTURE — CPU oracle for the cubehub/doppler hot path.
 *
 * A plain-C restatement of the reference's per-sample path, function by
 * function.  It is the checker for the HIP kernels: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 * Nothing in doppler_amd/ (the product) includes, links or loads it.
 *
 * Pinning status:
 *  - ccexpf: the reference's own four known-answer vectors
 *    (/root/reference/src/dsp.rs:57-83) pass, and when oracle/_ref/libcomplex.so
 *    exists (built by oracle/Makefile from /root/reference/src/complex.c, in
 *    place) the oracle calls THAT function per sample.
 *  - cexpf(0+i*theta) == restated glibc-2.35 sincosf: exhaustive, all 2^32
 *    floats (oracle/check_sincosf.c).
 *  - unpack / complex multiply / counter rule / i16 pack: the reference has no
 *    result-pinning test for these and its Rust sources cannot be compiled in
 *    this image (no cargo/rustc; three un-vendored git dependencies), so these
 *    rows are restated from the source text only: PARITY UNPINNED for them.
 *  - orbit (libgpredict range-rate): not in /root/reference, PARITY UNPINNED;
 *    the oracle takes range-rate as an input table.
 *
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math), link -lm.  Rust never
 * contracts a*b+c, so neither may this file.
 */
#ifndef DOPPLER_ORACLE_H
#define DOPPLER_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* /root/reference/src/complex.c:28-31 (RustComplex) == dsp.rs:41 LiquidComplex32 */
typedef struct { float re, im; } orc_complex;

#define ORC_FMT_I16 0   /* usage.rs:39-42 DataType::I16 */
#define ORC_FMT_F32 1   /* usage.rs:39-42 DataType::F32 */
#define ORC_BUFFER_SIZE 8192   /* main.rs:49 */

#define ORC_ERR_BLOCK_LEN (-1)  /* where the reference panics: dsp.rs:87,103 */
#define ORC_ERR_ARG (-2)

/* ---- A7: complex.c:33-39 ------------------------------------------------ */
typedef void (*orc_ccexpf_fn)(orc_complex *);
void orc_ccexpf(orc_complex *z);              /* in-place cexpf through libm */
void orc_set_ccexpf(orc_ccexpf_fn fn);        /* NULL restores orc_ccexpf */
/* 0: libm cexpf (default, or oracle/_ref's ccexpf once set);
 * 1: restated glibc sincosf, FMA variant; 2: restated, non-FMA variant */
void orc_set_corrector_mode(int mode);
/* `as i16` of main.rs:77-78: 0 = saturating, NaN -> 0 (Rust >= 1.45, default); 1 = CVTTSS2SI + low 16 bits (x86-64 code of a 2016 rustc) */
void orc_set_i16_cast(int mode);
int orc_get_i16_cast(void);
int orc_get_corrector_mode(void);

/* corrector for an array of angles: out[k] = ccexpf(0 + i*theta[k]) under `mode`
 * (same meaning as orc_set_corrector_mode; mode 0 goes through the ccexpf pointer) */
void orc_ccexpf_imag_array(const float *theta, size_t n, orc_complex *out, int mode);

/* general ccexpf for an array of arguments (any real part): mode 0 through the ccexpf pointer
 * (libm / reference complex.c), 1 / 2 the restated glibc cexpf (FMA / SSE2 builds of expf and sincosf) */
void orc_ccexpf_array(const orc_complex *z, size_t n, orc_complex *out, int mode);

/* ---- A1/A2: dsp.rs:85-99, 101-115 ---------------------------------------- */
/* return number of complex samples written, or ORC_ERR_BLOCK_LEN */
long orc_convert_iqi16_to_complex(const uint8_t *inbuf, size_t len, orc_complex *out);
long orc_convert_iqf32_to_complex(const uint8_t *inbuf, size_t len, orc_complex *out);

/* ---- A3/A4: dsp.rs:117-134 ----------------------------------------------- */
void orc_shift_frequency(const orc_complex *inbuf, size_t n, uint32_t *samplenum,
                         float shift_hz, uint32_t samplerate, orc_complex *out);
/* the counter rule alone (dsp.rs:125-130), n samples, no corrector */
void orc_advance_samplenum(uint32_t *samplenum, float shift_hz, uint32_t samplerate, uint64_t n);

/* ---- A5/A6: main.rs:72-94 ------------------------------------------------ */
void orc_pack_i16(const orc_complex *in, size_t n, uint8_t *out);   /* 4 B/sample */
void orc_pack_f32(const orc_complex *in, size_t n, uint8_t *out);   /* 8 B/sample */

/* ---- A8: main.rs:62-99, one call of the `shift` closure ------------------ */
/* in_len <= 8192 bytes were "read"; writes the packed output; *n_out = samples.
 * returns 1 if this was a short block (the loop must stop), 0 otherwise,
 * or a negative ORC_ERR_*. */
int orc_shift_block(const uint8_t *in, size_t in_len, int intype, int outtype,
                    uint32_t *samplenum, float shift_hz, uint32_t samplerate,
                    uint8_t *out, size_t *n_out);

/* ---- main.rs:102-119, `doppler const` over an in-memory stream ----------- */
/* out must hold (in_len / in_bytes_per_sample) * out_bytes_per_sample bytes.
 * *samplenum is the closure's captured state (main.rs:60: starts at 0). */
long orc_const_stream(const uint8_t *in, size_t in_len, int intype, int outtype,
                      int32_t shift, uint32_t samplerate, uint32_t *samplenum, uint8_t *out);

/* ---- main.rs:156-184, `doppler track --time` replay ---------------------- */
/* range_rate_km_s[t] stands in for predict.update(start_time + t seconds);
 * n_table entries, the last one is held beyond the table.  If shift_log is
 * non-NULL it receives the f32 shift applied to every block (caller sizes it
 * for ceil(in_len/8192)+1 entries) and *n_blocks the count. */
long orc_track_stream(const uint8_t *in, size_t in_len, int intype, int outtype,
                      uint32_t samplerate, uint32_t frequency_hz, int32_t offset_hz, int has_offset,
                      const double *range_rate_km_s, size_t n_table,
                      uint32_t *samplenum, uint8_t *out, float *shift_log, size_t *n_blocks);

/* ---- CPU baseline helper: const stream on n_threads host threads ---------- */
/* Thread t takes a block-aligned contiguous chunk; its starting samplenum is
 * obtained by running the counter rule sequentially over the preceding samples
 * (orc_advance_samplenum), so no closed form is assumed. n_threads==1 is the
 * reference's own structure. */
long orc_const_stream_mt(const uint8_t *in, size_t in_len, int intype, int outtype,
                         int32_t shift, uint32_t samplerate, uint8_t *out, int n_threads);

/* ---- checker for sharded / piecewise-constant streams --------------------- */
/* `segs`: runs of samples with one f32 shift each (main.rs:110 / main.rs:177 after merging equal
 * neighbouring blocks).  *samplenum is the carried counter (in/out); the counter at every thread's first
 * sample is obtained with the sequential rule, never a closed form.  Returns bytes written. */
typedef struct { uint64_t n_samples; float shift_hz; } orc_segment;
long orc_segments_stream_mt(const uint8_t *in, int intype, int outtype, uint32_t samplerate,
                            const orc_segment *segs, size_t n_segs, uint32_t *samplenum,
                            uint8_t *out, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
