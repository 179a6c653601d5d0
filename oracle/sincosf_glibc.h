/*
 * TEST INFRASTRUCTURE — CPU oracle. Not shipped, not on the product path.
 *
 * Restatement of glibc 2.35's single-precision sincosf, the routine that
 * libm's cexpf(0 + i*theta) reduces to and therefore the arithmetic that
 * /root/reference/src/complex.c:33-39 (ccexpf) performs for the argument
 * built at /root/reference/src/dsp.rs:121.
 *
 * Third-party algorithm, absent from /root/reference: glibc 2.35 libm
 * (Ubuntu GLIBC 2.35-0ubuntu3.11), sysdeps/ieee754/flt-32/s_sincosf.c +
 * sincosf.h + sysdeps/x86/fpu/sincosf_poly.h, i.e. Szabolcs Nagy's
 * "optimized-routines" sincosf (double-precision evaluation, 3 argument
 * ranges).  On x86-64 libm selects by ifunc between a build WITH fused
 * multiply-add contraction (__sincosf_fma, chosen when the CPU has FMA+AVX2)
 * and one without (__sincosf_sse2).  Both are restated here; which products
 * are fused was read off the disassembly of this container's libm.so.6
 * (every a+b*c in the polynomial and the x-n*hpi reduction is a single fma).
 * Constants were read from libm.so.6 .rodata (0xb3060: __inv_pio4,
 * 0xb30c0: __sincosf_table, 0x9da08: pi63).
 *
 * Pinned by oracle/check_sincosf.c: bit-for-bit equality with this host's
 * libm sincosf over all 2^32 float bit patterns.
 */
#ifndef ORC_SINCOSF_GLIBC_H
#define ORC_SINCOSF_GLIBC_H

#ifdef __cplusplus
extern "C" {
#endif

/* variant: 1 = contraction as in __sincosf_fma, 0 = as in __sincosf_sse2 */
void orc_sincosf_glibc235(float y, float *sinp, float *cosp, int fma_variant);

/* cexpf(0 + i*theta) as glibc 2.35 math/s_cexp_template.c evaluates it for a
 * zero real part: |theta| > FLT_MIN -> sincosf, else (cos,sin) = (1, theta). */
void orc_cexpf_imag_glibc235(float theta, float *re, float *im, int fma_variant);

/* glibc 2.35 expf (sysdeps/ieee754/flt-32/e_expf.c, Szabolcs Nagy's optimized-routines expf: double
 * evaluation, 32-entry 2^(i/32) table, cubic) — needed for ccexpf arguments with a real part
 * (/root/reference/src/dsp.rs:57-83 test_cexpf).  fma_variant as above: __expf_fma fuses
 * z*InvLn2N+SHIFT, InvLn2N*x-kd and every polynomial step; __expf_sse2 fuses nothing.
 * Constants read from libm.so.6 .rodata (0xb2b80: table, 0xb2ca0..0xb2cc0: SHIFT, InvLn2N, C0..C2). */
float orc_expf_glibc235(float x, int fma_variant);

/* glibc 2.35 cexpf (math/s_cexp_template.c) on top of the two functions above, preceded by the
 * argument construction of /root/reference/src/complex.c:34 (`real + imag * I`). */
void orc_ccexpf_glibc235(float re, float im, float *out_re, float *out_im, int fma_variant);

/* Which variant is bit-identical to this host's libm on a 1M-point probe:
 * 1 (fma), 0 (sse2) or -1 (neither; unexpected libm). */
int orc_detect_libm_variant(void);

#ifdef __cplusplus
}
#endif
#endif
