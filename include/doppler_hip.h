/*
 * doppler_hip.h — C ABI of the MI355X (gfx950) Doppler-correction hot path.
 *
 * This shared library (libdoppler_hip.so) stands behind the reference's operator
 * boundary `doppler::dsp` (reference src/lib.rs:35) exactly where the reference
 * already crosses into native code: src/dsp.rs:40-42 declares
 *     extern { pub fn ccexpf(z: *mut LiquidComplex32); }
 * and build.rs:28 links libcomplex.a.  A maintainer swaps that static library
 * for this one and binds the entry points below (the Rust `extern "C"` block is
 * shown in INTEGRATION.md).  Plain C types only: pointers, sizes, fixed-width
 * integers, float.  No panics cross the boundary: where the reference
 * `assert!`s (dsp.rs:87, dsp.rs:103) these functions return DPX_ERR_BLOCK_LEN.
 *
 * All arithmetic runs on the GPU in hand-written HIP kernels.  There is NO CPU
 * fallback: without a usable gfx950 device dpx_ctx_create fails with
 * DPX_ERR_NO_DEVICE and nothing else can be called.
 *
 * Data is little-endian interleaved IQ: i16 = 4 bytes/sample, f32 = 8 bytes/sample.
 * A context is bound to one GPU and is not thread-safe; distinct contexts are
 * independent (one context per GPU / per process rank).
 *
 * This header is the boundary proper (30 entry points: context, the operators of
 * doppler::dsp, plans over device buffers, the slab stream).  Two more headers ship
 * with the library and are NOT needed to bind the path:
 *   doppler_hip_host.h   host-only arithmetic of the callers either side of it: the
 *                        counter's closed form, the replay schedule of `doppler track`,
 *                        the SGP4 stand-in for libgpredict;
 *   doppler_hip_debug.h  measurement knobs, the planner's self-checks, device-memory
 *                        helpers for callers without a HIP binding (tests, the CLI).
 */
#ifndef DOPPLER_HIP_H
#define DOPPLER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5 (round 6): dpx_stream_create_opts / dpx_stream_describe (doppler_hip_debug.h); dpx_options.reserved became sub_lg and is
 * validated (round 5 changed its meaning without a bump: a caller that left garbage there is now refused, not obeyed);
 * dpx_layout.f32_i16_by_tiles covers both mixed pairs.  No entry point of versions 1-4 changed its signature. */
#define DPX_ABI_VERSION 5

/* reference src/usage.rs:39-42  enum DataType { F32, I16 } */
#define DPX_FMT_I16 0
#define DPX_FMT_F32 1

/* reference src/main.rs:49  const BUFFER_SIZE: usize = 8192 */
#define DPX_BUFFER_SIZE 8192

enum dpx_status {
    DPX_OK = 0,
    DPX_ERR_ARG = -1,        /* null pointer, unknown format, bad size            */
    DPX_ERR_BLOCK_LEN = -2,  /* byte length not a whole number of samples
                                (the reference panics: dsp.rs:87, dsp.rs:103)      */
    DPX_ERR_NO_DEVICE = -3,  /* no gfx950 GPU visible / device index out of range */
    DPX_ERR_HIP = -4,        /* a HIP runtime call failed; see dpx_last_error()   */
    DPX_ERR_CAPACITY = -5,   /* output buffer too small                           */
    DPX_ERR_PLAN = -6        /* run does not match the plan (sample count, ...)   */
};

/* reference src/complex.c:28-31 RustComplex == src/dsp.rs:41 LiquidComplex32 ==
 * num::complex::Complex<f32> memory layout */
typedef struct { float re, im; } dpx_complex32;

typedef struct dpx_ctx dpx_ctx;
typedef struct dpx_plan dpx_plan;

/* One constant-shift stretch of the stream (track mode: reference
 * src/main.rs:156-184 changes shift_hz only between 8192-byte blocks). */
typedef struct {
    uint64_t n_samples;
    float shift_hz;
} dpx_segment;

/* ------------------------------------------------------------------ context */
int dpx_abi_version(void);
const char *dpx_last_error(void);               /* thread-local message of the last failure */
int dpx_device_count(int *count);
int dpx_ctx_create(int device, dpx_ctx **ctx);  /* fails loudly without a gfx950 device */
void dpx_ctx_destroy(dpx_ctx *ctx);

/* ------------------------------------------- operator entry points (host I/O)
 * Synchronous, host pointers in and out; staging, kernels and copies happen
 * inside.  These are what the reference's FFI for this path binds. */

/* replaces dsp.rs:85-99  convert_iqi16_to_complex(&[u8]) -> Vec<Complex<f32>> */
int dpx_convert_iqi16_to_complex(dpx_ctx *ctx, const uint8_t *inbuf, size_t in_bytes,
                                 dpx_complex32 *out, size_t out_cap, size_t *n_out);

/* replaces dsp.rs:101-115 convert_iqf32_to_complex(&[u8]) -> Vec<Complex<f32>> */
int dpx_convert_iqf32_to_complex(dpx_ctx *ctx, const uint8_t *inbuf, size_t in_bytes,
                                 dpx_complex32 *out, size_t out_cap, size_t *n_out);

/* replaces dsp.rs:117-134 shift_frequency(&[Complex<f32>], &mut u32, f32, u32) -> Vec<..>
 * `samplenum` is the caller-owned counter of dsp.rs:125-130, updated in place. */
int dpx_shift_frequency(dpx_ctx *ctx, const dpx_complex32 *inbuf, size_t n, uint32_t *samplenum,
                        float shift_hz, uint32_t samplerate, dpx_complex32 *out);

/* replaces main.rs:72-87: Complex<f32> -> LE i16 bytes, (x*32767.0) as i16 */
int dpx_pack_iqi16(dpx_ctx *ctx, const dpx_complex32 *inbuf, size_t n, uint8_t *out, size_t out_cap);

/* The whole body of the `shift` closure (main.rs:65-94) fused into one kernel:
 * unpack -> mix -> pack, never materialising Complex<f32> in memory.  Any number
 * of bytes (not limited to 8192); `samplenum` in/out as above.  Per 8 KiB call: 12.7 us (see dpx_shift_block_async). */
int dpx_shift_block(dpx_ctx *ctx, const void *in, size_t in_bytes, int in_fmt,
                    void *out, size_t out_cap, int out_fmt, uint32_t *samplenum,
                    float shift_hz, uint32_t samplerate, size_t *n_samples_out);

/* Many blocks per call: what a caller that wants the GPU's rate rather than a block per call does with the loop of
 * main.rs:113-118 / 160-183 — read up to N blocks, one call.  in_bytes may end with a short block (the reference's last
 * read); shift_hz[b] is the shift of block b (8192 input bytes each, the granularity at which the reference can change
 * it), n_blocks = ceil(in_bytes / 8192).  The counter is carried through all blocks exactly as through N calls of
 * dpx_shift_block.  Measured per call (pageable host memory): 64 KiB 52 us, 1 MiB 133 us, 16 MiB 676 us. */
int dpx_shift_blocks(dpx_ctx *ctx, const void *in, size_t in_bytes, int in_fmt, void *out, size_t out_cap, int out_fmt,
                     uint32_t *samplenum, const float *shift_hz, size_t n_blocks, uint32_t samplerate, size_t *n_samples_out);

/* The same block, asynchronously: dpx_shift_block_async copies the block into one of four pinned, device-mapped staging
 * buffers, enqueues the fused kernel and returns at once with a ticket; `samplenum` is already the counter after the block
 * (it follows from the closed form, not from the kernel).  dpx_wait waits until that block is done and copies its output
 * out.  So the loop of main.rs:113-118 can read block k + 1 from stdin while block k is on the GPU:
 *     dpx_shift_block_async(ctx, blk[k+1], ...&t[k+1]);  dpx_wait(ctx, t[k], out, ...);  write(out);
 * Tickets complete in the order they were issued; at most four may be outstanding (DPX_ERR_PLAN beyond that).  Blocks of
 * up to 8192 samples (the reference's block is 2048 / 1024); every input dpx_shift_block takes is taken.
 * The blocks of both calls go to a RESIDENT kernel (one workgroup per staging buffer polling a doorbell in the buffer; it
 * leaves when idle for 2 ms or when the library launches anything else on that device; at most one per device and process —
 * contexts hand it over), so a block costs PCIe round trips, not a launch: 12.7 us per 8 KiB block alone, 3.2 us with four
 * in flight (a launch per block: 17.9 / 10.0 us; profiles/r04_cli.md).  DPX_RESIDENT=0 in the environment restores a launch
 * per block; INTEGRATION.md section 3d says what sharing a device with other GPU software means for it. */
typedef uint32_t dpx_ticket;
int dpx_shift_block_async(dpx_ctx *ctx, const void *in, size_t in_bytes, int in_fmt, int out_fmt, uint32_t *samplenum,
                          float shift_hz, uint32_t samplerate, dpx_ticket *ticket);
int dpx_wait(dpx_ctx *ctx, dpx_ticket ticket, void *out, size_t out_cap, size_t *n_samples_out);

/* replaces complex.c:33-39 ccexpf(z): z[k] <- cexpf(z[k].re + i*z[k].im), in place, any argument
 * (bit-identical to glibc 2.35 cexpf, incl. the overflow / inf / nan rules of s_cexp_template.c). */
int dpx_ccexpf(dpx_ctx *ctx, dpx_complex32 *z, size_t n);

/* the same for the only argument shape the hot path ever builds (dsp.rs:121: real part 0):
 * z[k] <- cexpf(0 + i*z[k].im); z[k].re is ignored. */
int dpx_ccexpf_imag(dpx_ctx *ctx, dpx_complex32 *z, size_t n);

/* ----------------------------------------------- bulk API (device pointers) */

/* constant shift over n_samples starting with counter samplenum0 */
int dpx_plan_const(dpx_ctx *ctx, float shift_hz, uint32_t samplerate, uint32_t samplenum0,
                   uint64_t n_samples, dpx_plan **plan);
/* piecewise-constant shift: consecutive segments, counter carried across them
 * exactly as the reference carries `samplenr` (main.rs:60) across blocks */
int dpx_plan_segments(dpx_ctx *ctx, const dpx_segment *segs, size_t n_segs, uint32_t samplerate,
                      uint32_t samplenum0, dpx_plan **plan);
int dpx_plan_final_samplenum(const dpx_plan *plan, uint32_t *samplenum);
void dpx_plan_destroy(dpx_plan *plan);

/* Launch the fused kernel on `hip_stream` (a hipStream_t; NULL = default stream).
 * d_in / d_out are device pointers, 16-byte aligned, holding plan.n_samples
 * samples in in_fmt / out_fmt.  Asynchronous: returns after the launch. */
int dpx_run_device(dpx_plan *plan, const void *d_in, int in_fmt, void *d_out, int out_fmt,
                   void *hip_stream);

/* ------------------------------------------------ streaming from host memory
 * What the reference's driver loop (src/main.rs:57-99: read a block from stdin, shift, write to
 * stdout) becomes when the blocks are gathered into slabs: a ring of pinned host slabs, each slab
 * one plan + one fused launch, so that filling, the two PCIe directions, the kernel and draining
 * overlap.  Zero-copy on the host side: the caller reads into / writes out of the pinned buffers.
 * The sample counter (`samplenr`, main.rs:60) is carried from slab to slab.  From pinned memory the
 * ring moves 11.8 Gsamples/s of i16 IQ (47 GB/s each way: 0.98 of the PCIe link) with slabs of
 * 16 MiB and more; how a slab crosses the link follows its size (small slabs: the kernel reads and
 * writes the pinned buffers itself, 13 us per 8 KiB slab) — doppler_hip_debug.h, dpx_stream_options.
 *
 *   dpx_stream_acquire  -> pinned input buffer of the next free slab (slab_bytes capacity); several slabs may be
 *                          acquired before the first is submitted (they can then be filled in parallel); when every
 *                          slab is in use it returns DPX_ERR_PLAN: drain one first
 *   dpx_stream_submit   -> the OLDEST acquired slab now holds in_bytes of IQ with the given constant-shift
 *                          segments (their sample counts must add up to in_bytes); asynchronous
 *   dpx_stream_next     -> oldest submitted slab not yet handed out: waits for it, returns its pinned output; several
 *                          outputs may be handed out before the first is released (they can be drained in parallel)
 *   dpx_stream_release  -> the OLDEST handed-out output has been consumed; its slab is free again
 * Outputs come back in submission order.
 * Threads: acquire + submit form the producer side, next the consumer side, release the recycling side; each side may
 * live on its own thread (one caller at a time per side), as in the `doppler` command. */
typedef struct dpx_stream dpx_stream;
int dpx_stream_create(dpx_ctx *ctx, int in_fmt, int out_fmt, uint32_t samplerate, uint32_t samplenum0,
                      size_t slab_bytes, int n_slabs, dpx_stream **stream);
/* The same ring over several GPUs of one node (the product form of the time-chunk sharding: slabs ARE time chunks).
 * ctxs: one context per GPU (a device may be listed more than once: separate contexts on it); slab k of the ring is
 * processed on context k mod n_ctx, so consecutive slabs run on consecutive GPUs and their PCIe copies and kernels
 * overlap.  The counter is carried on the host from slab to slab (closed form per slab: no GPU waits for another),
 * every GPU writes its output straight into its own pinned host slab (per-GPU D2H, no gather through one GPU), and
 * dpx_stream_next still hands the outputs back in submission order.  Tuning and libm choice are taken from ctxs[0]. */
int dpx_stream_create_multi(dpx_ctx *const *ctxs, int n_ctx, int in_fmt, int out_fmt, uint32_t samplerate,
                            uint32_t samplenum0, size_t slab_bytes, int slabs_per_ctx, dpx_stream **stream);
int dpx_stream_acquire(dpx_stream *s, void **pinned_in, size_t *capacity_bytes);
int dpx_stream_submit(dpx_stream *s, size_t in_bytes, const dpx_segment *segs, size_t n_segs);
int dpx_stream_next(dpx_stream *s, const void **pinned_out, size_t *out_bytes);
int dpx_stream_release(dpx_stream *s);
int dpx_stream_samplenum(const dpx_stream *s, uint32_t *samplenum);   /* counter after everything submitted */
void dpx_stream_destroy(dpx_stream *s);

/* Which build of glibc's sincosf the correctors reproduce bit-for-bit:
 * fma = 1 (default): the FMA build libm selects on every x86-64 CPU with FMA+AVX2;
 * fma = 0: the SSE2 build (differs on 34 of all 2^32 arguments). */
int dpx_set_libm_contraction(dpx_ctx *ctx, int fma);

/* What `(x * 32767.0) as i16` (reference src/main.rs:77-78) means when the product leaves the i16 range — rotated full-scale
 * samples reach sqrt(2) * 32767, and real recordings clip:
 *   DPX_CAST_SATURATE (default): truncate toward zero, saturate to [-32768, 32767], NaN -> 0 — the language's definition
 *       since Rust 1.45, i.e. what the reference does when built with any toolchain of the last years;
 *   DPX_CAST_LEGACY_X86: truncate toward zero, keep the low 16 bits (40000 -> -25536); NaN and |x| >= 2^31 -> 0 — the x86-64
 *       code a 2016 rustc (the toolchain of the reference's Cargo.lock) emitted for the then-undefined out-of-range case
 *       (`fptosi float to i16` = CVTTSS2SI into a 32-bit register, low half stored).  For byte-for-byte comparisons with
 *       output files produced by binaries of that time.
 * Applies to plans created afterwards and to the host-pointer operators; the mode is a launch-uniform flag of every kernel
 * (the same plans and rates as the default).  f32 output is not affected. */
#define DPX_CAST_SATURATE 0
#define DPX_CAST_LEGACY_X86 1
int dpx_set_i16_cast(dpx_ctx *ctx, int mode);

#ifdef __cplusplus
}
#endif
#endif
