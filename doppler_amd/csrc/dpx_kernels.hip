// dpx_kernels.hip — fused Doppler-shift kernels for MI355X (gfx950, CDNA4).
//
// One pass over HBM replaces the reference's three passes per 8 KiB block
// (unpack src/dsp.rs:85-115, mix src/dsp.rs:117-134, pack src/main.rs:72-94):
//     load 16-byte vectors of interleaved IQ  ->  unpack in registers
//     corrector(n) from an LDS table (one period) or evaluated on the fly
//     complex multiply with the reference's unfused f32 operation order
//     pack to i16 / f32  ->  16-byte stores
// The kernel is HBM-bandwidth bound by design: 8 B/sample for i16->i16.
// No MFMA: there is no contraction anywhere in this path.
//
// The sequential counter of dsp.rs:125-130 is replaced by the closed forms in
// dpx_types.h (DevSeg), so every sample's corrector depends only on its index.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/doppler_hip.h"
#include "dpx_sincos.cuh"
#include "dpx_types.h"

#pragma clang fp contract(off)

namespace dpx {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- sample math

// dsp.rs:92-93: ((hi as i16) << 8 | lo as i16) as f32 / 32768.  (exact: power of two)
__device__ __forceinline__ void unpack_i16(uint32_t w, float &re, float &im)
{
    re = (float)(int16_t)(w & 0xffffu) * 0x1p-15f;
    im = (float)(int16_t)(w >> 16) * 0x1p-15f;
}

// dsp.rs:123 with num-complex 0.1.35 Mul: (a*c - b*s, a*s + b*c); every product and
// the sum/difference individually rounded (Rust never contracts to fma).
__device__ __forceinline__ void mix(float a, float b, float c, float s, float &re, float &im)
{
    re = __fsub_rn(__fmul_rn(a, c), __fmul_rn(b, s));
    im = __fadd_rn(__fmul_rn(a, s), __fmul_rn(b, c));
}

// Rust `f32 as i16`: truncate toward zero, saturate, NaN -> 0.
__device__ __forceinline__ int f32_as_i16(float x)
{
    x = (x != x) ? 0.0f : x;
    x = fminf(fmaxf(x, -32768.0f), 32767.0f);
    return (int)x;
}

// main.rs:77-83: i = (re * 32767.0) as i16, little-endian I then Q.
__device__ __forceinline__ uint32_t pack_i16(float re, float im)
{
    const int i = f32_as_i16(__fmul_rn(re, 32767.0f));
    const int q = f32_as_i16(__fmul_rn(im, 32767.0f));
    return ((uint32_t)i & 0xffffu) | ((uint32_t)q << 16);
}

template <int FMT> struct Fmt;
template <> struct Fmt<DPX_FMT_I16> { static constexpr int kBytes = 4; static constexpr int kVecs = 1; };
template <> struct Fmt<DPX_FMT_F32> { static constexpr int kBytes = 8; static constexpr int kVecs = 2; };

// four consecutive samples of one lane, as raw 16-byte vectors
template <int FMT> struct Quad { u32x4 v[Fmt<FMT>::kVecs]; };

template <int FMT>
__device__ __forceinline__ Quad<FMT> load_quad(const uint8_t *base, uint64_t g)
{
    Quad<FMT> q;
    const u32x4 *p = reinterpret_cast<const u32x4 *>(base + g * Fmt<FMT>::kBytes);
#pragma unroll
    for (int i = 0; i < Fmt<FMT>::kVecs; ++i) q.v[i] = __builtin_nontemporal_load(p + i);
    return q;
}

template <int FMT>
__device__ __forceinline__ void store_quad(uint8_t *base, uint64_t g, const Quad<FMT> &q)
{
    u32x4 *p = reinterpret_cast<u32x4 *>(base + g * Fmt<FMT>::kBytes);
#pragma unroll
    for (int i = 0; i < Fmt<FMT>::kVecs; ++i) __builtin_nontemporal_store(q.v[i], p + i);
}

template <int FMT>
__device__ __forceinline__ void quad_get(const Quad<FMT> &q, int k, float &re, float &im)
{
    if constexpr (FMT == DPX_FMT_I16) {
        unpack_i16(q.v[0][k], re, im);
    } else {   // dsp.rs:108-109: the bytes ARE the f32
        re = __uint_as_float(q.v[k >> 1][(k & 1) * 2]);
        im = __uint_as_float(q.v[k >> 1][(k & 1) * 2 + 1]);
    }
}

template <int FMT>
__device__ __forceinline__ void quad_set(Quad<FMT> &q, int k, float re, float im)
{
    if constexpr (FMT == DPX_FMT_I16) {
        q.v[0][k] = pack_i16(re, im);
    } else {   // main.rs:91: raw reinterpret
        q.v[k >> 1][(k & 1) * 2] = __float_as_uint(re);
        q.v[k >> 1][(k & 1) * 2 + 1] = __float_as_uint(im);
    }
}

// one sample at a time: ragged tiles and stretch boundaries only
template <int FMT>
__device__ __forceinline__ void load_one(const uint8_t *base, uint64_t g, float &re, float &im)
{
    if constexpr (FMT == DPX_FMT_I16) {
        unpack_i16(*reinterpret_cast<const uint32_t *>(base + g * 4), re, im);
    } else {
        const u32x2 w = *reinterpret_cast<const u32x2 *>(base + g * 8);
        re = __uint_as_float(w[0]);
        im = __uint_as_float(w[1]);
    }
}

template <int FMT>
__device__ __forceinline__ void store_one(uint8_t *base, uint64_t g, float re, float im)
{
    if constexpr (FMT == DPX_FMT_I16) {
        *reinterpret_cast<uint32_t *>(base + g * 4) = pack_i16(re, im);
    } else {
        u32x2 w;
        w[0] = __float_as_uint(re);
        w[1] = __float_as_uint(im);
        *reinterpret_cast<u32x2 *>(base + g * 8) = w;
    }
}

// counter value for sample j of a stretch (dpx_types.h)
__device__ __forceinline__ uint32_t counter_at(const DevSeg &sg, uint64_t j)
{
    if (sg.period == 0) return sg.n_start + (uint32_t)j;
    return (uint32_t)(((uint64_t)(sg.n_start - 1u) + j) % sg.period) + 1u;
}

// ------------------------------------------------------------- fused kernel
//
// Work decomposition: the stream is cut into tiles of BLOCK*4*U samples; tile t
// goes to workgroup t mod gridDim (block-cyclic, so the whole grid sweeps one
// contiguous window of HBM at a time).  A lane owns 4 consecutive samples per
// vector and U vectors per tile, all U loads issued before the first use.
template <int IN_FMT, int OUT_FMT, bool FMA, int U>
__global__ __launch_bounds__(kBlock) void shift_kernel(const uint8_t *__restrict__ in,
                                                       uint8_t *__restrict__ out,
                                                       const DevSeg *__restrict__ segs,
                                                       uint32_t n_segs, uint64_t n_samples)
{
    extern __shared__ float2 lut[];   // correctors (cos, sin) of one table period
    constexpr uint32_t SPL = kSamplesPerLane;
    constexpr uint32_t TILE = kBlock * SPL * U;
    const uint32_t tid = threadIdx.x;
    const uint64_t n_tiles = (n_samples + TILE - 1) / TILE;

    uint32_t si = 0;           // current stretch (uniform across the workgroup)
    uint32_t lut_owner = ~0u;  // stretch whose table is in LDS

    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t t0 = tile * TILE;
        while (si + 1 < n_segs && segs[si].first + segs[si].count <= t0) ++si;
        const DevSeg sg = segs[si];
        const bool whole = (t0 + TILE <= n_samples) && (t0 >= sg.first) &&
                           (t0 + TILE <= sg.first + sg.count);
        if (whole) {
            const uint64_t j0 = t0 - sg.first;

            // issue every load of this tile first: U x 16 B (i16) or 2U x 16 B (f32) per lane
            Quad<IN_FMT> qin[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                qin[u] = load_quad<IN_FMT>(in, t0 + (uint64_t)(u * kBlock + tid) * SPL);

            if (sg.lut_len != 0) {
                // ---- corrector table of one period in LDS
                const uint32_t L = sg.lut_len;
                if (lut_owner != si) {
                    __syncthreads();   // previous table no longer in use
                    for (uint32_t e = tid; e < L; e += kBlock) {
                        float c, s;
                        corrector<FMA>(sg.ratio, (e % sg.period) + 1u, c, s);
                        lut[e] = make_float2(c, s);
                    }
                    __syncthreads();
                    lut_owner = si;
                }
                const uint32_t tb = (uint32_t)(((uint64_t)(sg.n_start - 1u) + j0) % L);
                const uint32_t step = (kBlock * SPL) % L;
                uint32_t t = (tb + tid * SPL) % L;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    Quad<OUT_FMT> qo;
#pragma unroll
                    for (int k = 0; k < (int)SPL; ++k) {
                        uint32_t e = t + k;
                        e = (e >= L) ? e - L : e;
                        const float2 cs = lut[e];
                        float a, b, re, im;
                        quad_get<IN_FMT>(qin[u], k, a, b);
                        mix(a, b, cs.x, cs.y, re, im);
                        quad_set<OUT_FMT>(qo, k, re, im);
                    }
                    store_quad<OUT_FMT>(out, t0 + (uint64_t)(u * kBlock + tid) * SPL, qo);
                    t += step;
                    t = (t >= L) ? t - L : t;
                }
            } else {
                // ---- corrector evaluated per sample (periodic with period >= 4, or linear)
                const uint32_t P = sg.period;
                uint32_t t, step;
                if (P != 0) {
                    const uint32_t tb = (uint32_t)(((uint64_t)(sg.n_start - 1u) + j0) % P);
                    step = (kBlock * SPL) % P;
                    t = (tb + tid * SPL) % P;
                } else {
                    step = kBlock * SPL;
                    t = sg.n_start + (uint32_t)j0 + tid * SPL;   // the counter itself
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    Quad<OUT_FMT> qo;
#pragma unroll
                    for (int k = 0; k < (int)SPL; ++k) {
                        uint32_t n;
                        if (P != 0) {
                            uint32_t e = t + k;
                            e = (e >= P) ? e - P : e;
                            n = e + 1u;
                        } else {
                            n = t + k;
                        }
                        float c, s, a, b, re, im;
                        corrector<FMA>(sg.ratio, n, c, s);
                        quad_get<IN_FMT>(qin[u], k, a, b);
                        mix(a, b, c, s, re, im);
                        quad_set<OUT_FMT>(qo, k, re, im);
                    }
                    store_quad<OUT_FMT>(out, t0 + (uint64_t)(u * kBlock + tid) * SPL, qo);
                    t += step;
                    if (P != 0) t = (t >= P) ? t - P : t;
                }
            }
        } else {
            // ---- ragged tile: stream tail, or a tile that straddles stretches
            for (uint32_t o = tid; o < TILE; o += kBlock) {
                const uint64_t g = t0 + o;
                if (g >= n_samples) break;
                uint32_t s2 = si;
                while (s2 + 1 < n_segs && segs[s2].first + segs[s2].count <= g) ++s2;
                const DevSeg sx = segs[s2];
                float c, s, a, b, re, im;
                corrector<FMA>(sx.ratio, counter_at(sx, g - sx.first), c, s);
                load_one<IN_FMT>(in, g, a, b);
                mix(a, b, c, s, re, im);
                store_one<OUT_FMT>(out, g, re, im);
            }
        }
    }
}

// ------------------------------------------------------- auxiliary kernels

// calibration: the same 16-byte non-temporal, block-cyclic stream with no math
__global__ __launch_bounds__(kBlock) void copy_kernel(const u32x4 *__restrict__ in,
                                                      u32x4 *__restrict__ out, uint64_t n_vec)
{
    constexpr int U = 4;
    const uint64_t tile = (uint64_t)kBlock * U;
    for (uint64_t t0 = (uint64_t)blockIdx.x * tile; t0 < n_vec; t0 += (uint64_t)gridDim.x * tile) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = t0 + (uint64_t)u * kBlock + threadIdx.x;
            if (i < n_vec) v[u] = __builtin_nontemporal_load(in + i);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = t0 + (uint64_t)u * kBlock + threadIdx.x;
            if (i < n_vec) __builtin_nontemporal_store(v[u], out + i);
        }
    }
}

// dsp.rs:85-99 on its own (the fused kernel never materialises this)
__global__ __launch_bounds__(kBlock) void unpack_i16_kernel(const uint32_t *__restrict__ in,
                                                            float2 *__restrict__ out, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        float re, im;
        unpack_i16(in[i], re, im);
        out[i] = make_float2(re, im);
    }
}

// main.rs:72-87 on its own
__global__ __launch_bounds__(kBlock) void pack_i16_kernel(const float2 *__restrict__ in,
                                                          uint32_t *__restrict__ out, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const float2 z = in[i];
        out[i] = pack_i16(z.x, z.y);
    }
}

// complex.c:33-39 for a purely imaginary argument: z <- (cos z.im, sin z.im)
template <bool FMA>
__global__ __launch_bounds__(kBlock) void ccexpf_imag_kernel(float2 *__restrict__ z, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        float s, c;
        sincosf_glibc<FMA>(z[i].y, s, c);
        z[i] = make_float2(c, s);
    }
}

// ------------------------------------------------------------ launch wrappers

template <int IN_FMT, int OUT_FMT, bool FMA>
static int launch_u(const void *d_in, void *d_out, const DevSeg *d_segs, uint32_t n_segs,
                    uint64_t n_samples, const LaunchGeom &g, hipStream_t st)
{
    const uint8_t *in = static_cast<const uint8_t *>(d_in);
    uint8_t *out = static_cast<uint8_t *>(d_out);
    switch (g.unroll) {
    case 1: shift_kernel<IN_FMT, OUT_FMT, FMA, 1><<<g.grid, kBlock, g.lds_bytes, st>>>(in, out, d_segs, n_segs, n_samples); break;
    case 2: shift_kernel<IN_FMT, OUT_FMT, FMA, 2><<<g.grid, kBlock, g.lds_bytes, st>>>(in, out, d_segs, n_segs, n_samples); break;
    case 4: shift_kernel<IN_FMT, OUT_FMT, FMA, 4><<<g.grid, kBlock, g.lds_bytes, st>>>(in, out, d_segs, n_segs, n_samples); break;
    case 8: shift_kernel<IN_FMT, OUT_FMT, FMA, 8><<<g.grid, kBlock, g.lds_bytes, st>>>(in, out, d_segs, n_segs, n_samples); break;
    default: return DPX_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;
}

template <bool FMA>
static int launch_f(const void *d_in, int in_fmt, void *d_out, int out_fmt, const DevSeg *d_segs,
                    uint32_t n_segs, uint64_t n_samples, const LaunchGeom &g, hipStream_t st)
{
    if (in_fmt == DPX_FMT_I16 && out_fmt == DPX_FMT_I16) return launch_u<DPX_FMT_I16, DPX_FMT_I16, FMA>(d_in, d_out, d_segs, n_segs, n_samples, g, st);
    if (in_fmt == DPX_FMT_I16 && out_fmt == DPX_FMT_F32) return launch_u<DPX_FMT_I16, DPX_FMT_F32, FMA>(d_in, d_out, d_segs, n_segs, n_samples, g, st);
    if (in_fmt == DPX_FMT_F32 && out_fmt == DPX_FMT_I16) return launch_u<DPX_FMT_F32, DPX_FMT_I16, FMA>(d_in, d_out, d_segs, n_segs, n_samples, g, st);
    if (in_fmt == DPX_FMT_F32 && out_fmt == DPX_FMT_F32) return launch_u<DPX_FMT_F32, DPX_FMT_F32, FMA>(d_in, d_out, d_segs, n_segs, n_samples, g, st);
    return DPX_ERR_ARG;
}

int launch_shift(const void *d_in, int in_fmt, void *d_out, int out_fmt, const DevSeg *d_segs,
                 uint32_t n_segs, uint64_t n_samples, bool fma, const LaunchGeom &g, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    return fma ? launch_f<true>(d_in, in_fmt, d_out, out_fmt, d_segs, n_segs, n_samples, g, st)
               : launch_f<false>(d_in, in_fmt, d_out, out_fmt, d_segs, n_segs, n_samples, g, st);
}

static int aux_grid(uint64_t n)
{
    uint64_t b = (n + kBlock - 1) / kBlock;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int launch_copy(const void *d_in, void *d_out, uint64_t n_bytes, int grid, void *stream)
{
    copy_kernel<<<grid, kBlock, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const u32x4 *>(d_in), static_cast<u32x4 *>(d_out), n_bytes / 16);
    return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;
}

int launch_unpack_i16(const void *d_in, void *d_out, uint64_t n, void *stream)
{
    unpack_i16_kernel<<<aux_grid(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const uint32_t *>(d_in), static_cast<float2 *>(d_out), n);
    return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;
}

int launch_pack_i16(const void *d_in, void *d_out, uint64_t n, void *stream)
{
    pack_i16_kernel<<<aux_grid(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const float2 *>(d_in), static_cast<uint32_t *>(d_out), n);
    return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;
}

int launch_ccexpf_imag(void *d_z, uint64_t n, bool fma, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (fma) ccexpf_imag_kernel<true><<<aux_grid(n), kBlock, 0, st>>>(static_cast<float2 *>(d_z), n);
    else     ccexpf_imag_kernel<false><<<aux_grid(n), kBlock, 0, st>>>(static_cast<float2 *>(d_z), n);
    return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;
}

}  // namespace dpx
