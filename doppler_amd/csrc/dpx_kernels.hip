// dpx_kernels.hip — fused Doppler-shift kernels for MI355X (gfx950, CDNA4).
//
// One pass over HBM replaces the reference's three passes per 8 KiB block
// (unpack src/dsp.rs:85-115, mix src/dsp.rs:117-134, pack src/main.rs:72-94):
//     load 16-byte vectors of interleaved IQ  ->  unpack in registers
//     corrector(n) from a precomputed table (one period) or evaluated on the fly
//     complex multiply with the reference's unfused f32 operation order
//       (three packed instructions, each product and the sum rounded on its own)
//     pack to i16 / f32  ->  16-byte stores
// The kernel is HBM-bandwidth bound by design: 8 B/sample for i16->i16.
// No MFMA: there is no contraction anywhere in this path.
//
// The sequential counter of dsp.rs:125-130 is replaced by the closed forms in
// dpx_types.h (DevSeg), so every sample's corrector depends only on its index.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/doppler_hip.h"
#include "dpx_sincos.h"
#include "dpx_types.h"

#pragma clang fp contract(off)

// gfx950 only: the inline assembly (v_pk_mul_f32 op_sel, v_bitop3_b32, v_cvt_pk_i16_i32), the kernarg preload and the
// LDS layouts below are written for CDNA4's 64-wide wavefronts; there is no other device path.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "dpx_kernels.hip is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

namespace dpx {

static_assert(kLargeQuickEnd == kThetaHuge, "the planner's path bounds (dpx_types.h) are those of the device sincos (dpx_sincos.h)");

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- sample math

// dsp.rs:92-93: ((hi as i16) << 8 | lo as i16) as f32 / 32768.  (exact: power of two)
//
// RAW (the fused i16 -> i16 path only): the division is left out here and folded into the constant of the pack below.
// Scaling by 2^-15 is exact and commutes with every rounding in between (the two products and the sum of the mix, each
// rounded on its own): with I, Q the integers, fl(fl(I c) - fl(Q s)) 2^-15 IS the reference's fl(fl(I' c) - fl(Q' s)) for
// I' = I 2^-15, and fl(X 2^-15 * 32767) = fl(X * (32767 * 2^-15)) — the same real product, rounded once, the constant
// exact in f32.  Only below 2^-126 (a denormal intermediate: |corrector| < 2^-111) could the two differ, and such a value
// truncates to the i16 0 either way.  One packed multiply less per sample.
template <bool RAW = false>
__device__ __forceinline__ void unpack_i16(uint32_t w, float &re, float &im)
{
    if constexpr (RAW) {
        re = (float)(int16_t)(w & 0xffffu);
        im = (float)(int16_t)(w >> 16);
    } else {
        re = (float)(int16_t)(w & 0xffffu) * 0x1p-15f;
        im = (float)(int16_t)(w >> 16) * 0x1p-15f;
    }
}
// the i16 -> i16 pair computes on the unscaled integers
template <int IN_FMT, int OUT_FMT> constexpr bool kRawI16 = IN_FMT == DPX_FMT_I16 && OUT_FMT == DPX_FMT_I16;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// dsp.rs:123 with num-complex 0.1.35 Mul: (a*c - b*s, a*s + b*c); every product and
// the sum/difference individually rounded (Rust never contracts to fma).
// Three packed instructions on the register pairs (a, b) and (c, s) as they come out of the loads:
//   t0 = (a*c, a*s)     t1 = (b*s, b*c)     result = (t0.lo - t1.lo, t0.hi + t1.hi)
// op_sel / op_sel_hi pick the half of each source pair that feeds the low / high result lane; neg_lo
// negates the low-lane input of the second operand, so a*c + (-(b*s)) is the IEEE subtraction itself.
__device__ __forceinline__ f32x2 mix2(f32x2 ab, f32x2 cs)
{
    f32x2 t0, t1, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t0) : "v"(ab), "v"(cs));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t1) : "v"(ab), "v"(cs));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(t0), "v"(t1));
    return r;
}

__device__ __forceinline__ void mix(float a, float b, float c, float s, float &re, float &im)
{
    const f32x2 r = mix2(f32x2{a, b}, f32x2{c, s});
    re = r.x;
    im = r.y;
}

// Rust `f32 as i16`: truncate toward zero, saturate, NaN -> 0.  v_cvt_i32_f32 truncates, saturates to the
// i32 range and turns NaN into 0; v_cvt_pk_i16_i32 then saturates both values to i16 and packs them.
__device__ __forceinline__ int f32_as_i32_sat(float x)
{
    int i;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(i) : "v"(x));
    return i;
}

// main.rs:77-83: i = (re * 32767.0) as i16, little-endian I then Q.
// LEGACY (dpx_set_i16_cast(DPX_CAST_LEGACY_X86); every kernel, round 4): the cast as a 2016 rustc compiled it for
// x86-64 — CVTTSS2SI into a 32-bit register, low half kept: truncate, then wrap modulo 2^16; NaN and |x| >= 2^31 give
// 0x80000000, low half 0.  v_cvt_i32_f32 differs from CVTTSS2SI only at x >= 2^31 (0x7fffffff: low half 0xffff).
// The mode reaches a kernel as a launch-uniform flag; the kernels branch on it ONCE per row / per four samples, around
// two instantiations of their pack code (a test inside every pack would cut the mix of four samples into pieces the
// scheduler cannot interleave: measured 2-3 points on the span kernel's short matrices).
template <bool RAW = false, bool LEGACY = false>
__device__ __forceinline__ uint32_t pack_i16(float re, float im)
{
    constexpr float K = RAW ? 0x1.fffcp-1f /* 32767 / 32768, exact */ : 32767.0f;
    const f32x2 sc = f32x2{re, im} * K;                  // one v_pk_mul_f32, each product rounded on its own
    if constexpr (LEGACY && RAW) {
        // i16 in: |I|, |Q| <= 32768 and |c|, |s| <= 1, so |sc| < 65537 — no NaN, nothing near 2^31: truncate and wrap,
        // as many instructions as the default cast (the replay under this mode ran 3 points behind with the tests below)
        return ((uint32_t)f32_as_i32_sat(sc.x) & 0xffffu) | ((uint32_t)f32_as_i32_sat(sc.y) << 16);
    } else if constexpr (LEGACY) {
        const uint32_t i = sc.x >= 2147483648.0f ? 0u : (uint32_t)f32_as_i32_sat(sc.x);
        const uint32_t q = sc.y >= 2147483648.0f ? 0u : (uint32_t)f32_as_i32_sat(sc.y);
        return (i & 0xffffu) | (q << 16);
    } else {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        const s16x2 p = __builtin_amdgcn_cvt_pk_i16(f32_as_i32_sat(sc.x), f32_as_i32_sat(sc.y));
        return __builtin_bit_cast(uint32_t, p);
    }
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

// the same at a byte address: the tile kernel keeps the tile's base in scalar registers and adds a 32-bit lane offset
// (saddr + voffset + immediate instead of 64-bit address arithmetic per lane and vector)
template <int FMT>
__device__ __forceinline__ Quad<FMT> load_quad_at(const uint8_t *p)
{
    Quad<FMT> q;
    const u32x4 *v = reinterpret_cast<const u32x4 *>(p);
#pragma unroll
    for (int i = 0; i < Fmt<FMT>::kVecs; ++i) q.v[i] = __builtin_nontemporal_load(v + i);
    return q;
}

template <int FMT>
__device__ __forceinline__ void store_quad_at(uint8_t *p, const Quad<FMT> &q)
{
    u32x4 *v = reinterpret_cast<u32x4 *>(p);
#pragma unroll
    for (int i = 0; i < Fmt<FMT>::kVecs; ++i) __builtin_nontemporal_store(q.v[i], v + i);
}

template <int FMT, bool RAW = false>
__device__ __forceinline__ void quad_get(const Quad<FMT> &q, int k, float &re, float &im)
{
    if constexpr (FMT == DPX_FMT_I16) {
        unpack_i16<RAW>(q.v[0][k], re, im);
    } else {   // dsp.rs:108-109: the bytes ARE the f32
        re = __uint_as_float(q.v[k >> 1][(k & 1) * 2]);
        im = __uint_as_float(q.v[k >> 1][(k & 1) * 2 + 1]);
    }
}

template <int FMT, bool RAW = false, bool LEGACY = false>
__device__ __forceinline__ void quad_set(Quad<FMT> &q, int k, float re, float im)
{
    if constexpr (FMT == DPX_FMT_I16) {
        q.v[0][k] = pack_i16<RAW, LEGACY>(re, im);
    } else {   // main.rs:91: raw reinterpret
        q.v[k >> 1][(k & 1) * 2] = __float_as_uint(re);
        q.v[k >> 1][(k & 1) * 2 + 1] = __float_as_uint(im);
    }
}

// four samples at once, the cast mode decided by ONE uniform branch (see pack_i16)
template <int FMT, bool RAW>
__device__ __forceinline__ void quad_set4(Quad<FMT> &q, const float (&re)[4], const float (&im)[4], bool legacy)
{
    if constexpr (FMT == DPX_FMT_I16) {
        if (legacy) {
#pragma unroll
            for (int k = 0; k < 4; ++k) quad_set<FMT, RAW, true>(q, k, re[k], im[k]);
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) quad_set<FMT, RAW, false>(q, k, re[k], im[k]);
}

// S packed i16 samples of a row vector, the cast mode decided by ONE uniform branch (see pack_i16)
template <bool RAW, bool LEGACY, int S>
__device__ __forceinline__ void pack_row(const float (&re)[S], const float (&im)[S], uint32_t (&o)[S])
{
#pragma unroll
    for (int k = 0; k < S; ++k) o[k] = pack_i16<RAW, LEGACY>(re[k], im[k]);
}
template <bool RAW, int S>
__device__ __forceinline__ void pack_row(const float (&re)[S], const float (&im)[S], uint32_t (&o)[S], bool legacy)
{
    if (legacy) pack_row<RAW, true, S>(re, im, o);
    else        pack_row<RAW, false, S>(re, im, o);
}

// one sample at a time: ragged tiles and stretch boundaries only
template <int FMT, bool RAW = false>
__device__ __forceinline__ void load_one(const uint8_t *base, uint64_t g, float &re, float &im)
{
    if constexpr (FMT == DPX_FMT_I16) {
        unpack_i16<RAW>(*reinterpret_cast<const uint32_t *>(base + g * 4), re, im);
    } else {
        const u32x2 w = *reinterpret_cast<const u32x2 *>(base + g * 8);
        re = __uint_as_float(w[0]);
        im = __uint_as_float(w[1]);
    }
}

template <int FMT, bool RAW = false>
__device__ __forceinline__ void store_one(uint8_t *base, uint64_t g, float re, float im, bool legacy = false)
{
    if constexpr (FMT == DPX_FMT_I16) {
        *reinterpret_cast<uint32_t *>(base + g * 4) = legacy ? pack_i16<RAW, true>(re, im) : pack_i16<RAW, false>(re, im);
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

// ---------------------------------------------------------------- the kernels
//
// Shape of the launch (measured, profiles/r01_membench.md): on MI355X a
// 1 GiB -> 1 GiB stream runs at 6.6-6.7 TB/s when it is issued as one-shot
// workgroups of 1-4 wavefronts that touch 1-4 KiB and exit, against 6.0-6.1 TB/s
// for every persistent grid-stride shape tried, and every extra vector-memory
// instruction per wavefront (a corrector table read) costs a few percent.  So:
//   * no persistent loop, tiny workgroups, addresses from blockIdx alone so the
//     sample loads are issued first;
//   * correctors of a periodic stretch are tabulated ONCE per plan in global
//     memory (one period) by build_lut_kernel with the bit-exact sincos;
//   * rows kernel (const mode): the stream is viewed as a matrix whose row
//     length is a multiple of the period, so the two rows a wavefront handles
//     share one 32-byte table read per lane and need no phase arithmetic;
//   * span kernel (track mode, hundreds of stretches in one launch): same idea
//     with rows shifted onto 128-byte lines and the 288 correctors of a column
//     window evaluated by the workgroup and shared by its rows through LDS;
//   * tile kernel: whatever the two above leave.

// ---- per-sample evaluation (ragged ranges and stretch boundaries)
template <int IN_FMT, int OUT_FMT, bool FMA>
__device__ __forceinline__ void one_sample(const uint8_t *in, uint8_t *out, const DevSeg *segs,
                                           uint32_t n_segs, uint32_t si, uint64_t g, bool legacy)
{
    while (si + 1 < n_segs && segs[si].first + segs[si].count <= g) ++si;
    const DevSeg sx = segs[si];
    float c, s, a, b, re, im;
    corrector<FMA>(sx.ratio, counter_at(sx, g - sx.first), c, s);
    load_one<IN_FMT, kRawI16<IN_FMT, OUT_FMT>>(in, g, a, b);
    mix(a, b, c, s, re, im);
    store_one<OUT_FMT, kRawI16<IN_FMT, OUT_FMT>>(out, g, re, im, legacy);
}

// ---- rows kernel: one wavefront, R rows of one tabulated periodic stretch
// S samples per lane per row: the wider of the two sides moves as one 16-byte vector per lane,
// the narrower side as 16 or 8 bytes (measured: 8-byte accesses on the narrow side are free,
// two 16-byte stores at a 32-byte lane stride halve the rate).
template <int IN_FMT, int OUT_FMT> struct RowVec {
    static constexpr int S = (IN_FMT == DPX_FMT_I16 && OUT_FMT == DPX_FMT_I16) ? 4 : 2;
};

// Argument order matters: the first 16 dwords are preloaded into SGPRs at wavefront
// launch (-amdgpu-kernarg-preload-count=16), and they are exactly what the matrix
// path needs — a one-shot wavefront issues its loads without waiting for any s_load.
template <int IN_FMT, int OUT_FMT, bool FMA, int R, bool COMPUTE>
__global__ __launch_bounds__(kRowsLanes) void rows_kernel(const uint8_t *__restrict__ in,
                                                          uint8_t *__restrict__ out,
                                                          const float2 *__restrict__ tab,   // table, origin = sample A
                                                          uint64_t A, uint32_t L, uint32_t cols,
                                                          uint64_t div_m, uint32_t div_s,
                                                          uint32_t n_extra, uint32_t P,
                                                          float ratio,                       // COMPUTE only
                                                          // ---- not preloaded
                                                          uint32_t idx0,                     // COMPUTE only
                                                          const DevSeg *__restrict__ segs,
                                                          RowsArgs ra)
{
    constexpr int S = RowVec<IN_FMT, OUT_FMT>::S;
    constexpr int IB = Fmt<IN_FMT>::kBytes, OB = Fmt<OUT_FMT>::kBytes;
    const uint32_t lane = threadIdx.x;
    const bool legacy = (div_s >> 31) != 0;                    // dpx_set_i16_cast, in the spare bits of a preloaded argument

    // the workgroups of the ragged ranges come first in the grid: their sincos work then overlaps the memory-bound
    // matrix instead of forming a tail
    if (blockIdx.x >= n_extra) {
        const uint32_t b = blockIdx.x - n_extra;
        // (row group, column slice) = divmod(b, cols), exact magic-number division
        const uint32_t rg = (uint32_t)(((uint64_t)b * div_m) >> (div_s & 63u));
        const uint32_t col = b - rg * cols;
        const uint32_t cs0 = (col * kRowsLanes + lane) * S;   // first sample of this lane in the row
        if (cs0 >= L) return;                                  // ragged last column slice
        const uint64_t g0 = A + (uint64_t)rg * (R * (uint64_t)L) + cs0;

        constexpr int QW = S * IB / 4;                         // input dwords per lane per row: 4 or 2
        typedef uint32_t qvec __attribute__((ext_vector_type(QW)));
        qvec q[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
            q[r] = __builtin_nontemporal_load(reinterpret_cast<const qvec *>(in + (g0 + (uint64_t)r * L) * IB));

        // S correctors, shared by the R rows.  The table holds ONE period (+3 entries so that a group of 4 never
        // wraps), origin = sample A; the row length is a multiple of the period, so the index is the column
        // modulo the period — the column itself when a row is exactly one period (the headline case).
        const uint32_t e0 = (L == P) ? cs0 : cs0 % P;
        u32x4 t[S / 2];
        if constexpr (COMPUTE) {
            // the same S (cos, sin) pairs, evaluated here: counter of column c = ((idx0 + c) mod P) + 1, P >= 4
            uint32_t e = idx0 + e0;
            e = e >= P ? e - P : e;
            uint32_t n[S];
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const uint32_t ek = e + (uint32_t)k;
                n[k] = (ek >= P ? ek - P : ek) + 1u;
            }
            if constexpr (S == 4) {
                f32x2 cs[4];
                corrector4<FMA>(ratio, n, cs);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    t[k >> 1][(k & 1) * 2] = __float_as_uint(cs[k].x);
                    t[k >> 1][(k & 1) * 2 + 1] = __float_as_uint(cs[k].y);
                }
            } else {
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    float c, sn;
                    corrector<FMA>(ratio, n[k], c, sn);
                    t[k >> 1][(k & 1) * 2] = __float_as_uint(c);
                    t[k >> 1][(k & 1) * 2 + 1] = __float_as_uint(sn);
                }
            }
        } else {
            const u32x4 *tp = reinterpret_cast<const u32x4 *>(tab + e0);
#pragma unroll
            for (int i = 0; i < S / 2; ++i) t[i] = tp[i];
        }

#pragma unroll
        for (int r = 0; r < R; ++r) {
            float re[S], im[S];
#pragma unroll
            for (int k = 0; k < S; ++k) {
                float a, bq;
                if constexpr (IN_FMT == DPX_FMT_I16) unpack_i16<kRawI16<IN_FMT, OUT_FMT>>(q[r][k], a, bq);
                else { a = __uint_as_float(q[r][2 * k]); bq = __uint_as_float(q[r][2 * k + 1]); }
                const float c = __uint_as_float(t[k >> 1][(k & 1) * 2]);
                const float s = __uint_as_float(t[k >> 1][(k & 1) * 2 + 1]);
                mix(a, bq, c, s, re[k], im[k]);
            }
            uint8_t *op = out + (g0 + (uint64_t)r * L) * OB;
            if constexpr (OUT_FMT == DPX_FMT_I16) {
                uint32_t pk[S];
                pack_row<kRawI16<IN_FMT, OUT_FMT>, S>(re, im, pk, legacy);
                if constexpr (S == 4) {
                    const u32x4 o = {pk[0], pk[1], pk[2], pk[3]};
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(op));
                } else {
                    const u32x2 o = {pk[0], pk[1]};
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x2 *>(op));
                }
            } else {
#pragma unroll
                for (int i = 0; i < S / 2; ++i) {
                    u32x4 o;
                    o[0] = __float_as_uint(re[2 * i]);     o[1] = __float_as_uint(im[2 * i]);
                    o[2] = __float_as_uint(re[2 * i + 1]); o[3] = __float_as_uint(im[2 * i + 1]);
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(op) + i);
                }
            }
        }
    } else {
        // ragged ranges [r0, A) and [B, r1): 4 samples per lane, evaluated one by one
        const uint64_t head = ra.A - ra.r0;
        const uint64_t total = head + (ra.r1 - ra.B);
        const uint64_t e0 = (uint64_t)blockIdx.x * (kRowsLanes * 4);
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const uint64_t idx = e0 + (uint64_t)k * kRowsLanes + lane;
            if (idx >= total) break;
            const uint64_t g = idx < head ? ra.r0 + idx : ra.B + (idx - head);
            one_sample<IN_FMT, OUT_FMT, FMA>(in, out, segs, ra.n_segs, ra.seg_lo, g, legacy);
        }
    }
}

// ---- tile kernel: any mixture of stretches; workgroup b handles tile tile_lo + b and exits
template <int IN_FMT, int OUT_FMT, bool FMA, int BLOCK, int V>
__global__ __launch_bounds__(BLOCK) void tile_kernel(const uint8_t *__restrict__ in,
                                                     uint8_t *__restrict__ out,
                                                     const DevSeg *__restrict__ segs,
                                                     uint32_t n_segs,
                                                     const uint32_t *__restrict__ hint,
                                                     const float2 *__restrict__ lut_pool,
                                                     TileArgs ta)
{
    constexpr uint32_t SPL = kSamplesPerLane;
    constexpr uint32_t TILE = BLOCK * SPL * V;
    const uint32_t tid = threadIdx.x;
    const bool legacy = ta.legacy != 0;                   // dpx_set_i16_cast: uniform
    const uint64_t tile = ta.tile_lo + blockIdx.x;
    const uint64_t t0 = tile * TILE;
    const bool in_mask = t0 >= ta.m0 && t0 + TILE <= ta.m1;

    // request this lane's input right away; nothing below is needed for the addresses.  Tile base: uniform (scalar
    // registers); lane offset: 32 bits; the V vectors of a lane: immediate offsets.
    constexpr uint32_t IBq = Fmt<IN_FMT>::kBytes * SPL, OBq = Fmt<OUT_FMT>::kBytes * SPL;      // bytes of four samples
    const uint8_t *tin = in + t0 * Fmt<IN_FMT>::kBytes;
    uint8_t *tout = out + t0 * Fmt<OUT_FMT>::kBytes;
    const uint32_t lane_in = tid * IBq, lane_out = tid * OBq;
    // f32 output (kPairs): a lane's four samples are two PAIRS, BLOCK * 2 samples apart — each vector of a wavefront
    // instruction is then adjacent to its neighbours' (16-byte stores: a KiB per instruction) instead of every other
    // 16 bytes of two KiB (16-byte accesses at a 32-byte lane stride: 5.0 TB/s where the span kernel's f32 -> i16 rows
    // were first tried that way, and 51-68 % of the peak on this kernel's per-sample path).  The input side follows:
    // two 16-byte loads (f32) or two 8-byte loads (i16: 512 contiguous bytes per instruction, what the rows kernel reads
    // for this pair).
    // f32 -> i16 (kXpose): pairs as well, 128 samples apart inside the 256 samples of the lane's WAVEFRONT, so that the
    // packed results can change lanes through a wavefront-private KiB of LDS and leave as one 16-byte store per lane
    // (pairs stored as they are would be 8-byte stores: 2.4 TB/s in the span kernel, hence its WalkVec::kTranspose).
    // Sample k of vector v sits at tile offset lane_sample(v, k).  (256 lanes x one vector use 72-78 vector registers,
    // six wavefronts per SIMD; a build held to eight by amdgpu_waves_per_eu spills and loses 1-3 points.)
    constexpr bool kPairs = OUT_FMT == DPX_FMT_F32;
    constexpr bool kXpose = IN_FMT == DPX_FMT_F32 && OUT_FMT == DPX_FMT_I16;
    constexpr bool kTwoPairs = kPairs || kXpose;          // a lane's samples: two pairs (else four consecutive ones)
    const uint32_t wv = tid >> 6, ln = tid & 63u;
    __shared__ __attribute__((aligned(16))) uint32_t xpose_lds[kXpose ? BLOCK * SPL : 4];
    auto lane_sample = [&](int v, int k) -> uint32_t {
        if constexpr (kPairs) return (uint32_t)v * (BLOCK * SPL) + (uint32_t)(k >> 1) * (BLOCK * 2u) + tid * 2u + (uint32_t)(k & 1);
        else if constexpr (kXpose) return (uint32_t)v * (BLOCK * SPL) + wv * 256u + (uint32_t)(k >> 1) * 128u + ln * 2u + (uint32_t)(k & 1);
        else                  return (uint32_t)(v * BLOCK + tid) * SPL + (uint32_t)k;
    };
    auto tile_load = [&](int v) -> Quad<IN_FMT> {
        if constexpr (kPairs) {
            Quad<IN_FMT> q;
            const uint8_t *pv = tin + (uint32_t)v * (BLOCK * IBq);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (IN_FMT == DPX_FMT_F32) {
                    q.v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(pv + ((uint32_t)i * (BLOCK * 16u) + tid * 16u)));
                } else {
                    const u32x2 h = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pv + ((uint32_t)i * (BLOCK * 8u) + tid * 8u)));
                    q.v[0][2 * i] = h[0];
                    q.v[0][2 * i + 1] = h[1];
                }
            }
            return q;
        } else if constexpr (kXpose) {
            Quad<IN_FMT> q;
            const uint8_t *pv = tin + ((uint32_t)v * (BLOCK * IBq) + wv * 2048u + ln * 16u);
#pragma unroll
            for (int i = 0; i < 2; ++i) q.v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(pv + (uint32_t)i * 1024u));
            return q;
        } else {
            return load_quad_at<IN_FMT>(tin + (lane_in + (uint32_t)v * (BLOCK * IBq)));
        }
    };
    auto tile_store = [&](int v, const Quad<OUT_FMT> &q) {
        if constexpr (kPairs) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_nontemporal_store(q.v[i], reinterpret_cast<u32x4 *>(tout + ((uint32_t)v * (BLOCK * OBq) + (uint32_t)i * (BLOCK * 16u) + tid * 16u)));
        } else if constexpr (kXpose) {
            // q.v[0] = {pair A, pair B} of this lane; the wavefront's 256 packed samples in order, then 16 bytes per lane
            // (same wavefront: LDS operations execute in order; the barrier keeps the compiler from moving them)
            uint32_t *xrow = xpose_lds + wv * 256u;
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<u32x2 *>(xrow + ln * 2u) = u32x2{q.v[0][0], q.v[0][1]};
            *reinterpret_cast<u32x2 *>(xrow + 128u + ln * 2u) = u32x2{q.v[0][2], q.v[0][3]};
            __builtin_amdgcn_wave_barrier();
            u32x4 o = *reinterpret_cast<const u32x4 *>(xrow + ln * 4u);
            asm volatile("" : "+v"(o));
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(tout + ((uint32_t)v * (BLOCK * OBq) + wv * 1024u + ln * 16u)));
        } else {
            store_quad_at<OUT_FMT>(tout + (lane_out + (uint32_t)v * (BLOCK * OBq)), q);
        }
    };
    Quad<IN_FMT> qin[V];
    if (in_mask) {
#pragma unroll
        for (int v = 0; v < V; ++v) qin[v] = tile_load(v);
    }

    // stretch holding the first produced sample (uniform: scalar loads)
    const uint64_t gs = t0 > ta.m0 ? t0 : ta.m0;
    uint32_t si = hint[gs >> kHintShift];
    while (si + 1 < n_segs && segs[si].first + segs[si].count <= gs) ++si;
    const DevSeg sg = segs[si];
    const bool whole = in_mask && (t0 >= sg.first) && (t0 + TILE <= sg.first + sg.count);

    if (whole && sg.lut_len != 0 && (sg.flags & kSegTileTable)) {
        // ---- tabulated correctors: phase of t0 within the period, then straight indexing
        const uint32_t P = sg.period;
        // (c0 + t0) mod P with t0 = tile * TILE; 32-bit while (tile mod P) * tmod < 2^18 * 2^11
        uint32_t ph;
        if (P <= (1u << 18)) ph = sg.c0 + (((uint32_t)tile % P) * sg.tmod) % P;   // tile < 2^32 (checked on the host)
        else                 ph = sg.c0 + (uint32_t)(t0 % P);
        ph = (ph >= P) ? ph - P : ph;
        const float2 *tab = lut_pool + sg.lut_off + ph;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float2 cs[SPL];
#pragma unroll
            for (int k = 0; k < (int)SPL; ++k) cs[k] = tab[lane_sample(v, k)];
            Quad<OUT_FMT> qo;
            float re[SPL], im[SPL];
#pragma unroll
            for (int k = 0; k < (int)SPL; ++k) {
                float a, b;
                quad_get<IN_FMT, kRawI16<IN_FMT, OUT_FMT>>(qin[v], k, a, b);
                mix(a, b, cs[k].x, cs[k].y, re[k], im[k]);
            }
            quad_set4<OUT_FMT, kRawI16<IN_FMT, OUT_FMT>>(qo, re, im, legacy);
            tile_store(v, qo);
        }
    } else if (whole && (sg.period == 0 || sg.period >= 4)) {
        // ---- sincos per sample: periodic with period >= 4, or linear.  A lane's four consecutive counters go
        // through corrector4 (packed theta products, one fast-path decision per four samples).
        const uint32_t P = sg.period;
        const uint64_t j0 = t0 - sg.first;
        uint32_t base;   // periodic: phase of t0 in [0, P); linear: the counter itself
        if (P != 0) {
            // (a 64-bit by 32-bit remainder is seventeen vector instructions per wavefront even for uniform operands; a
            // stream below 2^32 samples past the stretch's start — any stream a GPU holds — takes the 32-bit one)
            const uint64_t s64 = (uint64_t)(sg.n_start - 1u) + j0;
            base = (s64 >> 32) == 0 ? (uint32_t)s64 % P : (uint32_t)(s64 % P);
        } else {
            base = sg.n_start + (uint32_t)j0;
        }
        // whether the counter wraps inside this tile is the same for all its lanes (base is the tile's): almost every
        // tile takes the path with one addition per counter
        const bool wraps = P != 0 && base + TILE > P;
        const bool small = base < (1u << 24) - TILE - 1u;                 // fl32(n0 + k) = fl32(n0) + k for the whole tile
        // sincosf's argument path for the whole tile, from the stretch's three counters (scalar comparisons)
        int path = kPathAny;
        if (!wraps) {
            const uint32_t n_lo = P == 0 ? base : base + 1u, n_hi = n_lo + (TILE - 1u);
            if (n_hi >= n_lo) {
                if (n_lo >= sg.n_plain && n_hi < sg.n_large) path = kPathPlain;
                else if (n_lo >= sg.n_large && n_hi < sg.n_huge) path = kPathLarge;
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            uint32_t t = base + lane_sample(v, 0);                      // periodic: < P + TILE
            const uint32_t tb = base + lane_sample(v, 2);               // pairs: the lane's second pair
            f32x2 cs[SPL];
            if (!wraps) {
                const uint32_t n0 = P == 0 ? t : t + 1u;                // u32 arithmetic wraps like the reference's `+= 1`
                const uint32_t n2 = kTwoPairs ? (P == 0 ? tb : tb + 1u) : n0 + 2u;
                if (small) {                                             // every counter of the tile below 2^24 (uniform)
                    if constexpr (kTwoPairs) corrector4_pairs<FMA>(sg.ratio, n0, n2, cs, path);
                    else                  corrector4_consecutive<FMA>(sg.ratio, n0, cs, path);
                } else {
                    const uint32_t n[SPL] = {n0, n0 + 1u, n2, n2 + 1u};
                    corrector4<FMA>(sg.ratio, n, cs, path);
                }
            } else {
                uint32_t n[SPL];
                uint32_t tt[2] = {t, tb};
#pragma unroll
                for (int h = 0; h < (kTwoPairs ? 2 : 1); ++h) {
                    if (P >= TILE) tt[h] = tt[h] >= P ? tt[h] - P : tt[h];   // at most one wrap (uniform branch)
                    else           tt[h] %= P;
                }
#pragma unroll
                for (int k = 0; k < (int)SPL; ++k) {
                    const uint32_t e = kTwoPairs ? tt[k >> 1] + (uint32_t)(k & 1) : tt[0] + (uint32_t)k;   // P >= 4: at most one wrap
                    n[k] = (e >= P ? e - P : e) + 1u;
                }
                corrector4<FMA>(sg.ratio, n, cs);
            }
            Quad<OUT_FMT> qo;
            float re[SPL], im[SPL];
#pragma unroll
            for (int k = 0; k < (int)SPL; ++k) {
                float a, b;
                quad_get<IN_FMT, kRawI16<IN_FMT, OUT_FMT>>(qin[v], k, a, b);
                mix(a, b, cs[k].x, cs[k].y, re[k], im[k]);
            }
            quad_set4<OUT_FMT, kRawI16<IN_FMT, OUT_FMT>>(qo, re, im, legacy);
            tile_store(v, qo);
        }
    } else {
        // ---- ragged tile: mask edge, stream tail, or a tile that straddles stretches
        for (uint32_t o = tid; o < TILE; o += BLOCK) {
            const uint64_t g = t0 + o;
            if (g < ta.m0) continue;
            if (g >= ta.m1) break;
            one_sample<IN_FMT, OUT_FMT, FMA>(in, out, segs, n_segs, si, g, legacy);
        }
    }
}


// ---- column windows of the span kernel (below).
// f32 -> i16: loads are 16 bytes per lane (2 samples), which would make the stores 8 bytes per lane — and 8-byte stores
// run at 2.4 TB/s in this kernel (measured; 16-byte stores with two 16-byte loads per lane at a 32-byte lane stride:
// 5.0 TB/s).  So the packed results of a row go through a wavefront-private kilobyte of LDS and leave as one 16-byte
// store per lane: both sides fully coalesced.
template <int IN_FMT, int OUT_FMT> struct WalkVec {
    static constexpr int S = RowVec<IN_FMT, OUT_FMT>::S;
    static constexpr bool kTranspose = IN_FMT == DPX_FMT_F32 && OUT_FMT == DPX_FMT_I16;
    // f32 output: a 256-sample window is 2 KiB per row on the wide side — two vectors per lane per row, which measures
    // 8 points below one (75 % against 83 % of the HBM peak for the same stream on the rows kernel, which moves one
    // vector per lane per row).  Such a window is therefore shared by TWO workgroups, 128 columns each (the grid is
    // doubled at launch; the plan, which does not know the formats, is unchanged).
    static constexpr int kSplit = (S == 2 && !kTranspose) ? 2 : 1;
    // (The opposite — two adjacent windows per i16 -> i16 workgroup, two vectors per lane per row — measured 3-5 points
    // slower on replays and const-mode walks alike: profiles/r02_walk.md.)
    static constexpr uint32_t kCols = kWalkWindow / kSplit;              // columns a workgroup takes
    static constexpr uint32_t kEntries = kCols + kWalkPad;               // slice entries it needs
};

// LDS layout of a slice: S planes by entry index modulo S (entry e at plane e % S, position e / S): the S correctors a
// lane needs for one vector (entries off + S * lane + k) then sit at consecutive 8-byte positions across the lanes for
// every k — conflict-free ds_read_b64 whatever the row's shift (a plain array read at a lane stride of S entries is
// 4-way conflicted: 30 % of the LDS cycles in round 1).  The plane stride makes the fills conflict-free as well
// (planes 16 or 8 banks apart).
template <int S, uint32_t ENTRIES> struct SlicePlanes {
    static constexpr uint32_t kPlanes = S, kLog2 = S == 4 ? 2 : 1;
    // entries per plane + 1, rounded up to 4 (mod 8) for four planes / 8 (mod 16) for two: plane bases 8 / 16 banks apart
    static constexpr uint32_t kNeed = ENTRIES / S + 1;
    static constexpr uint32_t kStride = S == 4 ? ((kNeed + 3) / 8) * 8 + 4 : ((kNeed + 7) / 16) * 16 + 8;
    static_assert(kStride >= kNeed && ENTRIES % kPlanes == 0, "planes hold the slice");
    static __device__ __forceinline__ uint32_t index(uint32_t e) { return (e & (kPlanes - 1)) * kStride + (e >> kLog2); }
};

// one block of kLeftBlock samples of ONE stretch, sincos per sample (lead-ins, heads, tails, stretches too short for a matrix)
template <int IN_FMT, int OUT_FMT, bool FMA, int THREADS>
__device__ __forceinline__ void leftover_block(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const LeftRange *__restrict__ left,
                                               const uint32_t *__restrict__ lhint, const DevSeg *__restrict__ segs, uint32_t e, uint32_t tid, bool legacy)
{
    {
        // e: index of this block among all leftover blocks
        uint32_t li = lhint[e >> kLeftHintShift];
        while (left[li + 1].wg_off <= e) ++li;                    // sentinel at the end
        const LeftRange lr = left[li];
        const DevSeg sg = segs[lr.seg];
        const uint32_t o0 = (e - lr.wg_off) * kLeftBlock;         // offset of this block in the range
        const uint64_t g0 = lr.start + o0;
        const uint32_t P = sg.period;
        const uint64_t j0 = g0 - sg.first;
        uint32_t base;   // periodic: phase of g0 in [0, P); linear: the counter itself
        if (P != 0) {
            // (the tile kernel's note: a 64-bit remainder is seventeen vector instructions even for uniform operands)
            const uint64_t s64 = (uint64_t)(sg.n_start - 1u) + j0;
            base = (s64 >> 32) == 0 ? (uint32_t)s64 % P : (uint32_t)(s64 % P);
        } else {
            base = sg.n_start + (uint32_t)j0;
        }
        static_assert(kLeftBlock == 4 * 256, "a leftover block is four samples for each of 256 thread slots");
        // 256 thread slots q; a workgroup of fewer threads takes them in turns (compile-time trip count)
#pragma unroll
        for (uint32_t q = tid; q < 256u; q += THREADS) {
        if (o0 + kLeftBlock <= lr.len && (P == 0 || P >= 4)) {
            // a whole block: four samples per thread slot, moved as 16-byte vectors (a block starts wherever its range
            // does, so the vectors are only sample-aligned: the hardware takes unaligned global accesses).  i16 output:
            // four CONSECUTIVE samples; f32 output: two PAIRS half a block apart, so that the 16-byte stores of a wavefront
            // instruction are adjacent (the tile kernel's kPairs, same reason).
            constexpr int IB = Fmt<IN_FMT>::kBytes, OB = Fmt<OUT_FMT>::kBytes;
            constexpr bool kPairs = OUT_FMT == DPX_FMT_F32;
            typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));
            typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(4)));
            const uint32_t oa = kPairs ? q * 2u : q * 4u;                       // offsets of the lane's samples 0 and 2 in the block
            const uint32_t ob = kPairs ? kLeftBlock / 2u + q * 2u : q * 4u + 2u;
            Quad<IN_FMT> qi;
            if constexpr (!kPairs) {
#pragma unroll
                for (int i = 0; i < Fmt<IN_FMT>::kVecs; ++i) qi.v[i] = *(reinterpret_cast<const u32x4_u *>(in + (g0 + oa) * IB) + i);
            } else if constexpr (IN_FMT == DPX_FMT_F32) {
                qi.v[0] = *reinterpret_cast<const u32x4_u *>(in + (g0 + oa) * IB);
                qi.v[1] = *reinterpret_cast<const u32x4_u *>(in + (g0 + ob) * IB);
            } else {
                const u32x2_u ha = *reinterpret_cast<const u32x2_u *>(in + (g0 + oa) * IB);
                const u32x2_u hb = *reinterpret_cast<const u32x2_u *>(in + (g0 + ob) * IB);
                qi.v[0] = u32x4{ha[0], ha[1], hb[0], hb[1]};
            }
            f32x2 cs[4];
            uint32_t ta = base + oa, tb = base + ob;
            if ((P == 0 || base + kLeftBlock <= P) && base < (1u << 24) - kLeftBlock - 1u) {
                // no wrap inside the block and every counter below 2^24 (uniform): the usual case, lead-ins above all
                if constexpr (kPairs) corrector4_pairs<FMA>(sg.ratio, P == 0 ? ta : ta + 1u, P == 0 ? tb : tb + 1u, cs);
                else                  corrector4_consecutive<FMA>(sg.ratio, P == 0 ? ta : ta + 1u, cs);
            } else {
                uint32_t n[4];
                if (P == 0) {
                    n[0] = ta; n[1] = ta + 1u; n[2] = tb; n[3] = tb + 1u;
                } else {
                    if (P >= kLeftBlock) { ta = ta >= P ? ta - P : ta; tb = tb >= P ? tb - P : tb; }   // base < P: at most one wrap
                    else                 { ta %= P; tb %= P; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t ee = (k < 2 ? ta : tb) + (uint32_t)(k & 1);   // P >= 4: at most one wrap
                        n[k] = (ee >= P ? ee - P : ee) + 1u;
                    }
                }
                corrector4<FMA>(sg.ratio, n, cs);
            }
            Quad<OUT_FMT> qo;
            float re[4], im[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a, bq;
                quad_get<IN_FMT, kRawI16<IN_FMT, OUT_FMT>>(qi, k, a, bq);
                mix(a, bq, cs[k].x, cs[k].y, re[k], im[k]);
            }
            quad_set4<OUT_FMT, kRawI16<IN_FMT, OUT_FMT>>(qo, re, im, legacy);
            if constexpr (!kPairs) {
#pragma unroll
                for (int i = 0; i < Fmt<OUT_FMT>::kVecs; ++i) {
                    const u32x4_u o = qo.v[i];
                    *(reinterpret_cast<u32x4_u *>(out + (g0 + oa) * OB) + i) = o;
                }
            } else {
                const u32x4_u o0 = qo.v[0], o1 = qo.v[1];
                *reinterpret_cast<u32x4_u *>(out + (g0 + oa) * OB) = o0;
                *reinterpret_cast<u32x4_u *>(out + (g0 + ob) * OB) = o1;
            }
        } else {
            // the last block of a range (or a period below 4): sample by sample, o = q + k * 256
            if (o0 + q >= lr.len) continue;
            uint32_t n[4];
            bool have[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t o = q + (uint32_t)k * 256u;
                have[k] = o0 + o < lr.len;
                const uint32_t oo = have[k] ? o : q;
                if (P == 0) {
                    n[k] = base + oo;
                } else if (P >= kLeftBlock) {                         // base < P and o < kLeftBlock: at most one wrap
                    const uint64_t t = (uint64_t)base + oo;
                    n[k] = (uint32_t)(t >= P ? t - P : t) + 1u;
                } else {
                    n[k] = (uint32_t)(((uint64_t)base + oo) % P) + 1u;
                }
            }
            f32x2 cs[4];
            corrector4<FMA>(sg.ratio, n, cs);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!have[k]) continue;
                const uint32_t o = q + (uint32_t)k * 256u;
                float a, bq, re, im;
                load_one<IN_FMT, kRawI16<IN_FMT, OUT_FMT>>(in, g0 + o, a, bq);
                mix(a, bq, cs[k].x, cs[k].y, re, im);
                store_one<OUT_FMT, kRawI16<IN_FMT, OUT_FMT>>(out, g0 + o, re, im, legacy);
            }
        }
        }
    }
}

// ---- span kernel (round 3): the same matrices, the same descriptors, the same slice — but a workgroup keeps its
// column window for a whole SPAN of rows (up to WalkArgs::span, the whole matrix when it is shorter) instead of one
// chunk of WAVES x 2, its wavefronts taking two rows per turn until the span is done.
//
// Why (profiles/r03_walk.md, SQ counters on the 600 s replay): the walk kernel is not short of HBM requests, it is short
// of VALU issue slots.  30.7 vector instructions per sample against 12.6 for the rows kernel, the vector ALUs 79 % busy,
// and the shader clock down to 1.85 GHz under the 1400 W cap; half of those instructions are the slice — five wavefront
// rounds of the bit-exact sincos per workgroup, shared by the 5-8 rows a one-second matrix gives a chunk.  What the
// "fixed cost per workgroup" of round 2 measured was this arithmetic.  A span evaluates a slice once per 256 columns of
// up to `span` rows: 75 / rows instructions per sample instead of 75 / 6.4, and every row of a matrix — the ninth of nine
// as well — finds a wavefront of an already running workgroup.
//
// Shape of the row loop: the first two rows' loads are issued before the slice is evaluated (as in the walk kernel);
// after the barrier each wavefront is on its own: mix and store the rows it holds, request the next two (rows
// r + WAVES * 2), and so on — no further barrier, wavefronts of a span end independently.  Per row everything but the
// lane's column offset is uniform and lives in scalar registers (row origin, shift, length); a window that lies wholly
// inside a row (all but the last window of a matrix, and the rows a stretch end cuts) takes loads and stores without
// any per-lane test.  Every wavefront of the workgroup takes part in the slice and reaches the barrier (the walk
// kernel lets wavefronts without rows end before it).
struct RowGeo {
    uint64_t row0;      // first sample of the row's storage: the 32-sample boundary at or below A + r * L
    uint32_t rowlen;    // samples from row0 to the next row's boundary (or the end of the matrix)
    uint32_t off;       // slice entry of the row's column 0: kWalkPad - shift
};

__device__ __forceinline__ RowGeo row_geo(const WalkSeg &ws, uint32_t r)
{
    const uint64_t ideal = ws.A + (uint64_t)r * ws.L;
    const uint64_t nxt = (ideal + ws.L) & ~31ull;
    RowGeo g;
    g.row0 = ideal & ~31ull;
    g.rowlen = (uint32_t)((nxt < ws.E ? nxt : ws.E) - g.row0);
    g.off = kWalkPad - ((uint32_t)ideal & 31u);
    return g;
}

// slice entries a workgroup may hold: one window (+ the row shifts), or up to four adjacent windows (MULTI: the
// launches whose spans come from descriptors, dpx_types.h WalkSeg::wshift)
template <int IN_FMT, int OUT_FMT, bool MULTI> struct SpanSlice {
    static constexpr uint32_t kCols = WalkVec<IN_FMT, OUT_FMT>::kCols;
    static constexpr uint32_t kEntries = (MULTI ? kCols << kSpanMaxShift : kCols) + kWalkPad;
};

template <int IN_FMT, int OUT_FMT>
struct SpanTypes {
    typedef WalkVec<IN_FMT, OUT_FMT> WV;
    static constexpr int S = WV::S;
    static constexpr int NV = (int)WV::kCols / (kRowsLanes * S);
    static constexpr int QW = S * Fmt<IN_FMT>::kBytes / 4;
    typedef uint32_t qvec __attribute__((ext_vector_type(QW)));
};

// request one row's vectors; FULL: the window lies inside the row
template <int IN_FMT, int OUT_FMT, bool FULL>
__device__ __forceinline__ void span_load_row(const uint8_t *__restrict__ in, const RowGeo &g, uint32_t col0, uint32_t lane,
                                              typename SpanTypes<IN_FMT, OUT_FMT>::qvec (&q)[SpanTypes<IN_FMT, OUT_FMT>::NV])
{
    typedef SpanTypes<IN_FMT, OUT_FMT> T;
    constexpr int IB = Fmt<IN_FMT>::kBytes;
    const uint8_t *rowp = in + g.row0 * IB;                       // uniform
#pragma unroll
    for (int v = 0; v < T::NV; ++v) {
        uint32_t c = col0 + lane * T::S + (uint32_t)v * (kRowsLanes * T::S);
        if constexpr (!FULL) c = c < g.rowlen ? c : 0u;           // a lane past the row reads the row's first vector instead
        q[v] = __builtin_nontemporal_load(reinterpret_cast<const typename T::qvec *>(rowp + c * IB));
    }
}

// mix one row with the slice and store it
template <int IN_FMT, int OUT_FMT, bool FULL, bool MULTI>
__device__ __forceinline__ void span_finish_row(uint8_t *__restrict__ out, const RowGeo &g, uint32_t col0, uint32_t lane,
                                                const typename SpanTypes<IN_FMT, OUT_FMT>::qvec (&q)[SpanTypes<IN_FMT, OUT_FMT>::NV],
                                                const float2 *slice, uint32_t *xrow, bool legacy)
{
    typedef SpanTypes<IN_FMT, OUT_FMT> T;
    constexpr int S = T::S, NV = T::NV;
    constexpr int OB = Fmt<OUT_FMT>::kBytes;
    constexpr bool XP = T::WV::kTranspose;
    constexpr bool RAW = kRawI16<IN_FMT, OUT_FMT>;
    typedef SlicePlanes<S, SpanSlice<IN_FMT, OUT_FMT, MULTI>::kEntries> SP;
    uint8_t *rowo = out + g.row0 * OB;                             // uniform
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const uint32_t c = col0 + lane * S + (uint32_t)v * (kRowsLanes * S);
        float re[S], im[S];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const uint32_t ok = g.off + (uint32_t)k;              // uniform: plane and base position of corrector k
            const float2 cs = slice[(ok & (SP::kPlanes - 1)) * SP::kStride + (ok >> SP::kLog2) + (uint32_t)v * kRowsLanes + lane];
            float a, bq;
            if constexpr (IN_FMT == DPX_FMT_I16) unpack_i16<RAW>(q[v][k], a, bq);
            else { a = __uint_as_float(q[v][2 * k]); bq = __uint_as_float(q[v][2 * k + 1]); }
            mix(a, bq, cs.x, cs.y, re[k], im[k]);
        }
        const bool active = FULL || c < g.rowlen;
        if constexpr (OUT_FMT == DPX_FMT_I16) {
            uint32_t pk[S];
            pack_row<RAW, S>(re, im, pk, legacy);
            if constexpr (S == 4) {
                u32x4 o = {pk[0], pk[1], pk[2], pk[3]};
                asm volatile("" : "+v"(o));   // keeps the vectoriser from rebuilding the store without its nt flag
                if (active) __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(rowo + c * OB));
            } else if constexpr (XP) {
                const u32x2 o = {pk[0], pk[1]};
                *reinterpret_cast<u32x2 *>(xrow + (uint32_t)v * (kRowsLanes * S) + lane * S) = o;
            } else {
                const u32x2 o = {pk[0], pk[1]};
                if (active) __builtin_nontemporal_store(o, reinterpret_cast<u32x2 *>(rowo + c * OB));
            }
        } else {
#pragma unroll
            for (int i = 0; i < S / 2; ++i) {
                u32x4 o;
                o[0] = __float_as_uint(re[2 * i]);     o[1] = __float_as_uint(im[2 * i]);
                o[2] = __float_as_uint(re[2 * i + 1]); o[3] = __float_as_uint(im[2 * i + 1]);
                if (active) __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(rowo + c * OB) + i);
            }
        }
    }
    if constexpr (XP) {
        // f32 -> i16: the row's 256 packed samples are in LDS (same wavefront: LDS operations execute in order) and
        // leave as one 16-byte store per lane
        __builtin_amdgcn_wave_barrier();
        u32x4 o = *reinterpret_cast<const u32x4 *>(xrow + lane * 4);
        asm volatile("" : "+v"(o));
        const uint32_t c4 = col0 + lane * 4;
        if (FULL || c4 < g.rowlen) __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(rowo + c4 * OB));
    }
}

// U: rows a wavefront takes per turn.  MULTI: the span's workgroups take 2^ws.wshift adjacent windows, WAVES >> wshift
// wavefronts each (a wave-uniform split: everything below stays in scalar registers); `g.off` of a row then counts from
// the first of them, sub * kCols entries further for the wavefronts of window `sub`.
template <int IN_FMT, int OUT_FMT, bool FMA, int WAVES, int U, bool MULTI>
__device__ __forceinline__ void span_body(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const WalkSeg &ws,
                                          uint32_t w, uint32_t half, uint32_t wave, uint32_t lane, uint32_t tid,
                                          float2 *slice, uint32_t *xpose, bool legacy)
{
    typedef SpanTypes<IN_FMT, OUT_FMT> T;
    typedef typename T::WV WV;
    constexpr int NV = T::NV;
    constexpr int THREADS = WAVES * 64;
    constexpr uint32_t kCols = WV::kCols, kSplit = WV::kSplit;
    typedef SlicePlanes<T::S, SpanSlice<IN_FMT, OUT_FMT, MULTI>::kEntries> SP;
    // the workgroup's windows, and this wavefront's place among them
    const uint32_t wshift = MULTI ? ws.wshift : 0u;                   // uniform
    uint32_t sub = 0, wiw = wave, wpwin = WAVES;                       // window of the workgroup, wavefront of the window, wavefronts per window
    if (MULTI && wshift != 0) {
        wpwin = (uint32_t)WAVES >> wshift;
        sub = wave / wpwin;                                            // scalar: WAVES and wshift are tiny
        wiw = wave - sub * wpwin;
    }
    const uint32_t colbase = ((w * kSplit + half) << wshift) * kCols;  // first column of this workgroup
    const uint32_t col0 = colbase + sub * kCols;                       // ... and of this wavefront
    const uint32_t n_entries = (kCols << wshift) + kWalkPad;           // the workgroup's slice
    const uint32_t stride = wpwin * U;

    uint32_t r = ws.row0 + wiw * U;                              // this wavefront's rows: r .. r + U - 1, then r + stride ...
    typename T::qvec q[U][NV];
    auto request = [&](uint32_t rr) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (rr + u < ws.row_end) {                           // uniform
                const RowGeo g = row_geo(ws, rr + u);
                if (col0 >= g.rowlen) continue;                  // the matrix's last row ends before this window
                if (col0 + kCols <= g.rowlen) span_load_row<IN_FMT, OUT_FMT, true>(in, g, col0, lane, q[u]);
                else                          span_load_row<IN_FMT, OUT_FMT, false>(in, g, col0, lane, q[u]);
            }
        }
    };
    request(r);

    // the slice: thread j evaluates entry j, j + THREADS, ... with the bit-exact sincos, after the first rows' loads have
    // been issued; entry j = corrector of column colbase + j - kWalkPad
    {
        const uint32_t P = ws.period;
        // counter of entry j, minus one: (phase + colbase + j - kWalkPad) mod P.  The part that does not depend on j is
        // reduced once (uniform; rows are multiples of the period, so colbase may exceed it many times), entry by entry
        // one conditional subtraction is left — periods shorter than a slice take the modulo per entry.
        const uint32_t ub = (ws.phase + colbase + P * kWalkPad - kWalkPad) % P;  // 32-bit: phase < P, colbase <= L + 1024, P <= 2^22 (kLutMaxEntries)
        constexpr int kRounds = ((int)SpanSlice<IN_FMT, OUT_FMT, MULTI>::kEntries + THREADS - 1) / THREADS;
#pragma unroll
        for (int it = 0; it < kRounds; ++it) {
            if (MULTI && (uint32_t)it * THREADS >= n_entries) break;             // uniform
            const uint32_t j = tid + (uint32_t)it * THREADS;
            if (j < n_entries) {
                uint32_t t = ub + j;
                if (P > n_entries) t = t >= P ? t - P : t;       // uniform: j < n_entries < P
                else               t %= P;
                float c, sn;
                corrector<FMA>(ws.ratio, t + 1u, c, sn);
                slice[SP::index(j)] = make_float2(c, sn);
            }
        }
    }
    __syncthreads();

    const float2 *wslice = slice + sub * (kCols / T::S);           // entry e of this window = entry e + sub * kCols of the slice
    while (r < ws.row_end) {                                     // uniform per wavefront
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u < ws.row_end) {
                const RowGeo g = row_geo(ws, r + u);
                if (col0 >= g.rowlen) continue;
                uint32_t *xrow = xpose + (wave * U + u) * kWalkWindow;
                if (col0 + kCols <= g.rowlen) span_finish_row<IN_FMT, OUT_FMT, true, MULTI>(out, g, col0, lane, q[u], wslice, xrow, legacy);
                else                          span_finish_row<IN_FMT, OUT_FMT, false, MULTI>(out, g, col0, lane, q[u], wslice, xrow, legacy);
            }
        }
        r += stride;
        request(r);
    }
}

// UNI: a launch of ONE matrix (WalkUni): the matrix comes with the kernel arguments, the span is blockIdx.y, the window
// blockIdx.x; grid rows past the last span hold the leftover blocks (head and tail of the stretch).
template <int IN_FMT, int OUT_FMT, bool FMA, int WAVES, bool UNI>
__global__ __launch_bounds__(WAVES * 64) void span_kernel(const uint8_t *__restrict__ in,
                                                            uint8_t *__restrict__ out,
                                                            WalkUni uni,                           // UNI only
                                                            const WalkSeg *__restrict__ wdesc,     // !UNI only
                                                            uint32_t n_left_wg,
                                                            uint32_t cast_legacy,                  // dpx_set_i16_cast
                                                            uint32_t wg_off,                       // sub-launch: first span (UNI: grid row) / first workgroup of this launch
                                                            // ---- leftover path only
                                                            const LeftRange *__restrict__ left,
                                                            const uint32_t *__restrict__ lhint,
                                                            const DevSeg *__restrict__ segs)
{
    constexpr int S = WalkVec<IN_FMT, OUT_FMT>::S;
    constexpr int THREADS = WAVES * 64;
    constexpr bool XP = WalkVec<IN_FMT, OUT_FMT>::kTranspose;
    constexpr bool MULTI = !UNI;                                  // spans from descriptors may take several windows per workgroup
    constexpr int kMaxU = 2;                                      // rows per wavefront per turn
    typedef SlicePlanes<S, SpanSlice<IN_FMT, OUT_FMT, MULTI>::kEntries> SPK;
    __shared__ float2 slice[SPK::kPlanes * SPK::kStride];
    __shared__ __attribute__((aligned(16))) uint32_t xpose[XP ? WAVES * kMaxU * (int)kWalkWindow : 4];   // packed i16 samples of one row per (wavefront, u); read back as 16-byte vectors
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t kSplit = WalkVec<IN_FMT, OUT_FMT>::kSplit;
    static_assert(kSplit == span_split(IN_FMT, OUT_FMT), "the planner's simulation walks the same grid");
    const uint32_t half = blockIdx.x % kSplit;
    const bool legacy = cast_legacy != 0;                         // uniform
    if constexpr (UNI) {
        const uint32_t w = blockIdx.x / kSplit, c = blockIdx.y + wg_off;
        if (c < uni.n_spans) {
            if (w >= uni.seg.nw) return;                          // padding
            WalkSeg ws = uni.seg;
            ws.row0 = c * uni.base + (c < uni.rem ? c : uni.rem);
            ws.row_end = ws.row0 + uni.base + (c < uni.rem ? 1u : 0u);
            const uint32_t lane = tid & (kRowsLanes - 1), wave = __builtin_amdgcn_readfirstlane(tid / kRowsLanes);
            span_body<IN_FMT, OUT_FMT, FMA, WAVES, 2, false>(in, out, ws, w, half, wave, lane, tid, slice, xpose, legacy);
        } else {
            const uint32_t e = (c - uni.n_spans) * uni.nw8 + w;   // leftover block
            if (half != 0 || e >= n_left_wg) return;
            leftover_block<IN_FMT, OUT_FMT, FMA, THREADS>(in, out, left, lhint, segs, e, tid, legacy);
        }
    } else {
        // ONE scalar load before the first sample load: the descriptor of this group of 8 workgroups — a span of a matrix
        // (replicated per group; spans start on multiples of 8 workgroups) or a group of leftover blocks.  The leftover
        // groups (sincos per sample, VALU-bound) are spread evenly between the spans by the planner, so that they run
        // beside memory-bound workgroups.
        const uint32_t b = blockIdx.x / kSplit + wg_off;
        const WalkSeg ws = wdesc[b >> kWalkHintShift];
        const uint32_t w = b - ws.wg_base;
        if (w >= ws.nwg) return;                                  // padding
        if (ws.upw != 0) {
            // the wavefront index is uniform: telling the compiler so keeps all the row geometry in scalar registers
            const uint32_t lane = tid & (kRowsLanes - 1), wave = __builtin_amdgcn_readfirstlane(tid / kRowsLanes);
            span_body<IN_FMT, OUT_FMT, FMA, WAVES, 2, true>(in, out, ws, w, half, wave, lane, tid, slice, xpose, legacy);
        } else {
            if (half != 0) return;                                // a leftover block is one workgroup whatever the grid scaling
            leftover_block<IN_FMT, OUT_FMT, FMA, THREADS>(in, out, left, lhint, segs, ws.row0 + w, tid, legacy);
        }
    }
}

// ---- resident block kernel (dpx_types.h, BlockCtl): workgroup s serves staging slot s, polling its doorbell in host
// memory.  Everything it touches but its LDS and the shared idle clock is host-mapped (fine-grained: uncached on the device).
template <int IN_FMT, int OUT_FMT, bool FMA>
__global__ __launch_bounds__(kResidentThreads) void resident_block_kernel(ResidentArgs ra)
{
    __shared__ DevSeg s_segs[kResidentMaxSegs];
    __shared__ uint32_t s_door, s_leave, s_n, s_nsegs, s_legacy;
    const uint32_t tid = threadIdx.x, slot = blockIdx.x;
    BlockCtl *ctl = ra.ctl[slot];
    const uint8_t *in = ra.in[slot];
    uint8_t *out = ra.out[slot];
    const DevSeg *segs_host = ra.segs[slot];
    if (tid == 0) {
        s_door = __hip_atomic_load(&ctl->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&ra.shared->activity, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    uint32_t last = s_door, served = 0, empty = 0;
    __syncthreads();
    for (;;) {
        if (tid == 0) {
            // ticket, payload (sample count, stretch count, cast mode, kernel instance) and the ticket again are the first 16
            // bytes of the host-written line: ONE PCIe read.  The word proves itself (dpx_types.h, ctl_word_valid): a read
            // that caught the line between two of the host's stores is taken as "not rung yet".
            const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(ctl));
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            const uint32_t d = ctl_word_valid(w[0], w[1], w[3]) ? w[0] : last;
            uint32_t leave = d == kDoorExit ? 1u : 0u;
            if (d != last && d != kDoorExit && ((w[1] >> 20) & 7u) != ctl_instance(IN_FMT, OUT_FMT, FMA)) {
                // a ticket rung for another instance of this kernel (format pair, libm build): not ours to serve — everyone
                // leaves, the host starts the right kernel and that one finds the doorbell rung (dpx_resident.cpp)
                __hip_atomic_store(&ra.shared->leaving, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                leave = 1;
            }
            if (d == last) {                                      // (a block that has been rung is always finished first)
                if (__hip_atomic_load(&ra.shared->leaving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    leave = 1;
                } else if ((empty & 15u) == 15u) {
                    const unsigned long long act = __hip_atomic_load(&ra.shared->activity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long now = (unsigned long long)wall_clock64();
                    if (now > act && now - act > ra.idle_ticks) {
                        __hip_atomic_store(&ra.shared->leaving, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        leave = 1;
                    }
                }
            }
            s_door = d;
            s_leave = leave;
            s_n = w[1] & 0x3fffu;
            s_nsegs = (w[1] >> 14) & 0x1fu;
            s_legacy = (w[1] >> 19) & 1u;
        }
        __syncthreads();
        const uint32_t d = s_door, leave = s_leave, n = s_n, n_segs = s_nsegs;
        const bool legacy = s_legacy != 0;
        __syncthreads();                                          // the words are rewritten by the next poll
        if (leave) break;
        if (d == last) {
            // nothing new: poll again — at once for the first polls after a block (the next one is usually on its way),
            // then every few microseconds (each poll is a PCIe read)
            if (++empty > 4096) __builtin_amdgcn_s_sleep(127);
            else if (empty > 64) __builtin_amdgcn_s_sleep(16);
            continue;
        }
        empty = 0;
        // Everything the block needs comes over PCIe; all of it is requested at once (one round trip, not three): the
        // stretch list, and every thread's first quad — the reference's block is 2048 (i16) or 1024 (f32) samples: one quad
        // per thread of the 512; a quad past the block's end reads slot memory that is there and is not used.
        u32x4 seg_piece = {0, 0, 0, 0};
        if (tid < kResidentMaxSegs * (uint32_t)(sizeof(DevSeg) / 16)) seg_piece = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(segs_host) + tid);
        const Quad<IN_FMT> pre = load_quad<IN_FMT>(in, (uint64_t)tid * 4u);
        if (tid < kResidentMaxSegs * (uint32_t)(sizeof(DevSeg) / 16)) reinterpret_cast<u32x4 *>(s_segs)[tid] = seg_piece;
        __syncthreads();
        // four consecutive samples per thread where they lie inside one stretch without a wrap; sample by sample elsewhere
        const uint32_t nq = n >> 2;
        uint32_t turn = 0;
        for (uint32_t q = tid; q < nq + 1; q += kResidentThreads, ++turn) {
            const uint64_t g = (uint64_t)q * 4u;
            if (q == nq) {                                        // the last n mod 4 samples
                for (uint64_t gg = g; gg < n; ++gg) one_sample<IN_FMT, OUT_FMT, FMA>(in, out, s_segs, n_segs, 0, gg, legacy);
                break;
            }
            uint32_t si = 0;
            while (si + 1 < n_segs && s_segs[si].first + s_segs[si].count <= g) ++si;
            const DevSeg &sg = s_segs[si];
            const uint32_t P = sg.period;
            const uint64_t j = g - sg.first;
            const uint32_t base = P != 0 ? (uint32_t)(((uint64_t)(sg.n_start - 1u) + j) % P) : sg.n_start + (uint32_t)j;
            const bool together = g + 4 <= sg.first + sg.count && (P == 0 ? base < 0xfffffffcu : (P >= 4 && base + 4 <= P));
            if (!together) {
                for (uint64_t gg = g; gg < g + 4; ++gg) one_sample<IN_FMT, OUT_FMT, FMA>(in, out, s_segs, n_segs, si, gg, legacy);
                continue;
            }
            const uint32_t n0 = P == 0 ? base : base + 1u;
            const uint32_t cn[4] = {n0, n0 + 1u, n0 + 2u, n0 + 3u};
            f32x2 cs[4];
            corrector4<FMA>(sg.ratio, cn, cs);
            const Quad<IN_FMT> qi = turn == 0 ? pre : load_quad<IN_FMT>(in, g);
            Quad<OUT_FMT> qo;
            float re[4], im[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a, b;
                quad_get<IN_FMT, kRawI16<IN_FMT, OUT_FMT>>(qi, k, a, b);
                mix(a, b, cs[k].x, cs[k].y, re[k], im[k]);
            }
            quad_set4<OUT_FMT, kRawI16<IN_FMT, OUT_FMT>>(qo, re, im, legacy);
            store_quad<OUT_FMT>(out, g, qo);
        }
        __threadfence_system();                                   // the block's output before its completion word
        __syncthreads();
        last = d;
        ++served;
        if (tid == 0) {
            ctl->blocks = served;
            __hip_atomic_store(&ctl->done, d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&ra.shared->activity, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) __hip_atomic_store(&ctl->state, kResidentParked, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// plan-time: entry e of a table = corrector(((n_first - 1 + e) mod period) + 1)
template <bool FMA>
__global__ __launch_bounds__(256) void build_lut_kernel(float2 *__restrict__ tab, uint32_t period,
                                                        uint32_t n_first, uint32_t n_entries, float ratio)
{
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < n_entries; e += gridDim.x * 256u) {
        float c, s;
        const uint32_t n = (uint32_t)(((uint64_t)(n_first - 1u) + e) % period) + 1u;
        corrector<FMA>(ratio, n, c, s);
        tab[e] = make_float2(c, s);
    }
}

// ------------------------------------------------------- auxiliary kernels

constexpr int kBlock = 256;

// calibration: the same one-shot 16-byte non-temporal stream with no math
__global__ __launch_bounds__(kBlock) void copy_kernel(const u32x4 *__restrict__ in,
                                                      u32x4 *__restrict__ out, uint64_t n_vec)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n_vec) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

// A span launch of many matrices finds its descriptor with ONE scalar load: desc[blockIdx.x >> kWalkHintShift], i.e. the list
// of spans written out once per group of 8 workgroups (2.6 MB for a 600 s replay).  The host uploads the spans themselves
// (92 KB) and the index of every group (165 KB); this writes the rest on the device, 16 bytes per lane (round 6: the
// 2.6 MB went over PCIe from pageable memory before, 240 of a plan's 840 us).
__global__ __launch_bounds__(kBlock) void span_descriptors_kernel(const u32x4 *__restrict__ spans, const uint32_t *__restrict__ index,
                                                             u32x4 *__restrict__ desc, uint32_t n_desc)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_desc * 4u) desc[i] = spans[index[i >> 2] * 4u + (i & 3u)];
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
                                                          uint32_t *__restrict__ out, uint64_t n, int legacy)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const float2 z = in[i];
        out[i] = legacy ? pack_i16<false, true>(z.x, z.y) : pack_i16(z.x, z.y);
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

// complex.c:33-39 for any argument: z <- cexpf(z.re + i*z.im)
template <bool FMA>
__global__ __launch_bounds__(kBlock) void ccexpf_kernel(float2 *__restrict__ z, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        float re, im;
        ccexpf_glibc<FMA>(z[i].x, z[i].y, re, im);
        z[i] = make_float2(re, im);
    }
}

// ------------------------------------------------------------ launch wrappers

#define DPX_DISPATCH_FMT(FN, ...)                                                                           \
    do {                                                                                                    \
        if (in_fmt == DPX_FMT_I16 && out_fmt == DPX_FMT_I16) return FN<DPX_FMT_I16, DPX_FMT_I16>(__VA_ARGS__); \
        if (in_fmt == DPX_FMT_I16 && out_fmt == DPX_FMT_F32) return FN<DPX_FMT_I16, DPX_FMT_F32>(__VA_ARGS__); \
        if (in_fmt == DPX_FMT_F32 && out_fmt == DPX_FMT_I16) return FN<DPX_FMT_F32, DPX_FMT_I16>(__VA_ARGS__); \
        if (in_fmt == DPX_FMT_F32 && out_fmt == DPX_FMT_F32) return FN<DPX_FMT_F32, DPX_FMT_F32>(__VA_ARGS__); \
        return DPX_ERR_ARG;                                                                                 \
    } while (0)

template <int IN_FMT, int OUT_FMT>
static int tiles_t(const void *d_in, void *d_out, const DevSeg *d_segs, uint32_t n_segs,
                   const uint32_t *d_hint, const void *d_lut, const TileArgs &t_in, bool fma,
                   const LaunchGeom &g, hipStream_t st)
{
    TileArgs t = t_in;
    t.legacy = (g.legacy_cast && OUT_FMT == DPX_FMT_I16) ? 1u : 0u;
    const uint8_t *in = static_cast<const uint8_t *>(d_in);
    uint8_t *out = static_cast<uint8_t *>(d_out);
    const float2 *lut = static_cast<const float2 *>(d_lut);
    if (t.n_tiles == 0) return DPX_OK;
    if (t.n_tiles > 0x7fffffffull) return DPX_ERR_ARG;
    // (one launch whatever the length: cutting a tile launch into sub-launches as span_t does changes nothing here —
    // configs[4]'s chunk 77.7 / 77.9 / 77.4 % as one, two, four launches, the per-sample path 62.8 / 62.5: profiles/raw/r05_ab_sub_tiles.log)
    const dim3 grid((uint32_t)t.n_tiles);
#define DPX_CASE(B, Vv)                                                                                      \
    if (g.block == B && g.vecs == Vv) {                                                                      \
        if (fma) tile_kernel<IN_FMT, OUT_FMT, true, B, Vv><<<grid, B, 0, st>>>(in, out, d_segs, n_segs, d_hint, lut, t);  \
        else     tile_kernel<IN_FMT, OUT_FMT, false, B, Vv><<<grid, B, 0, st>>>(in, out, d_segs, n_segs, d_hint, lut, t); \
        return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;                                       \
    }
    DPX_CASE(128, 2) DPX_CASE(256, 1)
    if constexpr (IN_FMT == DPX_FMT_I16 && OUT_FMT == DPX_FMT_I16) {       // one wavefront x 16 samples per lane: built for this pair only
        DPX_CASE(64, 4)
    }
#undef DPX_CASE
    return DPX_ERR_ARG;
}

int launch_tiles(const void *d_in, int in_fmt, void *d_out, int out_fmt, const DevSeg *d_segs,
                 uint32_t n_segs, const uint32_t *d_hint, const void *d_lut, const TileArgs &t,
                 bool fma, const LaunchGeom &g, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    DPX_DISPATCH_FMT(tiles_t, d_in, d_out, d_segs, n_segs, d_hint, d_lut, t, fma, g, st);
}

template <int IN_FMT, int OUT_FMT>
static int rows_t(const void *d_in, void *d_out, const DevSeg *d_segs, const void *d_lut,
                  const RowsArgs &r, bool fma, int legacy_cast, hipStream_t st)
{
    const uint8_t *in = static_cast<const uint8_t *>(d_in);
    uint8_t *out = static_cast<uint8_t *>(d_out);
    const float2 *lut = static_cast<const float2 *>(d_lut);
    constexpr uint32_t S = RowVec<IN_FMT, OUT_FMT>::S;
    const uint32_t cols = (r.L + kRowsLanes * S - 1) / (kRowsLanes * S);
    const uint64_t n_main = r.n_rg * cols;
    const uint64_t ragged = (r.A - r.r0) + (r.r1 - r.B);
    const uint64_t n_extra = (ragged + kRowsLanes * 4 - 1) / (kRowsLanes * 4);
    if (n_main + n_extra == 0) return DPX_OK;
    if (n_main + n_extra > 0x7fffffffull) return DPX_ERR_ARG;
    // exact b / cols for b < 2^31: M = ceil(2^(31+s) / cols), s = ceil(log2 cols)
    uint32_t s = 0;
    while ((1u << s) < cols) ++s;
    const uint32_t sh = (31 + s) | ((legacy_cast && OUT_FMT == DPX_FMT_I16) ? 0x80000000u : 0u);   // bit 31: the legacy i16 cast
    const uint64_t M = ((1ull << (31 + s)) + cols - 1) / cols;
    const dim3 grid((uint32_t)(n_main + n_extra));
    const float2 *tab = lut + r.tab_off;
    // table or evaluation (dpx_planner.cpp, kRowsComputeMinP): since round 5 the same choice for every format pair
    const bool comp = r.compute != 0;
#define DPX_ROWS_CASE(RR, CC)                                                                                                               \
    if (r.R == RR && comp == CC) {                                                                                                           \
        if (fma) rows_kernel<IN_FMT, OUT_FMT, true, RR, CC><<<grid, kRowsLanes, 0, st>>>(in, out, tab, r.A, r.L, cols, M, sh, (uint32_t)n_extra, r.P, r.ratio, r.idx0, d_segs, r);  \
        else     rows_kernel<IN_FMT, OUT_FMT, false, RR, CC><<<grid, kRowsLanes, 0, st>>>(in, out, tab, r.A, r.L, cols, M, sh, (uint32_t)n_extra, r.P, r.ratio, r.idx0, d_segs, r); \
        return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;                                                                      \
    }
    // (rounds 2-4 also built 8 rows per wavefront: never a plan's choice — table 6194 GB/s against 6615 under two rows,
    // evaluation 73.7 % against 80.2 — and a sixth of the code object)
    DPX_ROWS_CASE(2, false) DPX_ROWS_CASE(4, false) DPX_ROWS_CASE(2, true) DPX_ROWS_CASE(4, true)
#undef DPX_ROWS_CASE
    return DPX_ERR_ARG;
}

int launch_rows(const void *d_in, int in_fmt, void *d_out, int out_fmt, const DevSeg *d_segs,
                const void *d_lut, const RowsArgs &r, bool fma, int legacy_cast, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    DPX_DISPATCH_FMT(rows_t, d_in, d_out, d_segs, d_lut, r, fma, legacy_cast, st);
}

// Sub-launches.  One launch over a stream of 4 GiB runs 2-3 points below the same launch cut into pieces of 1 GiB — span
// kernel, const 5001 Hz, one box: 75.6 % whole against 78.8 % as four launches back to back (79.9-78.5 each alone); 8 GiB:
// 77.7 against 80.7 (profiles/r05_generality.md; the rows kernel shows nothing of the kind: 84.7 whole, 83.4 cut).  The
// counters name no cause (same requests, fewer DRAM-credit stalls, address translation misses 0.02 per thousand samples in
// both), so the remedy is the measured one: the grid is dealt out in pieces that cover about 2^sub_lg samples each, back to
// back on the stream.  Workgroups are independent and find their work from (wg_off + blockIdx), so a cut anywhere is valid.
template <int IN_FMT, int OUT_FMT>
static int span_t(const void *d_in, void *d_out, const DevSeg *d_segs, const WalkSeg *d_wdesc,
                  const LeftRange *d_left, const uint32_t *d_lhint, const WalkArgs &w, bool fma, int legacy_cast, hipStream_t st)
{
    const uint32_t lg = (legacy_cast && OUT_FMT == DPX_FMT_I16) ? 1u : 0u;
    const uint8_t *in = static_cast<const uint8_t *>(d_in);
    uint8_t *out = static_cast<uint8_t *>(d_out);
    constexpr uint32_t kSplit = WalkVec<IN_FMT, OUT_FMT>::kSplit;
    // the whole grid: spans and the groups of leftover blocks between them; x2 where a window is shared by two workgroups
    const uint64_t n_wg = (uint64_t)w.n_walk_wg * kSplit;
    if (n_wg == 0) return DPX_OK;
    if (n_wg > 0x7fffffffull || w.span == 0) return DPX_ERR_ARG;
    // what this format pair makes of the plan's shape (dpx_planner.cpp: the same function the planner's simulation walks)
    SpanLaunch sl;
    if (!span_launch_shape(w, IN_FMT, OUT_FMT, &sl)) return DPX_ERR_ARG;
    const bool uni = sl.uni.n_spans != 0;
    // pieces: grid rows (one matrix: a row is a span or a row of leftover blocks) or workgroups (multiples of 8: descriptor groups)
    const uint32_t units = uni ? sl.uni.n_spans + sl.left_rows : w.n_walk_wg;
    const uint32_t pieces = sub_launch_pieces(w.cover, w.sub_lg);
    uint32_t per = (units + pieces - 1) / pieces;
    if (!uni) per = (per + 7u) & ~7u;
    if (per == 0) per = units;
    for (uint32_t off = 0; off < units; off += per) {
        const uint32_t cnt = units - off < per ? units - off : per;
        const dim3 grid = uni ? dim3(sl.uni.nw8 * kSplit, cnt) : dim3(cnt * kSplit);
#define DPX_SPAN_CASE(WW, UU)                                                                                                          \
        if (sl.waves == WW && uni == UU) {                                                                                             \
            if (fma) span_kernel<IN_FMT, OUT_FMT, true, WW, UU><<<grid, WW * 64, 0, st>>>(in, out, sl.uni, d_wdesc, w.n_left_wg, lg, off, d_left, d_lhint, d_segs);   \
            else     span_kernel<IN_FMT, OUT_FMT, false, WW, UU><<<grid, WW * 64, 0, st>>>(in, out, sl.uni, d_wdesc, w.n_left_wg, lg, off, d_left, d_lhint, d_segs);  \
            if (hipGetLastError() != hipSuccess) return DPX_ERR_HIP;                                                                   \
            continue;                                                                                                                  \
        }
        DPX_SPAN_CASE(4, true) DPX_SPAN_CASE(4, false) DPX_SPAN_CASE(5, true) DPX_SPAN_CASE(2, true) DPX_SPAN_CASE(8, false)
        DPX_SPAN_CASE(5, false) DPX_SPAN_CASE(2, false)
#undef DPX_SPAN_CASE
        return DPX_ERR_ARG;
    }
    return DPX_OK;
}

int launch_span(const void *d_in, int in_fmt, void *d_out, int out_fmt, const DevSeg *d_segs,
                const WalkSeg *d_walk_desc, const LeftRange *d_left, const uint32_t *d_left_hint,
                const WalkArgs &w, bool fma, int legacy_cast, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    DPX_DISPATCH_FMT(span_t, d_in, d_out, d_segs, d_walk_desc, d_left, d_left_hint, w, fma, legacy_cast, st);
}

template <int IN_FMT, int OUT_FMT>
static int resident_t(const ResidentArgs &args, bool fma, hipStream_t st)
{
    if (fma) resident_block_kernel<IN_FMT, OUT_FMT, true><<<kResidentSlots, kResidentThreads, 0, st>>>(args);
    else     resident_block_kernel<IN_FMT, OUT_FMT, false><<<kResidentSlots, kResidentThreads, 0, st>>>(args);
    return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;
}

int launch_resident_block(const ResidentArgs &args, int in_fmt, int out_fmt, bool fma, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    DPX_DISPATCH_FMT(resident_t, args, fma, st);
}

int launch_build_lut(void *d_lut_entries, uint32_t period, uint32_t n_first, uint32_t n_entries,
                     float ratio, bool fma, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint32_t grid = (n_entries + 255u) / 256u;
    float2 *tab = static_cast<float2 *>(d_lut_entries);
    if (fma) build_lut_kernel<true><<<grid, 256, 0, st>>>(tab, period, n_first, n_entries, ratio);
    else     build_lut_kernel<false><<<grid, 256, 0, st>>>(tab, period, n_first, n_entries, ratio);
    return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;
}

static int aux_grid(uint64_t n)
{
    uint64_t b = (n + kBlock - 1) / kBlock;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int launch_expand_walk(const void *d_spans, const void *d_index, void *d_desc, uint32_t n_desc, void *stream)
{
    static_assert(sizeof(WalkSeg) == 4 * sizeof(u32x4), "a descriptor is four 16-byte vectors");
    if (n_desc == 0) return DPX_OK;
    span_descriptors_kernel<<<(n_desc * 4u + kBlock - 1) / kBlock, kBlock, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const u32x4 *>(d_spans), static_cast<const uint32_t *>(d_index), static_cast<u32x4 *>(d_desc), n_desc);
    return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;
}

int launch_copy(const void *d_in, void *d_out, uint64_t n_bytes, void *stream)
{
    const uint64_t n_vec = n_bytes / 16;
    const uint64_t grid = (n_vec + kBlock - 1) / kBlock;
    if (grid == 0) return DPX_OK;
    if (grid > 0x7fffffffull) return DPX_ERR_ARG;
    copy_kernel<<<(uint32_t)grid, kBlock, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const u32x4 *>(d_in), static_cast<u32x4 *>(d_out), n_vec);
    return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;
}

int launch_unpack_i16(const void *d_in, void *d_out, uint64_t n, void *stream)
{
    unpack_i16_kernel<<<aux_grid(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const uint32_t *>(d_in), static_cast<float2 *>(d_out), n);
    return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;
}

int launch_pack_i16(const void *d_in, void *d_out, uint64_t n, void *stream, int legacy_cast)
{
    pack_i16_kernel<<<aux_grid(n), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const float2 *>(d_in), static_cast<uint32_t *>(d_out), n, legacy_cast);
    return hipGetLastError() == hipSuccess ? DPX_OK : DPX_ERR_HIP;
}

int launch_ccexpf(void *d_z, uint64_t n, bool fma, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (fma) ccexpf_kernel<true><<<aux_grid(n), kBlock, 0, st>>>(static_cast<float2 *>(d_z), n);
    else     ccexpf_kernel<false><<<aux_grid(n), kBlock, 0, st>>>(static_cast<float2 *>(d_z), n);
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
