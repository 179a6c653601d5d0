// doppler_dsp.hpp — C++ mirror of the reference's operator module `doppler::dsp`
// (reference src/dsp.rs, exported at src/lib.rs:35), header-only, over the C ABI of doppler_hip.h.
//
// Same function names, argument order and meaning as the Rust functions:
//   dsp.rs:85   pub fn convert_iqi16_to_complex(inbuf: &[u8]) -> Vec<Complex<f32>>
//   dsp.rs:101  pub fn convert_iqf32_to_complex(inbuf: &[u8]) -> Vec<Complex<f32>>
//   dsp.rs:117  pub fn shift_frequency(inbuf: &[Complex<f32>], samplenum: &mut u32, shift_hz: f32,
//                                      samplerate: u32) -> Vec<Complex<f32>>
//   dsp.rs:40   extern { pub fn ccexpf(z: *mut LiquidComplex32); }      (in place, returns nothing)
// Failure behaviour: where the Rust code panics (assert! at dsp.rs:87 / dsp.rs:103) these throw
// doppler::dsp::Panic carrying the same message; any other library error throws Error.
// All arithmetic runs on the GPU through libdoppler_hip.so; there is no host implementation here.
#pragma once
#include <complex>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "doppler_hip.h"

namespace doppler {
namespace dsp {

using Complex32 = std::complex<float>;   // layout {re, im}: num::complex::Complex<f32> / RustComplex (complex.c:28-31)
static_assert(sizeof(Complex32) == sizeof(dpx_complex32), "Complex<f32> is two consecutive f32");

struct Panic : std::logic_error { using std::logic_error::logic_error; };   // the reference would panic here
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

namespace detail {
inline dpx_ctx *context()
{
    static dpx_ctx *ctx = [] {
        dpx_ctx *c = nullptr;
        const int rc = dpx_ctx_create(0, &c);
        if (rc != DPX_OK) throw Error(std::string("doppler_hip: ") + dpx_last_error());   // no CPU fallback exists
        return c;
    }();
    return ctx;
}
inline void check(int rc, const char *panic_msg = nullptr)
{
    if (rc == DPX_OK) return;
    if (rc == DPX_ERR_BLOCK_LEN && panic_msg) throw Panic(panic_msg);
    throw Error(std::string("doppler_hip error ") + std::to_string(rc) + ": " + dpx_last_error());
}
}  // namespace detail

inline std::vector<Complex32> convert_iqi16_to_complex(const std::vector<uint8_t> &inbuf)
{
    std::vector<Complex32> out(inbuf.size() / 4 + 1);
    size_t n = 0;
    detail::check(dpx_convert_iqi16_to_complex(detail::context(), inbuf.data(), inbuf.size(),
                                               reinterpret_cast<dpx_complex32 *>(out.data()), out.size(), &n),
                  "assertion failed: inbuf.len() % 4 == 0");
    out.resize(n);
    return out;
}

inline std::vector<Complex32> convert_iqf32_to_complex(const std::vector<uint8_t> &inbuf)
{
    std::vector<Complex32> out(inbuf.size() / 8 + 1);
    size_t n = 0;
    detail::check(dpx_convert_iqf32_to_complex(detail::context(), inbuf.data(), inbuf.size(),
                                               reinterpret_cast<dpx_complex32 *>(out.data()), out.size(), &n),
                  "assertion failed: inbuf.len() % 8 == 0");
    out.resize(n);
    return out;
}

inline std::vector<Complex32> shift_frequency(const std::vector<Complex32> &inbuf, uint32_t &samplenum,
                                              float shift_hz, uint32_t samplerate)
{
    std::vector<Complex32> out(inbuf.size());
    detail::check(dpx_shift_frequency(detail::context(), reinterpret_cast<const dpx_complex32 *>(inbuf.data()),
                                      inbuf.size(), &samplenum, shift_hz, samplerate,
                                      reinterpret_cast<dpx_complex32 *>(out.data())));
    return out;
}

// dsp.rs:40-42 / complex.c:33-39: z <- cexpf(z), in place, returns nothing.
inline void ccexpf(Complex32 *z)
{
    detail::check(dpx_ccexpf(detail::context(), reinterpret_cast<dpx_complex32 *>(z), 1));
}

// the fused body of the `shift` closure (main.rs:65-94): returns the packed output bytes
enum DataType { I16 = DPX_FMT_I16, F32 = DPX_FMT_F32 };   // usage.rs:38-42
inline std::vector<uint8_t> shift_block(const std::vector<uint8_t> &invec, DataType intype, DataType outtype,
                                        uint32_t &samplenr, float shift_hz, uint32_t samplerate)
{
    const size_t ibs = intype == I16 ? 4 : 8, obs = outtype == I16 ? 4 : 8;
    std::vector<uint8_t> out(invec.size() / ibs * obs + 8);
    size_t n = 0;
    detail::check(dpx_shift_block(detail::context(), invec.data(), invec.size(), intype, out.data(), out.size(), outtype,
                                  &samplenr, shift_hz, samplerate, &n),
                  intype == I16 ? "assertion failed: inbuf.len() % 4 == 0" : "assertion failed: inbuf.len() % 8 == 0");
    out.resize(n * obs);
    return out;
}

}  // namespace dsp
}  // namespace doppler
