// test_dsp.cpp — the reference's own in-file tests (reference src/dsp.rs:57-83 test_cexpf,
// src/dsp.rs:136-157 test_bench_shift_frequency), written against the C++ mirror of `doppler::dsp`
// (include/doppler_dsp.hpp) and checked against the CPU oracle (oracle/liboracle.so; test infrastructure).
// Needs an MI355X: run by tests/test_gpu_cpp_mirror.py under `-m gpu`.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/doppler_dsp.hpp"
#include "../../oracle/doppler_oracle.h"

using doppler::dsp::Complex32;

static int failures = 0;
#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) {                                                      \
            printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);          \
            ++failures;                                                     \
        }                                                                   \
    } while (0)

static bool same_bits(const Complex32 *a, const orc_complex *b, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        float ar = a[i].real(), ai = a[i].imag();
        if (memcmp(&ar, &b[i].re, 4) && !(ar != ar && b[i].re != b[i].re)) return false;
        if (memcmp(&ai, &b[i].im, 4) && !(ai != ai && b[i].im != b[i].im)) return false;
    }
    return true;
}

// dsp.rs:49-55
static void assert_eq_delta(float a, float b, float delta)
{
    const float relative_error = fabsf((a - b) / b);
    if (relative_error >= delta) {
        printf("FAIL `(left == right)` (left: `%g`, right: `%g`)\n", a, b);
        ++failures;
    }
}

// dsp.rs:57-83, vector for vector, same tolerance; then bit-exactness against the oracle's ccexpf
// (libm cexpf through the reference's own complex.c when oracle/_ref is built).
static void test_cexpf()
{
    Complex32 a(0.0f, 0.0f);
    doppler::dsp::ccexpf(&a);
    assert_eq_delta(a.real(), 1.0f, 0.000001f);
    CHECK(a.imag() == 0.0f);             // the reference divides by b = 0.0 here: NaN >= delta is false, so it passes on any value

    a = Complex32(1.0f, 1.0f);
    doppler::dsp::ccexpf(&a);
    assert_eq_delta(a.real(), 1.468694f, 0.000001f);
    assert_eq_delta(a.imag(), 2.2873552f, 0.000001f);

    a = Complex32(70.0f, 70.0f);
    doppler::dsp::ccexpf(&a);
    assert_eq_delta(a.real(), 1593075600000000000000000000000.0f, 0.000001f);
    assert_eq_delta(a.imag(), 1946674600000000000000000000000.0f, 0.000001f);

    a = Complex32(1000000.0f, 1000000.0f);
    doppler::dsp::ccexpf(&a);
    CHECK(a.real() == INFINITY);
    CHECK(a.imag() == -INFINITY);

    const float vals[] = {0.0f, -0.0f, 1.0f, -1.0f, 70.0f, 88.0f, 88.5f, 89.0f, 176.5f, 265.0f, 1e6f, -103.5f, -104.0f, -150.0f,
                          -31.415928f, 119.99999f, 120.0f, 3450.123f, 1e-13f, 1e-45f, 8388608.0f, INFINITY, -INFINITY, NAN};
    for (float re : vals)
        for (float im : vals) {
            Complex32 z(re, im);
            doppler::dsp::ccexpf(&z);
            orc_complex o = {re, im};
            orc_ccexpf(&o);
            if (!same_bits(&z, &o, 1)) {
                printf("FAIL ccexpf(%g, %g): got (%g, %g), oracle (%g, %g)\n", re, im, z.real(), z.imag(), o.re, o.im);
                ++failures;
            }
        }
}

// dsp.rs:136-157: the reference asserts nothing here; the oracle supplies the expected values.
static void test_bench_shift_frequency()
{
    uint32_t samplenr = 0, samplenr_o = 0;
    const float shift_hz = 815000.0f;
    const uint32_t samplerate = 2400000;
    const std::vector<uint8_t> input(1000000, 0xAA);
    const std::vector<Complex32> complex_input = doppler::dsp::convert_iqf32_to_complex(input);
    CHECK(complex_input.size() == 125000);
    std::vector<orc_complex> oin(complex_input.size()), oout(complex_input.size());
    CHECK(orc_convert_iqf32_to_complex(input.data(), input.size(), oin.data()) == 125000);
    CHECK(same_bits(complex_input.data(), oin.data(), oin.size()));
    int iterator = 0;
    for (;;) {
        const std::vector<Complex32> out = doppler::dsp::shift_frequency(complex_input, samplenr, shift_hz, samplerate);
        orc_shift_frequency(oin.data(), oin.size(), &samplenr_o, shift_hz, samplerate, oout.data());
        CHECK(out.size() == oout.size());
        CHECK(samplenr == samplenr_o);
        if (!same_bits(out.data(), oout.data(), oout.size())) {
            printf("FAIL shift_frequency differs from the oracle at call %d\n", iterator);
            ++failures;
            break;
        }
        iterator += 1;
        if (iterator > 300) break;
    }
}

static void test_operators_and_panics()
{
    std::vector<uint8_t> raw(8192);
    for (size_t i = 0; i < raw.size(); ++i) raw[i] = (uint8_t)(i * 2654435761u >> 13);
    const std::vector<Complex32> cx = doppler::dsp::convert_iqi16_to_complex(raw);
    std::vector<orc_complex> ocx(2048);
    CHECK(orc_convert_iqi16_to_complex(raw.data(), raw.size(), ocx.data()) == 2048);
    CHECK(cx.size() == 2048 && same_bits(cx.data(), ocx.data(), 2048));
    // fused closure body vs the oracle's three passes (main.rs:62-99)
    uint32_t sn = 0, sn_o = 0;
    std::vector<uint8_t> want(8192);
    size_t cnt = 0;
    for (int blk = 0; blk < 3; ++blk) {
        const std::vector<uint8_t> got = doppler::dsp::shift_block(raw, doppler::dsp::I16, doppler::dsp::I16, sn, 5000.0f, 1024000);
        CHECK(orc_shift_block(raw.data(), raw.size(), ORC_FMT_I16, ORC_FMT_I16, &sn_o, 5000.0f, 1024000, want.data(), &cnt) == 0);
        CHECK(got.size() == 8192 && cnt == 2048 && sn == sn_o && memcmp(got.data(), want.data(), 8192) == 0);
    }
    bool panicked = false;
    try { raw.resize(8190); doppler::dsp::convert_iqi16_to_complex(raw); } catch (const doppler::dsp::Panic &p) {
        panicked = strstr(p.what(), "inbuf.len() % 4 == 0") != nullptr;
    }
    CHECK(panicked);
    panicked = false;
    try { raw.resize(8188); doppler::dsp::convert_iqf32_to_complex(raw); } catch (const doppler::dsp::Panic &) { panicked = true; }
    CHECK(panicked);
}

int main()
{
    try {
        test_cexpf();
        test_bench_shift_frequency();
        test_operators_and_panics();
    } catch (const std::exception &e) {
        printf("FAIL exception: %s\n", e.what());
        return 2;
    }
    printf(failures ? "test_dsp: %d failure(s)\n" : "test_dsp: all passed (%d failures)\n", failures);
    return failures ? 1 : 0;
}
