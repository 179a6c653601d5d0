// main.cpp — `doppler const` / `doppler track`: stdin -> MI355X -> stdout  (SURVEY.md section 8f, N1/N3/N4).
//
// Keeps the observable behaviour of the reference driver (reference src/main.rs:51-210):
//   * the stream is consumed in 8192-byte blocks; the shift may change only between blocks;
//   * the loop ends with the first short (possibly empty) read (main.rs:98,115-117);
//   * a final block that is not a whole number of samples aborts before producing output for that
//     block (the reference's assert!, dsp.rs:87/103) — exit status 101 like a Rust panic;
//   * the sample counter `samplenr` (main.rs:60) is carried through the whole run.
// What changes is the granularity of the work: blocks are gathered into slabs (as many complete
// blocks as are available without waiting, up to DOPPLER_SLAB_BYTES), each slab is one plan + one
// fused launch, and three slabs rotate (the dpx_stream_* ring of the C ABI) so that reading, PCIe
// copies, the kernel and writing overlap.
// A live 1 Msps pipe therefore still moves in ~8-64 KiB steps, a file at PCIe speed.
#include <errno.h>
#include <poll.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <fstream>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/doppler_hip.h"
#include "../host/orbit.h"
#include "../host/schedule.h"
#include "args.h"

namespace {

// fern format of the reference (main.rs:220-223): "{ts}.{ms:3} [{level:<6} {module:<30} {line:>3}]  {msg}"
void info(const char *fmt, ...)
{
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    struct tm tmv;
    localtime_r(&tv.tv_sec, &tmv);
    char ts[32];
    strftime(ts, sizeof(ts), "%Y-%m-%dT%H:%M:%S", &tmv);
    fprintf(stderr, "%s.%3d [%-6s %-30s %3d]  ", ts, (int)(tv.tv_usec / 1000), "INFO", "doppler", 0);
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

#define DPXCHK(call)                                                                     \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != DPX_OK) {                                                             \
            fprintf(stderr, "doppler: %s failed (%d): %s\n", #call, rc_, dpx_last_error()); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

bool write_all(int fd, const char *p, size_t n)
{
    while (n) {
        const ssize_t w = write(fd, p, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += w;
        n -= (size_t)w;
    }
    return true;
}

// Blocks until at least one full 8192-byte block or EOF, then keeps taking what is already
// there (never waits for more) up to `cap`.  Returns bytes read; *eof set at end of input.
size_t gather(int fd, char *buf, size_t cap, bool *eof)
{
    size_t n = 0;
    while (n < cap) {
        if (n > 0 && n % DPX_BUFFER_SIZE == 0) {     // on a block boundary: take more only if it is already there
            struct pollfd p = {fd, POLLIN, 0};
            if (poll(&p, 1, 0) <= 0 || !(p.revents & (POLLIN | POLLHUP))) break;
        }
        const ssize_t r = read(fd, buf + n, cap - n);
        if (r < 0) {
            if (errno == EINTR) continue;
            fprintf(stderr, "doppler collect error\n");    // main.rs:63
            exit(101);
        }
        if (r == 0) {
            *eof = true;
            break;
        }
        n += (size_t)r;
    }
    return n;
}

}  // namespace

int main(int argc, char **argv)
{
    dpx::CommandArgs args;
    bool exit_now = false;
    const int st = dpx::parse_args(argc, argv, &args, &exit_now);
    if (exit_now) return st;

    const int in_fmt = args.inputtype == dpx::DataType::I16 ? DPX_FMT_I16 : DPX_FMT_F32;
    const int out_fmt = args.outputtype == dpx::DataType::I16 ? DPX_FMT_I16 : DPX_FMT_F32;
    const size_t ibs = in_fmt == DPX_FMT_I16 ? 4 : 8;

    info("doppler %s (MI355X hot path)\n\n", "1.1.10");
    std::function<double(int64_t)> range_rate;      // replay: seconds since --time
    std::unique_ptr<dpx::Sgp4> sgp4;
    dpx::Observer observer;
    std::vector<double> rr_table;
    if (args.mode == dpx::Mode::Const) {
        info("constant shift mode");                                        // main.rs:103-107
        info("\tIQ samplerate   : %u", args.samplerate);
        info("\tIQ input type   : %s", dpx::datatype_name(args.inputtype));
        info("\tIQ output type  : %s\n", dpx::datatype_name(args.outputtype));
        info("\tfrequency shift : %d Hz", args.shift);
    } else {
        info("tracking mode");                                              // main.rs:123-134
        info("\tIQ samplerate   : %u", args.samplerate);
        info("\tIQ input type   : %s", dpx::datatype_name(args.inputtype));
        info("\tIQ output type  : %s\n", dpx::datatype_name(args.outputtype));
        if (!args.range_rate_file.empty()) {
            std::ifstream f(args.range_rate_file);
            double v;
            while (f >> v) rr_table.push_back(v);
            if (rr_table.empty()) {
                info("cannot read range rates from %s", args.range_rate_file.c_str());
                return 1;
            }
            info("\trange-rate table: %s (%zu s)", args.range_rate_file.c_str(), rr_table.size());
            range_rate = [&rr_table](int64_t dt) {
                const size_t i = dt < 0 ? 0 : ((uint64_t)dt >= rr_table.size() ? rr_table.size() - 1 : (size_t)dt);
                return rr_table[i];
            };
        } else {
            info("\tTLE file        : %s", args.tlefile.c_str());
            info("\tTLE name        : %s", args.tlename.c_str());
            info("\tlocation        : Location { lat: %g, lon: %g, alt: %g }", args.location.lat, args.location.lon, args.location.alt);
            dpx::Tle tle;
            std::string err;
            if (!dpx::tle_from_file(args.tlefile.c_str(), args.tlename.c_str(), &tle, &err)) {
                info("%s", err.c_str());                                     // main.rs:143-146
                return 1;
            }
            sgp4.reset(new dpx::Sgp4);
            if (!sgp4->init(tle, &err)) {
                info("%s", err.c_str());
                return 1;
            }
            observer.lat_deg = args.location.lat;
            observer.lon_deg = args.location.lon;
            observer.alt_m = args.location.alt;
        }
        if (args.has_time) {
            time_t t = (time_t)args.time_unix;
            struct tm g;
            gmtime_r(&t, &g);
            char b[40];
            strftime(b, sizeof(b), "%Y-%m-%dT%H:%M:%SZ", &g);
            info("\ttime            : %s", b);
        }
        info("\tfrequency       : %u Hz", args.frequency);
        info("\toffset          : %d Hz\n\n\n", args.has_offset ? args.offset : 0);
    }

    dpx_ctx *ctx = nullptr;
    if (dpx_ctx_create(0, &ctx) != DPX_OK) {
        fprintf(stderr, "doppler: %s\n", dpx_last_error());
        return 1;
    }

    size_t slab_bytes = 8u << 20;       // measured on the GPU box: 4-8 MiB slabs beat larger ones end to end
    if (const char *e = getenv("DOPPLER_SLAB_BYTES")) slab_bytes = strtoull(e, nullptr, 0);
    slab_bytes = (slab_bytes / DPX_BUFFER_SIZE) * DPX_BUFFER_SIZE;
    if (slab_bytes < DPX_BUFFER_SIZE) slab_bytes = DPX_BUFFER_SIZE;

    // three pinned slabs in rotation (include/doppler_hip.h, "streaming from host memory")
    dpx_stream *stream = nullptr;
    DPXCHK(dpx_stream_create(ctx, in_fmt, out_fmt, args.samplerate, /*samplenr, main.rs:60*/ 0, slab_bytes, 3, &stream));   // kSlabs below

    // The ring is driven from two threads: this one reads stdin into slabs and submits them, the writer thread
    // below waits for the oldest slab, writes it to stdout and frees it — read(), the GPU and write() overlap.
    constexpr int kSlabs = 3;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t submitted = 0, drained = 0;    // guarded by mu
    bool input_done = false;
    std::thread writer([&]() {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return submitted > drained || input_done; });
                if (submitted == drained) return;       // input_done and nothing left
            }
            const void *out = nullptr;
            size_t nbytes = 0;
            DPXCHK(dpx_stream_next(stream, &out, &nbytes));
            if (!write_all(STDOUT_FILENO, static_cast<const char *>(out), nbytes)) {
                info("doppler stdout.write error: %s", strerror(errno));       // main.rs:86
                exit(1);
            }
            DPXCHK(dpx_stream_release(stream));
            {
                std::lock_guard<std::mutex> lk(mu);
                ++drained;
            }
            cv.notify_all();
        }
    });

    // the reference's loop state
    const bool replay = args.mode == dpx::Mode::Track && args.has_time;
    std::unique_ptr<dpx::ReplaySchedule> sched;
    if (args.mode == dpx::Mode::Track) {
        if (!range_rate) {
            const double t0 = (double)args.time_unix;
            range_rate = [&sgp4, &observer, t0](int64_t dt) { return sgp4->observe(observer, t0 + (double)dt).range_rate_km_s; };
        }
        if (replay) sched.reset(new dpx::ReplaySchedule(range_rate, args.samplerate, args.frequency, args.has_offset, args.offset));
    }
    int64_t last_log = 0;
    double last_wall_log = 0;

    bool eof = false, ragged = false;
    uint64_t total_samples = 0;
    struct timeval tv_start;
    gettimeofday(&tv_start, nullptr);
    std::vector<dpx_segment> segs;
    while (!eof) {
        void *buf = nullptr;
        size_t cap = 0;
        {   // every slab in flight: wait for the writer to free the oldest
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return submitted - drained < (uint64_t)kSlabs; });
        }
        DPXCHK(dpx_stream_acquire(stream, &buf, &cap));
        const size_t n = gather(STDIN_FILENO, static_cast<char *>(buf), cap, &eof);
        // main.rs:63-68: complete blocks always; the trailing short block only if it is whole samples
        // (gather() stops only on block boundaries unless the input ended, so tail != 0 implies eof)
        const size_t full = n / DPX_BUFFER_SIZE * DPX_BUFFER_SIZE;
        size_t tail = n - full;
        if (tail % ibs != 0) {
            ragged = true;                 // the reference panics on this block: no output for it
            tail = 0;
        }
        const size_t use = full + tail;
        const size_t n_samples = use / ibs;
        total_samples += n_samples;

        // shift schedule for the blocks of this slab
        segs.clear();
        const size_t spb = DPX_BUFFER_SIZE / ibs;
        const size_t n_blocks = (use + DPX_BUFFER_SIZE - 1) / DPX_BUFFER_SIZE;
        if (args.mode == dpx::Mode::Const) {
            if (n_samples) segs.push_back({(uint64_t)n_samples, (float)args.shift});   // main.rs:110
        } else {
            float wall_shift = 0;
            if (!replay) {                 // main.rs:186-205: wall clock; one evaluation per slab
                struct timeval tv;
                gettimeofday(&tv, nullptr);
                const double now = tv.tv_sec + tv.tv_usec * 1e-6;
                double rr;
                dpx::LookAngles la;
                if (sgp4) { la = sgp4->observe(observer, now); rr = la.range_rate_km_s; }
                else rr = range_rate(0);
                const double doppler_hz = (rr * 1000.0 / 299792458.) * (double)args.frequency * (-1.0);
                wall_shift = (float)doppler_hz + (float)(args.has_offset ? args.offset : 0);
                if (now - last_wall_log >= 1.0) {
                    last_wall_log = now;
                    if (sgp4) {
                        info("az                  : %.2f\xC2\xB0", la.az_deg);
                        info("el                  : %.2f\xC2\xB0", la.el_deg);
                        info("range               : %.0f km", la.range_km);
                    }
                    info("range rate          : %.3f km/sec", rr);
                    info("doppler@%.3f MHz : %.2f Hz\n", (float)args.frequency / 1000000.0f, doppler_hz);
                }
            }
            for (size_t b = 0; b < n_blocks; ++b) {
                const size_t cnt = std::min(spb, n_samples - b * spb);
                float hz = wall_shift;
                if (replay) {
                    hz = sched->next_block_shift();
                    if (sched->dt_seconds() - last_log >= 5) {             // main.rs:167-175
                        last_log = sched->dt_seconds();
                        info("time                : +%lld s", (long long)sched->dt_seconds());
                        info("range rate          : %.3f km/sec", sched->last_range_rate());
                        info("doppler@%.3f MHz : %.2f Hz\n", (float)args.frequency / 1000000.0f, sched->last_doppler_hz());
                    }
                    sched->block_done(cnt);
                }
                if (!segs.empty() && memcmp(&segs.back().shift_hz, &hz, sizeof(float)) == 0) segs.back().n_samples += cnt;
                else segs.push_back({(uint64_t)cnt, hz});
            }
        }
        DPXCHK(dpx_stream_submit(stream, use, segs.data(), segs.size()));
        {
            std::lock_guard<std::mutex> lk(mu);
            ++submitted;
        }
        cv.notify_all();        // the writer hands every slab over as soon as it is done: on a live pipe the latency is one slab
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        input_done = true;
    }
    cv.notify_all();
    writer.join();

    if (getenv("DOPPLER_STATS")) {      // steady-state rate: first read to last write, start-up excluded
        struct timeval tv_end;
        gettimeofday(&tv_end, nullptr);
        const double dt = (tv_end.tv_sec - tv_start.tv_sec) + (tv_end.tv_usec - tv_start.tv_usec) * 1e-6;
        fprintf(stderr, "doppler stats: %llu samples in %.6f s = %.1f Msamples/s (stdin -> stdout, start-up excluded)\n",
                (unsigned long long)total_samples, dt, total_samples / dt / 1e6);
    }
    dpx_stream_destroy(stream);
    dpx_ctx_destroy(ctx);
    if (ragged) {
        fprintf(stderr, "thread 'main' panicked at 'assertion failed: inbuf.len() %% %zu == 0'\n", ibs);
        return 101;
    }
    return 0;
}
