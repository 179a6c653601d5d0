// main.cpp — `doppler const` / `doppler track`: stdin -> MI355X -> stdout  (SURVEY.md section 8f, N1/N3/N4).
//
// Keeps the observable behaviour of the reference driver (reference src/main.rs:51-210):
//   * the stream is consumed in 8192-byte blocks; the shift may change only between blocks;
//   * the loop ends with the first short (possibly empty) read (main.rs:98,115-117);
//   * a final block that is not a whole number of samples aborts before producing output for that
//     block (the reference's assert!, dsp.rs:87/103) — exit status 101 like a Rust panic;
//   * the sample counter `samplenr` (main.rs:60) is carried through the whole run;
//   * a failed read or write is a panic in the reference (`expect` / `unwrap`, main.rs:63,86-95): status 101.
// What changes is the granularity of the work: blocks are gathered into slabs, each slab is one plan + one fused
// launch, and a ring of slabs rotates (the dpx_stream_* ring of the C ABI) so that reading, PCIe copies in both
// directions, the kernels and writing overlap.  Three shapes of I/O:
//   * a pipe or terminal on stdin: ONE reader takes as many complete blocks as are available without waiting, up to
//     DOPPLER_SLAB_BYTES — a live 1 Msps pipe still moves in ~8-64 KiB steps, a fast producer in whole slabs;
//   * a regular file on stdin: the slabs are filled in parallel by DOPPLER_IO_THREADS workers with pread() at the
//     offsets the sequential loop would have reached (a single read() stream is what bounded round 1 at 1.4 Gsamples/s);
//   * a regular file on stdout: drained the same way with pwrite(); a pipe gets one ordered writer.
// With --gpus N (extension) the ring spans N GPUs: slab k runs on GPU k mod N (dpx_stream_create_multi) — the slabs ARE
// the time chunks of the sharded design, the counter is carried on the host, every GPU copies its own output back
// (--gather rccl: the outputs of GPUs 1..N-1 go over RCCL into the first GPU and leave from there).
// Live track mode (no --time) evaluates the orbit once per 8192-byte block, like the reference (main.rs:186-205): its
// slabs are a single block.
#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/uio.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/doppler_hip.h"
#include "../../../include/doppler_hip_debug.h"   // dpx_stream_get_stats (the --stats line), dpx_set_tuning (DOPPLER_VARIANT)
#include "../../../include/doppler_hip_host.h"    // track schedule, orbit
#include "../host/orbit.h"
#include "../host/schedule.h"
#include "args.h"

namespace {

// fern format of the reference (main.rs:220-223): "{ts}.{ms:3} [{level:<6} {module:<30} {line:>3}]  {msg}"
// module: the binary's crate name, as module_path!() gives it for main.rs; line: the call site (this file's — the
// reference prints main.rs's own line numbers, e.g. 103 for the banner)
std::mutex g_log_mu;
#define info(...) info_at(__LINE__, __VA_ARGS__)
void info_at(int line, const char *fmt, ...)
{
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    struct tm tmv;
    localtime_r(&tv.tv_sec, &tmv);
    char ts[32];
    strftime(ts, sizeof(ts), "%Y-%m-%dT%H:%M:%S", &tmv);
    std::lock_guard<std::mutex> lk(g_log_mu);
    fprintf(stderr, "%s.%3d [%-6s %-30s %3d]  ", ts, (int)(tv.tv_usec / 1000), "INFO", "doppler", line);
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

// Wall clock of live track mode.  DOPPLER_FAKE_CLOCK=<unix seconds>,<seconds per call> (tests only) replaces it with a
// clock that advances by a fixed step per query, which makes the per-block cadence of main.rs:186-205 observable.
double wall_clock()
{
    static const char *fake = getenv("DOPPLER_FAKE_CLOCK");
    if (fake) {
        static double t0 = 0, step = 0, t = 0;
        static bool init = false;
        if (!init) {
            sscanf(fake, "%lf,%lf", &t0, &step);
            t = t0;
            init = true;
        } else {
            t += step;
        }
        return t;
    }
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + tv.tv_usec * 1e-6;
}

// Reads into buf until at least one full 8192-byte block or EOF, then keeps taking what is already there (never waits
// for more) up to `cap`.  Returns bytes read, or -1 on a read error; *eof set at end of input.
// `stop` (optional): set by another thread when the run has failed elsewhere — a reader waiting for input that may never
// come (a pipe whose writer neither writes nor closes) gives up within a fifth of a second instead of blocking in read().
ssize_t gather(int fd, char *buf, size_t cap, bool *eof, const std::atomic<int> *stop = nullptr)
{
    size_t n = 0;
    while (n < cap) {
        if (n > 0 && n % DPX_BUFFER_SIZE == 0) {     // on a block boundary: take more only if it is already there
            struct pollfd p = {fd, POLLIN, 0};
            if (poll(&p, 1, 0) <= 0 || !(p.revents & (POLLIN | POLLHUP))) break;
        } else if (stop) {
            struct pollfd p = {fd, POLLIN, 0};
            int pr;
            while ((pr = poll(&p, 1, 200)) == 0 && *stop == 0) {}
            if (pr == 0) break;                      // the run is being wound down: hand over what there is
        }
        const ssize_t r = read(fd, buf + n, cap - n);
        if (r < 0) {
            if (errno == EINTR) continue;
            return -1;
        }
        if (r == 0) {
            *eof = true;
            break;
        }
        n += (size_t)r;
    }
    return (ssize_t)n;
}

// A pipe on stdin / stdout is given the largest buffer the system allows (default 64 KiB: a context switch per 16 Ki
// samples; with 1 MiB and more the reader and the writer move megabytes per system call).  Returns the size in effect.
long grow_pipe(int fd, bool grow = true)
{
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISFIFO(st.st_mode)) return 0;
    if (!grow) return fcntl(fd, F_GETPIPE_SZ);
    long cap = 1 << 20;
    if (FILE *f = fopen("/proc/sys/fs/pipe-max-size", "r")) {
        long v = 0;
        if (fscanf(f, "%ld", &v) == 1 && v > 0) cap = v;
        fclose(f);
    }
    // DOPPLER_PIPE_BYTES=N: ask for that much and no more (a pipeline that wants small buffers; the tests)
    if (const char *e = getenv("DOPPLER_PIPE_BYTES")) {
        const long v = atol(e);
        if (v >= (64 << 10)) { (void)fcntl(fd, F_SETPIPE_SZ, (int)std::min(v, cap)); return fcntl(fd, F_GETPIPE_SZ); }
    }
    // a privileged process may go beyond pipe-max-size: try 16 MiB first, then the limit, then halves of it
    for (long want : {16L << 20, cap, cap / 2, cap / 4})
        if (want >= (64 << 10) && fcntl(fd, F_SETPIPE_SZ, (int)want) >= 0) break;
    return fcntl(fd, F_GETPIPE_SZ);
}

bool pread_all(int fd, char *buf, size_t n, off_t off)
{
    while (n) {
        const ssize_t r = pread(fd, buf, n, off);
        if (r < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        if (r == 0) return false;       // the file shrank under us
        buf += r;
        off += r;
        n -= (size_t)r;
    }
    return true;
}

bool write_all(int fd, const char *p, size_t n, bool positioned, off_t off)
{
    while (n) {
        const ssize_t w = positioned ? pwrite(fd, p, n, off) : write(fd, p, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += w;
        off += w;
        n -= (size_t)w;
    }
    return true;
}

// Output into a pipe without write()'s page allocation and kernel copy: the bytes are copied (in user space) into a
// staging ring of ordinary pages, which vmsplice() then lends to the pipe.  A lent page may be reused once the reader has
// taken it; the pipe holds at most `slots` pages, every piece starts on a page of its own, so a page that lies more than
// `slots` pages behind the write position has left the pipe — the ring is four times that long.  (The pinned output slab
// itself is never lent: it is recycled at once, and its pages belong to the GPU runtime.)
//
// OPT-IN (DOPPLER_VMSPLICE=1; round 4).  vmsplice gives no completion signal, and "has left doppler's pipe" is not "has been
// consumed": a reader that forwards pipe buffers by reference — splice() into another pipe or a socket, tee(2); `pv` does
// so by default — keeps pointing at ring pages after they left this pipe, and a downstream queue longer than the ring's
// slack (a later stage that enlarged its own pipe while ours stayed at 64 KiB) would see them overwritten: silent
// corruption.  The reference writes with a plain `stdout.write` (src/main.rs:86,92) and has no such hazard, so that is
// the default here too; lending is for pipelines whose next stage is known to COPY (read()), where it is worth 2-5 x
// (profiles/r03_cli.md).  Even then it is refused when the output pipe could not be grown beyond the default 64 KiB.
class PipeLender {
public:
    bool open(int fd, long pipe_bytes)
    {
        page_ = (size_t)sysconf(_SC_PAGESIZE);
        fd_ = fd;
        return resize((size_t)pipe_bytes / page_);
    }
    // (rings are never unmapped while the process runs: a page may still sit in the pipe; the pipe keeps it alive anyway)
    bool resize(size_t slots)
    {
        if (slots < 16) return false;
        void *m = mmap(nullptr, 4 * slots * page_, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) return false;
        ring_ = static_cast<char *>(m);
        slots_ = slots;
        pages_ = 4 * slots;
        pos_ = 0;
        memset(ring_, 0, pages_ * page_);             // touch: the first lending pass should not page-fault
        return true;
    }
    bool active() const { return ring_ != nullptr; }
    // false: the write failed (errno set); *unsupported: vmsplice is not available here, nothing was written
    bool write(const char *p, size_t n, bool *unsupported)
    {
        *unsupported = false;
        while (n) {
            // the reader may have enlarged the pipe (F_SETPIPE_SZ on its end) since the ring was sized: the reuse distance
            // must follow the pipe's real capacity, so it is read again before every piece (a piece is at most `slots_`
            // pages and the ring four times that, so one piece lent into a pipe that has just grown cannot wrap onto a
            // page still queued), and a larger pipe gets a larger ring
            const long cap = fcntl(fd_, F_GETPIPE_SZ);
            if (cap > 0 && (size_t)cap / page_ > slots_ && !resize((size_t)cap / page_)) {
                if (first_) { *unsupported = true; ring_ = nullptr; }      // nothing lent yet: the caller writes instead
                errno = ENOMEM;
                return false;
            }
            const size_t piece_pages = std::min(slots_, (n + page_ - 1) / page_);
            if (pos_ + piece_pages > pages_) pos_ = 0;                  // pieces do not wrap around the ring
            const size_t bytes = std::min(n, piece_pages * page_);
            char *dst = ring_ + pos_ * page_;
            memcpy(dst, p, bytes);
            struct iovec iov = {dst, bytes};
            while (iov.iov_len) {
                const ssize_t w = vmsplice(fd_, &iov, 1, 0);
                if (w < 0) {
                    if (errno == EINTR) continue;
                    if (first_ && (errno == EINVAL || errno == ENOSYS || errno == EBADF)) {
                        *unsupported = true;
                        ring_ = nullptr;                                  // not lending from now on: active() says so, no further attempts
                        return false;
                    }
                    return false;
                }
                first_ = false;
                iov.iov_base = static_cast<char *>(iov.iov_base) + w;
                iov.iov_len -= (size_t)w;
            }
            pos_ += piece_pages;
            p += bytes;
            n -= bytes;
        }
        return true;
    }
private:
    int fd_ = -1;
    char *ring_ = nullptr;
    size_t page_ = 4096, slots_ = 0, pages_ = 0, pos_ = 0;
    bool first_ = true;
};

// a tiny pool: jobs run on `n` threads, in any order
class Workers {
public:
    explicit Workers(int n)
    {
        for (int i = 0; i < n; ++i) th_.emplace_back([this] { run(); });
    }
    ~Workers() { stop(); }
    void push(std::function<void()> job)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            q_.push_back(std::move(job));
        }
        cv_.notify_one();
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            done_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : th_) if (t.joinable()) t.join();
    }

private:
    void run()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return done_ || !q_.empty(); });
                if (q_.empty()) return;
                job = std::move(q_.front());
                q_.pop_front();
            }
            job();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    std::vector<std::thread> th_;
    bool done_ = false;
};

}  // namespace

int main(int argc, char **argv)
{
    dpx::CommandArgs args;
    bool exit_now = false;
    const int st = dpx::parse_args(argc, argv, &args, &exit_now);
    if (exit_now) return st;

    const int in_fmt = args.inputtype == dpx::DataType::I16 ? DPX_FMT_I16 : DPX_FMT_F32;
    const int out_fmt = args.outputtype == dpx::DataType::I16 ? DPX_FMT_I16 : DPX_FMT_F32;
    const size_t ibs = in_fmt == DPX_FMT_I16 ? 4 : 8, obs = out_fmt == DPX_FMT_I16 ? 4 : 8;

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

    // ---- GPUs: --gpus N / DOPPLER_GPUS (devices 0..N-1), or an explicit list in DOPPLER_DEVICES (a device may repeat:
    // separate contexts on it — how the multi-GPU path is exercised on a one-GPU box)
    std::vector<int> devices;
    if (const char *e = getenv("DOPPLER_DEVICES")) {
        for (const char *p = e; *p;) {
            char *end = nullptr;
            const long d = strtol(p, &end, 10);
            if (end == p || d < 0 || d > 1023 || devices.size() >= 64) {       // not a number: strtol made no progress
                fprintf(stderr, "doppler: DOPPLER_DEVICES must be a comma-separated list of device numbers, got \"%s\"\n", e);
                return 1;
            }
            devices.push_back((int)d);
            p = end;
            while (*p == ',' || *p == ' ') ++p;
        }
    }
    uint32_t n_gpus = args.gpus;
    if (n_gpus == 0) {
        const char *e = getenv("DOPPLER_GPUS");
        n_gpus = e ? (uint32_t)atoi(e) : (devices.empty() ? 1u : (uint32_t)devices.size());
    }
    if (n_gpus < 1 || n_gpus > 64) n_gpus = 1;
    if (devices.empty()) for (uint32_t i = 0; i < n_gpus; ++i) devices.push_back((int)i);
    devices.resize(n_gpus, devices.back());
    std::vector<dpx_ctx *> ctxs;
    auto destroy_ctxs = [&] { for (dpx_ctx *c : ctxs) dpx_ctx_destroy(c); };
    for (int d : devices) {
        dpx_ctx *c = nullptr;
        if (dpx_ctx_create(d, &c) != DPX_OK) {
            fprintf(stderr, "doppler: %s\n", dpx_last_error());
            destroy_ctxs();
            return 1;
        }
        ctxs.push_back(c);
    }
    if (n_gpus > 1) info("\tGPUs            : %u", n_gpus);
    // DOPPLER_I16_CAST=legacy: `as i16` as a 2016 rustc compiled it for x86-64 (out-of-range samples wrap instead of
    // saturating) — for byte-for-byte comparisons with files written by binaries of that time (include/doppler_hip.h)
    if (const char *e = getenv("DOPPLER_I16_CAST")) {
        const bool legacy = !strcmp(e, "legacy") || !strcmp(e, "wrap");
        if (!legacy && strcmp(e, "saturate") != 0) {
            fprintf(stderr, "doppler: DOPPLER_I16_CAST must be \"saturate\" (default) or \"legacy\", got \"%s\"\n", e);
            destroy_ctxs();
            return 1;
        }
        for (dpx_ctx *c : ctxs) dpx_set_i16_cast(c, legacy ? DPX_CAST_LEGACY_X86 : DPX_CAST_SATURATE);
    }

    const bool replay = args.mode == dpx::Mode::Track && args.has_time;
    const bool live_track = args.mode == dpx::Mode::Track && !args.has_time;
    size_t slab_bytes = 8u << 20;       // measured on the GPU box: 4-8 MiB slabs beat larger ones end to end
    if (const char *e = getenv("DOPPLER_SLAB_BYTES")) slab_bytes = strtoull(e, nullptr, 0);
    slab_bytes = (slab_bytes / DPX_BUFFER_SIZE) * DPX_BUFFER_SIZE;
    if (slab_bytes < DPX_BUFFER_SIZE) slab_bytes = DPX_BUFFER_SIZE;
    if (live_track) slab_bytes = DPX_BUFFER_SIZE;          // main.rs:186-205: predict.update() before every block

    struct stat sin, sout;
    const bool in_file = fstat(STDIN_FILENO, &sin) == 0 && S_ISREG(sin.st_mode) && !live_track && !getenv("DOPPLER_NO_PREAD");
    // (an O_APPEND descriptor ignores pwrite offsets: it gets the single ordered writer)
    const bool out_file = fstat(STDOUT_FILENO, &sout) == 0 && S_ISREG(sout.st_mode) && !(fcntl(STDOUT_FILENO, F_GETFL) & O_APPEND) &&
                          !getenv("DOPPLER_NO_PWRITE");
    int io_threads = 8;                  // measured, file -> /dev/null, 4 GiB: 4 / 8 / 16 / 24 workers 6.3 / 6.1 / 6.1 / 5.7 Gsamples/s
    if (const char *e = getenv("DOPPLER_IO_THREADS")) io_threads = std::max(1, std::min(64, atoi(e)));
    // a regular file arrives as fast as memory allows: larger slabs (16 MiB: 6.1-6.3 Gsamples/s against 3.6-6.0 with 8 MiB)
    if (in_file && !getenv("DOPPLER_SLAB_BYTES")) {
        // ... but no more pinned memory than the file needs: about one slab per worker for small files, at least 1 MiB
        const off_t pos0 = lseek(STDIN_FILENO, 0, SEEK_CUR);
        const uint64_t left = sin.st_size > (pos0 < 0 ? 0 : pos0) ? (uint64_t)(sin.st_size - (pos0 < 0 ? 0 : pos0)) : 0;
        const uint64_t share = (left / (uint64_t)io_threads + DPX_BUFFER_SIZE - 1) / DPX_BUFFER_SIZE * DPX_BUFFER_SIZE;
        slab_bytes = (size_t)std::min<uint64_t>(16u << 20, std::max<uint64_t>(1u << 20, share));
    }
    // pipes (the reference's only I/O mode, main.rs:57-58): the largest pipe buffers the system gives, and a ring deep
    // enough that the reader, the GPU and the writer each hold a slab with one to spare
    const bool grow = !getenv("DOPPLER_NO_PIPE_GROW");
    const long pipe_in = live_track || !grow ? 0 : grow_pipe(STDIN_FILENO);
    const long pipe_out = grow_pipe(STDOUT_FILENO, grow);
    // a pipe hands over at most its buffer per read: slabs of that size (measured, profiles/r03_cli.md: 3.4 Gsamples/s with
    // 1 MiB slabs against 2.5 with 8 MiB through two 1 MiB pipes) and a deeper ring
    if (!in_file && !live_track && pipe_in > 0 && !getenv("DOPPLER_SLAB_BYTES"))
        slab_bytes = std::max<size_t>(1u << 20, (size_t)pipe_in / DPX_BUFFER_SIZE * DPX_BUFFER_SIZE);
    const int slabs_per_gpu = in_file || out_file ? std::max(3, (2 * io_threads + (int)n_gpus - 1) / (int)n_gpus + 1) : live_track ? 3 : 6;
    const int n_slabs = slabs_per_gpu * (int)n_gpus;

    dpx_stream *stream = nullptr;
    dpx_stream_options sopt = {};
    bool gather_rccl = args.gather_rccl;
    if (const char *e = getenv("DOPPLER_GATHER")) gather_rccl = gather_rccl || strcmp(e, "rccl") == 0;
    sopt.gather = gather_rccl ? DPX_STREAM_GATHER_RCCL : DPX_STREAM_GATHER_D2H;
    if (gather_rccl && getenv("DOPPLER_GATHER_SELF")) sopt.gather |= DPX_STREAM_GATHER_SELF;       // tests on a one-GPU box
    if (const char *e = getenv("DPX_STREAM_PATH")) sopt.path = (uint32_t)atoi(e);
    if (gather_rccl) info("\tgather          : RCCL into GPU %d", devices[0]);
    if (dpx_stream_create_opts(ctxs.data(), (int)ctxs.size(), in_fmt, out_fmt, args.samplerate, /*samplenr, main.rs:60*/ 0, slab_bytes,
                               slabs_per_gpu, &sopt, &stream) != DPX_OK) {
        fprintf(stderr, "doppler: %s\n", dpx_last_error());
        destroy_ctxs();
        return 1;
    }

    // ---- shared state of the three sides (producer = this thread; consumer; recycling under `mu`)
    std::mutex mu;
    std::condition_variable cv;
    uint64_t acquired = 0, submitted = 0, handed = 0, released = 0;     // slab counts, guarded by mu
    bool input_done = false;
    std::atomic<int> failure{0};            // exit status to use once everything has been wound down (0 = none)
    std::string failure_msg;
    auto fail_with = [&](int status, const std::string &msg) {
        std::lock_guard<std::mutex> lk(mu);
        if (failure == 0) {
            failure = status;
            failure_msg = msg;
        }
        cv.notify_all();
    };

    struct SlabIo {             // per ring position
        char *in = nullptr;
        size_t filled = 0;      // bytes in the slab once `ready`
        bool ready = false;     // fill finished (guarded by mu)
        bool written = false;   // drain finished (guarded by mu)
    };
    std::vector<SlabIo> io((size_t)n_slabs);
    std::unique_ptr<Workers> fillers, drainers;
    if (in_file) fillers.reset(new Workers(io_threads));
    if (out_file) drainers.reset(new Workers(io_threads));
    const off_t out_base = out_file ? lseek(STDOUT_FILENO, 0, SEEK_CUR) : 0;
    // File to file, the output size is known up front: the output file is sized once and mapped, and the drain workers
    // copy into the mapping — page faults of different threads proceed in parallel, whereas write()/pwrite() on ONE file
    // serialise on its inode lock (measured on tmpfs: more pwrite workers made the command slower, not faster).
    char *out_map = nullptr;
    size_t out_map_len = 0;
    bool out_grown = false;             // this run extended the output file up front: a ragged tail trims it again
    if (in_file && out_file && out_base >= 0 && out_base % 4096 == 0 && !getenv("DOPPLER_NO_MMAP")) {
        const off_t in_pos = lseek(STDIN_FILENO, 0, SEEK_CUR);
        const uint64_t in_total = sin.st_size > (in_pos < 0 ? 0 : in_pos) ? (uint64_t)(sin.st_size - (in_pos < 0 ? 0 : in_pos)) : 0;
        out_map_len = (size_t)(in_total / ibs * obs);
        // The blocks are ALLOCATED before the file is mapped (posix_fallocate, never a sparse ftruncate): a full disk then
        // shows up here — the mapping is skipped and the ordered pwrite path reports the failed write as the reference's
        // `stdout.write` does (status 101) — not as a SIGBUS inside a drain worker's memcpy.  A file that is already
        // longer (`1<>file`: opened without O_TRUNC) keeps its length: the reference only overwrites the prefix.
        const off_t want = out_base + (off_t)out_map_len;
        out_grown = out_map_len != 0 && sout.st_size < want;
        if (out_map_len && (!out_grown || posix_fallocate(STDOUT_FILENO, out_base, (off_t)out_map_len) == 0)) {
            // a shell's `> file` is write-only, and a shared mapping needs a readable descriptor: reopen the same file
            const int rw = open("/proc/self/fd/1", O_RDWR);
            void *m = mmap(nullptr, out_map_len, PROT_READ | PROT_WRITE, MAP_SHARED, rw >= 0 ? rw : STDOUT_FILENO, out_base);
            if (m != MAP_FAILED) out_map = static_cast<char *>(m);
            if (rw >= 0) close(rw);
        } else if (out_grown) {
            if (ftruncate(STDOUT_FILENO, sout.st_size) != 0) {}  // the allocation failed half way: back to the old length
            out_grown = false;
        }
        if (!out_map) out_map_len = 0;
    }

    PipeLender lender;
    {
        // lending pages to the output pipe: only when asked for, and only into a pipe that did grow (see PipeLender)
        const char *e = getenv("DOPPLER_VMSPLICE");
        if (pipe_out > 65536 && e && atoi(e) != 0 && !getenv("DOPPLER_NO_VMSPLICE")) (void)lender.open(STDOUT_FILENO, pipe_out);
    }
    // ---- consumer: oldest slab -> stdout.  Pipe: write here, in order.  File: hand the slab to a drain worker with
    // its offset; the recycling (dpx_stream_release, in order) happens as the oldest writes complete.
    std::thread consumer([&]() {
        off_t out_off = out_base < 0 ? 0 : out_base;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return submitted > handed || input_done || failure != 0; });
                if (failure != 0 || submitted == handed) break;        // error, or input_done and nothing left
            }
            const void *out = nullptr;
            size_t nbytes = 0;
            if (dpx_stream_next(stream, &out, &nbytes) != DPX_OK) {
                fail_with(1, std::string("dpx_stream_next: ") + dpx_last_error());
                break;
            }
            uint64_t k;
            {
                std::lock_guard<std::mutex> lk(mu);
                k = handed++;
            }
            auto finish = [&, k](bool ok) {
                std::lock_guard<std::mutex> lk(mu);
                if (!ok && failure == 0) {
                    failure = 101;                                       // main.rs:86-95: unwrap() on a failed write
                    failure_msg = std::string("doppler stdout.write error: ") + strerror(errno);
                }
                io[k % io.size()].written = true;
                while (released < handed && io[released % io.size()].written) {   // recycle in order
                    io[released % io.size()].written = false;
                    if (dpx_stream_release(stream) != DPX_OK && failure == 0) {
                        failure = 1;
                        failure_msg = std::string("dpx_stream_release: ") + dpx_last_error();
                    }
                    ++released;
                }
                cv.notify_all();
            };
            const char *p = static_cast<const char *>(out);
            if (out_map && (uint64_t)(out_off - out_base) + nbytes <= out_map_len) {
                char *dst = out_map + (out_off - out_base);
                drainers->push([=] { memcpy(dst, p, nbytes); finish(true); });
            } else if (out_file) {
                const off_t off = out_off;
                drainers->push([=] { finish(write_all(STDOUT_FILENO, p, nbytes, true, off)); });
            } else if (lender.active()) {
                bool unsupported = false;
                bool ok = lender.write(p, nbytes, &unsupported);
                if (unsupported) ok = write_all(STDOUT_FILENO, p, nbytes, false, 0);      // (first piece: nothing was written yet)
                finish(ok);
            } else {
                finish(write_all(STDOUT_FILENO, p, nbytes, false, 0));
            }
            out_off += (off_t)nbytes;
        }
        if (drainers) drainers->stop();             // all queued writes are done when this returns
        if (out_map) {
            // write-back errors of the mapping belong to this run: report them like a failed write (main.rs:86-95)
            if (msync(out_map, out_map_len, MS_SYNC) != 0 && failure == 0) {
                std::lock_guard<std::mutex> lk(mu);
                failure = 101;
                failure_msg = std::string("doppler stdout.write error: ") + strerror(errno);
            }
            (void)munmap(out_map, out_map_len);
        }
        // a ragged tail (or a failure) produced less than was allocated up front: the file ends where the output does —
        // whether the workers went through the mapping or, after a failed mmap, through pwrite
        if (out_grown && ftruncate(STDOUT_FILENO, std::max<off_t>(out_off, sout.st_size)) != 0)
            info("doppler: cannot trim the output file: %s", strerror(errno));
        if (out_file && failure == 0 && out_off > 0) (void)lseek(STDOUT_FILENO, out_off, SEEK_SET);
    });

    // the reference's loop state
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
    auto log_time = [](double unix_s) {            // "{:}" of Tm::rfc3339() in UTC: 2015-01-22T19:48:05Z
        time_t t = (time_t)unix_s;
        struct tm g;
        gmtime_r(&t, &g);
        char b[40];
        strftime(b, sizeof(b), "%Y-%m-%dT%H:%M:%SZ", &g);
        info("time                : %s", b);
    };
    const double live_start = live_track ? wall_clock() : 0;

    bool ragged = false;
    uint64_t total_samples = 0;
    struct timeval tv_start;
    gettimeofday(&tv_start, nullptr);
    std::vector<dpx_segment> segs;

    // what happens to a slab once its bytes are in: schedule for its blocks, then submit (always in stream order)
    auto submit_slab = [&](size_t n) -> bool {
        // main.rs:63-68: complete blocks always; the trailing short block only if it is whole samples
        const size_t full = n / DPX_BUFFER_SIZE * DPX_BUFFER_SIZE;
        size_t tail = n - full;
        if (tail % ibs != 0) {
            ragged = true;                 // the reference panics on this block: no output for it
            tail = 0;
        }
        const size_t use = full + tail;
        const size_t n_samples = use / ibs;
        total_samples += n_samples;
        segs.clear();
        const size_t spb = DPX_BUFFER_SIZE / ibs;
        const size_t n_blocks = (use + DPX_BUFFER_SIZE - 1) / DPX_BUFFER_SIZE;
        if (args.mode == dpx::Mode::Const) {
            if (n_samples) segs.push_back({(uint64_t)n_samples, (float)args.shift});   // main.rs:110
        } else {
            float wall_shift = 0;
            if (live_track) {              // main.rs:186-205: wall clock, evaluated for this block (the slab IS one block)
                const double now = wall_clock();
                double rr;
                dpx::LookAngles la;
                if (sgp4) { la = sgp4->observe(observer, now); rr = la.range_rate_km_s; }
                else rr = range_rate((int64_t)(now - live_start));          // table extension: whole seconds since start-up
                const double doppler_hz = (rr * 1000.0 / 299792458.) * (double)args.frequency * (-1.0);
                wall_shift = (float)doppler_hz + (float)(args.has_offset ? args.offset : 0);
                if (now - last_wall_log >= 1.0) {                              // main.rs:190-198: once per second of wall time
                    last_wall_log = now;
                    log_time(now);
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
                        log_time((double)args.time_unix + (double)sched->dt_seconds());
                        if (sgp4) {                                            // what predict.sat held: updated one block earlier
                            const dpx::LookAngles la = sgp4->observe(observer, (double)args.time_unix + (double)sched->update_dt_seconds());
                            info("az                  : %.2f\xC2\xB0", la.az_deg);
                            info("el                  : %.2f\xC2\xB0", la.el_deg);
                            info("range               : %.0f km", la.range_km);
                        }
                        info("range rate          : %.3f km/sec", sched->last_range_rate());
                        info("doppler@%.3f MHz : %.2f Hz\n", (float)args.frequency / 1000000.0f, sched->last_doppler_hz());
                    }
                    sched->block_done(cnt);
                }
                if (!segs.empty() && memcmp(&segs.back().shift_hz, &hz, sizeof(float)) == 0) segs.back().n_samples += cnt;
                else segs.push_back({(uint64_t)cnt, hz});
            }
        }
        if (dpx_stream_submit(stream, use, segs.data(), segs.size()) != DPX_OK) {
            fail_with(1, std::string("dpx_stream_submit: ") + dpx_last_error());
            return false;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            ++submitted;
        }
        cv.notify_all();        // the consumer hands every slab over as soon as it is done: on a live pipe the latency is one slab
        return true;
    };

    if (!in_file && !live_track) {
        // ---- pipe / terminal, not live: a reader thread does nothing but move bytes from the pipe into pinned slabs (it is
        // the pipe's copy speed that bounds this path: one thread, ~9 GB/s), this thread schedules and submits them in
        // order.  The reader still takes whatever complete blocks are there and never waits past a block boundary.
        bool reader_done = false;                                           // guarded by mu
        std::vector<char> read_failed((size_t)n_slabs, 0);
        std::thread reader([&] {
            bool eof = false;
            uint64_t k = 0;
            while (!eof && failure == 0) {
                {   // every slab in use: wait for the oldest to be recycled
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return acquired - released < (uint64_t)n_slabs || failure != 0; });
                    if (failure != 0) break;
                }
                void *buf = nullptr;
                size_t cap = 0;
                if (dpx_stream_acquire(stream, &buf, &cap) != DPX_OK) {
                    fail_with(1, std::string("dpx_stream_acquire: ") + dpx_last_error());
                    break;
                }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ++acquired;
                    io[k % io.size()].ready = false;
                }
                const ssize_t n = gather(STDIN_FILENO, static_cast<char *>(buf), cap, &eof, &failure);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    io[k % io.size()].filled = n < 0 ? 0 : (size_t)n;
                    read_failed[k % io.size()] = n < 0;
                    io[k % io.size()].ready = true;
                }
                cv.notify_all();
                if (n < 0) break;
                ++k;
            }
            std::lock_guard<std::mutex> lk(mu);
            reader_done = true;
            cv.notify_all();
        });
        for (;;) {
            size_t n = 0;
            bool bad = false, have = false;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return (submitted < acquired && io[submitted % io.size()].ready) || (reader_done && submitted == acquired); });
                if (submitted < acquired) {
                    have = true;
                    n = io[submitted % io.size()].filled;
                    bad = read_failed[submitted % io.size()] != 0;
                }
            }
            if (!have) break;                                               // the reader is done and everything it acquired went in
            if (bad || failure != 0) {                                      // a failed read, or a failure elsewhere: hand the slab back empty
                if (bad) fail_with(101, "doppler collect error");           // main.rs:63 expect()
                (void)dpx_stream_submit(stream, 0, nullptr, 0);
                std::lock_guard<std::mutex> lk(mu);
                ++submitted;
                cv.notify_all();
                continue;
            }
            if (!submit_slab(n)) {
                // the slab stays acquired in the ring; nothing more can be submitted in order: let the reader wind down
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return reader_done; });
                break;
            }
        }
        reader.join();
    } else if (!in_file) {
        // ---- live track mode: one block per slab, the orbit evaluated for every block as it arrives (main.rs:186-205)
        bool eof = false;
        while (!eof && failure == 0) {
            {   // every slab in use: wait for the oldest to be recycled
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return submitted - released < (uint64_t)n_slabs || failure != 0; });
                if (failure != 0) break;
            }
            void *buf = nullptr;
            size_t cap = 0;
            if (dpx_stream_acquire(stream, &buf, &cap) != DPX_OK) {
                fail_with(1, std::string("dpx_stream_acquire: ") + dpx_last_error());
                break;
            }
            const ssize_t n = gather(STDIN_FILENO, static_cast<char *>(buf), cap, &eof);
            if (n < 0) {
                fail_with(101, "doppler collect error");                    // main.rs:63 expect()
                (void)dpx_stream_submit(stream, 0, nullptr, 0);             // hand the acquired slab back empty
                std::lock_guard<std::mutex> lk(mu);
                ++submitted;
                break;
            }
            if (!submit_slab((size_t)n)) break;
        }
    } else {
        // ---- regular file: the loop of main.rs reads min(8192, remaining) per block until a short (or empty) read,
        // i.e. it consumes the whole file; slab k holds bytes [k * slab_bytes, ...) and is filled by a worker.
        const off_t start = lseek(STDIN_FILENO, 0, SEEK_CUR);
        const uint64_t total = sin.st_size > (start < 0 ? 0 : start) ? (uint64_t)(sin.st_size - (start < 0 ? 0 : start)) : 0;
        // the reference's last read is short or empty: when the size is a multiple of 8192 that is an extra, empty block
        const uint64_t n_fill = total / slab_bytes + 1;                     // the last slab may be empty: it ends the stream
        uint64_t next_fill = 0;
        while (failure == 0) {
            bool progressed = false;
            // hand free slabs to the fill workers
            for (;;) {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (next_fill >= n_fill || acquired - released >= (uint64_t)n_slabs) break;
                }
                void *buf = nullptr;
                size_t cap = 0;
                if (dpx_stream_acquire(stream, &buf, &cap) != DPX_OK) {
                    fail_with(1, std::string("dpx_stream_acquire: ") + dpx_last_error());
                    break;
                }
                const uint64_t k = next_fill++;
                const uint64_t off = k * (uint64_t)slab_bytes;
                const size_t n = (size_t)std::min<uint64_t>(slab_bytes, total - std::min(total, off));
                SlabIo &s = io[k % io.size()];
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ++acquired;
                    s.in = static_cast<char *>(buf);
                    s.filled = n;
                    s.ready = false;
                }
                char *dst = static_cast<char *>(buf);
                const off_t foff = (off_t)((start < 0 ? 0 : start) + off);
                fillers->push([&, dst, n, foff, k] {
                    const bool ok = n == 0 || pread_all(STDIN_FILENO, dst, n, foff);
                    std::lock_guard<std::mutex> lk(mu);
                    if (!ok && failure == 0) {
                        failure = 101;                                  // main.rs:63 expect("doppler collect error")
                        failure_msg = "doppler collect error";
                    }
                    io[k % io.size()].ready = true;
                    cv.notify_all();
                });
                progressed = true;
            }
            // submit the filled slabs, in order
            bool ready;
            size_t n = 0;
            {
                std::lock_guard<std::mutex> lk(mu);
                ready = submitted < acquired && io[submitted % io.size()].ready;
                if (ready) n = io[submitted % io.size()].filled;
            }
            if (ready) {
                if (!submit_slab(n)) break;
                progressed = true;
                std::lock_guard<std::mutex> lk(mu);
                if (submitted == n_fill) break;
            }
            if (!progressed) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] {
                    return failure != 0 || (submitted < acquired && io[submitted % io.size()].ready) ||
                           (next_fill < n_fill && acquired - released < (uint64_t)n_slabs);
                });
            }
        }
        if (fillers) fillers->stop();
        (void)lseek(STDIN_FILENO, (off_t)((start < 0 ? 0 : start) + total), SEEK_SET);
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        input_done = true;
    }
    cv.notify_all();
    consumer.join();

    if (getenv("DOPPLER_STATS")) {      // steady-state rate: first read to last write, start-up excluded
        struct timeval tv_end;
        gettimeofday(&tv_end, nullptr);
        const double dt = (tv_end.tv_sec - tv_start.tv_sec) + (tv_end.tv_usec - tv_start.tv_usec) * 1e-6;
        fprintf(stderr, "doppler stats: %llu samples in %.6f s = %.1f Msamples/s (stdin -> stdout, start-up excluded; %u GPU(s), %d slabs of %zu bytes, %s in, %s out)\n",
                (unsigned long long)total_samples, dt, total_samples / dt / 1e6, n_gpus, n_slabs, slab_bytes,
                in_file ? "pread workers" : "one reader", out_map ? "mapped-file workers" : out_file ? "pwrite workers" : "one writer");
        if (pipe_in || pipe_out) fprintf(stderr, "doppler stats: pipe buffers %ld bytes in, %ld bytes out%s\n", pipe_in, pipe_out, lender.active() ? " (output lent to the pipe with vmsplice)" : "");
        dpx_stream_stats st;
        if (dpx_stream_get_stats(stream, &st) == DPX_OK && st.slabs)
            fprintf(stderr, "doppler stats: dpx_stream_submit %.1f us per slab over %llu slabs (plan %.1f, device image %.1f, enqueue %.1f; %llu plans reused)\n",
                    st.total_us / st.slabs, (unsigned long long)st.slabs, st.plan_us / st.slabs, st.upload_us / st.slabs,
                    st.enqueue_us / st.slabs, (unsigned long long)st.plans_reused);
    }
    (void)obs;
    dpx_stream_destroy(stream);
    destroy_ctxs();
    if (failure != 0) {
        if (failure == 101) fprintf(stderr, "thread 'main' panicked at '%s'\n", failure_msg.c_str());
        else fprintf(stderr, "doppler: %s\n", failure_msg.c_str());
        return failure;
    }
    if (ragged) {
        fprintf(stderr, "thread 'main' panicked at 'assertion failed: inbuf.len() %% %zu == 0'\n", ibs);
        return 101;
    }
    return 0;
}
