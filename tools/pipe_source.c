/* pipe_source — writes N bytes of pseudo-random i16 IQ to stdout as fast as a pipe takes them (measurement tool:
 * the figure of `... | doppler | ...` should be doppler's, not cat's).   pipe_source BYTES [vmsplice]
 * Default: plain write() of a 64 MiB buffer, over and over.  With "vmsplice": the buffer's pages are mapped into the pipe
 * without a copy (the buffer is never modified, so that is safe here). */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/uio.h>
#include <unistd.h>

int main(int argc, char **argv)
{
    const uint64_t total = argc > 1 ? strtoull(argv[1], NULL, 0) : (1ull << 30);
    const int splice = argc > 2 && !strcmp(argv[2], "vmsplice");
    const size_t cap = 64u << 20;
    int16_t *buf = NULL;
    if (posix_memalign((void **)&buf, 4096, cap)) return 2;
    uint32_t x = 12345;
    for (size_t i = 0; i < cap / 2; ++i) {
        x = x * 1664525u + 1013904223u;
        buf[i] = (int16_t)((int32_t)(x >> 16) % 23171);
    }
    (void)fcntl(STDOUT_FILENO, F_SETPIPE_SZ, 16 << 20);
    if (fcntl(STDOUT_FILENO, F_GETPIPE_SZ) < (1 << 20)) (void)fcntl(STDOUT_FILENO, F_SETPIPE_SZ, 1 << 20);
    uint64_t left = total;
    while (left) {
        size_t n = left < cap ? (size_t)left : cap;
        const char *p = (const char *)buf;
        while (n) {
            ssize_t w;
            if (splice) {
                struct iovec iov = {(void *)p, n};
                w = vmsplice(STDOUT_FILENO, &iov, 1, 0);
            } else {
                w = write(STDOUT_FILENO, p, n);
            }
            if (w <= 0) return 1;
            p += w;
            n -= (size_t)w;
            left -= (uint64_t)w;
        }
    }
    return 0;
}
