/* pipe_sink — consumes stdin as fast as a pipe gives it and prints the byte count (measurement tool).
 * pipe_sink [splice]: default read() into a 16 MiB buffer; "splice": splice() to /dev/null, no copy at all. */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

int main(int argc, char **argv)
{
    const int use_splice = argc > 1 && !strcmp(argv[1], "splice");
    (void)fcntl(STDIN_FILENO, F_SETPIPE_SZ, 16 << 20);
    if (fcntl(STDIN_FILENO, F_GETPIPE_SZ) < (1 << 20)) (void)fcntl(STDIN_FILENO, F_SETPIPE_SZ, 1 << 20);
    uint64_t total = 0;
    if (use_splice) {
        const int nul = open("/dev/null", O_WRONLY);
        for (;;) {
            const ssize_t r = splice(STDIN_FILENO, NULL, nul, NULL, 16 << 20, SPLICE_F_MOVE);
            if (r <= 0) break;
            total += (uint64_t)r;
        }
    } else {
        const size_t cap = 16u << 20;
        char *buf = malloc(cap);
        for (;;) {
            const ssize_t r = read(STDIN_FILENO, buf, cap);
            if (r <= 0) break;
            total += (uint64_t)r;
        }
    }
    fprintf(stderr, "pipe_sink: %llu bytes\n", (unsigned long long)total);
    return 0;
}
