/*
 * ref_bench.c -- TIMING ONLY.  The reference's own flake_encode_frame() loop (the caller's loop of
 * flake/flake.c:624-663) over an in-memory synthetic batch, linked against the reference's
 * libflake_static.a as the reference's CMake builds it (build/ref/, git-ignored; oracle/Makefile
 * target `refbench`).  bench.py's cpu_baseline leg runs this binary as a child process and reports
 * its rate as kind "reference (timing only)".  It is NOT a parity witness: no test loads it, nothing
 * compares its bytes with anything, and no product path knows it exists.
 *
 * The PCM is the repository's generator (flake_amd/host/synth.c, compiled in), so the reference
 * encodes the very frames the GPU leg encodes.
 *
 * usage: ref_bench channels bps rate block_size order_method stereo_method prediction_type min_order
 *                  max_order min_porder max_porder variable_block_size allow_vbs frames seconds transient
 *        (every FlakeEncodeParams field the hot path reads, flake.h:59-160, as the caller sets them)
 * prints one JSON line: frames encoded, samples, seconds, bytes written.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "flake.h"       /* the reference's: -I/root/reference/libflake */

void flake_amd_synth_pcm(int32_t *pcm, int64_t first_frame, int nframes, int n, int channels, int bps);

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv)
{
    if (argc < 17) {
        fprintf(stderr, "usage: ref_bench ch bps rate block order_method stereo_method prediction_type min_order max_order "
                        "min_porder max_porder vbs allow_vbs frames seconds transient\n");
        return 2;
    }
    const int ch = atoi(argv[1]), bps = atoi(argv[2]), rate = atoi(argv[3]);
    const int frames = atoi(argv[14]);
    const double seconds = atof(argv[15]);
    const int transient = atoi(argv[16]);

    FlakeContext s;
    memset(&s, 0, sizeof(s));
    s.channels = ch;
    s.sample_rate = rate;
    s.bits_per_sample = bps;
    s.samples = 0;
    s.params.compression = 5;
    if (flake_set_defaults(&s.params) < 0) { fprintf(stderr, "flake_set_defaults failed\n"); return 1; }
    s.params.block_size = atoi(argv[4]);
    s.params.order_method = atoi(argv[5]);
    s.params.stereo_method = atoi(argv[6]);
    s.params.prediction_type = atoi(argv[7]);
    s.params.min_prediction_order = atoi(argv[8]);
    s.params.max_prediction_order = atoi(argv[9]);
    s.params.min_partition_order = atoi(argv[10]);
    s.params.max_partition_order = atoi(argv[11]);
    s.params.variable_block_size = atoi(argv[12]);
    s.params.allow_vbs = atoi(argv[13]);
    s.params.padding_size = 0;
    if (flake_validate_params(&s) < 0) { fprintf(stderr, "flake_validate_params failed\n"); return 1; }
    if (flake_encode_init(&s) < 0) { fprintf(stderr, "flake_encode_init failed\n"); return 1; }
    const int n = s.params.block_size;

    int32_t *pcm = malloc((size_t)frames * n * ch * sizeof(int32_t));
    if (!pcm) return 1;
    flake_amd_synth_pcm(pcm, 0, frames, n, ch, bps);
    if (transient) {
        /* bench.py's VBS rows: every third block drops by 24 dB half way (something to split) */
        for (int f = 0; f < frames; f += 3)
            for (size_t i = (size_t)(n / 2) * ch; i < (size_t)n * ch; i++)
                pcm[(size_t)f * n * ch + i] >>= 4;
    }

    long long done = 0, bytes = 0;
    const double t0 = now();
    double dt = 0;
    do {
        for (int f = 0; f < frames; f++) {
            const int w = flake_encode_frame(&s, pcm + (size_t)f * n * ch, n);
            if (w < 0) { fprintf(stderr, "flake_encode_frame failed at frame %d\n", f); return 1; }
            bytes += w;
        }
        done += frames;
        dt = now() - t0;
    } while (dt < seconds);
    flake_encode_close(&s);
    printf("{\"frames\": %lld, \"block_size\": %d, \"channels\": %d, \"samples\": %lld, \"seconds\": %.6f, \"bytes\": %lld}\n",
           done, n, ch, done * (long long)n * ch, dt, bytes);
    free(pcm);
    return 0;
}
