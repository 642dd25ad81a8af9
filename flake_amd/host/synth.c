/*
 * synth.c -- deterministic synthetic PCM for benchmarks and tests
 * (SURVEY.md 8d).  Integer-only so that every consumer (C, Python through
 * ctypes, the CLI) sees the same samples:
 *
 *   per (frame f, channel c): PCG32 stream, seed 0xF1A4E000 + 8f + c
 *   noise   e[t]  = triangular (sum of two 16-bit draws) scaled to 2^(bps-6)
 *   even c: x[t]  = ((a1*x[t-1] - a2*x[t-2]) >> 14) + e[t]   two-pole resonator,
 *                   pole pair (f + c/2) mod 8, clamped to the sample range
 *   odd  c: built from y, the channel before it, by one of four pairings chosen
 *           by (f >> 3) & 3, so that every stereo decorrelation mode occurs:
 *             0: y - (y >> 3) + e'/4      1: an independent resonator
 *             2: (y >> 1) + e'            3: y + e'/16
 *
 * The generator restarts every frame, so frames are independent and any shard
 * of a batch can be produced without the frames before it.
 */
#include <stdint.h>
#include <stdlib.h>

#include "flake_amd.h"

typedef struct { uint64_t state, inc; } pcg32_t;

static uint32_t pcg32_next(pcg32_t *g)
{
    uint64_t old = g->state;
    g->state = old * 6364136223846793005ULL + g->inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
}

static void pcg32_seed(pcg32_t *g, uint64_t seed, uint64_t seq)
{
    g->state = 0;
    g->inc = (seq << 1u) | 1u;
    pcg32_next(g);
    g->state += seed;
    pcg32_next(g);
}

static const int32_t pole_a1[8] = {30000, 28000, 25000, 21000, 16000, 10000, 3000, -8000};
static const int32_t pole_a2[8] = {16000, 16050, 16100, 16150, 16200, 16250, 16280, 16300};

static int32_t tri_noise(pcg32_t *g, int amp_shift)
{
    int32_t a = (int32_t)(pcg32_next(g) >> 16), b = (int32_t)(pcg32_next(g) >> 16);
    int64_t t = (int64_t)(a + b - 65535);          /* [-65535, 65535] */
    if (amp_shift >= 0) return (int32_t)((t << amp_shift) >> 16);
    return (int32_t)(t >> (16 - amp_shift));
}

static int32_t clampi(int64_t v, int32_t lim)
{
    if (v > lim) return lim;
    if (v < -lim) return -lim;
    return (int32_t)v;
}

FLAKE_AMD_API void
flake_amd_synth_pcm(int32_t *pcm, int64_t first_frame, int nframes, int n, int channels, int bps)
{
    const int32_t lim = (int32_t)((1u << (bps - 1)) - 1u);
    const int amp_shift = bps - 6;
    for (int f = 0; f < nframes; f++) {
        const int64_t fr = first_frame + f;
        int32_t *dst = pcm + (size_t)f * n * channels;
        for (int c = 0; c < channels; c++) {
            pcg32_t g;
            const uint64_t seq = (uint64_t)(8 * fr + c);
            pcg32_seed(&g, 0xF1A4E000ULL + seq, seq);
            const int pairing = (int)((fr >> 3) & 3);
            if ((c & 1) == 0 || pairing == 1) {
                const int pi = (int)((fr + c / 2 + 3 * (c & 1)) & 7);
                const int64_t a1 = pole_a1[pi], a2 = pole_a2[pi];
                int64_t x1 = 0, x2 = 0;
                for (int t = 0; t < n; t++) {
                    int64_t x = ((a1 * x1 - a2 * x2) >> 14) + tri_noise(&g, amp_shift);
                    int32_t v = clampi(x, lim);
                    dst[(size_t)t * channels + c] = v;
                    x2 = x1;
                    x1 = v;
                }
            } else {
                for (int t = 0; t < n; t++) {
                    int64_t y = dst[(size_t)t * channels + c - 1];
                    int64_t x;
                    if (pairing == 0) x = y - (y >> 3) + tri_noise(&g, amp_shift - 2);
                    else if (pairing == 2) x = (y >> 1) + tri_noise(&g, amp_shift);
                    else x = y + tri_noise(&g, amp_shift - 4);
                    dst[(size_t)t * channels + c] = clampi(x, lim);
                }
            }
        }
    }
}
