/*
 * flake_amd_cli.c -- minimal command-line encoder on top of the host layer:
 * the block loop of the reference CLI (flake/flake.c:495-689: read a block of
 * samples, flake_encode_frame(), write the frame, rewrite STREAMINFO at the
 * end), with the frames of many blocks encoded per GPU batch.
 *
 *   flake_amd_cli [-0..-12] [-b blocksize] in.wav out.flac
 *   flake_amd_cli [-0..-12] --synth FRAMES [--channels C] [--bps B] out.flac
 *
 * Only canonical PCM WAV (8/16/24/32 bit) is read; this is a harness for the
 * host API, not a replacement for the reference's libpcm_io.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "flake_amd.h"

static uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

typedef struct { FILE *f; int channels, rate, bps; uint32_t frames; } wav_t;

static int wav_open(wav_t *w, const char *path)
{
    uint8_t h[12], ck[8];
    w->f = fopen(path, "rb");
    if (!w->f || fread(h, 1, 12, w->f) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) return -1;
    int have_fmt = 0;
    while (fread(ck, 1, 8, w->f) == 8) {
        uint32_t sz = rd32(ck + 4);
        if (!memcmp(ck, "fmt ", 4)) {
            uint8_t f[40];
            uint32_t take = sz < 40 ? sz : 40;
            if (fread(f, 1, take, w->f) != take) return -1;
            if (sz > take) fseek(w->f, (long)(sz - take), SEEK_CUR);
            int tag = rd16(f);
            if (tag != 1 && tag != 0xFFFE) return -1;
            w->channels = rd16(f + 2); w->rate = (int)rd32(f + 4); w->bps = rd16(f + 14);
            have_fmt = 1;
        } else if (!memcmp(ck, "data", 4)) {
            if (!have_fmt) return -1;
            w->frames = sz / (uint32_t)(w->channels * ((w->bps + 7) / 8));
            return 0;
        } else {
            fseek(w->f, (long)(sz + (sz & 1)), SEEK_CUR);
        }
    }
    return -1;
}

/* up to `frames` sample-frames as interleaved int32, sign-extended (pcm_io.c:155-277) */
static uint32_t wav_read(wav_t *w, int32_t *dst, uint32_t frames)
{
    const int bytes = (w->bps + 7) / 8;
    const size_t want = (size_t)frames * w->channels;
    uint8_t *raw = (uint8_t *)malloc(want * bytes);
    size_t got = fread(raw, (size_t)bytes, want, w->f);
    for (size_t i = 0; i < got; i++) {
        const uint8_t *p = raw + i * bytes;
        int32_t v;
        if (bytes == 1) v = (int32_t)p[0] - 128;
        else if (bytes == 2) v = (int16_t)rd16(p);
        else if (bytes == 3) v = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8;
        else v = (int32_t)rd32(p);
        dst[i] = v;
    }
    free(raw);
    return (uint32_t)(got / w->channels);
}

int main(int argc, char **argv)
{
    FlakeAmdContext s;
    memset(&s, 0, sizeof s);
    int level = 5, bsize = -1, synth = 0, channels = 2, bps = 16;
    const char *in = NULL, *out = NULL;
    for (int i = 1; i < argc; i++) {
        if (argv[i][0] == '-' && argv[i][1] >= '0' && argv[i][1] <= '9') level = atoi(argv[i] + 1);
        else if (!strcmp(argv[i], "-b") && i + 1 < argc) bsize = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--synth") && i + 1 < argc) synth = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--channels") && i + 1 < argc) channels = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--bps") && i + 1 < argc) bps = atoi(argv[++i]);
        else if (!in && !synth) in = argv[i];
        else out = argv[i];
    }
    if (!out || (!in && !synth)) {
        fprintf(stderr, "usage: %s [-0..-12] [-b blocksize] (in.wav | --synth FRAMES [--channels C] [--bps B]) out.flac\n", argv[0]);
        return 2;
    }
    wav_t w;
    memset(&w, 0, sizeof w);
    if (in) {
        if (wav_open(&w, in)) { fprintf(stderr, "cannot read %s as PCM WAV\n", in); return 1; }
        s.channels = w.channels; s.sample_rate = w.rate; s.bits_per_sample = w.bps; s.samples = w.frames;
    } else {
        s.channels = channels; s.sample_rate = 44100; s.bits_per_sample = bps;
    }
    s.params.compression = level;
    if (flake_amd_set_defaults(&s.params)) return 1;                 /* flake.c:528 */
    if (bsize > 0) s.params.block_size = bsize;
    if (synth) s.samples = (unsigned)synth * (unsigned)s.params.block_size;
    if (flake_amd_validate_params(&s) < 0) { fprintf(stderr, "invalid parameters\n"); return 1; }
    int hlen = flake_amd_encode_init(&s);                            /* flake.c:558 */
    if (hlen < 0) { fprintf(stderr, "encoder init failed (%d)\n", hlen); return 1; }
    FILE *fo = fopen(out, "wb");
    if (!fo) { perror(out); return 1; }
    fwrite(s.header, 1, (size_t)hlen, fo);

    const int bs = s.params.block_size, nch = s.channels, batch = 256;
    int32_t *pcm = (int32_t *)malloc(sizeof(int32_t) * (size_t)batch * bs * nch);
    const size_t cap = (size_t)batch * bs * nch * 5 + 65536;
    uint8_t *buf = (uint8_t *)malloc(cap);
    uint64_t total_in = 0, total_out = (uint64_t)hlen;
    int64_t synth_left = synth, synth_pos = 0;
    for (;;) {
        uint32_t frames;
        if (synth) {
            int nb = synth_left < batch ? (int)synth_left : batch;
            if (nb <= 0) break;
            flake_amd_synth_pcm(pcm, synth_pos, nb, bs, nch, s.bits_per_sample);
            synth_pos += nb; synth_left -= nb;
            frames = (uint32_t)nb * (uint32_t)bs;
        } else {
            frames = wav_read(&w, pcm, (uint32_t)batch * (uint32_t)bs);
            if (!frames) break;
        }
        const int nblocks = (int)(frames / (uint32_t)bs), tail = (int)(frames % (uint32_t)bs);
        long long n = flake_amd_encode_frames(&s, pcm, nblocks, bs, tail, buf, cap, NULL);   /* flake.c:633 */
        if (n < 0) { fprintf(stderr, "encode error: %s\n", flake_amd_last_error(&s)); return 1; }
        fwrite(buf, 1, (size_t)n, fo);
        total_in += frames; total_out += (uint64_t)n;
        if (tail) break;
    }
    /* rewrite STREAMINFO with the final MD5 / max frame size (flake.c:668-679) */
    FlakeAmdStreaminfo si;
    uint8_t sib[34];
    if (!s.samples) s.samples = (unsigned)total_in;
    if (!flake_amd_get_streaminfo(&s, &si)) {
        si.samples = (unsigned)total_in;
        flake_amd_write_streaminfo(&si, sib);
        fseek(fo, 8, SEEK_SET);
        fwrite(sib, 1, 34, fo);
    }
    fclose(fo);
    fprintf(stderr, "%llu sample-frames -> %llu bytes (ratio %.3f)\n", (unsigned long long)total_in,
            (unsigned long long)total_out,
            total_in ? (double)total_out / ((double)total_in * nch * ((s.bits_per_sample + 7) / 8)) : 0.0);
    flake_amd_encode_close(&s);
    free(pcm); free(buf);
    return 0;
}
