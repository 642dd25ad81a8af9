/*
 * dropin_client.c -- a libflake CLIENT, compiled against the REFERENCE's own
 * public header (/root/reference/libflake/flake.h) and linked against our
 * libflake.so.  TEST INFRASTRUCTURE: it shows the link-level drop-in -- the
 * call sequence is the one util/api_example.c and flake/flake.c:528-663 use
 * (set_defaults, validate, encode_init, encode_frame per block, get_buffer,
 * streaminfo rewrite, close).  Built by oracle/Makefile into oracle/_ref/
 * (only where /root/reference exists; the binary travels to the GPU box).
 *
 *   dropin_client LEVEL NBLOCKS out.flac   (stereo 16-bit synthetic input)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "flake.h"                      /* the reference's header */

/* provided by libflake.so as well (include/flake_amd.h) */
extern void flake_amd_synth_pcm(int *pcm, long long first_frame, int nframes, int n, int channels, int bps);

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    FlakeContext s;
    memset(&s, 0, sizeof s);
    s.channels = 2; s.sample_rate = 44100; s.bits_per_sample = 16;
    s.params.compression = atoi(argv[1]);
    const int nblocks = atoi(argv[2]);
    if (flake_set_defaults(&s.params)) return 1;
    s.samples = (unsigned)nblocks * (unsigned)s.params.block_size;
    if (flake_validate_params(&s) < 0) return 1;
    const int hlen = flake_encode_init(&s);
    if (hlen < 0) return 1;
    FILE *fo = fopen(argv[3], "wb");
    if (!fo) return 1;
    fwrite(s.header, 1, (size_t)hlen, fo);
    const int bs = s.params.block_size;
    int *pcm = (int *)malloc(sizeof(int) * (size_t)bs * 2);
    unsigned char *frame = (unsigned char *)flake_get_buffer(&s);
    for (int b = 0; b < nblocks; b++) {
        flake_amd_synth_pcm(pcm, b, 1, bs, 2, 16);
        const int fs = flake_encode_frame(&s, pcm, bs);
        if (fs < 0) { fprintf(stderr, "Error encoding frame\n"); return 1; }
        fwrite(frame, 1, (size_t)fs, fo);
    }
    FlakeStreaminfo si;
    if (!flake_get_streaminfo(&s, &si)) {
        unsigned char d[34];
        flake_write_streaminfo(&si, d);
        fseek(fo, 8, SEEK_SET);
        fwrite(d, 1, 34, fo);
    }
    fclose(fo);
    flake_encode_close(&s);
    free(pcm);
    printf("%s %d blocks ok\n", flake_get_version(), nblocks);
    return 0;
}
