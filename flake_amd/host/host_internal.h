/* host_internal.h -- shared by the host C sources; not installed. */
#ifndef FLAKE_AMD_HOST_INTERNAL_H
#define FLAKE_AMD_HOST_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

typedef struct fa_md5 {
    uint32_t h[4];
    uint64_t len;
    uint8_t buf[64];
    int fill;
} fa_md5;

void fa_md5_init(fa_md5 *m);
void fa_md5_update(fa_md5 *m, const uint8_t *data, size_t n);
void fa_md5_final(const fa_md5 *m, uint8_t out[16]);
void fa_md5_pcm(fa_md5 *m, const int32_t *pcm, size_t nvalues, int bps);

#endif
