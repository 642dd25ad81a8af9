/*
 * md5.c -- MD5 (RFC 1321) of the raw little-endian PCM stream, the STREAMINFO
 * checksum libflake keeps on the host (md5.c:281-320 md5_accumulate packs
 * (bps+7)/8 bytes per sample, low byte first).  Own implementation from the
 * RFC; sequential over the whole stream, so it stays on the CPU.
 */
#include <stdint.h>
#include <string.h>

#include "host_internal.h"

/* one 64-byte block; the 64 steps written out (RFC 1321 section 3.4) */
#define ROL(x, s) (((x) << (s)) | ((x) >> (32 - (s))))
#define F1(b, c, d) ((d) ^ ((b) & ((c) ^ (d))))
#define F2(b, c, d) ((c) ^ ((d) & ((b) ^ (c))))
#define F3(b, c, d) ((b) ^ (c) ^ (d))
#define F4(b, c, d) ((c) ^ ((b) | ~(d)))
#define STEP(f, a, b, c, d, w, k, s) do { (a) += f((b), (c), (d)) + (w) + (k); (a) = ROL((a), (s)) + (b); } while (0)

static void md5_block(fa_md5 *m, const uint8_t *p)
{
    uint32_t w[16], a = m->h[0], b = m->h[1], c = m->h[2], d = m->h[3];
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
    memcpy(w, p, 64);
#else
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) |
               ((uint32_t)p[4 * i + 3] << 24);
#endif
    STEP(F1, a, b, c, d, w[0], 0xd76aa478, 7);   STEP(F1, d, a, b, c, w[1], 0xe8c7b756, 12);
    STEP(F1, c, d, a, b, w[2], 0x242070db, 17);  STEP(F1, b, c, d, a, w[3], 0xc1bdceee, 22);
    STEP(F1, a, b, c, d, w[4], 0xf57c0faf, 7);   STEP(F1, d, a, b, c, w[5], 0x4787c62a, 12);
    STEP(F1, c, d, a, b, w[6], 0xa8304613, 17);  STEP(F1, b, c, d, a, w[7], 0xfd469501, 22);
    STEP(F1, a, b, c, d, w[8], 0x698098d8, 7);   STEP(F1, d, a, b, c, w[9], 0x8b44f7af, 12);
    STEP(F1, c, d, a, b, w[10], 0xffff5bb1, 17); STEP(F1, b, c, d, a, w[11], 0x895cd7be, 22);
    STEP(F1, a, b, c, d, w[12], 0x6b901122, 7);  STEP(F1, d, a, b, c, w[13], 0xfd987193, 12);
    STEP(F1, c, d, a, b, w[14], 0xa679438e, 17); STEP(F1, b, c, d, a, w[15], 0x49b40821, 22);

    STEP(F2, a, b, c, d, w[1], 0xf61e2562, 5);   STEP(F2, d, a, b, c, w[6], 0xc040b340, 9);
    STEP(F2, c, d, a, b, w[11], 0x265e5a51, 14); STEP(F2, b, c, d, a, w[0], 0xe9b6c7aa, 20);
    STEP(F2, a, b, c, d, w[5], 0xd62f105d, 5);   STEP(F2, d, a, b, c, w[10], 0x02441453, 9);
    STEP(F2, c, d, a, b, w[15], 0xd8a1e681, 14); STEP(F2, b, c, d, a, w[4], 0xe7d3fbc8, 20);
    STEP(F2, a, b, c, d, w[9], 0x21e1cde6, 5);   STEP(F2, d, a, b, c, w[14], 0xc33707d6, 9);
    STEP(F2, c, d, a, b, w[3], 0xf4d50d87, 14);  STEP(F2, b, c, d, a, w[8], 0x455a14ed, 20);
    STEP(F2, a, b, c, d, w[13], 0xa9e3e905, 5);  STEP(F2, d, a, b, c, w[2], 0xfcefa3f8, 9);
    STEP(F2, c, d, a, b, w[7], 0x676f02d9, 14);  STEP(F2, b, c, d, a, w[12], 0x8d2a4c8a, 20);

    STEP(F3, a, b, c, d, w[5], 0xfffa3942, 4);   STEP(F3, d, a, b, c, w[8], 0x8771f681, 11);
    STEP(F3, c, d, a, b, w[11], 0x6d9d6122, 16); STEP(F3, b, c, d, a, w[14], 0xfde5380c, 23);
    STEP(F3, a, b, c, d, w[1], 0xa4beea44, 4);   STEP(F3, d, a, b, c, w[4], 0x4bdecfa9, 11);
    STEP(F3, c, d, a, b, w[7], 0xf6bb4b60, 16);  STEP(F3, b, c, d, a, w[10], 0xbebfbc70, 23);
    STEP(F3, a, b, c, d, w[13], 0x289b7ec6, 4);  STEP(F3, d, a, b, c, w[0], 0xeaa127fa, 11);
    STEP(F3, c, d, a, b, w[3], 0xd4ef3085, 16);  STEP(F3, b, c, d, a, w[6], 0x04881d05, 23);
    STEP(F3, a, b, c, d, w[9], 0xd9d4d039, 4);   STEP(F3, d, a, b, c, w[12], 0xe6db99e5, 11);
    STEP(F3, c, d, a, b, w[15], 0x1fa27cf8, 16); STEP(F3, b, c, d, a, w[2], 0xc4ac5665, 23);

    STEP(F4, a, b, c, d, w[0], 0xf4292244, 6);   STEP(F4, d, a, b, c, w[7], 0x432aff97, 10);
    STEP(F4, c, d, a, b, w[14], 0xab9423a7, 15); STEP(F4, b, c, d, a, w[5], 0xfc93a039, 21);
    STEP(F4, a, b, c, d, w[12], 0x655b59c3, 6);  STEP(F4, d, a, b, c, w[3], 0x8f0ccc92, 10);
    STEP(F4, c, d, a, b, w[10], 0xffeff47d, 15); STEP(F4, b, c, d, a, w[1], 0x85845dd1, 21);
    STEP(F4, a, b, c, d, w[8], 0x6fa87e4f, 6);   STEP(F4, d, a, b, c, w[15], 0xfe2ce6e0, 10);
    STEP(F4, c, d, a, b, w[6], 0xa3014314, 15);  STEP(F4, b, c, d, a, w[13], 0x4e0811a1, 21);
    STEP(F4, a, b, c, d, w[4], 0xf7537e82, 6);   STEP(F4, d, a, b, c, w[11], 0xbd3af235, 10);
    STEP(F4, c, d, a, b, w[2], 0x2ad7d2bb, 15);  STEP(F4, b, c, d, a, w[9], 0xeb86d391, 21);
    m->h[0] += a; m->h[1] += b; m->h[2] += c; m->h[3] += d;
}

void fa_md5_init(fa_md5 *m)
{
    m->h[0] = 0x67452301; m->h[1] = 0xefcdab89; m->h[2] = 0x98badcfe; m->h[3] = 0x10325476;
    m->len = 0;
    m->fill = 0;
}

void fa_md5_update(fa_md5 *m, const uint8_t *data, size_t n)
{
    m->len += n;
    if (m->fill) {
        size_t take = 64 - m->fill;
        if (take > n) take = n;
        memcpy(m->buf + m->fill, data, take);
        m->fill += (int)take; data += take; n -= take;
        if (m->fill == 64) { md5_block(m, m->buf); m->fill = 0; }
    }
    while (n >= 64) { md5_block(m, data); data += 64; n -= 64; }
    if (n) { memcpy(m->buf, data, n); m->fill = (int)n; }
}

void fa_md5_final(const fa_md5 *m0, uint8_t out[16])
{
    fa_md5 m = *m0;                       /* the running state stays usable (metadata.c:61-62) */
    uint64_t bits = m.len * 8;
    uint8_t pad[72] = {0x80};
    size_t padlen = (m.fill < 56) ? (size_t)(56 - m.fill) : (size_t)(120 - m.fill);
    for (int i = 0; i < 8; i++) pad[padlen + i] = (uint8_t)(bits >> (8 * i));
    fa_md5_update(&m, pad, padlen + 8);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(m.h[i] >> (8 * j));
}

/* md5.c:281-320: (bps+7)/8 bytes per sample, little-endian, interleaved order */
void fa_md5_pcm(fa_md5 *m, const int32_t *pcm, size_t nvalues, int bps)
{
    uint8_t tmp[16384];
    const int bpsamp = (bps + 7) >> 3;
    size_t k = 0;
    if (bpsamp == 2) {
        /* the common case, packed 2048 samples at a time */
        for (size_t i = 0; i < nvalues;) {
            const size_t cnt = (nvalues - i < 8192) ? nvalues - i : 8192;
            for (size_t j = 0; j < cnt; j++) {
                const uint32_t x = (uint32_t)pcm[i + j];
                tmp[2 * j] = (uint8_t)x;
                tmp[2 * j + 1] = (uint8_t)(x >> 8);
            }
            fa_md5_update(m, tmp, 2 * cnt);
            i += cnt;
        }
        return;
    }
    for (size_t i = 0; i < nvalues; i++) {
        uint32_t x = (uint32_t)pcm[i];
        for (int b = 0; b < bpsamp; b++) { tmp[k++] = (uint8_t)x; x >>= 8; }
        if (k > sizeof(tmp) - 4) { fa_md5_update(m, tmp, k); k = 0; }
    }
    if (k) fa_md5_update(m, tmp, k);
}
