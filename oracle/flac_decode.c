/*
 * flac_decode.c -- minimal, independent FLAC frame decoder.
 *
 * TEST INFRASTRUCTURE ONLY.  Written from the FLAC format (frame header,
 * subframe types, Rice / Rice2 residual coding, stereo decorrelation, CRC-8,
 * CRC-16), not from any encoder: it is the check that what the host layer and
 * the oracle emit is a valid lossless stream (the reference's only check is
 * the external `flac -t`, util/flake-test.sh:10, which this image lacks).
 *
 *   long fd_decode_frames(const uint8_t *data, size_t len, int channels, int bps,
 *                         int32_t *pcm_out, size_t pcm_cap_frames,
 *                         int *nframes_out, int *block_sizes, int max_frames)
 * decodes back-to-back frames; returns decoded sample-frames, or a negative
 * error code: -1 sync, -2 CRC-8, -3 CRC-16, -4 reserved/unsupported, -5 overrun.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const uint8_t *d; size_t len; size_t pos; int err; } br_t;   /* pos in bits */

static uint32_t rd(br_t *b, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) {
        size_t byte = b->pos >> 3;
        if (byte >= b->len) { b->err = 1; return 0; }
        v = (v << 1) | ((b->d[byte] >> (7 - (b->pos & 7))) & 1u);
        b->pos++;
    }
    return v;
}

static int32_t rds(br_t *b, int n)
{
    if (n == 0) return 0;
    uint32_t v = rd(b, n);
    if (n < 32 && (v & (1u << (n - 1)))) v |= ~((1u << n) - 1u);
    return (int32_t)v;
}

static uint32_t unary(br_t *b)
{
    uint32_t q = 0;
    while (!b->err && rd(b, 1) == 0) q++;
    return q;
}

static uint8_t crc8(const uint8_t *d, size_t n)
{
    uint8_t c = 0;
    for (size_t i = 0; i < n; i++) {
        c ^= d[i];
        for (int k = 0; k < 8; k++) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1);
    }
    return c;
}

static uint16_t crc16(const uint8_t *d, size_t n)
{
    uint16_t c = 0;
    for (size_t i = 0; i < n; i++) {
        c ^= (uint16_t)(d[i] << 8);
        for (int k = 0; k < 8; k++) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1);
    }
    return c;
}

static int residual(br_t *b, int32_t *res, int n, int order)
{
    int method = (int)rd(b, 2);
    if (method > 1) return -4;
    int porder = (int)rd(b, 4);
    int pbits = method ? 5 : 4;
    int psz = n >> porder;
    int i = order;
    for (int p = 0; p < (1 << porder); p++) {
        int k = (int)rd(b, pbits);
        int cnt = psz - (p == 0 ? order : 0);
        if (k == (method ? 31 : 15)) {                 /* escape: raw bits */
            int raw = (int)rd(b, 5);
            for (int j = 0; j < cnt; j++) res[i++] = rds(b, raw);
        } else {
            for (int j = 0; j < cnt; j++) {
                uint32_t q = unary(b);
                uint32_t u = (q << k) | (k ? rd(b, k) : 0);
                res[i++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
            }
        }
        if (b->err) return -5;
    }
    return 0;
}

static int subframe(br_t *b, int32_t *out, int n, int bps)
{
    if (rd(b, 1)) return -4;
    int type = (int)rd(b, 6);
    int wasted = 0;
    if (rd(b, 1)) { wasted = 1 + (int)unary(b); }
    bps -= wasted;
    if (type == 0) {
        int32_t v = rds(b, bps);
        for (int i = 0; i < n; i++) out[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < n; i++) out[i] = rds(b, bps);
    } else if (type >= 8 && type <= 12) {
        int order = type - 8;
        for (int i = 0; i < order; i++) out[i] = rds(b, bps);
        int rc = residual(b, out, n, order);
        if (rc) return rc;
        for (int i = order; i < n; i++) {
            int64_t p = 0;
            if (order == 1) p = out[i - 1];
            else if (order == 2) p = 2LL * out[i - 1] - out[i - 2];
            else if (order == 3) p = 3LL * out[i - 1] - 3LL * out[i - 2] + out[i - 3];
            else if (order == 4) p = 4LL * out[i - 1] - 6LL * out[i - 2] + 4LL * out[i - 3] - out[i - 4];
            out[i] = (int32_t)(out[i] + p);
        }
    } else if (type >= 32) {
        int order = type - 31;
        int32_t coef[32];
        for (int i = 0; i < order; i++) out[i] = rds(b, bps);
        int prec = (int)rd(b, 4) + 1;
        int shift = rds(b, 5);
        if (prec == 16 || shift < 0) return -4;
        for (int i = 0; i < order; i++) coef[i] = rds(b, prec);
        int rc = residual(b, out, n, order);
        if (rc) return rc;
        for (int i = order; i < n; i++) {
            int64_t p = 0;
            for (int j = 0; j < order; j++) p += (int64_t)coef[j] * out[i - 1 - j];
            out[i] = (int32_t)(out[i] + (p >> shift));
        }
    } else {
        return -4;
    }
    if (wasted) for (int i = 0; i < n; i++) out[i] = (int32_t)((uint32_t)out[i] << wasted);
    return b->err ? -5 : 0;
}

long fd_decode_frames(const uint8_t *data, size_t len, int channels, int bps,
                      int32_t *pcm_out, size_t pcm_cap_frames,
                      int *nframes_out, int *block_sizes, int max_frames)
{
    static const int bs_tab[16] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048,
                                   4096, 8192, 16384, 32768};
    size_t off = 0, done = 0;
    int nf = 0;
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * 65536 * 2);
    while (off + 6 <= len) {
        br_t b = {data + off, len - off, 0, 0};
        if (rd(&b, 14) != 0x3FFE) { free(tmp); return -1; }
        rd(&b, 1);
        rd(&b, 1);                                   /* blocking strategy */
        int bs_code = (int)rd(&b, 4), sr_code = (int)rd(&b, 4), ch_code = (int)rd(&b, 4);
        int bps_code = (int)rd(&b, 3);
        (void)bps_code;
        if (rd(&b, 1)) { free(tmp); return -4; }
        /* UTF-8 coded number */
        uint32_t first = rd(&b, 8);
        int extra = 0;
        if (first >= 0x80) {
            int ones = 0;
            while (ones < 8 && (first & (0x80u >> ones))) ones++;
            extra = ones - 1;
        }
        for (int i = 0; i < extra; i++) rd(&b, 8);
        int n = bs_tab[bs_code];
        if (bs_code == 6) n = (int)rd(&b, 8) + 1;
        else if (bs_code == 7) n = (int)rd(&b, 16) + 1;
        if (sr_code == 12) rd(&b, 8);
        else if (sr_code == 13 || sr_code == 14) rd(&b, 16);
        size_t hdr_bytes = b.pos >> 3;
        uint8_t c8 = (uint8_t)rd(&b, 8);
        if (b.err || n <= 0) { free(tmp); return -5; }
        if (crc8(data + off, hdr_bytes) != c8) { free(tmp); return -2; }
        int nch = (ch_code < 8) ? ch_code + 1 : 2;
        if (ch_code > 10 || nch != channels) { free(tmp); return -4; }
        if (done + (size_t)n > pcm_cap_frames) { free(tmp); return -5; }
        int32_t *chan[8];
        int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * (size_t)nch);
        for (int c = 0; c < nch; c++) {
            chan[c] = buf + (size_t)c * n;
            int cb = bps;
            if ((ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1)) cb++;
            int rc = subframe(&b, chan[c], n, cb);
            if (rc) { free(buf); free(tmp); return rc; }
        }
        if (b.pos & 7) b.pos += 8 - (b.pos & 7);
        size_t body = b.pos >> 3;
        uint16_t c16 = (uint16_t)rd(&b, 16);
        if (b.err) { free(buf); free(tmp); return -5; }
        if (crc16(data + off, body) != c16) { free(buf); free(tmp); return -3; }
        for (int i = 0; i < n; i++) {
            int32_t l, r;
            if (ch_code == 8) { l = chan[0][i]; r = l - chan[1][i]; }
            else if (ch_code == 9) { r = chan[1][i]; l = chan[0][i] + r; }
            else if (ch_code == 10) {
                int32_t mid = chan[0][i], side = chan[1][i];
                mid = (int32_t)(((uint32_t)mid << 1) | ((uint32_t)side & 1u));
                l = (mid + side) >> 1; r = (mid - side) >> 1;
            } else { l = chan[0][i]; r = nch > 1 ? chan[1][i] : 0; }
            if (nch == 2) {
                pcm_out[(done + i) * 2] = l; pcm_out[(done + i) * 2 + 1] = r;
            } else {
                for (int c = 0; c < nch; c++) pcm_out[(done + i) * nch + c] = chan[c][i];
            }
        }
        free(buf);
        if (nf < max_frames && block_sizes) block_sizes[nf] = n;
        nf++;
        done += (size_t)n;
        off += body + 2;
    }
    free(tmp);
    if (nframes_out) *nframes_out = nf;
    return (long)done;
}
