/*
 * flake_host.c -- the host C layer above the HIP C ABI (include/flakehip.h):
 * what libflake keeps on the CPU around encode_residual().
 *
 *   presets / validation        flake_set_defaults, flake_validate_params  encode.c:158-373
 *   stream header               write_headers + metadata.c                 encode.c:51-156
 *   per frame                   init_frame codes, frame header + CRC-8,
 *                               subframe headers, warm-up samples, coefs,
 *                               the device-made residual section appended
 *                               bit for bit, CRC-16, verbatim fallback,
 *                               frame counter, max frame size, MD5         encode.c:490-536,
 *                                                                          696-977, 1006
 *   variable block size         split_frame_v1 + encode_frame_vbs          vbs.c:36-119
 *
 * Everything per-sample that the reference does inside encode_residual() and
 * its feeders happens on the GPU (fhip_encode_subframes); this file only moves
 * side information and already-packed bits.  Written from the behaviour of
 * the reference; citations are relative to the reference tree.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "flake_amd.h"
#include "flakehip.h"
#include "host_internal.h"

#define VBS_PARTS 8                       /* VBS_MAX_FRAMES, vbs.h:26 */
#define MIN_BLOCK 16                      /* FLAC_MIN_BLOCKSIZE, encode.h:34 */
#define MAX_BLOCK 65535

typedef struct host_ctx {
    fhip_ctx *hip;
    fhip_ctx *hip2;                       /* second handle: chunks of a large batch alternate between two
                                             host threads so that uploads, kernels and downloads overlap */
    int chunk_frames;                     /* FLAKE_AMD_CHUNK (default 1024); 0 = one handle, one pass */
    int hip2_state;                       /* 0 not created yet, 1 created, -1 not wanted or creation failed */
    int device;
    fhip_params hp;
    int max_batch;                        /* blocks per GPU batch */
    int sr_code[2], bps_code, ch_code;
    int max_frame_size;
    uint32_t frame_count;
    int last_frame;
    int host_assembly;                    /* FLAKE_AMD_HOST_ASSEMBLY=1: build frames on the CPU */
    int host_vbs;                         /* FLAKE_AMD_HOST_VBS=1: split blocks on the CPU */
    int vbs_host_gather;                  /* FLAKE_AMD_VBS_HOST_GATHER=1: the round-1 VBS path (pieces gathered
                                             on the host, one upload / download per piece length) */
    int md5_off;                          /* FLAKE_AMD_MD5=0: STREAMINFO carries the all-zero "not computed" MD5 */
    int trace;                            /* FLAKE_AMD_TRACE=1: phase times of every batch on stderr */
    /* FLAKE_AMD_LOOKAHEAD=N: flake_encode_frame() queues up to N whole blocks and
     * encodes them as one GPU batch (see flake_amd_encode_frame) */
    int lookahead;
    int32_t *q_pcm;
    int q_count;
    unsigned long long q_seen;            /* samples per channel handed in so far */
    fa_md5 md5;
    /* single-frame buffer of flake_amd_encode_frame */
    uint8_t *frame_buffer;
    int frame_buffer_size;
    /* batch staging (host) */
    fhip_subframe_info *info;
    uint8_t *bits;
    int64_t slot;
    uint8_t *frames;                      /* device-assembled frames of the last GPU launches */
    int64_t fstride;
    int32_t *fbytes;
    uint32_t *fnum;
    int32_t *gather;                      /* ragged VBS batches, contiguous per size */
    /* page-locked staging (round 4).  The look-ahead queue and its frame buffer are the library's own: allocated
     * page-locked (fhip_host_alloc).  The batch entry's PCM and output belong to the caller, who may page-lock them
     * in place through flake_amd_pin_buffers() -- an explicit promise that the ranges stay mapped; nothing is
     * registered behind a caller's back (a cached registration of memory the caller has since freed would be
     * a stale device mapping).  FLAKE_AMD_PIN=0 turns all of it off. */
    int pin_off;
    int q_pinned, fb_pinned;
    struct { void *ptr; size_t bytes; int ok; } pin[2];      /* 0: PCM in, 1: stream out */
    char err[256];
} host_ctx;

/* ------------------------------------------------------------------ */
/* CRC-8 / CRC-16 (crc.c:24-94: poly 0x07 and 0x8005, MSB first, init 0) */
/* ------------------------------------------------------------------ */
static uint8_t crc8_tab[256];
static uint16_t crc16_tab[256];
static int crc_ready;

static void crc_setup(void)
{
    if (crc_ready) return;
    for (int i = 0; i < 256; i++) {
        uint8_t c8 = (uint8_t)i;
        uint16_t c16 = (uint16_t)(i << 8);
        for (int b = 0; b < 8; b++) {
            c8 = (uint8_t)((c8 & 0x80) ? ((c8 << 1) ^ 0x07) : (c8 << 1));
            c16 = (uint16_t)((c16 & 0x8000) ? ((c16 << 1) ^ 0x8005) : (c16 << 1));
        }
        crc8_tab[i] = c8;
        crc16_tab[i] = c16;
    }
    crc_ready = 1;
}

static uint8_t crc8(const uint8_t *d, size_t n)
{
    uint8_t c = 0;
    while (n--) c = crc8_tab[c ^ *d++];
    return c;
}

static uint16_t crc16(const uint8_t *d, size_t n)
{
    uint16_t c = 0;
    while (n--) c = (uint16_t)((c << 8) ^ crc16_tab[(c >> 8) ^ *d++]);
    return c;
}

/* ------------------------------------------------------------------ */
/* MSB-first bit sink (same byte stream as bitio.h:83-141)             */
/* ------------------------------------------------------------------ */
typedef struct { uint8_t *buf; size_t cap, pos; uint64_t acc; int nacc; int over; } sink;

static void sink_init(sink *s, uint8_t *buf, size_t cap)
{
    s->buf = buf; s->cap = cap; s->pos = 0; s->acc = 0; s->nacc = 0; s->over = 0;
}

static void sink_put(sink *s, int nb, uint32_t v)      /* wider than 32: zero-extended */
{
    if (nb == 0) return;
    if (nb > 32) { sink_put(s, nb - 32, 0); nb = 32; }
    s->acc = (s->acc << nb) | (uint64_t)(nb == 32 ? v : (v & ((1u << nb) - 1u)));
    s->nacc += nb;
    while (s->nacc >= 8) {
        if (s->pos >= s->cap) { s->over = 1; s->nacc -= 8; continue; }
        s->nacc -= 8;
        s->buf[s->pos++] = (uint8_t)(s->acc >> s->nacc);
    }
}

static void sink_put_signed(sink *s, int nb, int32_t v) { sink_put(s, nb, (uint32_t)v); }

/* append nbits of an MSB-first bit string that starts at bit 0 of src */
static void sink_append(sink *s, const uint8_t *src, int64_t nbits)
{
    int64_t full = nbits >> 3;
    if (s->nacc == 0) {                   /* byte aligned: plain copy */
        if (s->pos + (size_t)full > s->cap) { s->over = 1; return; }
        memcpy(s->buf + s->pos, src, (size_t)full);
        s->pos += (size_t)full;
    } else {
        for (int64_t i = 0; i < full; i++) sink_put(s, 8, src[i]);
    }
    int rem = (int)(nbits & 7);
    if (rem) sink_put(s, rem, (uint32_t)(src[full] >> (8 - rem)));
}

static void sink_align(sink *s) { if (s->nacc) sink_put(s, 8 - s->nacc, 0); }

/* ------------------------------------------------------------------ */
/* presets and validation                                             */
/* ------------------------------------------------------------------ */
static int ilog2u(uint32_t v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }

FLAKE_AMD_API int flake_amd_set_defaults(FlakeAmdEncodeParams *p)
{
    if (!p) return -1;
    const int lvl = p->compression;
    if (lvl < 0 || lvl > 12) return -1;
    /* level 5 is the base row of the table (encode.c:170-181) */
    p->order_method = 1; p->stereo_method = 1; p->block_size = 4096; p->prediction_type = 2;
    p->min_prediction_order = 1; p->max_prediction_order = 8;
    p->min_partition_order = 0; p->max_partition_order = 5;
    p->padding_size = 8192; p->variable_block_size = 0; p->allow_vbs = 0;
    static const int fixed_min[3] = {2, 2, 0}, fixed_max[3] = {2, 4, 4};
    if (lvl <= 2) {
        p->block_size = 1152; p->prediction_type = 1;
        p->min_prediction_order = fixed_min[lvl]; p->max_prediction_order = fixed_max[lvl];
        p->max_partition_order = 3;
        if (lvl == 0) p->stereo_method = 0;
    } else if (lvl == 3) {
        p->stereo_method = 0; p->max_prediction_order = 6; p->max_partition_order = 4;
    } else if (lvl == 4) {
        p->max_partition_order = 4;
    } else if (lvl == 6 || lvl == 7) {
        p->max_partition_order = 6;
        if (lvl == 7) p->order_method = 3;
    } else if (lvl >= 8) {
        p->order_method = (lvl == 10 || lvl == 12) ? 5 : 6;
        p->max_prediction_order = (lvl >= 11) ? 32 : 12;
        p->max_partition_order = (lvl == 8) ? 6 : 8;
        if (lvl >= 11) p->block_size = 8192;
        if (lvl >= 9) { p->allow_vbs = 1; p->variable_block_size = 1; }
    }
    return 0;
}

FLAKE_AMD_API int flake_amd_validate_params(const FlakeAmdContext *s)
{
    if (!s) return -1;
    const FlakeAmdEncodeParams *p = &s->params;
    int subset = 0;
    if (s->channels < 1 || s->channels > 8) return -1;
    if (s->sample_rate < 1 || s->sample_rate > 655350) return -1;
    if (s->bits_per_sample < 4 || s->bits_per_sample > 32) return -1;
    if (s->bits_per_sample < 8 || s->bits_per_sample > 24 || s->bits_per_sample % 4) subset = 1;
    if (p->compression < 0 || p->compression > 12) return -1;
    if (p->order_method < 0 || p->order_method > 6) return -1;
    if (p->stereo_method < 0 || p->stereo_method > 1) return -1;
    if (p->block_size < MIN_BLOCK || p->block_size > MAX_BLOCK) return -1;
    if (s->sample_rate <= 48000 && p->block_size > 4608) subset = 1;
    if (p->prediction_type < 0 || p->prediction_type > 2) return -1;
    if (p->min_prediction_order > p->max_prediction_order) return -1;
    if (p->prediction_type == 1) {
        if (p->min_prediction_order < 0 || p->min_prediction_order > 4) return -1;
        if (p->max_prediction_order < 0 || p->max_prediction_order > 4) return -1;
    } else {
        if (p->min_prediction_order < 1 || p->min_prediction_order > 32) return -1;
        if (p->max_prediction_order < 1 || p->max_prediction_order > 32) return -1;
        if (s->sample_rate <= 48000 && p->max_prediction_order > 12) subset = 1;
    }
    if (p->min_partition_order > p->max_partition_order) return -1;
    if (p->min_partition_order < 0 || p->min_partition_order > 8) return -1;
    if (p->max_partition_order < 0 || p->max_partition_order > 8) return -1;
    if (p->padding_size < 0 || p->padding_size >= (1 << 24)) return -1;
    if (p->variable_block_size < 0 || p->variable_block_size > 1) return -1;
    if (p->variable_block_size > 0 && !p->allow_vbs) return -1;
    if (p->block_size < VBS_PARTS * MIN_BLOCK && p->allow_vbs) return -1;
    return subset;
}

FLAKE_AMD_API const char *flake_amd_get_version(void) { return "flake-amd 0.1"; }

FLAKE_AMD_API const char *flake_amd_last_error(const FlakeAmdContext *s)
{
    return (s && s->private_ctx) ? ((host_ctx *)s->private_ctx)->err : "";
}

/* ------------------------------------------------------------------ */
/* stream header: encode.c:51-156 + metadata.c:32-229                  */
/* ------------------------------------------------------------------ */
static int frame_verbatim_size(const host_ctx *c, int n)          /* encode.c:521-527 */
{
    const int bps = c->hp.bits_per_sample;
    if (c->hp.channels == 2) return 16 + ((n * (bps + bps + 1) + 7) >> 3);
    return 16 + ((n * c->hp.channels * bps + 7) >> 3);
}

FLAKE_AMD_API int flake_amd_get_streaminfo(const FlakeAmdContext *s, FlakeAmdStreaminfo *si)
{
    if (!s || !si || !s->private_ctx || flake_amd_validate_params(s) < 0) return -1;
    const host_ctx *c = (const host_ctx *)s->private_ctx;
    si->min_block_size = (s->params.variable_block_size || s->params.allow_vbs) ? 16u
                                                                                : (unsigned)s->params.block_size;
    si->max_block_size = (unsigned)s->params.block_size;
    si->min_frame_size = 0;
    si->max_frame_size = (unsigned)c->max_frame_size;
    si->sample_rate = (unsigned)s->sample_rate;
    si->channels = (unsigned)s->channels;
    si->bits_per_sample = (unsigned)s->bits_per_sample;
    si->samples = s->samples;
    if (c->md5_off) memset(si->md5sum, 0, 16);
    else fa_md5_final(&c->md5, si->md5sum);
    return 0;
}

FLAKE_AMD_API void flake_amd_write_streaminfo(const FlakeAmdStreaminfo *si, unsigned char *d)
{
    sink s;
    memset(d, 0, 34);
    sink_init(&s, d, 34);
    sink_put(&s, 16, si->min_block_size);
    sink_put(&s, 16, si->max_block_size);
    sink_put(&s, 24, si->min_frame_size);
    sink_put(&s, 24, si->max_frame_size);
    sink_put(&s, 20, si->sample_rate);
    sink_put(&s, 3, si->channels - 1);
    sink_put(&s, 5, si->bits_per_sample - 1);
    sink_put(&s, 4, 0);
    sink_put(&s, 32, si->samples);
    memcpy(d + 18, si->md5sum, 16);
}

static void put_block_header(uint8_t *d, int last, int type, int size)
{
    d[0] = (uint8_t)((last << 7) | type);
    d[1] = (uint8_t)(size >> 16); d[2] = (uint8_t)(size >> 8); d[3] = (uint8_t)size;
}

static int write_stream_header(FlakeAmdContext *s, uint8_t *h)
{
    int n = 0;
    memcpy(h, "fLaC", 4); n += 4;
    put_block_header(h + n, 0, 0, 34); n += 4;
    FlakeAmdStreaminfo si;
    flake_amd_get_streaminfo(s, &si);
    flake_amd_write_streaminfo(&si, h + n); n += 34;
    /* Vorbis comment with the vendor string only (metadata.c:86-229) */
    const char *vendor = "flake-amd 0.1";
    const int vlen = (int)strlen(vendor);
    const int last_vc = s->params.padding_size == 0;
    put_block_header(h + n, last_vc, 4, 8 + vlen); n += 4;
    h[n++] = (uint8_t)vlen; h[n++] = 0; h[n++] = 0; h[n++] = 0;
    memcpy(h + n, vendor, (size_t)vlen); n += vlen;
    memset(h + n, 0, 4); n += 4;
    if (s->params.padding_size > 0) {
        put_block_header(h + n, 1, 1, s->params.padding_size); n += 4;
        memset(h + n, 0, (size_t)s->params.padding_size); n += s->params.padding_size;
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* init / close                                                       */
/* ------------------------------------------------------------------ */
static const int sr_table[16] = {0, 0, 0, 0, 8000, 16000, 22050, 24000, 32000, 44100, 48000,
                                 96000, 0, 0, 0, 0};              /* encode.c:33-37 */
static const int bd_table[8] = {0, 8, 12, 0, 16, 20, 24, 0};      /* encode.c:39-41 */
static const int bs_table[15] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048,
                                 4096, 8192, 16384};              /* encode.c:43-49 */

FLAKE_AMD_API int flake_amd_encode_init(FlakeAmdContext *s)
{
    if (!s) return -1;
    s->header = NULL;
    s->private_ctx = NULL;
    if (flake_amd_validate_params(s) < 0) return -1;
    crc_setup();
    host_ctx *c = (host_ctx *)calloc(1, sizeof(host_ctx));
    if (!c) return -1;
    s->private_ctx = c;

    fhip_params *hp = &c->hp;
    hp->channels = s->channels; hp->sample_rate = s->sample_rate;
    hp->bits_per_sample = s->bits_per_sample; hp->block_size = s->params.block_size;
    hp->order_method = s->params.order_method; hp->stereo_method = s->params.stereo_method;
    hp->prediction_type = s->params.prediction_type;
    hp->min_prediction_order = s->params.min_prediction_order;
    hp->max_prediction_order = s->params.max_prediction_order;
    hp->min_partition_order = s->params.min_partition_order;
    hp->max_partition_order = s->params.max_partition_order;
    hp->variable_block_size = s->params.variable_block_size;
    hp->allow_vbs = s->params.allow_vbs;
    hp->lpc_precision = 15;                                       /* encode.c:443 */

    /* sample-rate / bit-depth codes, encode.c:400-438 */
    c->sr_code[0] = 0; c->sr_code[1] = 0;
    for (int i = 4; i < 12; i++) if (s->sample_rate == sr_table[i]) { c->sr_code[0] = i; break; }
    if (!c->sr_code[0]) {
        const int sr = s->sample_rate;
        if (sr % 1000 == 0 && sr <= 255000) { c->sr_code[0] = 12; c->sr_code[1] = sr / 1000; }
        else if (sr % 10 == 0 && sr <= 655350) { c->sr_code[0] = 14; c->sr_code[1] = sr / 10; }
        else if (sr < 65535) { c->sr_code[0] = 13; c->sr_code[1] = sr; }
    }
    c->bps_code = 0;
    for (int i = 1; i < 8; i++) if (s->bits_per_sample == bd_table[i]) { c->bps_code = i; break; }
    c->ch_code = s->channels - 1;
    c->max_frame_size = frame_verbatim_size(c, s->params.block_size);   /* encode.c:446-450 */
    c->frame_buffer_size = c->max_frame_size * 3 / 2;
    {
        /* the look-ahead queue needs the stream length (to know the last block) */
        const char *el = getenv("FLAKE_AMD_LOOKAHEAD");
        c->lookahead = (el && s->samples > 0) ? atoi(el) : 0;
        if (c->lookahead < 2) c->lookahead = 0;
        if (c->lookahead > 4096) c->lookahead = 4096;
        if (c->lookahead) {
            const long long one = (long long)c->frame_buffer_size + 8 * 32;    /* a VBS block: up to 8 frames */
            if (one * c->lookahead > 0x7FFFFFFFLL) c->lookahead = (int)(0x7FFFFFFFLL / one);
            c->frame_buffer_size = (int)(one * c->lookahead);
            { const char *ev = getenv("FLAKE_AMD_PIN"); c->pin_off = ev && ev[0] == '0'; }
            const size_t qb = sizeof(int32_t) * (size_t)(c->lookahead + 1) * (size_t)s->params.block_size * (size_t)s->channels;
            if (!c->pin_off) c->q_pcm = (int32_t *)fhip_host_alloc(qb);
            c->q_pinned = c->q_pcm != NULL;
            if (!c->q_pcm) c->q_pcm = (int32_t *)malloc(qb);
        }
    }
    { const char *ev = getenv("FLAKE_AMD_PIN"); c->pin_off = ev && ev[0] == '0'; }
    if (c->lookahead && !c->pin_off) {
        c->frame_buffer = (uint8_t *)fhip_host_alloc((size_t)c->frame_buffer_size);
        c->fb_pinned = c->frame_buffer != NULL;
        if (c->frame_buffer) memset(c->frame_buffer, 0, (size_t)c->frame_buffer_size);
    }
    if (!c->frame_buffer) c->frame_buffer = (uint8_t *)calloc((size_t)c->frame_buffer_size, 1);
    fa_md5_init(&c->md5);

    const char *eb = getenv("FLAKE_AMD_BATCH"), *ed = getenv("FLAKE_AMD_DEVICE");
    const char *eh = getenv("FLAKE_AMD_HOST_ASSEMBLY");
    c->host_assembly = eh && eh[0] == '1';
    { const char *ev = getenv("FLAKE_AMD_HOST_VBS"); c->host_vbs = ev && ev[0] == '1'; }
    { const char *ev = getenv("FLAKE_AMD_VBS_HOST_GATHER"); c->vbs_host_gather = ev && ev[0] == '1'; }
    { const char *ev = getenv("FLAKE_AMD_MD5"); c->md5_off = ev && ev[0] == '0'; }
    { const char *ev = getenv("FLAKE_AMD_TRACE"); c->trace = ev && ev[0] == '1'; }
    c->max_batch = eb ? atoi(eb) : 1024;
    if (c->max_batch < 1) c->max_batch = 1;
    if (c->lookahead > c->max_batch) c->max_batch = c->lookahead;
    /* a VBS block may turn into up to 8 frames of the smallest size */
    const int max_frames = c->max_batch * (s->params.variable_block_size ? VBS_PARTS : 1);
    int rc = fhip_create(&c->hip, ed ? atoi(ed) : 0, hp, max_frames);
    if (rc != FHIP_OK) {
        snprintf(c->err, sizeof c->err, "fhip_create: %s", fhip_strerror(rc));
        flake_amd_encode_close(s);
        return rc;
    }
    {
        /* large uniform batches run in chunks through two handles (run_chunked) */
        const char *ec = getenv("FLAKE_AMD_CHUNK");
        c->chunk_frames = ec ? atoi(ec) : 1024;
        if (c->chunk_frames < 0) c->chunk_frames = 0;
        /* (the second handle and its workspaces are created by the first batch that qualifies) */
        c->hip2_state = (c->chunk_frames > 0 && c->max_batch >= 2 * c->chunk_frames && !c->host_assembly) ? 0 : -1;
        c->device = ed ? atoi(ed) : 0;
    }
    c->slot = (frame_verbatim_size(c, s->params.block_size) + 3) & ~3;
    const size_t nsub = (size_t)max_frames * (size_t)s->channels;
    c->info = (fhip_subframe_info *)malloc(nsub * sizeof(fhip_subframe_info));
    c->bits = (uint8_t *)malloc(nsub * (size_t)c->slot);
    c->gather = (int32_t *)malloc(sizeof(int32_t) * (size_t)c->max_batch *
                                  (size_t)s->params.block_size * (size_t)s->channels);
    c->fstride = fhip_frame_stride(hp, s->params.block_size);
    c->frames = (uint8_t *)malloc((size_t)max_frames * (size_t)c->fstride);
    c->fbytes = (int32_t *)malloc(sizeof(int32_t) * (size_t)max_frames);
    c->fnum = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)max_frames);
    s->header = (unsigned char *)calloc((size_t)s->params.padding_size + 1024, 1);
    if (!c->frame_buffer || !c->info || !c->bits || !c->gather || !s->header || !c->frames ||
        !c->fbytes || !c->fnum) {
        flake_amd_encode_close(s);
        return -1;
    }
    return write_stream_header(s, s->header);
}

FLAKE_AMD_API void *flake_amd_get_buffer(const FlakeAmdContext *s)
{
    return (s && s->private_ctx) ? ((host_ctx *)s->private_ctx)->frame_buffer : NULL;
}

FLAKE_AMD_API void flake_amd_encode_close(FlakeAmdContext *s)
{
    if (!s) return;
    host_ctx *c = (host_ctx *)s->private_ctx;
    if (c) {
        if (c->hip) (void)fhip_sync(c->hip);
        if (c->hip2) (void)fhip_sync(c->hip2);
        for (int k = 0; k < 2; k++) if (c->pin[k].ok) (void)fhip_host_unregister(c->pin[k].ptr);
        if (c->q_pinned) fhip_host_free(c->q_pcm); else free(c->q_pcm);
        if (c->fb_pinned) fhip_host_free(c->frame_buffer); else free(c->frame_buffer);
        if (c->hip) fhip_destroy(c->hip);
        if (c->hip2) fhip_destroy(c->hip2);
        free(c->info); free(c->bits); free(c->gather);
        free(c->frames); free(c->fbytes); free(c->fnum);
        free(c);
    }
    free(s->header);
    s->header = NULL;
    s->private_ctx = NULL;
}

/* ------------------------------------------------------------------ */
/* one frame from the device's side information                        */
/* ------------------------------------------------------------------ */
static void put_utf8(sink *s, uint32_t v)                          /* encode.c:696-716 */
{
    if (v < 0x80) { sink_put(s, 8, v); return; }
    const int bytes = (ilog2u(v) + 4) / 5;
    int sh = (bytes - 1) * 6;
    sink_put(s, 8, ((256u - (256u >> bytes)) | (v >> sh)) & 0xFFu);
    while (sh >= 6) { sh -= 6; sink_put(s, 8, 0x80u | ((v >> sh) & 0x3Fu)); }
}

static void put_frame_header(sink *s, const host_ctx *c, uint32_t number, int n, int ch_mode)
{
    int bs0 = -1, bs1 = -1;                                        /* encode.c:502-519 */
    for (int i = 0; i < 15; i++) if (n == bs_table[i]) { bs0 = i; break; }
    if (bs0 < 0) { bs0 = (n <= 256) ? 6 : 7; bs1 = n - 1; }
    const size_t start = s->pos;
    sink_put(s, 15, 0x7FFC);                                       /* encode.c:718-764 */
    sink_put(s, 1, (uint32_t)c->hp.allow_vbs);
    sink_put(s, 4, (uint32_t)bs0);
    sink_put(s, 4, (uint32_t)c->sr_code[0]);
    sink_put(s, 4, (uint32_t)(ch_mode == FHIP_CH_NOT_STEREO ? c->ch_code : ch_mode));
    sink_put(s, 3, (uint32_t)c->bps_code);
    sink_put(s, 1, 0);
    put_utf8(s, number);
    if (bs1 >= 0) sink_put(s, bs1 < 256 ? 8 : 16, (uint32_t)bs1);
    if (c->sr_code[1] > 0) sink_put(s, c->sr_code[1] < 256 ? 8 : 16, (uint32_t)c->sr_code[1]);
    sink_put(s, 8, s->over ? 0 : crc8(s->buf + start, s->pos - start));
}

/* samples of one channel after decorrelation and wasted-bits shift, recomputed
 * on the host for the rare VERBATIM subframes (encode.c:648-694, :558-593) */
static void verbatim_samples(const host_ctx *c, const int32_t *pcm, int n, int ch,
                             const fhip_subframe_info *i, int32_t *dst)
{
    const int nch = c->hp.channels;
    for (int t = 0; t < n; t++) {
        int32_t v;
        if (nch == 2 && i->ch_mode != FHIP_CH_LEFT_RIGHT) {
            const int32_t l = pcm[2 * t], r = pcm[2 * t + 1];
            const int32_t side = (int32_t)((uint32_t)l - (uint32_t)r);
            if (i->ch_mode == FHIP_CH_MID_SIDE) v = ch ? side : ((int32_t)((uint32_t)l + (uint32_t)r) >> 1);
            else if (i->ch_mode == FHIP_CH_LEFT_SIDE) v = ch ? side : l;
            else v = ch ? r : side;                                /* RIGHT_SIDE */
        } else {
            v = pcm[(size_t)t * nch + ch];
        }
        dst[t] = v >> i->wasted;
    }
}

/* encode.c:800-905 output_subframes for one channel */
static void put_subframe(sink *s, const host_ctx *c, const fhip_subframe_info *i,
                         const uint8_t *rice, const int32_t *pcm, int n, int ch, int force_verbatim,
                         int32_t *scratch)
{
    const int type = force_verbatim ? FHIP_SUB_VERBATIM : i->type;
    sink_put(s, 1, 0);
    sink_put(s, 6, (uint32_t)(force_verbatim ? FHIP_SUB_VERBATIM : i->type_code));
    if (i->wasted) { sink_put(s, 1, 1); sink_put(s, i->wasted - 1, 0); sink_put(s, 1, 1); }
    else sink_put(s, 1, 0);
    switch (type) {
    case FHIP_SUB_CONSTANT:
        sink_put_signed(s, i->obits, i->warmup[0]);
        break;
    case FHIP_SUB_VERBATIM:
        verbatim_samples(c, pcm, n, ch, i, scratch);
        for (int t = 0; t < n; t++) sink_put_signed(s, i->obits, scratch[t]);
        break;
    case FHIP_SUB_FIXED:
        for (int t = 0; t < i->order; t++) sink_put_signed(s, i->obits, i->warmup[t]);
        sink_append(s, rice, i->rice_nbits);
        break;
    case FHIP_SUB_LPC:
        for (int t = 0; t < i->order; t++) sink_put_signed(s, i->obits, i->warmup[t]);
        sink_put(s, 4, (uint32_t)(c->hp.lpc_precision - 1));
        sink_put_signed(s, 5, i->shift);
        for (int t = 0; t < i->order; t++) sink_put_signed(s, c->hp.lpc_precision, i->coefs[t]);
        sink_append(s, rice, i->rice_nbits);
        break;
    }
}

/* encode.c:944-964: header + subframes + footer; verbatim re-encode when the
 * frame is larger than its verbatim size or a section did not fit its slot */
static int assemble_frame(host_ctx *c, uint32_t number, const int32_t *pcm, int n,
                          const fhip_subframe_info *info, const uint8_t *bits,
                          uint8_t *out, size_t cap, int32_t *scratch)
{
    const int nch = c->hp.channels;
    int overflow = 0;
    for (int ch = 0; ch < nch; ch++) if (info[ch].rice_nbits < 0) overflow = 1;
    for (int pass = overflow ? 1 : 0; pass < 2; pass++) {
        sink s;
        sink_init(&s, out, cap);
        put_frame_header(&s, c, number, n, info[0].ch_mode);
        for (int ch = 0; ch < nch; ch++)
            put_subframe(&s, c, &info[ch], bits + (size_t)ch * (size_t)c->slot, pcm, n, ch, pass, scratch);
        sink_align(&s);
        if (!s.over) {
            const uint16_t crc = crc16(out, s.pos);
            sink_put(&s, 16, crc);
        }
        if (!s.over && (pass == 1 || (int)s.pos <= frame_verbatim_size(c, n))) return (int)s.pos;
        if (pass == 1) return -1;
    }
    return -1;
}

/* ------------------------------------------------------------------ */
/* vbs.c:36-83 split_frame_v1                                          */
/* ------------------------------------------------------------------ */
static int vbs_split(const int32_t *pcm, int channels, int block_size, int sizes[VBS_PARTS])
{
    const int n = block_size / VBS_PARTS;
    int64_t score[VBS_PARTS];
    for (int p = 0; p < VBS_PARTS; p++) {
        const int32_t *b = pcm + (size_t)p * n * channels;
        int64_t acc = 0;
        for (int ch = 0; ch < channels; ch++)
            for (int j = 2; j < n; j++) {
                const uint32_t x0 = (uint32_t)b[(size_t)j * channels + ch];
                const uint32_t x1 = (uint32_t)b[(size_t)(j - 1) * channels + ch];
                const uint32_t x2 = (uint32_t)b[(size_t)(j - 2) * channels + ch];
                const int32_t d = (int32_t)(x0 - 2u * x1 + x2);
                acc += (int64_t)(d < 0 ? (int32_t)(0u - (uint32_t)d) : d);
            }
        score[p] = acc / channels + 1;
    }
    int nf = 0;
    memset(sizes, 0, sizeof(int) * VBS_PARTS);
    for (int p = 0; p < VBS_PARTS; p++) {
        int cut = (p == 0);
        if (p > 0) {
            /* vbs.c:69 with its int abs() and 32-bit multiply */
            int32_t diff = (int32_t)(uint32_t)(uint64_t)(score[p - 1] - score[p]);
            if (diff < 0) diff = (int32_t)(0u - (uint32_t)diff);
            const int32_t scaled = (int32_t)((uint32_t)diff * 200u);
            cut = ((int64_t)scaled / score[p - 1]) > 50;
        }
        if (cut) nf++;
        sizes[nf - 1] += n;
    }
    return nf;
}

/* ------------------------------------------------------------------ */
/* batches                                                            */
/* ------------------------------------------------------------------ */
typedef struct { const int32_t *pcm; int n; int block; } piece;       /* one FLAC frame to make */

static int run_gpu(host_ctx *c, const int32_t *pcm, int nframes, int n, size_t first_sub,
                   size_t first_frame)
{
    fhip_batch b;
    memset(&b, 0, sizeof b);
    b.pcm = pcm; b.nframes = nframes; b.block_size = n;
    if (c->host_assembly) {
        /* the CPU builds the frames from side information + residual sections */
        b.info = c->info + first_sub;
        b.rice_bits = c->bits + first_sub * (size_t)c->slot;
    }
    b.rice_slot_bytes = c->slot;
    /* whole frames come back assembled (K4); numbers were filled by the caller */
    b.frames = c->frames + first_frame * (size_t)c->fstride;
    b.frame_stride = c->fstride;
    b.frame_bytes = c->fbytes + first_frame;
    b.frame_numbers = c->fnum + first_frame;
    int rc = fhip_encode_subframes(c->hip, &b);
    if (rc != FHIP_OK)
        snprintf(c->err, sizeof c->err, "fhip_encode_subframes: %s (%s)", fhip_strerror(rc),
                 fhip_last_error(c->hip));
    return rc;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* encode.c:1006 md5_accumulate over the batch's input, on a helper thread: the
 * hash is sequential over the whole stream (~0.5 GB/s) and would otherwise sit
 * behind every GPU batch. */
typedef struct { fa_md5 *m; const int32_t *pcm; size_t nvalues; int bps; } md5_job;
static void *md5_worker(void *arg)
{
    md5_job *j = (md5_job *)arg;
    fa_md5_pcm(j->m, j->pcm, j->nvalues, j->bps);
    return NULL;
}

/* A uniform batch in chunks through two handles (one host thread each): chunk k's upload and
 * kernels run while chunk k-1's frames come back -- PCIe is full duplex -- and a chunk's frames
 * go to `out` as soon as the chunks before it have reported their sizes. */
typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    long long *end;                       /* end[k]: stream offset behind chunk k, -1 = not known yet */
    int failed;
    int upload_turn;                      /* the chunk whose upload may run now */
    double t0;                            /* FLAKE_AMD_TRACE: start of the batch */
} chunk_sync;
typedef struct {
    host_ctx *c;
    fhip_ctx *h;
    chunk_sync *sy;
    const int32_t *pcm;
    int n, nch, np, chunk, first, nchunks;
    int split_last, all_chunks;           /* the last full-size chunk runs as two halves (chunks all_chunks-2, all_chunks-1) */
    uint8_t *out;
    long long cap;
    int rc;
} chunk_job;

static void *chunk_worker(void *arg)
{
    chunk_job *j = (chunk_job *)arg;
    host_ctx *c = j->c;
    for (int k = j->first; k < j->nchunks; k += 2) {
        int f0 = k * j->chunk, nf = (j->np - f0 < j->chunk) ? j->np - f0 : j->chunk;
        if (j->split_last && k >= j->all_chunks - 2) {
            /* what is left behind the last upload -- kernels, the download of the packed frames -- is a chunk's
             * worth of time nothing overlaps: the batch ends on two half chunks */
            const int g0 = (j->all_chunks - 2) * j->chunk, left = j->np - g0, h = left / 2;
            f0 = (k == j->all_chunks - 2) ? g0 : g0 + h;
            nf = (k == j->all_chunks - 2) ? h : left - h;
        }
        fhip_batch b;
        memset(&b, 0, sizeof b);
        b.pcm = j->pcm + (size_t)f0 * (size_t)j->n * (size_t)j->nch;
        b.nframes = nf; b.block_size = j->n;
        b.frame_bytes = c->fbytes + f0;
        b.frame_numbers = c->fnum + f0;
        int64_t total = 0;
        /* uploads take turns in chunk order: two handles uploading at once share the link and then sit in their
         * kernel and download phases together; one behind the other, a chunk's kernels and download run beside the
         * next chunk's upload (round 4: 3.3 -> 2.7 ms per 4096 frames) */
        pthread_mutex_lock(&j->sy->mu);
        while (!j->sy->failed && j->sy->upload_turn != k) pthread_cond_wait(&j->sy->cv, &j->sy->mu);
        pthread_mutex_unlock(&j->sy->mu);
        /* (the next upload queued behind this one's marker on the device instead -- no host round trip between
         * the copies -- measured slower: 3.0-3.1 ms; the cross-queue wait costs more than the hand-over) */
        const double tu0 = now_ms();
        int rc = j->sy->failed ? FHIP_E_GENERIC : fhip_frames_packed_upload(j->h, &b);
        const double tu1 = now_ms();
        pthread_mutex_lock(&j->sy->mu);
        j->sy->upload_turn = k + 1;
        pthread_cond_broadcast(&j->sy->cv);
        pthread_mutex_unlock(&j->sy->mu);
        if (rc == FHIP_OK) rc = fhip_frames_packed_begin(j->h, &b, &total);
        pthread_mutex_lock(&j->sy->mu);
        while (!j->sy->failed && k > 0 && j->sy->end[k - 1] < 0) pthread_cond_wait(&j->sy->cv, &j->sy->mu);
        const long long start = (k > 0) ? j->sy->end[k - 1] : 0;
        if (rc != FHIP_OK || j->sy->failed || start + total > j->cap) {
            if (rc == FHIP_OK && !j->sy->failed)
                snprintf(c->err, sizeof c->err, "output buffer too small: chunk %d ends at byte %lld of %lld",
                         k, start + (long long)total, j->cap);
            j->sy->failed = 1;
            j->sy->end[k] = 0;
            pthread_cond_broadcast(&j->sy->cv);
            pthread_mutex_unlock(&j->sy->mu);
            j->rc = (rc != FHIP_OK) ? rc : FHIP_E_INVALID;
            (void)fhip_frames_packed_fetch_wait(j->h);       /* nothing of ours still writes to `out` */
            return NULL;
        }
        j->sy->end[k] = start + total;
        pthread_cond_broadcast(&j->sy->cv);
        pthread_mutex_unlock(&j->sy->mu);
        const double tf0 = now_ms();
        /* (not waited for here: the download runs beside this handle's next upload; the worker waits once, below) */
        rc = fhip_frames_packed_fetch_async(j->h, j->out + start, j->cap - start);
        if (c->trace)
            fprintf(stderr, "flake_amd chunk %d (%d frames): upload %.3f .. %.3f ms, kernels done %.3f, fetch (%lld bytes) done %.3f\n",
                    k, nf, tu0 - j->sy->t0, tu1 - j->sy->t0, tf0 - j->sy->t0, (long long)total, now_ms() - j->sy->t0);
        if (rc != FHIP_OK) {
            pthread_mutex_lock(&j->sy->mu);
            j->sy->failed = 1;
            pthread_cond_broadcast(&j->sy->cv);
            pthread_mutex_unlock(&j->sy->mu);
            j->rc = rc;
            (void)fhip_frames_packed_fetch_wait(j->h);
            return NULL;
        }
    }
    j->rc = fhip_frames_packed_fetch_wait(j->h);
    return NULL;
}

/* returns bytes written, or -1 */
static long long run_chunked(host_ctx *c, const int32_t *pcm, int np, int n, int nch, uint8_t *out, size_t cap)
{
    const int chunk = c->chunk_frames;
    int nchunks = (np + chunk - 1) / chunk;
    /* end on two half chunks when the last one is (nearly) a full one */
    const int last = np - (nchunks - 1) * chunk;
    const int split_last = nchunks >= 2 && last > chunk / 2 && last >= 64;
    if (split_last) nchunks++;
    chunk_sync sy;
    long long *end = (long long *)malloc(sizeof(long long) * (size_t)nchunks);
    if (!end) { snprintf(c->err, sizeof c->err, "chunked batch: out of host memory"); return -1; }
    c->err[0] = 0;
    for (int k = 0; k < nchunks; k++) end[k] = -1;
    pthread_mutex_init(&sy.mu, NULL);
    pthread_cond_init(&sy.cv, NULL);
    sy.end = end; sy.failed = 0; sy.upload_turn = 0; sy.t0 = now_ms();
    chunk_job ja = { c, c->hip, &sy, pcm, n, nch, np, chunk, 0, nchunks, split_last, nchunks, out, (long long)cap, FHIP_OK };
    chunk_job jb = ja;
    jb.h = c->hip2; jb.first = 1;
    pthread_t tb;
    const int have_b = pthread_create(&tb, NULL, chunk_worker, &jb) == 0;
    if (!have_b) {                         /* no second thread: this one walks every chunk */
        for (int k = 0; k < nchunks && ja.rc == FHIP_OK; k++) {
            chunk_job one = ja;
            one.first = k; one.nchunks = k + 1;
            chunk_worker(&one);
            ja.rc = one.rc;
        }
    } else {
        chunk_worker(&ja);
        pthread_join(tb, NULL);
    }
    long long total = -1;
    if (!sy.failed && ja.rc == FHIP_OK && (!have_b || jb.rc == FHIP_OK)) total = end[nchunks - 1];
    else if (!c->err[0])
        snprintf(c->err, sizeof c->err, "chunked batch failed: %s (%s) / %s (%s)", fhip_strerror(ja.rc),
                 fhip_last_error(c->hip), have_b ? fhip_strerror(jb.rc) : "-", c->hip2 ? fhip_last_error(c->hip2) : "");
    pthread_mutex_destroy(&sy.mu);
    pthread_cond_destroy(&sy.cv);
    free(end);
    return total;
}

/* Encode `count` blocks starting at pcm (each block_size samples/channel). */
static long long encode_batch(FlakeAmdContext *s, host_ctx *c, const int32_t *pcm, int count,
                              int block_size, uint8_t *out, size_t cap, int *frame_sizes)
{
    const int nch = c->hp.channels;
    const size_t bstride = (size_t)block_size * nch;
    const int vbs = s->params.variable_block_size > 0 && (block_size % VBS_PARTS) == 0 &&
                    block_size >= VBS_PARTS * MIN_BLOCK;           /* encode.c:997-999 */
    piece *pieces = (piece *)malloc(sizeof(piece) * (size_t)count * VBS_PARTS);
    int32_t *scratch = (int32_t *)malloc(sizeof(int32_t) * (size_t)block_size);
    if (!pieces || !scratch) { free(pieces); free(scratch); return -1; }
    /* hash a copy of the state; it is committed only if the batch succeeds */
    fa_md5 md5_next = c->md5;
    md5_job job = { &md5_next, pcm, (size_t)count * bstride, c->hp.bits_per_sample };
    pthread_t md5_thread;
    int md5_running = 0, md5_done = 0;
    if (!c->md5_off) md5_running = pthread_create(&md5_thread, NULL, md5_worker, &job) == 0;
    int np = 0;
    int32_t *dev_nf = NULL, *dev_sizes = NULL;
    double t_begin = now_ms(), t_split = 0, t_gather = 0, t_gpu = 0, t_out = 0;
    int ngroups = 0;
    if (vbs && !c->host_vbs && !c->host_assembly && !c->vbs_host_gather) {
        /* the whole variable-block-size batch on the device: split, gather of the pieces, one pass
         * of the path per piece length, frames packed in stream order, one download */
        int64_t wrote = 0;
        int32_t mx = 0;
        uint32_t next = c->frame_count;
        const int rc = fhip_encode_blocks_vbs_packed(c->hip, pcm, count, block_size, c->frame_count, out,
                                                     (int64_t)cap, frame_sizes ? frame_sizes : c->fbytes, NULL,
                                                     &wrote, &mx, &next);
        free(pieces); free(scratch);
        pieces = NULL; scratch = NULL;
        if (rc != FHIP_OK) {
            snprintf(c->err, sizeof c->err, "fhip_encode_blocks_vbs_packed: %s (%s)", fhip_strerror(rc),
                     fhip_last_error(c->hip));
            if (md5_running) pthread_join(md5_thread, NULL);
            return -1;
        }
        if (mx > c->max_frame_size) c->max_frame_size = mx;        /* encode.c:967 */
        c->frame_count = next;                                     /* encode.c:969-975 */
        if (c->trace)
            fprintf(stderr, "flake_amd batch: %d vbs blocks on the device, %.2f ms\n", count, now_ms() - t_begin);
        if (!c->md5_off) {
            if (md5_running) { pthread_join(md5_thread, NULL); md5_running = 0; }
            else md5_worker(&job);
            c->md5 = md5_next;
        }
        return (long long)wrote;
    }
    if (vbs && !c->host_vbs) {
        /* split_frame_v1 on the device (K-vbs) for the whole batch */
        dev_nf = (int32_t *)malloc(sizeof(int32_t) * (size_t)count);
        dev_sizes = (int32_t *)malloc(sizeof(int32_t) * (size_t)count * VBS_PARTS);
        if (!dev_nf || !dev_sizes ||
            fhip_vbs_split(c->hip, pcm, count, block_size, dev_nf, dev_sizes) != FHIP_OK) {
            snprintf(c->err, sizeof c->err, "fhip_vbs_split failed");
            if (md5_running) pthread_join(md5_thread, NULL);
            free(dev_nf); free(dev_sizes); free(pieces); free(scratch);
            return -1;
        }
    }
    for (int b = 0; b < count; b++) {
        int sizes[VBS_PARTS], nf = 1;
        sizes[0] = block_size;
        if (vbs) {
            if (dev_nf) {
                nf = dev_nf[b];
                memcpy(sizes, dev_sizes + (size_t)b * VBS_PARTS, sizeof sizes);
            } else {
                nf = vbs_split(pcm + b * bstride, nch, block_size, sizes);
            }
            if (nf <= 1) { nf = 1; sizes[0] = block_size; }        /* vbs.c:100, encode.c:1001 */
        }
        int pos = 0;
        for (int f = 0; f < nf; f++) {
            pieces[np].pcm = pcm + b * bstride + (size_t)pos * nch;
            pieces[np].n = sizes[f];
            pieces[np].block = b;
            pos += sizes[f];
            np++;
        }
    }
    t_split = now_ms() - t_begin;
    /* GPU: one launch per distinct frame length; pieces keep their order in info[] */
    long long total = -1;
    int *slot_of = (int *)malloc(sizeof(int) * (size_t)np);
    char *done = (char *)calloc((size_t)np, 1);
    uint32_t *num_of = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)np);
    if (!slot_of || !done || !num_of) goto out;
    {
        /* frame numbers in stream order (encode.c:969-975) */
        uint32_t fc = c->frame_count;
        for (int i = 0; i < np; i++) { num_of[i] = fc; fc += s->params.allow_vbs ? (uint32_t)pieces[i].n : 1u; }
    }
    /* every piece the same length and in place (no VBS split happened): the device packs the
     * frames back to back and the copy over PCIe lands them in `out` directly */
    {
        int uniform = !c->host_assembly && np > 0;
        for (int i = 1; i < np && uniform; i++)
            uniform = pieces[i].n == pieces[0].n &&
                      pieces[i].pcm == pieces[0].pcm + (size_t)i * (size_t)pieces[0].n * nch;
        if (uniform) {
            const double tg0 = now_ms();
            fhip_batch b;
            memset(&b, 0, sizeof b);
            b.pcm = pieces[0].pcm; b.nframes = np; b.block_size = pieces[0].n;
            b.frame_bytes = c->fbytes;
            for (int i = 0; i < np; i++) c->fnum[i] = num_of[i];
            b.frame_numbers = c->fnum;
            int64_t wrote = 0;
            if (c->hip2_state == 0 && c->chunk_frames > 0 && np >= 2 * c->chunk_frames) {
                /* the chunk handle only ever runs uniform batches (fhip_encode_frames_packed_begin): without
                 * variable_block_size its subframe-indexed workspaces are sized for max_frames, not for the 2.5 x
                 * max_frames slots a variable-block-size batch's eight bins need (allow_vbs, which the frame
                 * headers depend on, stays) */
                fhip_params hp2 = c->hp;
                hp2.variable_block_size = 0;
                c->hip2_state = (fhip_create(&c->hip2, c->device, &hp2, c->chunk_frames) == FHIP_OK) ? 1 : -1;
            }
            if (c->hip2_state < 0) c->hip2 = NULL;                 /* fine: one handle, one pass */
            if (c->hip2 && c->chunk_frames > 0 && np >= 2 * c->chunk_frames) {
                wrote = run_chunked(c, pieces[0].pcm, np, pieces[0].n, nch, out, cap);
                if (wrote < 0) goto out;
            } else {
                const int rc = fhip_encode_frames_packed(c->hip, &b, out, (int64_t)cap, &wrote);
                if (rc != FHIP_OK) {
                    snprintf(c->err, sizeof c->err, "fhip_encode_frames_packed: %s (%s)", fhip_strerror(rc),
                             fhip_last_error(c->hip));
                    goto out;
                }
            }
            t_gpu = now_ms() - tg0; ngroups = 1;
            int cur_block = -1;
            for (int i = 0; i < np; i++) {
                const int fs = c->fbytes[i];
                if (fs <= 0) { snprintf(c->err, sizeof c->err, "frame %d was not encoded", i); goto out; }
                if (fs > c->max_frame_size) c->max_frame_size = fs;    /* encode.c:967 */
                c->frame_count += s->params.allow_vbs ? (uint32_t)pieces[i].n : 1u;   /* encode.c:969-975 */
                if (frame_sizes) {
                    if (pieces[i].block != cur_block) { cur_block = pieces[i].block; frame_sizes[cur_block] = 0; }
                    frame_sizes[cur_block] += fs;
                }
            }
            total = (long long)wrote;
            goto hashed;
        }
    }
    {
        int next_slot = 0;
        for (int i = 0; i < np; i++) {
            if (done[i]) continue;
            const int n = pieces[i].n;
            /* gather all pieces of this length */
            int cnt = 0, contiguous = 1;
            const int32_t *base = pieces[i].pcm;
            for (int j = i; j < np; j++) {
                if (done[j] || pieces[j].n != n) continue;
                if (pieces[j].pcm != base + (size_t)cnt * n * nch) contiguous = 0;
                cnt++;
            }
            const int32_t *src = base;
            const double tg0 = now_ms();
            if (!contiguous) {
                int k = 0;
                for (int j = i; j < np; j++) {
                    if (done[j] || pieces[j].n != n) continue;
                    memcpy(c->gather + (size_t)k * n * nch, pieces[j].pcm, sizeof(int32_t) * (size_t)n * nch);
                    k++;
                }
                src = c->gather;
            }
            {
                int k = 0;
                for (int j = i; j < np; j++) {
                    if (done[j] || pieces[j].n != n) continue;
                    c->fnum[next_slot + k] = num_of[j];
                    k++;
                }
            }
            const double tg1 = now_ms();
            if (run_gpu(c, src, cnt, n, (size_t)next_slot * nch, (size_t)next_slot) != FHIP_OK) goto out;
            t_gather += tg1 - tg0; t_gpu += now_ms() - tg1; ngroups++;
            int k = 0;
            for (int j = i; j < np; j++) {
                if (done[j] || pieces[j].n != n) continue;
                slot_of[j] = next_slot + k;
                done[j] = 1;
                k++;
            }
            next_slot += cnt;
        }
    }
    /* host: frames in stream order */
    {
        const double to0 = now_ms();
        size_t pos = 0;
        int cur_block = -1;
        for (int i = 0; i < np; i++) {
            int fs;
            if (c->host_assembly) {
                const size_t sub = (size_t)slot_of[i] * nch;
                fs = assemble_frame(c, c->frame_count, pieces[i].pcm, pieces[i].n,
                                    c->info + sub, c->bits + sub * (size_t)c->slot,
                                    out + pos, cap - pos, scratch);
            } else {
                fs = c->fbytes[slot_of[i]];
                if (fs > 0 && (size_t)fs <= cap - pos)
                    memcpy(out + pos, c->frames + (size_t)slot_of[i] * (size_t)c->fstride, (size_t)fs);
                else fs = -1;
            }
            if (fs < 0) { snprintf(c->err, sizeof c->err, "output buffer too small"); goto out; }
            if (fs > c->max_frame_size) c->max_frame_size = fs;    /* encode.c:967 */
            c->frame_count += s->params.allow_vbs ? (uint32_t)pieces[i].n : 1u;   /* encode.c:969-975 */
            if (frame_sizes) {
                if (pieces[i].block != cur_block) { cur_block = pieces[i].block; frame_sizes[cur_block] = 0; }
                frame_sizes[cur_block] += fs;
            }
            pos += (size_t)fs;
        }
        total = (long long)pos;
        t_out = now_ms() - to0;
    }
hashed:
    if (c->trace)
        fprintf(stderr, "flake_amd batch: %d blocks -> %d frames in %d size groups; split %.2f ms, gather %.2f, "
                        "gpu (H2D + kernels + D2H) %.2f, copy-out %.2f\n", count, np, ngroups, t_split, t_gather,
                t_gpu, t_out);
    if (!c->md5_off) {
        if (md5_running) { pthread_join(md5_thread, NULL); md5_running = 0; }
        else md5_worker(&job);                         /* no thread: hash here */
        md5_done = 1;
    }
out:
    if (md5_running) pthread_join(md5_thread, NULL);
    if (md5_done && total >= 0) c->md5 = md5_next;
    free(dev_nf); free(dev_sizes);
    free(pieces); free(scratch); free(slot_of); free(done); free(num_of);
    return total;
}

FLAKE_AMD_API long long flake_amd_encode_frames(FlakeAmdContext *s, const int *samples, int nblocks,
                                                int block_size, int tail_size, unsigned char *out,
                                                size_t out_size, int *frame_sizes)
{
    if (!s || !samples || !s->private_ctx || !out) return -1;
    host_ctx *c = (host_ctx *)s->private_ctx;
    c->err[0] = 0;
    if (nblocks < 0 || block_size < 1 || block_size > s->params.block_size) return -1;   /* encode.c:987 */
    if (tail_size < 0 || tail_size >= block_size + (nblocks == 0)) return -1;
    if (c->last_frame) return -1;                                                        /* encode.c:989 */
    if (!s->params.allow_vbs && nblocks > 0 && block_size != s->params.block_size) {
        /* a short block latches the end of the stream (encode.c:991-992) */
        if (nblocks > 1 || tail_size) return -1;
        c->last_frame = 1;
    }
    const size_t bstride = (size_t)block_size * (size_t)s->channels;
    long long total = 0;
    for (int b0 = 0; b0 < nblocks; b0 += c->max_batch) {
        const int cnt = (nblocks - b0 < c->max_batch) ? nblocks - b0 : c->max_batch;
        long long w = encode_batch(s, c, (const int32_t *)samples + (size_t)b0 * bstride, cnt, block_size,
                                   out + total, out_size - (size_t)total,
                                   frame_sizes ? frame_sizes + b0 : NULL);
        if (w < 0) return -1;
        total += w;
    }
    if (tail_size > 0) {
        if (!s->params.allow_vbs) c->last_frame = 1;
        long long w = encode_batch(s, c, (const int32_t *)samples + (size_t)nblocks * bstride, 1, tail_size,
                                   out + total, out_size - (size_t)total,
                                   frame_sizes ? frame_sizes + nblocks : NULL);
        if (w < 0) return -1;
        total += w;
    }
    return total;
}

/* Page-lock the caller's batch buffers in place (either may be NULL: left as it is; bytes = 0 releases).
 * The ranges must stay mapped until they are released here, replaced by another call, or the stream is
 * closed.  Returns 0, or -1 when the runtime refused a range (the copies then run from pageable memory
 * as before: not an error for the stream). */
FLAKE_AMD_API int flake_amd_pin_buffers(FlakeAmdContext *s, const int *samples, size_t sample_bytes,
                                        unsigned char *out, size_t out_bytes)
{
    if (!s || !s->private_ctx) return -1;
    host_ctx *c = (host_ctx *)s->private_ctx;
    void *ptr[2] = {(void *)samples, (void *)out};
    const size_t bytes[2] = {sample_bytes, out_bytes};
    int rc = 0;
    if (c->hip) (void)fhip_sync(c->hip);
    if (c->hip2) (void)fhip_sync(c->hip2);
    for (int k = 0; k < 2; k++) {
        if (!ptr[k]) continue;
        if (c->pin[k].ok) { (void)fhip_host_unregister(c->pin[k].ptr); c->pin[k].ok = 0; }
        if (bytes[k] == 0 || c->pin_off) continue;
        c->pin[k].ptr = ptr[k]; c->pin[k].bytes = bytes[k];
        c->pin[k].ok = fhip_host_register(ptr[k], bytes[k]) == FHIP_OK;
        if (!c->pin[k].ok) rc = -1;
    }
    return rc;
}

/* flake_encode_frame(), flake.h:229 / encode.c:979-1008.
 *
 * With FLAKE_AMD_LOOKAHEAD=N (and FlakeContext.samples known) an unmodified
 * caller loop (flake/flake.c:624-663: read block, flake_encode_frame, write fs
 * bytes from flake_get_buffer() if fs > 0) gets GPU batching: whole blocks are
 * copied into a queue and the call returns 0 -- which libflake's callers already
 * treat as "nothing to write" -- until N blocks are queued, the stream's last
 * sample has been handed in, or a short block arrives; that call encodes the
 * queue as one batch and returns all its frames back to back in the buffer,
 * exactly as a VBS call returns several frames (vbs.c:104-116).  The stream is
 * byte-identical to the unqueued one.  A caller that stops before `samples` are
 * in loses the queued blocks, so the switch is opt-in. */
FLAKE_AMD_API int flake_amd_encode_frame(FlakeAmdContext *s, const int *samples, int block_size)
{
    if (!s || !samples || !s->private_ctx) return -1;
    host_ctx *c = (host_ctx *)s->private_ctx;
    if (block_size < 1 || block_size > s->params.block_size) return -1;
    const int short_block = block_size != s->params.block_size;
    if (c->lookahead && c->q_pcm && !c->last_frame) {
        const size_t bstride = (size_t)s->params.block_size * (size_t)s->channels;
        memcpy(c->q_pcm + (size_t)c->q_count * bstride, samples,
               sizeof(int32_t) * (size_t)block_size * (size_t)s->channels);
        c->q_seen += (unsigned long long)block_size;
        if (!short_block) c->q_count++;
        const int flush = short_block || c->q_count >= c->lookahead || c->q_seen >= s->samples;
        if (!flush) return 0;
        const int cnt = c->q_count;
        c->q_count = 0;
        if (c->q_seen >= s->samples) c->lookahead = 0;          /* past the announced end: block by block */
        const long long w = flake_amd_encode_frames(s, c->q_pcm, cnt, s->params.block_size,
                                                    short_block ? block_size : 0, c->frame_buffer,
                                                    (size_t)c->frame_buffer_size, NULL);
        return (w < 0 || w > 0x7FFFFFFFLL) ? -1 : (int)w;
    }
    long long w = flake_amd_encode_frames(s, samples, short_block ? 0 : 1, short_block ? s->params.block_size : block_size,
                                          short_block ? block_size : 0, c->frame_buffer,
                                          (size_t)c->frame_buffer_size, NULL);
    return (int)w;
}


/* ------------------------------------------------------------------ */
/* libflake's own symbol names (flake.h:217-295)                       */
/* ------------------------------------------------------------------ */
/* Built only into libflake.so (-DFLAKE_AMD_EXPORT_FLAKE_NAMES): a program
 * written and compiled against the reference's flake.h links against this
 * library unchanged -- FlakeContext / FlakeEncodeParams / FlakeStreaminfo have
 * the layouts of the FlakeAmd* structs above. */
#ifdef FLAKE_AMD_EXPORT_FLAKE_NAMES
FLAKE_AMD_API int flake_set_defaults(FlakeAmdEncodeParams *p) { return flake_amd_set_defaults(p); }
FLAKE_AMD_API int flake_validate_params(const FlakeAmdContext *s) { return flake_amd_validate_params(s); }
FLAKE_AMD_API int flake_encode_init(FlakeAmdContext *s)
{
    const int rc = flake_amd_encode_init(s);
    return rc < 0 ? -1 : rc;                       /* libflake knows only -1 */
}
FLAKE_AMD_API void *flake_get_buffer(const FlakeAmdContext *s) { return flake_amd_get_buffer(s); }
FLAKE_AMD_API int flake_encode_frame(FlakeAmdContext *s, const int *samples, int block_size)
{
    const int rc = flake_amd_encode_frame(s, samples, block_size);
    return rc < 0 ? -1 : rc;
}
FLAKE_AMD_API void flake_encode_close(FlakeAmdContext *s) { flake_amd_encode_close(s); }
FLAKE_AMD_API const char *flake_get_version(void) { return flake_amd_get_version(); }
FLAKE_AMD_API int flake_get_streaminfo(const FlakeAmdContext *s, FlakeAmdStreaminfo *si)
{
    return flake_amd_get_streaminfo(s, si);
}
FLAKE_AMD_API void flake_write_streaminfo(const FlakeAmdStreaminfo *si, unsigned char *data)
{
    flake_amd_write_streaminfo(si, data);
}
#endif
