/*
 * flake_amd.h -- host-side C layer above the HIP C ABI (flakehip.h).
 *
 * Mirrors libflake's public surface (flake.h:217-234) for the path this
 * project accelerates: the structs below are layout-compatible with
 * FlakeEncodeParams (flake.h:59-161) and FlakeContext (flake.h:163-215), and
 * each function names the libflake function it stands in for.  The host layer
 * owns what libflake keeps on the CPU around encode_residual(): frame/subframe
 * headers, CRC-8/16, the verbatim fallback, the frame counter and MD5.
 *
 * File:line citations are relative to the reference tree (/root/reference).
 */
#ifndef FLAKE_AMD_H
#define FLAKE_AMD_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define FLAKE_AMD_API __attribute__((visibility("default")))
#else
#define FLAKE_AMD_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- libflake-compatible encoder surface ----------------------------- */

/* Layout-compatible with FlakeEncodeParams (flake.h:59-161); same field
 * meaning and ranges. */
typedef struct FlakeAmdEncodeParams {
    int compression;             /* 0..12 */
    int order_method;            /* FLAKE_ORDER_METHOD_* 0..6 */
    int stereo_method;           /* 0 independent, 1 estimate */
    int block_size;
    int padding_size;
    int min_prediction_order;
    int max_prediction_order;
    int prediction_type;         /* 0 none, 1 fixed, 2 levinson */
    int min_partition_order;
    int max_partition_order;
    int variable_block_size;
    int allow_vbs;
} FlakeAmdEncodeParams;

/* Layout-compatible with FlakeContext (flake.h:163-215). */
typedef struct FlakeAmdContext {
    int channels;
    int sample_rate;
    int bits_per_sample;
    unsigned int samples;        /* total stream samples, 0 = unknown */
    FlakeAmdEncodeParams params;
    unsigned char *header;       /* allocated by init, freed by close */
    void *private_ctx;
} FlakeAmdContext;

/* flake.h:241-251 FlakeStreaminfo */
typedef struct FlakeAmdStreaminfo {
    unsigned int min_block_size, max_block_size;
    unsigned int min_frame_size, max_frame_size;
    unsigned int sample_rate, channels, bits_per_sample, samples;
    unsigned char md5sum[16];
} FlakeAmdStreaminfo;

/* flake_set_defaults(), encode.c:158-266: params->compression must be set */
FLAKE_AMD_API int flake_amd_set_defaults(FlakeAmdEncodeParams *params);
/* flake_validate_params(), encode.c:268-373: -1 error, 0 ok, 1 ok but non-Subset */
FLAKE_AMD_API int flake_amd_validate_params(const FlakeAmdContext *s);
/* flake_encode_init(), encode.c:378-472: returns the header length (bytes in
 * s->header) or a negative code.  The HIP device is FLAKE_AMD_DEVICE (default
 * 0); up to FLAKE_AMD_BATCH (default 1024) blocks are encoded per GPU batch.
 * FLAKE_AMD_LOOKAHEAD=N (with FlakeContext.samples set) makes the one-block
 * flake_amd_encode_frame() queue N blocks per GPU batch: it returns 0 while it
 * queues and all queued frames at once when it flushes (see flake_host.c).
 * FLAKE_AMD_MD5=0 skips the stream MD5 (STREAMINFO then carries the all-zero
 * "not computed" signature); FLAKE_AMD_HOST_ASSEMBLY=1 / FLAKE_AMD_HOST_VBS=1 move
 * frame assembly / block splitting back to the CPU (for comparison);
 * FLAKE_AMD_TRACE=1 prints the phase times of every batch on stderr. */
FLAKE_AMD_API int flake_amd_encode_init(FlakeAmdContext *s);
/* flake_get_buffer(), encode.c:474-485: frame buffer of flake_amd_encode_frame */
FLAKE_AMD_API void *flake_amd_get_buffer(const FlakeAmdContext *s);
/* flake_encode_frame(), encode.c:979-1008: one block; bytes written or -1 */
FLAKE_AMD_API int flake_amd_encode_frame(FlakeAmdContext *s, const int *samples, int block_size);
/*
 * Batched form of the same call: nblocks consecutive blocks of block_size
 * samples per channel (interleaved), optionally followed by one shorter last
 * block of tail_size samples (0 = none).  Frames are written back to back
 * into out; frame_sizes (optional, [nblocks + (tail_size > 0)], VBS may write
 * several FLAC frames per block -- their total is recorded per block) and the
 * return value give byte counts.  Negative on error.
 */
FLAKE_AMD_API long long flake_amd_encode_frames(FlakeAmdContext *s, const int *samples,
                                                int nblocks, int block_size, int tail_size,
                                                unsigned char *out, size_t out_size,
                                                int *frame_sizes);
/* Page-lock the buffers a caller hands to flake_amd_encode_frames() batch after batch (the loop of
 * flake.c:622-663 reads every block into the same buffer), in place: copies from and to pageable memory run
 * at about two thirds of the link's rate.  Either pointer may be NULL (left as it is); bytes = 0 releases a
 * range.  The ranges must stay mapped until released, replaced or the stream is closed.  Returns 0, or -1
 * when a range was refused (encoding works as before).  The look-ahead queue of flake_amd_encode_frame()
 * and its frame buffer are page-locked by the library itself.  FLAKE_AMD_PIN=0 disables both. */
FLAKE_AMD_API int flake_amd_pin_buffers(FlakeAmdContext *s, const int *samples, size_t sample_bytes,
                                        unsigned char *out, size_t out_bytes);
/* flake_encode_close(), encode.c:1010-1026 */
FLAKE_AMD_API void flake_amd_encode_close(FlakeAmdContext *s);
/* flake_get_streaminfo() / flake_write_streaminfo(), metadata.c:32-84 */
FLAKE_AMD_API int flake_amd_get_streaminfo(const FlakeAmdContext *s, FlakeAmdStreaminfo *si);
FLAKE_AMD_API void flake_amd_write_streaminfo(const FlakeAmdStreaminfo *si, unsigned char *data34);
FLAKE_AMD_API const char *flake_amd_get_version(void);
/* Text of the last error of this context ("" if none). */
FLAKE_AMD_API const char *flake_amd_last_error(const FlakeAmdContext *s);

/* Deterministic synthetic PCM, channel-interleaved int32 as
 * flake_encode_frame() expects: nframes blocks of n samples per channel,
 * starting at absolute frame index first_frame. */
FLAKE_AMD_API void flake_amd_synth_pcm(int32_t *pcm, int64_t first_frame, int nframes,
                                       int n, int channels, int bps);

#ifdef __cplusplus
}
#endif
#endif /* FLAKE_AMD_H */
